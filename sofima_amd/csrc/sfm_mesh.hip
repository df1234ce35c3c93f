// Elastic spring-mesh relaxer for gfx950.
//
// Device replacement for mesh.py of the reference:
//   inplane_force     (mesh.py:42-169)   -> link stencil, 4 in-plane families
//   elastic_mesh_3d   (mesh.py:192-279)  -> link stencil, up to 13 families
//   velocity_verlet   (mesh.py:371-521)  -> one "advance" + one "integrate"
//                                           kernel per step, FIRE scalars kept
//                                           on the device
//
// Data layout: state arrays x, v, a, prev are float [C, B, Z, Y, X] (C = 2|3
// vector components, x fastest).  One thread owns one node (all C components)
// and gathers its neighbours; every spring is evaluated by both of its end
// nodes, so there are no atomics and results are run-to-run deterministic.
// This translation unit is compiled with -ffp-contract=off so that the f32
// arithmetic follows the reference's operation order without FMA fusion.
//
// Step structure (FIRE; the damped-Verlet path skips the scalar logic):
//   advance(k):   [k > 0] every block reduces the per-block partials of step
//                 k-1 in a fixed order -> power, drift sums -> updates
//                 (dt, alpha, n_pos, cap), gates v, removes drift;
//                 then x += dt v + dt^2/2 a
//   integrate(k): a' = F(x) + clip(-k0 (x - prev)); v = VV(v, a, a');
//                 partial power = sum a'.v; FIRE velocity mixing;
//                 per-block partials -> global
//   finish:       the pending gate / drift / scalar update of the last step,
//                 then e_kin and v_max.
#include "sfm_common.h"
#include "sfm_target.h"

#include <cmath>
#include <cstdlib>
#include <cstring>


namespace {

typedef unsigned long long u64;
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;
// Volumes below this many nodes keep integrate_kernel<3>: a column of planes is a
// serial chain of barriers, and a small volume has too few columns to fill the CUs
// (per step, per-node kernel against z-march: [3,1,64^3] 28 us / 55, [3,8,48^3] 66 / 78,
// [3,1,100^3] 75 / 80, [3,2,100^3] 132 / 106, [3,1,128^3] 137 / 110, [3,4,100^3] 251 / 185,
// [3,1,160^3] 257 / 183: tools/measure/march3d_sizes.py).
constexpr long long kMarch3dMinNodes = 1500000;
// Minimum waves per SIMD the register allocator must leave room for
// (__launch_bounds__ second argument).  Measured on [3,4,100^3] / [2,64,204^2]:
// the volumetric stencil takes 180 VGPRs unconstrained (2 waves per SIMD, 369 us
// per step); 3 waves (167 VGPRs, no scratch) 334 us; 4 and 5 waves spill and are
// slower (456 / 638 us).  The tiled in-plane step (127 VGPRs, 4 waves) only
// loses with 5, 6 or 8 waves (73 -> 87 / 110 / 134 us): both are bound by the
// latency of their long dependent chains, not by occupancy.
#ifndef SFM_LB3
#define SFM_LB3 3
#endif
// integrate_shared2d_kernel outside band mode: four workgroups per CU (128
// VGPRs; 130 without the bound).  Measured: 1-3 % over three once the SGPR
// spills were gone; with them (and 20 bytes of scratch) it had been 7 % slower.
#ifndef SFM_LB_SHARED
#define SFM_LB_SHARED 4
#endif
constexpr int kNP = 8;  // partials per block: power, sx[3], sv[3], pad

struct MeshParams {
  int ncomp;
  int B, Z, Y, X;
  long long N;  // nodes = B*Z*Y*X
  int n_links;
  int order2d;  // reference summation order of inplane_force
  int default_links;  // ncomp 3 with MESH_LINK_DIRECTIONS: compile-time unrolled path
  int dir[SFM_MESH_MAX_LINKS][3];     // xyz
  float rest[SFM_MESH_MAX_LINKS][3];  // xyz rest vector
  float neg_k[SFM_MESH_MAX_LINKS];    // -k_eff
  int prefer;
  float neg_k0;
  int has_prev;
  // integrator
  int fire;
  int remove_drift;
  int drift_cols;  // reference quirk for 5-D arrays: drift means per x column
  float n_col;     // nodes per x column (B * Z * Y) as the f32 mean divisor
  float gamma;
  float vv_dt;  // damped Verlet: fixed dt
  float f_alpha, f_inc, f_dec, alpha0;
  int n_min;
  float dt_cap;
  float final_cap, cap_scale;
  int cap_every;
  float n_f;  // N as the f32 mean divisor (mean = sum / N)
  int own_y0, own_y1;   // band shards: rows that count in sums / statistics
  int force_kind;       // SFM_FORCE_*
  const float* cx;      // tile mesh: desired offset to the +x tile, [C, B, Y, X]
  const float* cy;      // tile mesh: desired offset to the +y tile
  const float* ext;     // external force, [C, N]
};

typedef sfm::MeshScalars Scalars;

// Correctly rounded square root and division without the sub-normal scaling of
// the compiler's IEEE sequences (15 -> 9 and 11 -> 8 instructions; the spring
// evaluations are VALU bound).  Results are bit-identical to sqrtf() / operator/
// for normal-range operands (12.6 * 10^9 random samples per function on an
// MI355X, scratch/sqdiv.hip: no mismatch), for 0, inf and NaN inputs of the
// square root, and for the division whenever the quotient is used by the
// spring law: a zero length gives NaN instead of inf (both end as a zero
// force), an infinite length gives exactly 0 like IEEE.  Lengths below 1e-19
// (sub-normal squares) are outside the contract.
__device__ __forceinline__ float sfm_sqrt(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float lo = __int_as_float(__float_as_int(s) - 1);
  const float hi = __int_as_float(__float_as_int(s) + 1);
  const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
  s = rl <= 0.f ? lo : s;
  s = rh > 0.f ? hi : s;
  return s;
}

__device__ __forceinline__ float sfm_div(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  const float q = a * r;
  const float d = __builtin_fmaf(-b, q, a);
  const float out = __builtin_fmaf(d, r, q);
  return isinf(b) ? 0.f * a : out;  // finite / inf = 0 (NaN stays NaN)
}

__device__ __forceinline__ float vec_len(const float* d, int c) {
  float s = d[0] * d[0] + d[1] * d[1];
  if (c == 3) s = s + d[2] * d[2];
  return sfm_sqrt(s);
}

// Force of one spring given d = x_far - x_near + rest (mesh.py:107-117,
// 252-270).  Non-finite components become 0 like nan_to_num(posinf=0,
// neginf=0).
// One IEEE division per spring: (l0 * m) / l == (l0 / l) * m bit for bit when
// m is -1, 0 or 1 (sign flips and zeros commute with rounding; the NaN / inf
// cases give NaN both ways and end up 0), so the per-component quotients of
// the reference (mesh.py:109-116) share r = l0 / l.
template <int C>
__device__ __forceinline__ void spring(const float* d, const float* rest,
                                       const int* dir, float neg_k, int prefer,
                                       float* f) {
  const float l = vec_len(d, C);
  const float l0 = vec_len(rest, C);
  const float r = sfm_div(l0, l);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float t = r;
    if (prefer && dir[c] != 0) {
      const float sg = d[c] > 0.f ? 1.f : (d[c] < 0.f ? -1.f : 0.f);
      t = r * (static_cast<float>(dir[c]) * sg);
    }
    const float u = 1.0f - t;
    float v = (neg_k * u) * d[c];
    if (!isfinite(v)) v = 0.f;
    f[c] = v;
  }
}

// In-plane spring with compile-time link direction (DX, DY).
template <int DX, int DY>
__device__ __forceinline__ void spring_xy(float d0, float d1, float l0, float neg_k,
                                          int prefer, float* f) {
  const float l = sfm_sqrt(d0 * d0 + d1 * d1);
  const float r = sfm_div(l0, l);
  float t0 = r, t1 = r;
  if (prefer) {
    if (DX != 0) {
      const float sg = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f);
      t0 = r * (static_cast<float>(DX) * sg);
    }
    if (DY != 0) {
      const float sg = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
      t1 = r * (static_cast<float>(DY) * sg);
    }
  }
  float v0 = (neg_k * (1.0f - t0)) * d0;
  float v1 = (neg_k * (1.0f - t1)) * d1;
  if (!isfinite(v0)) v0 = 0.f;
  if (!isfinite(v1)) v1 = 0.f;
  f[0] = v0;
  f[1] = v1;
}

// The same spring with a run-time link direction (dx, dy in {-1, 0, 1}): the
// operations and their order are those of spring_xy<DX, DY>, so the result is
// bit-identical (used where the link differs from lane to lane).
__device__ __forceinline__ void spring_xy_rt(float d0, float d1, float l0, float neg_k,
                                             int prefer, int dx, int dy, float* f) {
  const float l = sfm_sqrt(d0 * d0 + d1 * d1);
  const float r = sfm_div(l0, l);
  float t0 = r, t1 = r;
  if (prefer) {
    const float sg0 = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f);
    const float sg1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
    const float m0 = r * (static_cast<float>(dx) * sg0);
    const float m1 = r * (static_cast<float>(dy) * sg1);
    t0 = dx != 0 ? m0 : r;
    t1 = dy != 0 ? m1 : r;
  }
  float v0 = (neg_k * (1.0f - t0)) * d0;
  float v1 = (neg_k * (1.0f - t1)) * d1;
  if (!isfinite(v0)) v0 = 0.f;
  if (!isfinite(v1)) v1 = 0.f;
  f[0] = v0;
  f[1] = v1;
}

// inplane_force at one node from an LDS tile: the four link families of
// build_params (ncomp 2) unrolled, same operation order as node_force_at with
// order2d.  `ctr` indexes the node inside a tile of row pitch TW.
template <int TW>
__device__ __forceinline__ void node_force_tile2d(const float* xt0, const float* xt1,
                                                  int ctr, const MeshParams& p, int xi,
                                                  int yi, float s0, float s1,
                                                  const float* l0, float* out) {
  float acc0 = 0.f, acc1 = 0.f;
  // Branch free: every spring is evaluated (the neighbour cell always exists in
  // the LDS tile, possibly with stale contents) and a missing one contributes
  // +0: eight independent chains the scheduler can interleave, instead of
  // eight exec-masked blocks padded with hazard nops.
#define SFM_FAR(L, DX, DY)                                                         \
  {                                                                                \
    const bool ok = xi - (DX) >= 0 && xi - (DX) < p.X && yi - (DY) >= 0 &&         \
                    yi - (DY) < p.Y;                                               \
    const int m = ctr - (DY) * TW - (DX);                                          \
    float f[2];                                                                    \
    spring_xy<DX, DY>(s0 - xt0[m] + p.rest[L][0], s1 - xt1[m] + p.rest[L][1],      \
                      l0[L], p.neg_k[L], p.prefer, f);                             \
    acc0 = acc0 + (ok ? f[0] : 0.f);                                               \
    acc1 = acc1 + (ok ? f[1] : 0.f);                                               \
  }
#define SFM_NEAR(L, DX, DY)                                                        \
  {                                                                                \
    const bool ok = xi + (DX) >= 0 && xi + (DX) < p.X && yi + (DY) >= 0 &&         \
                    yi + (DY) < p.Y;                                               \
    const int m = ctr + (DY) * TW + (DX);                                          \
    float f[2];                                                                    \
    spring_xy<DX, DY>(xt0[m] - s0 + p.rest[L][0], xt1[m] - s1 + p.rest[L][1],      \
                      l0[L], p.neg_k[L], p.prefer, f);                             \
    acc0 = acc0 - (ok ? f[0] : 0.f);                                               \
    acc1 = acc1 - (ok ? f[1] : 0.f);                                               \
  }
  SFM_FAR(0, 1, 0) SFM_FAR(1, 0, 1) SFM_FAR(2, 1, 1) SFM_FAR(3, -1, 1)
  SFM_NEAR(0, 1, 0) SFM_NEAR(1, 0, 1) SFM_NEAR(2, 1, 1) SFM_NEAR(3, -1, 1)
#undef SFM_FAR
#undef SFM_NEAR
  out[0] = acc0;
  out[1] = acc1;
}

// Net spring force on the node at (xi, yi, zi).  `ld(c, dx, dy, dz)` returns
// component c of the node at that offset (only called for in-range offsets).
template <int C, typename Load>
__device__ __forceinline__ void node_force_at(Load ld, const MeshParams& p, int xi,
                                              int yi, int zi, const float* self,
                                              float* out) {
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;

  auto far_side = [&](int L, float* f) -> bool {
    // This node is the far end; the near end is node - dir.
    const int nx = xi - p.dir[L][0], ny = yi - p.dir[L][1],
              nz = zi - p.dir[L][2];
    if (nx < 0 || nx >= p.X || ny < 0 || ny >= p.Y || nz < 0 || nz >= p.Z)
      return false;
    float d[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
      d[c] = self[c] - ld(c, -p.dir[L][0], -p.dir[L][1], -p.dir[L][2]) + p.rest[L][c];
    spring<C>(d, p.rest[L], p.dir[L], p.neg_k[L], p.prefer, f);
    return true;
  };
  auto near_side = [&](int L, float* f) -> bool {
    const int nx = xi + p.dir[L][0], ny = yi + p.dir[L][1],
              nz = zi + p.dir[L][2];
    if (nx < 0 || nx >= p.X || ny < 0 || ny >= p.Y || nz < 0 || nz >= p.Z)
      return false;
    float d[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
      d[c] = ld(c, p.dir[L][0], p.dir[L][1], p.dir[L][2]) - self[c] + p.rest[L][c];
    spring<C>(d, p.rest[L], p.dir[L], p.neg_k[L], p.prefer, f);
    return true;
  };

  float f[C];
  if (p.order2d) {
    // f1p + f2p + f3p + f4p - f1n - f2n - f3n - f4n   (mesh.py:169)
    for (int L = 0; L < p.n_links; ++L)
      if (far_side(L, f)) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = acc[c] + f[c];
      }
    for (int L = 0; L < p.n_links; ++L)
      if (near_side(L, f)) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = acc[c] - f[c];
      }
  } else {
    // per link: += fp; -= fn   (mesh.py:271-277)
    for (int L = 0; L < p.n_links; ++L) {
      if (far_side(L, f)) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = acc[c] + f[c];
      }
      if (near_side(L, f)) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = acc[c] - f[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) out[c] = acc[c];
}

// Volumetric spring with compile-time link direction (see spring<C>).
template <int DX, int DY, int DZ>
__device__ __forceinline__ void spring_xyz(const float* d, float l0, float neg_k,
                                           int prefer, float* f) {
  const float l = vec_len(d, 3);
  const float r = sfm_div(l0, l);
  constexpr int dir[3] = {DX, DY, DZ};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float t = r;
    if (prefer && dir[c] != 0) {
      const float sg = d[c] > 0.f ? 1.f : (d[c] < 0.f ? -1.f : 0.f);
      t = r * (static_cast<float>(dir[c]) * sg);
    }
    float v = (neg_k * (1.0f - t)) * d[c];
    if (!isfinite(v)) v = 0.f;
    f[c] = v;
  }
}

// The 13 default links of a volumetric mesh have rest = dir * stride (exactly: dir
// is -1, 0 or 1) and fall into seven classes by the axes they span -- x, y, z, xy,
// xz, yz, xyz -- with one rest length and one spring constant per class.  Reading
// rest[13][3] and neg_k[13] as 52 separate wave-uniform values had the 3-D kernels
// spill 75-239 SGPRs (one v_readlane in eight VALU instructions of
// integrate_kernel<3>); three strides and seven constants are the same numbers.
#define SFM_CLASS3(DX, DY, DZ)                                                        \
  ((DX) != 0 && (DY) != 0 && (DZ) != 0 ? 6                                           \
   : (DY) != 0 && (DZ) != 0 ? 5 : (DX) != 0 && (DZ) != 0 ? 4 : (DX) != 0 && (DY) != 0 ? 3 \
   : (DZ) != 0 ? 2 : (DY) != 0 ? 1 : 0)
struct DefLinks3 {
  float st[3];    // +stride per axis = the rest vectors of links 0, 1, 2
  float l0c[7];   // rest length per class (same device function as the live lengths)
  float nkc[7];   // -k_eff per class (representative links 0, 1, 2, 3, 5, 7, 9)
  __device__ __forceinline__ explicit DefLinks3(const MeshParams& p) {
    st[0] = p.rest[0][0];
    st[1] = p.rest[1][1];
    st[2] = p.rest[2][2];
    const int cls[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float r[3] = {rest(cls[k][0], 0), rest(cls[k][1], 1), rest(cls[k][2], 2)};
      l0c[k] = vec_len(r, 3);
    }
    nkc[0] = p.neg_k[0];
    nkc[1] = p.neg_k[1];
    nkc[2] = p.neg_k[2];
    nkc[3] = p.neg_k[3];
    nkc[4] = p.neg_k[5];
    nkc[5] = p.neg_k[7];
    nkc[6] = p.neg_k[9];
  }
  __device__ __forceinline__ float rest(int d, int c) const {
    return d == 0 ? 0.f : (d > 0 ? st[c] : -st[c]);
  }
};

// elastic_mesh_3d with the 13 default links (MESH_LINK_DIRECTIONS), unrolled with
// compile-time directions and branch free.  Same per-link order as node_force_at:
// += far end, -= near end (mesh.py:271-277).  26 independent chains instead of 26
// exec-masked blocks: the small 3-D meshes of a volumetric montage are bound by this
// kernel's latency.  (r4: class constants instead of 52 wave-uniform rest / k values:
// spills 233 -> 119, [3,4,100^3] 334 -> 309 us per step.)
// A spring whose other end lies outside the mesh is evaluated against the node
// itself: d = rest, l = l0, l0 / l = 1 exactly, force = k * 0 * d = +-0 (a NaN position
// gives NaN, which spring_xyz turns into 0); adding / subtracting that +-0 to a sum that
// started at +0 gives what the reference's masked +0 gives -- no select.  The 26 "other
// end exists" conditions are six VGPR words combined with v_and and AND-ed into the
// partner offset (as booleans they were SGPR pairs, most of the spilled SGPRs).
// Offsets are 32-bit bytes: build_params routes a mesh whose three component planes
// exceed 4 GB to the generic link loop (node_force_at).
__device__ __forceinline__ void node_force_default3d(const float* __restrict__ x,
                                                        const MeshParams& p, unsigned n,
                                                        int xi, int yi, int zi,
                                                        const float* self, float* out) {
  float acc[3] = {0.f, 0.f, 0.f};
  const int syb = p.X * 4, szb = p.X * p.Y * 4;
  const unsigned Nb = static_cast<unsigned>(p.N) * 4u;
  const unsigned nb = n * 4u;
  const unsigned kxm = xi > 0 ? ~0u : 0u, kxp = xi + 1 < p.X ? ~0u : 0u;
  const unsigned kym = yi > 0 ? ~0u : 0u, kyp = yi + 1 < p.Y ? ~0u : 0u;
  const unsigned kzm = zi > 0 ? ~0u : 0u, kzp = zi + 1 < p.Z ? ~0u : 0u;
  const DefLinks3 dl(p);
  auto ld = [&](unsigned b) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + b);
  };
#define SFM_LINK(DX, DY, DZ)                                                        \
  {                                                                                  \
    constexpr int kc = SFM_CLASS3(DX, DY, DZ);                                       \
    const float l0 = dl.l0c[kc];                                                     \
    const float rest[3] = {dl.rest(DX, 0), dl.rest(DY, 1), dl.rest(DZ, 2)};          \
    /* the far end exists where the node has a neighbour at -dir, the near end at +dir */ \
    const unsigned kn = ((DX) > 0 ? kxp : (DX) < 0 ? kxm : ~0u) &                    \
                        ((DY) > 0 ? kyp : (DY) < 0 ? kym : ~0u) &                    \
                        ((DZ) > 0 ? kzp : (DZ) < 0 ? kzm : ~0u);                     \
    const unsigned kf = ((DX) > 0 ? kxm : (DX) < 0 ? kxp : ~0u) &                    \
                        ((DY) > 0 ? kym : (DY) < 0 ? kyp : ~0u) &                    \
                        ((DZ) > 0 ? kzm : (DZ) < 0 ? kzp : ~0u);                     \
    const unsigned off = static_cast<unsigned>((DX) * 4 + (DY) * syb + (DZ) * szb);  \
    const unsigned mf = nb - (off & kf), mn = nb + (off & kn);                       \
    float df[3], dn[3], ff[3], fn[3];                                                \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                  \
      df[c] = self[c] - ld(mf + c * Nb) + rest[c];                                   \
      dn[c] = ld(mn + c * Nb) - self[c] + rest[c];                                   \
    }                                                                                \
    spring_xyz<DX, DY, DZ>(df, l0, dl.nkc[kc], p.prefer, ff);                        \
    spring_xyz<DX, DY, DZ>(dn, l0, dl.nkc[kc], p.prefer, fn);                        \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                  \
      acc[c] = acc[c] + ff[c];                                                       \
      acc[c] = acc[c] - fn[c];                                                       \
    }                                                                                \
  }
  SFM_LINK(1, 0, 0) SFM_LINK(0, 1, 0) SFM_LINK(0, 0, 1) SFM_LINK(1, 1, 0)
  SFM_LINK(-1, 1, 0) SFM_LINK(1, 0, 1) SFM_LINK(-1, 0, 1) SFM_LINK(0, 1, 1)
  SFM_LINK(0, -1, 1) SFM_LINK(1, 1, 1) SFM_LINK(1, 1, -1) SFM_LINK(1, -1, 1)
  SFM_LINK(-1, 1, 1)
#undef SFM_LINK
  out[0] = acc[0];
  out[1] = acc[1];
  out[2] = acc[2];
}

// jnp.nan_to_num with its defaults: nan -> 0, +-inf -> +-FLT_MAX.
__device__ __forceinline__ float nan_to_num_default(float v) {
  if (isnan(v)) return 0.f;
  if (isinf(v)) return v > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

// stitch_rigid.elastic_tile_mesh (stitch_rigid.py:330-388) / elastic_tile_mesh_3d
// (:391-473) at one node.  Every node is a tile; the pair (i, i+1) along x
// contributes t = nan_to_num((x_c[i+1] - x_c[i]) - cx_c[i]) to node i and -t to
// node i+1, likewise along y with cy, for every vector component c.  The
// reference adds the terms family by family (f_tot += pad(f); f_tot -= pad(f)):
// component 0: x pairs, then y pairs; component 1: y pairs, then x pairs;
// component 2: x pairs, then y pairs.  Sections (z) are independent.
template <int C>
__device__ __forceinline__ void tile_mesh_force(const float* __restrict__ x,
                                                const MeshParams& p, long long n,
                                                int xi, int yi, float* out) {
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float* xc = x + c * p.N;
    const float s = xc[n];
    float ax_p = 0.f, ax_m = 0.f, ay_p = 0.f, ay_m = 0.f;
    if (xi + 1 < p.X) ax_p = nan_to_num_default((xc[n + 1] - s) - p.cx[c * p.N + n]);
    if (xi > 0) ax_m = nan_to_num_default((s - xc[n - 1]) - p.cx[c * p.N + n - 1]);
    if (yi + 1 < p.Y) ay_p = nan_to_num_default((xc[n + p.X] - s) - p.cy[c * p.N + n]);
    if (yi > 0) ay_m = nan_to_num_default((s - xc[n - p.X]) - p.cy[c * p.N + n - p.X]);
    float acc = 0.f;
    if (c == 1) {
      acc = acc + ay_p;
      acc = acc - ay_m;
      acc = acc + ax_p;
      acc = acc - ax_m;
    } else {
      acc = acc + ax_p;
      acc = acc - ax_m;
      acc = acc + ay_p;
      acc = acc - ay_m;
    }
    out[c] = acc;
  }
}

template <int C>
__device__ void node_force(const float* __restrict__ x, const MeshParams& p,
                           long long n, float* out) {
  if (p.force_kind == SFM_FORCE_EXTERNAL) {
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = p.ext[c * p.N + n];
    return;
  }
  const int xi = static_cast<int>(n % p.X);
  long long r = n / p.X;
  const int yi = static_cast<int>(r % p.Y);
  r /= p.Y;
  const int zi = static_cast<int>(r % p.Z);
  if (p.force_kind == SFM_FORCE_TILE_MESH) {
    tile_mesh_force<C>(x, p, n, xi, yi, out);
    return;
  }
  float self[C];
#pragma unroll
  for (int c = 0; c < C; ++c) self[c] = x[c * p.N + n];
  if (C == 3 && p.default_links) {
    node_force_default3d(x, p, static_cast<unsigned>(n), xi, yi, zi, self, out);
    return;
  }
  node_force_at<C>(
      [&](int c, int dx, int dy, int dz) {
        return x[c * p.N + n + dx + (long long)dy * p.X + (long long)dz * p.X * p.Y];
      },
      p, xi, yi, zi, self, out);
}

// clip(-k0 * nan_to_num(x - prev), -cap, cap)   (mesh.py:432-433)
__device__ __forceinline__ float prev_pull(float x, float prev, float neg_k0,
                                           float cap) {
  float d = x - prev;
  if (isnan(d)) d = 0.f;
  if (isinf(d)) d = d > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  const float pl = neg_k0 * d;
  return fminf(fmaxf(pl, -cap), cap);
}

template <int C>
__global__ void __launch_bounds__(kBlock, C == 3 ? SFM_LB3 : 1)
force_kernel(const float* __restrict__ x, const float* __restrict__ prev,
             float* __restrict__ out, MeshParams p, float cap, int add_prev) {
  for (long long n = blockIdx.x * (long long)kBlock + threadIdx.x; n < p.N;
       n += (long long)gridDim.x * kBlock) {
    float f[C];
    node_force<C>(x, p, n, f);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = f[c];
      if (add_prev)
        v = v + prev_pull(x[c * p.N + n], prev[c * p.N + n], p.neg_k0, cap);
      out[c * p.N + n] = v;
    }
  }
}

// Fixed-order block reduction of `nv` values per thread (nv <= kNP).
__device__ void block_sum(float* vals, int nv, float* lds) {
  for (int i = 0; i < nv; ++i) lds[i * kBlock + threadIdx.x] = vals[i];
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int i = 0; i < nv; ++i)
        lds[i * kBlock + threadIdx.x] =
            lds[i * kBlock + threadIdx.x] + lds[i * kBlock + threadIdx.x + s];
    __syncthreads();
  }
  for (int i = 0; i < nv; ++i) vals[i] = lds[i * kBlock];
  __syncthreads();
}

// Advances the FIRE scalars from the summed partials of the previous step
// (mesh.py:455-497).
__device__ void scalars_from_sums(const Scalars& in, const float* acc,
                                  const MeshParams& p, Scalars* out) {
  const float power = acc[0];
  Scalars s = in;
  const bool downhill = power >= 0.f;
  s.n_pos = downhill ? in.n_pos + 1 : 0;
  if (downhill) {
    if (s.n_pos > p.n_min) {
      s.dt = fminf(in.dt * p.f_inc, p.dt_cap);
      s.alpha = in.alpha * p.f_alpha;
    }
    if (s.n_pos > 0 && (s.n_pos % p.cap_every) == 0) s.cap = p.cap_scale * in.cap;
  } else {
    s.dt = in.dt * p.f_dec;
    s.alpha = p.alpha0;
  }
  s.cap = fminf(s.cap, p.final_cap);
  s.gate = downhill ? 1.f : 0.f;
  for (int c = 0; c < 3; ++c) {
    s.mx[c] = p.remove_drift ? acc[1 + c] / p.n_f : 0.f;
    s.mv[c] = p.remove_drift ? (acc[4 + c] / p.n_f) * s.gate : 0.f;
  }
  *out = s;
}

// Reduces the partials of the previous step and advances the FIRE scalars.
// Every block computes the identical result.
__device__ void update_scalars(const Scalars& in, const float* partials,
                               int n_part_rows, const MeshParams& p,
                               float* lds, Scalars* out) {
  float acc[kNP];
  for (int i = 0; i < kNP; ++i) acc[i] = 0.f;
  for (int r = threadIdx.x; r < n_part_rows; r += kBlock)
    for (int i = 0; i < 7; ++i) acc[i] = acc[i] + partials[r * kNP + i];
  block_sum(acc, 7, lds);
  scalars_from_sums(in, acc, p, out);
}

// x += dt v + dt^2/2 a, after applying the pending gate / drift of the
// previous step (mesh.py:439, 492-497).
template <int C>
__global__ void __launch_bounds__(kBlock)
advance_kernel(float* __restrict__ x, float* __restrict__ v,
               const float* __restrict__ a, MeshParams p,
               const Scalars* __restrict__ scal_in, Scalars* __restrict__ scal_out,
               const float* __restrict__ partials, int n_part_rows,
               int pending, const float* __restrict__ colsum) {
  __shared__ float lds[kNP * kBlock];
  Scalars s;
  if (p.fire) {
    if (pending == 1) {
      update_scalars(*scal_in, partials, n_part_rows, p, lds, &s);
    } else if (pending == 2) {
      s = *scal_in;  // reduced by the last workgroup of the tiled integrator
    } else {
      s = *scal_in;
      s.gate = 1.f;
      for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *scal_out = s;
  } else {
    s.dt = p.vv_dt;
    s.gate = 1.f;
    for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
  }
  const float dt = s.dt;
  const float c2 = 0.5f * (dt * dt);
  const bool store_v = s.gate != 1.f || p.drift_cols || p.remove_drift;
  for (long long n = blockIdx.x * (long long)kBlock + threadIdx.x; n < p.N;
       n += (long long)gridDim.x * kBlock) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float xv = x[c * p.N + n];
      float vv = v[c * p.N + n];
      if (p.fire && pending) {
        vv = vv * s.gate;
        if (p.drift_cols) {
          const int xi = static_cast<int>(n % p.X);
          xv = xv - colsum[c * p.X + xi];
          vv = vv - colsum[(3 + c) * p.X + xi] * s.gate;
        } else if (p.remove_drift) {
          xv = xv - s.mx[c];
          vv = vv - s.mv[c];
        }
        // a downhill step without drift removal leaves v as it is (v * 1 = v): no store
        // (12 of the kernel's 60 bytes per node)
        if (store_v) v[c * p.N + n] = vv;
      }
      x[c * p.N + n] = xv + (dt * vv + c2 * a[c * p.N + n]);
    }
  }
}

template <int C>
__global__ void __launch_bounds__(kBlock, C == 3 ? SFM_LB3 : 1)
integrate_kernel(const float* __restrict__ x, float* __restrict__ v,
                 float* __restrict__ a, const float* __restrict__ prev,
                 MeshParams p, const Scalars* __restrict__ scal,
                 float fixed_cap, float* __restrict__ partials) {
  __shared__ float lds[kNP * kBlock];
  float dt, alpha, cap;
  if (p.fire) {
    dt = scal->dt;
    alpha = scal->alpha;
    cap = scal->cap;
  } else {
    dt = p.vv_dt;
    alpha = 0.f;
    cap = fixed_cap;
  }
  const float hdtg = (0.5f * dt) * p.gamma;
  const float fact0 = 1.0f / (1.0f + hdtg);
  const float fact1 = 1.0f - hdtg;
  const float hdt = 0.5f * dt;
  float part[kNP];
  for (int i = 0; i < kNP; ++i) part[i] = 0.f;
  for (long long n = blockIdx.x * (long long)kBlock + threadIdx.x; n < p.N;
       n += (long long)gridDim.x * kBlock) {
    float f[C], vn[C];
    node_force<C>(x, p, n, f);
    // halo rows of a band shard are integrated (their values are replaced by
    // the owner's at the next exchange) but do not count in the sums
    const int yrow = static_cast<int>((n / p.X) % p.Y);
    const bool own = yrow >= p.own_y0 && yrow < p.own_y1;
    float a2 = 0.f, v2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float xv = x[c * p.N + n];
      if (p.has_prev) f[c] = f[c] + prev_pull(xv, prev[c * p.N + n], p.neg_k0, cap);
      const float a_old = a[c * p.N + n];
      vn[c] = fact0 * (v[c * p.N + n] * fact1 + hdt * (a_old + f[c]));
      a[c * p.N + n] = f[c];
      a2 = a2 + f[c] * f[c];
      v2 = v2 + vn[c] * vn[c];
      if (p.fire && own) {
        part[0] = part[0] + f[c] * vn[c];
        part[1 + c] = part[1 + c] + xv;
      }
    }
    if (p.fire) {
      const float a_norm = sqrtf(a2) + 1e-6f;
      const float v_norm = sqrtf(v2);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        vn[c] = vn[c] + alpha * (f[c] / a_norm * v_norm - vn[c]);
        if (own) part[4 + c] = part[4 + c] + vn[c];
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) v[c * p.N + n] = vn[c];
  }
  if (p.fire) {
    block_sum(part, 7, lds);
    if (threadIdx.x == 0)
      for (int i = 0; i < kNP; ++i) partials[blockIdx.x * kNP + i] = part[i];
  }
}

// ---------------------------------------------------------------------------
// integrate_kernel<3> for the 13 default links, every spring evaluated ONCE.
//
// integrate_kernel<3> evaluates the 26 springs of a node from the node's own side
// (2 500 VALU instructions per node: the kernel runs at 0.14 of the HBM roofline).
// A spring seen from its two ends is the same float expression: with o = n - off,
//   far side of n  : d = x[n] - x[o] + rest        (df at node n)
//   near side of o : d = x[o + off] - x[o] + rest  (dn at node o)
// so the owner of a spring can publish the force and the other end can read it.
// Here a workgroup owns a column of the volume -- a tile of the (y, x) plane, T
// threads = tile positions INCLUDING a halo (one column on each side, one row on top)
// -- and marches along z.  In iteration z a thread holds its node of plane z (`self`)
// and of plane z + 1 (`next`) and owns 13 springs, every one between planes z / z + 1
// or inside plane z, chosen so that a reader finds its owner in its own tile row or the
// row above (offsets (-1,0), (+1,0), (0,-1), (-1,-1), (+1,-1): no halo row below):
//   links 0, 1, 3, 4       (dz = 0)   near form at `self`, partner in plane z
//                                     -> LDS `P`, read by the partner in this iteration
//   links 2, 5, 6, 7, 9, 12 (dz = +1) near form at `self`, partner in plane z + 1
//                                     -> LDS `U` (link 2: a register), read by the partner
//                                     in the NEXT iteration
//   links 8 = (0,-1,1), 11 = (1,-1,1): FAR form at `next` (the upper end), partner
//   link 10 = (1,1,-1):                near form at `next`,  in plane z at (., y+1)
//                                     -> LDS `P` for the partner (this iteration) and a
//                                     register for this thread's own node of plane z + 1
// so every owned spring needs the positions of planes z and z + 1 only, and a node adds
// its 26 terms in the reference's link order (mesh.py:271-277: += far side, -= near
// side, link by link) from registers, P and U: the same floats in the same order as
// node_force_default3d -- `a` and (without FIRE) every later state are bit-identical.
// The FIRE sums are added in a different order (a thread's column, then the
// workgroup tree) like those of the tiled in-plane integrator.
// Halo threads evaluate their springs (a value any tile computes is the same float)
// and take no part in the sums; a run of planes starts with one extra plane that only
// produces U and the registers.  LDS: 36 floats per thread.
// ---------------------------------------------------------------------------
struct March3dArgs {
  int txh, tyh;   // thread tile, the halo included (core = txh - 2 by tyh - 1)
  int ntx, nty;   // tiles per plane
  int cols;       // columns = B * nty * ntx
  int run;        // planes per workgroup: workgroup w owns planes [w * run, (w + 1) * run)
                  // of the sequence (column 0: z = 0 .. Z-1, column 1: ...)
};

// Owned spring of link (DX, DY, DZ) at node n.  FAR = false: the near form, partner
// n + off; FAR = true: the far form, partner n - off.  A missing partner is replaced
// by the node itself (d = rest; the result is masked where it is used).
// float at byte offset `b` (32 bits: an SGPR base + VGPR offset load)
__device__ __forceinline__ float ld_b(const float* __restrict__ base, unsigned b) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + b);
}

template <int DX, int DY, int DZ, bool FAR, bool PREFER>
__device__ __forceinline__ void march_spring(const float* __restrict__ x0,
                                             const float* __restrict__ x1,
                                             const float* __restrict__ x2, int syb, int szb,
                                             const DefLinks3& dl, unsigned nb, unsigned ok,
                                             const float* self, float* f) {
  constexpr int kc = SFM_CLASS3(DX, DY, DZ);
  // (formed here from the two strides: thirteen hoisted offsets were spilled SGPRs)
  asm volatile("" : "+s"(syb), "+s"(szb));
  const int offb = (DX) * 4 + (DY) * syb + (DZ) * szb;  // bytes
  // ok: all ones where the partner exists, else 0 (lane masks as SGPR pairs, one per
  // link, were a third of the kernel's spilled SGPRs)
  const unsigned mb = nb + (static_cast<unsigned>(FAR ? -offb : offb) & ok);
  const float rest[3] = {dl.rest(DX, 0), dl.rest(DY, 1), dl.rest(DZ, 2)};
  const float o[3] = {ld_b(x0, mb), ld_b(x1, mb), ld_b(x2, mb)};
  float d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = FAR ? self[c] - o[c] + rest[c] : o[c] - self[c] + rest[c];
  spring_xyz<DX, DY, DZ>(d, dl.l0c[kc], dl.nkc[kc], PREFER ? 1 : 0, f);
}

template <int T>
__device__ void block_sum_t(float* vals, int nv, float* lds) {
  for (int i = 0; i < nv; ++i) lds[i * T + threadIdx.x] = vals[i];
  __syncthreads();
  for (int s = T / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s)
      for (int i = 0; i < nv; ++i)
        lds[i * T + threadIdx.x] = lds[i * T + threadIdx.x] + lds[i * T + threadIdx.x + s];
    __syncthreads();
  }
  for (int i = 0; i < nv; ++i) vals[i] = lds[i * T];
  __syncthreads();
}

// v where the lane's mask is all ones, +0.0f where it is zero
__device__ __forceinline__ float keep_if(float v, unsigned mask) {
  return __uint_as_float(__float_as_uint(v) & mask);
}

#ifndef SFM_MARCH_LB
#define SFM_MARCH_LB(T) 4  // waves per SIMD: 1, 2, 4 workgroups of 1024, 512, 256 per CU
#endif
// A spring owned through the node of the NEXT plane (links 8, 10, 11, see the kernel):
// `next` is that node's position, the partner lies in plane z at byte offset offb from
// this column's plane-z node; a missing partner is replaced by the next-plane node itself
// (d = rest).  FAR: d = next - o + rest, else d = o - next + rest.
template <int DX, int DY, int DZ, bool FAR, bool PREFER>
__device__ __forceinline__ void march_spring_up(const float* __restrict__ x0,
                                                const float* __restrict__ x1,
                                                const float* __restrict__ x2, int offb, int szb,
                                                const DefLinks3& dl, unsigned nb, unsigned ok,
                                                const float* next, float* f) {
  constexpr int kc = SFM_CLASS3(DX, DY, DZ);
  asm volatile("" : "+s"(offb), "+s"(szb));  // (formed at its use, see march_spring)
  const unsigned mb = nb + static_cast<unsigned>(szb) + (static_cast<unsigned>(offb - szb) & ok);
  const float rest[3] = {dl.rest(DX, 0), dl.rest(DY, 1), dl.rest(DZ, 2)};
  const float o[3] = {ld_b(x0, mb), ld_b(x1, mb), ld_b(x2, mb)};
  float d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = FAR ? next[c] - o[c] + rest[c] : o[c] - next[c] + rest[c];
  spring_xyz<DX, DY, DZ>(d, dl.l0c[kc], dl.nkc[kc], PREFER ? 1 : 0, f);
}

template <int T, bool PREFER>
__global__ void __launch_bounds__(T, SFM_MARCH_LB(T))
integrate_march3d_kernel(const float* __restrict__ x, float* __restrict__ v,
                         float* __restrict__ a, const float* __restrict__ prev,
                         MeshParams p, const Scalars* __restrict__ scal,
                         float fixed_cap, float* __restrict__ partials, March3dArgs g) {
  extern __shared__ float march_lds[];
  // P: forces exchanged inside an iteration -- links 0, 1, 3, 4 of plane z and links 8,
  // 10, 11 between planes z and z + 1 (slots 4, 5, 6); U: links 5, 6, 7, 9, 12 of the
  // plane below, for the next iteration.
  float* P = march_lds;           // [7][3][T]
  float* U = march_lds + 21 * T;  // [5][3][T]
  const int tid = threadIdx.x;
  float dt, alpha, cap;
  if (p.fire) {
    dt = scal->dt;
    alpha = scal->alpha;
    cap = scal->cap;
  } else {
    dt = p.vv_dt;
    alpha = 0.f;
    cap = fixed_cap;
  }
  const float hdtg = (0.5f * dt) * p.gamma;
  const float fact0 = 1.0f / (1.0f + hdtg);
  const float fact1 = 1.0f - hdtg;
  const float hdt = 0.5f * dt;
  const DefLinks3 dl(p);
  const int W = g.txh;
  const int tx = tid % W, ty = tid / W;
  const int cxw = g.txh - 2, cyw = g.tyh - 1;  // halo: one column each side, one row on top
  const int sy = p.X, sz = p.X * p.Y;
  const int syb = sy * 4, szb = sz * 4;  // bytes
  const unsigned N = static_cast<unsigned>(p.N);
  const unsigned Nb = N * 4u;  // bytes per component plane (3 N floats < 4 GB: plan_march3d)
  const float* __restrict__ x0 = x;
  const float* __restrict__ x1 = x + N;
  const float* __restrict__ x2 = x + 2 * (size_t)N;
  float part[kNP];
  for (int i = 0; i < kNP; ++i) part[i] = 0.f;
  const long long all_planes = (long long)g.cols * p.Z;
  const long long pos1 = min(all_planes, (long long)(blockIdx.x + 1) * g.run);
  for (long long pos = (long long)blockIdx.x * g.run; pos < pos1;) {
    int r = static_cast<int>(pos / p.Z);
    const int z0 = static_cast<int>(pos - (long long)r * p.Z);
    const int z1 = static_cast<int>(min<long long>(p.Z, z0 + (pos1 - pos)));
    pos += z1 - z0;
    const int tix = r % g.ntx;
    r /= g.ntx;
    const int tiy = r % g.nty;
    const int b = r / g.nty;
    const int xi = tix * cxw - 1 + tx, yi = tiy * cyw - 1 + ty;
    const bool act = ty < g.tyh && xi >= 0 && xi < p.X && yi >= 0 && yi < p.Y;
    const bool core = act && tx >= 1 && tx < g.txh - 1 && ty >= 1;
    const bool own = yi >= p.own_y0 && yi < p.own_y1;
    // all ones where the neighbour in that direction exists
    const unsigned kxp = xi + 1 < p.X ? ~0u : 0u, kxm = xi > 0 ? ~0u : 0u;
    const unsigned kyp = yi + 1 < p.Y ? ~0u : 0u, kym = yi > 0 ? ~0u : 0u;
    int z = z0 > 0 ? z0 - 1 : 0;
    unsigned n = act ? static_cast<unsigned>(xi + sy * yi) + static_cast<unsigned>(sz) *
                           static_cast<unsigned>(b * p.Z + z)
                     : 0u;
    float self[3] = {0.f, 0.f, 0.f};
    unsigned nb = n * 4u;
    if (act) {
      self[0] = ld_b(x0, nb);
      self[1] = ld_b(x1, nb);
      self[2] = ld_b(x2, nb);
    }
    // forces this thread computed one iteration ago for what is now its node: the far
    // sides of links 2, 8, 11 and the near side of link 10 (+0 at the bottom of the volume)
    float c2[3] = {0.f, 0.f, 0.f}, c8[3] = {0.f, 0.f, 0.f}, c10[3] = {0.f, 0.f, 0.f},
          c11[3] = {0.f, 0.f, 0.f};
    // A slot nobody owns (outside the mesh) reads as +0: the side of a link whose owner
    // does not exist; so does the plane below the volume.
#pragma unroll
    for (int k = 0; k < 36; ++k) march_lds[k * T + tid] = 0.f;
    for (; z < z1; ++z, n += sz, nb += szb) {
      const bool zp = z + 1 < p.Z;
      const unsigned kzp = zp ? ~0u : 0u;
      const bool sum = z >= z0;
      float next[3] = {self[0], self[1], self[2]};
      float q8[3] = {0.f, 0.f, 0.f}, q10[3] = {0.f, 0.f, 0.f}, q11[3] = {0.f, 0.f, 0.f};
      if (act) {
        if (zp) {
          next[0] = ld_b(x0, nb + szb);
          next[1] = ld_b(x1, nb + szb);
          next[2] = ld_b(x2, nb + szb);
          // links that reach DOWN in y are owned through their upper end, this column's
          // node of plane z + 1, so that every reader finds its owner in its own row or the
          // row above (no halo row below the tile):
          //   8  ( 0,-1, 1): far form at `next`, partner (x, y+1) of plane z
          //   10 ( 1, 1,-1): near form at `next`, partner (x+1, y+1) of plane z
          //   11 ( 1,-1, 1): far form at `next`, partner (x-1, y+1) of plane z
          march_spring_up<0, -1, 1, true, PREFER>(x0, x1, x2, syb, szb, dl, nb, kyp, next, q8);
          march_spring_up<1, 1, -1, false, PREFER>(x0, x1, x2, 4 + syb, szb, dl, nb, kxp & kyp, next, q10);
          march_spring_up<1, -1, 1, true, PREFER>(x0, x1, x2, syb - 4, szb, dl, nb, kxm & kyp, next, q11);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          P[(4 * 3 + c) * T + tid] = q8[c];
          P[(5 * 3 + c) * T + tid] = q10[c];
          P[(6 * 3 + c) * T + tid] = q11[c];
        }
        if (sum) {
          float fin[4][3];  // links 0, 1, 3, 4
          march_spring<1, 0, 0, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxp, self, fin[0]);
          march_spring<0, 1, 0, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kyp, self, fin[1]);
          march_spring<1, 1, 0, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxp & kyp, self, fin[2]);
          march_spring<-1, 1, 0, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxm & kyp, self, fin[3]);
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) P[(k * 3 + c) * T + tid] = fin[k][c];
        }
      }
      __syncthreads();  // P of this iteration; U of the plane below
      float up[6][3];   // links 2, 5, 6, 7, 9, 12 (near form)
      if (act) {
        {
          // link 2: the partner is this column's next node
          constexpr int kc = SFM_CLASS3(0, 0, 1);
          const float rest[3] = {dl.rest(0, 0), dl.rest(0, 1), dl.rest(1, 2)};
          float d[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) d[c] = next[c] - self[c] + rest[c];
          spring_xyz<0, 0, 1>(d, dl.l0c[kc], dl.nkc[kc], PREFER ? 1 : 0, up[0]);
        }
        march_spring<1, 0, 1, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxp & kzp, self, up[1]);
        march_spring<-1, 0, 1, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxm & kzp, self, up[2]);
        march_spring<0, 1, 1, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kyp & kzp, self, up[3]);
        march_spring<1, 1, 1, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxp & kyp & kzp, self, up[4]);
        march_spring<-1, 1, 1, false, PREFER>(x0, x1, x2, syb, szb, dl, nb, kxm & kyp & kzp, self, up[5]);
        // A spring without partner was evaluated against the node itself: d = rest, l = l0,
        // l0 / l = 1 exactly, force = k * 0 * d = +-0 (a NaN position gives NaN, which
        // spring_xyz turns into 0) -- and subtracting +-0 from a sum that started at +0
        // gives what subtracting the reference's +0 gives.  No mask on the near sides.
        if (core && sum) {
          float acc[3] = {0.f, 0.f, 0.f};
          // link by link: += the far side, -= the near side (mesh.py:271-277)
#define SFM_TERM(FAR_EXPR, NEAR_EXPR)                                                     \
  _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                       \
    acc[c] = acc[c] + (FAR_EXPR);                                                       \
    acc[c] = acc[c] - (NEAR_EXPR);                                                      \
  }
#define SFM_P(K, IDX) P[((K) * 3 + c) * T + (IDX)]
#define SFM_U(K, IDX) U[((K) * 3 + c) * T + (IDX)]
          // clang-format off
          SFM_TERM(SFM_P(0, tid - 1),     SFM_P(0, tid))      // 0  ( 1, 0, 0)
          SFM_TERM(SFM_P(1, tid - W),     SFM_P(1, tid))      // 1  ( 0, 1, 0)
          SFM_TERM(c2[c],                 up[0][c])           // 2  ( 0, 0, 1)
          SFM_TERM(SFM_P(2, tid - 1 - W), SFM_P(2, tid))      // 3  ( 1, 1, 0)
          SFM_TERM(SFM_P(3, tid + 1 - W), SFM_P(3, tid))      // 4  (-1, 1, 0)
          SFM_TERM(SFM_U(0, tid - 1),     up[1][c])           // 5  ( 1, 0, 1)
          SFM_TERM(SFM_U(1, tid + 1),     up[2][c])           // 6  (-1, 0, 1)
          SFM_TERM(SFM_U(2, tid - W),     up[3][c])           // 7  ( 0, 1, 1)
          SFM_TERM(c8[c],                 SFM_P(4, tid - W))  // 8  ( 0,-1, 1)
          SFM_TERM(SFM_U(3, tid - 1 - W), up[4][c])           // 9  ( 1, 1, 1)
          SFM_TERM(SFM_P(5, tid - 1 - W), c10[c])             // 10 ( 1, 1,-1)
          SFM_TERM(c11[c],                SFM_P(6, tid + 1 - W))  // 11 ( 1,-1, 1)
          SFM_TERM(SFM_U(4, tid + 1 - W), up[5][c])           // 12 (-1, 1, 1)
          // clang-format on
#undef SFM_TERM
#undef SFM_P
#undef SFM_U
          // ---- integrate_kernel's node update ----
          float vn[3];
          float a2 = 0.f, v2 = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float xv = self[c];
            float f = acc[c];
            // one base pointer per array, the component in the 32-bit offset
            const unsigned nbc = nb + c * Nb;
            if (p.has_prev) f = f + prev_pull(xv, ld_b(prev, nbc), p.neg_k0, cap);
            const float a_old = ld_b(a, nbc);
            vn[c] = fact0 * (ld_b(v, nbc) * fact1 + hdt * (a_old + f));
            *reinterpret_cast<float*>(reinterpret_cast<char*>(a) + nbc) = f;
            acc[c] = f;
            a2 = a2 + f * f;
            v2 = v2 + vn[c] * vn[c];
            if (p.fire && own) {
              part[0] = part[0] + f * vn[c];
              part[1 + c] = part[1 + c] + xv;
            }
          }
          if (p.fire) {
            const float a_norm = sqrtf(a2) + 1e-6f;
            const float v_norm = sqrtf(v2);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              vn[c] = vn[c] + alpha * (acc[c] / a_norm * v_norm - vn[c]);
              if (own) part[4 + c] = part[4 + c] + vn[c];
            }
          }
#pragma unroll
          for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float*>(reinterpret_cast<char*>(v) + (nb + c * Nb)) = vn[c];
        }
      }
      __syncthreads();  // every read of P and U is done
      if (act) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
          for (int c = 0; c < 3; ++c) U[(k * 3 + c) * T + tid] = up[k + 1][c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          c2[c] = up[0][c];
          c8[c] = q8[c];
          c10[c] = q10[c];
          c11[c] = q11[c];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) self[c] = next[c];
    }
  }
  if (p.fire) {
    __syncthreads();
    block_sum_t<T>(part, 7, march_lds);
    if (threadIdx.x == 0)
      for (int i = 0; i < kNP; ++i) partials[blockIdx.x * kNP + i] = part[i];
  }
}

// Meshes of at most one workgroup's worth of nodes (tile meshes of rigid
// stitching, single small volumes): all `iters` steps of the advance /
// integrate pair above in ONE launch of ONE workgroup -- a thread owns a node,
// the neighbours' positions are exchanged through global memory (one CU, one
// L1) across workgroup barriers.  Same arithmetic in the same order as the
// two kernels with a grid of one block: bit-identical results, ~8 us of
// launch-bound step become ~1.5 us.
template <int C>
__global__ void __launch_bounds__(kBlock)
mesh_small_kernel(float* x, float* v, float* a, const float* prev, MeshParams p,
                  Scalars* scal, float fixed_cap, float* partials, int iters) {
  __shared__ float lds[kNP * kBlock];
  const long long n = threadIdx.x;
  const bool live = n < p.N;
  Scalars s = scal[0];
  float part[kNP];
  for (int i = 0; i < kNP; ++i) part[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    // -- advance_kernel (pending = it > 0) --
    if (p.fire) {
      if (it > 0) {
        // update_scalars over one row of partials: the block reduction adds
        // zeros to it (x + 0 keeps every value; -0 becomes +0 like there)
        float acc[kNP];
        for (int i = 0; i < kNP; ++i) acc[i] = part[i] + 0.f;
        Scalars sn;
        scalars_from_sums(s, acc, p, &sn);
        s = sn;
      } else {
        s.gate = 1.f;
        for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
      }
    } else {
      s.dt = p.vv_dt;
      s.gate = 1.f;
      for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
    }
    const float dt = s.dt;
    const float c2 = 0.5f * (dt * dt);
    if (live) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float xv = x[c * p.N + n];
        float vv = v[c * p.N + n];
        if (p.fire && it > 0) {
          vv = vv * s.gate;
          if (p.remove_drift) {
            xv = xv - s.mx[c];
            vv = vv - s.mv[c];
          }
          v[c * p.N + n] = vv;
        }
        x[c * p.N + n] = xv + (dt * vv + c2 * a[c * p.N + n]);
      }
    }
    __syncthreads();  // every position of this step is visible
    // -- integrate_kernel --
    float alpha, cap;
    if (p.fire) {
      alpha = s.alpha;
      cap = s.cap;
    } else {
      alpha = 0.f;
      cap = fixed_cap;
    }
    const float hdtg = (0.5f * dt) * p.gamma;
    const float fact0 = 1.0f / (1.0f + hdtg);
    const float fact1 = 1.0f - hdtg;
    const float hdt = 0.5f * dt;
    for (int i = 0; i < kNP; ++i) part[i] = 0.f;
    if (live) {
      float f[C], vn[C];
      node_force<C>(x, p, n, f);
      float a2 = 0.f, v2 = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float xv = x[c * p.N + n];
        if (p.has_prev) f[c] = f[c] + prev_pull(xv, prev[c * p.N + n], p.neg_k0, cap);
        const float a_old = a[c * p.N + n];
        vn[c] = fact0 * (v[c * p.N + n] * fact1 + hdt * (a_old + f[c]));
        a[c * p.N + n] = f[c];
        a2 = a2 + f[c] * f[c];
        v2 = v2 + vn[c] * vn[c];
        if (p.fire) {
          part[0] = part[0] + f[c] * vn[c];
          part[1 + c] = part[1 + c] + xv;
        }
      }
      if (p.fire) {
        const float a_norm = sqrtf(a2) + 1e-6f;
        const float v_norm = sqrtf(v2);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          vn[c] = vn[c] + alpha * (f[c] / a_norm * v_norm - vn[c]);
          part[4 + c] = part[4 + c] + vn[c];
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) v[c * p.N + n] = vn[c];
    }
    if (p.fire)
      block_sum(part, 7, lds);  // ends with a barrier: positions may move again
    else
      __syncthreads();
  }
  if (threadIdx.x == 0) {
    // what the last advance / integrate pair leaves behind for finish_kernel
    scal[0] = s;
    scal[1] = s;
    for (int i = 0; i < kNP; ++i) partials[i] = part[i];
  }
}

// The reducing (last) workgroup of a tiled step gathers the per-tile partial
// sums.  Every producer's stores were acknowledged before it took its ticket,
// so the granules are there: the loads of a batch are issued back to back
// (independent, one round trip per batch instead of one per granule -- the
// first version's 91 dependent round trips per thread were ~20 % of a step on
// [2,64,204,204]) and only a granule with a stale tag is polled.  Without
// drift removal only the power sum is read.  Rows are added in tile order.
__device__ __forceinline__ void tile_tail_gather(const u64* __restrict__ partials, int rows,
                                                 int nval, unsigned epoch, float* acc) {
  for (int i = 0; i < kNP; ++i) acc[i] = 0.f;
  constexpr int kBatchRows = 8;
  for (int r0 = threadIdx.x; r0 < rows; r0 += kBlock * kBatchRows) {
    u64 gr[kBatchRows][7];
#pragma unroll
    for (int u = 0; u < kBatchRows; ++u) {
      const int r = r0 + u * kBlock;
#pragma unroll
      for (int i = 0; i < 7; ++i)
        gr[u][i] = (r < rows && i < nval)
                       ? __hip_atomic_load(&partials[r * kNP + i], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT)
                       : (static_cast<u64>(epoch) << 32);
    }
#pragma unroll
    for (int u = 0; u < kBatchRows; ++u) {
      const int r = r0 + u * kBlock;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        u64 g = gr[u][i];
        for (int spin = 0; static_cast<unsigned>(g >> 32) != epoch && spin < (1 << 22);
             ++spin)
          g = __hip_atomic_load(&partials[r * kNP + i], __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
        acc[i] = acc[i] + __uint_as_float(static_cast<unsigned>(g));
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Tiled step for large in-plane meshes (state far beyond the caches), every
// spring evaluated ONCE.
//
// One workgroup owns a 16 x 62 tile of one section.  It loads x (FUSED: and v,
// a) of the tile plus a one-node halo with coalesced loads that are all in
// flight together and writes x, v, a of its own nodes once: ~59 bytes per node
// update against the algorithmic 56 (the advance / integrate pair moves 88 and
// waits for every neighbour load separately).  FUSED = true is the whole FIRE /
// Verlet step in one launch: the position update x += dt v + dt^2/2 a of the
// halo nodes is recomputed from the neighbours' (x, v, a), so the state
// ping-pongs between two buffer sets.  FUSED = false integrates positions that
// advance_kernel (and the native prev_fn) already produced, in place.  The
// per-tile partial sums are reduced by the LAST workgroup to finish (ticket
// counter), in tile order, so the result does not depend on which one is last;
// it leaves the updated FIRE scalars for the next launch.
//
// A spring between node n and n + dir is the "near side" term of n and the "far
// side" term of n + dir: the same d = x[n + dir] - x[n] + rest, the same force,
// bit for bit (mesh.py:107-169 adds it to one end and subtracts it from the
// other).  Evaluating it at both ends costs 8 evaluations of ~54 VALU slots per
// node (the first tiled kernel of this file, rounds 1-2, did).  Here a lane
// owns one COLUMN of a 16-row tile (a wave = 4 consecutive rows x 64 columns:
// 62 owned + the two halo columns), evaluates only the four near-side springs
// of its nodes (and three of the row above its rows), and receives the
// far-side terms from the lane to its left / right with DPP wave shifts (the
// vertical one from its own registers): 19 evaluations + 24 shifts per four
// nodes instead of 32 evaluations.  Sums are taken in the reference's order
// (far sides of links 0..3, then near sides), so the forces are bit-identical
// to the other kernels'.
// ---------------------------------------------------------------------------
constexpr int kSX = 62, kSY = 16;

// value of the lane to the left / right (lane 0 / 63 receive 0)
__device__ __forceinline__ float lane_left(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
}
__device__ __forceinline__ float lane_right(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));  // wave_shl:1
}

// Band mode of the tiled step (one mesh split into bands of rows, possibly on
// several GPUs: sfm_mesh_relax_banded).  `sums` holds the partial sums of ALL
// bands of the previous step, in band order; every workgroup of every band
// adds them up in that order (pending == 3), so all bands take the same FIRE
// branch.  The reducing workgroup leaves this band's sums of the step in
// `my_sums` instead of advancing the scalars.  A step may be split into two
// launches -- the tile rows at the band's edges first, so that their rows can
// travel to the neighbour while the interior is computed: `ty_mode` 1 = only
// tile rows ty_a / ty_b, 2 = all other tile rows, 0 = all; `total_tiles` counts
// the tiles of both launches (ticket / partial-sum slots are per tile).
// One band of a multi-band launch (device memory, one entry per local band and
// state parity): every local band of a mesh is stepped by the SAME launch -- a
// workgroup finds its band from its block index -- and a band writes the rows
// at its edges straight into the neighbour's halo rows (or into the packed
// send buffer of a neighbour on another GPU), so a step of any number of local
// bands is one launch (two when the edge tile rows go first).
struct BandDev {
  const float* in[3];   // x, v, a of the step's input set
  float* out[3];
  const float* prev;
  u64* partials;
  int* ticket;
  float* my_sums;
  const Scalars* scal_in;
  Scalars* scal_out;
  float* nb[2][3];      // destination arrays of my first / last owned row (or null)
  long long nb_n[2];    // their component stride,
  long long nb_plane[2];  // plane stride
  long long nb_off[2];  // and row offset (row * X)
  long long N;
  int Y, own_y0, own_y1;
  int nty, tiles, ty_a, ty_b;
  int base[3];          // first block of this band in a launch of ty_mode 0 / 1 / 2
};

struct BandArgs {
  const float* sums;   // [n_bands, kNP] or nullptr: not in band mode
  float* my_sums;      // [kNP]
  int n_bands;
  int total_tiles;     // 0: gridDim.x
  int ty_mode, ty_a, ty_b;
  const BandDev* multi;  // multi-band launch: the per-band fields above and the
  int n_multi;           // kernel's array / scalar arguments come from here
  int xcd_map;           // blocks of one XCD (block % 8) take a contiguous run of tiles
  // Packed remainder column (outside band mode).  When X is not a multiple of kSX
  // the last tile column uses rem = X % kSX of its 62 lanes (204 columns: 18).  With
  // pack_S >= 2 a workgroup of that column takes pack_S tile rows instead, side by
  // side in lane segments of pack_segw = rem + 2 lanes (own columns + one halo lane
  // each side): a lane's row offset is its segment's.  Workgroups per plane:
  // nty * pack_ntxw full-width tiles, then ceil(nty / pack_S) packed ones.
  int pack_S, pack_segw, pack_ntxw, pack_recip;   // recip: ceil(2^16 / segw)
};

// Sum of the bands' partial sums in band order + the FIRE update from them
// (wave-uniform addresses: scalar loads).
__device__ __forceinline__ void band_scalars(const Scalars& in, const float* sums, int n_bands,
                                             const MeshParams& p, Scalars* out) {
  float acc[kNP];
  for (int i = 0; i < kNP; ++i) acc[i] = 0.f;
  const int nval = p.remove_drift ? 7 : 1;
  for (int r = 0; r < n_bands; ++r)
    for (int i = 0; i < nval; ++i) acc[i] = acc[i] + sums[r * kNP + i];
  scalars_from_sums(in, acc, p, out);
}

// The chunk's last pending update in band mode (finish_kernel then runs with
// pending == 2): same order as the step kernels'.
__global__ void band_scalars_kernel(const Scalars* __restrict__ scal_in,
                                    Scalars* __restrict__ scal_out,
                                    const float* __restrict__ sums, int n_bands, MeshParams p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Scalars o;
  band_scalars(*scal_in, sums, n_bands, p, &o);
  *scal_out = o;
}

// BAND: band mode (sfm_mesh_relax_banded).  Its own instantiation: the band
// tables, neighbour-row pointers and strides are wave-uniform values that stay
// live through the whole kernel, and the plain kernel was spilling 111 SGPRs
// (475 v_readlane / v_writelane in a VALU-bound kernel) with them.
template <bool FUSED, bool BAND = false>
__global__ void __launch_bounds__(kBlock, FUSED ? SFM_LB_SHARED : 1)
integrate_shared2d_kernel(const float* x_in, const float* v_in, const float* a_in,
                          const float* prev, float* x_out, float* v_out,
                          float* a_out, MeshParams p,
                          const Scalars* scal_in,
                          Scalars* scal_out, float fixed_cap,
                          u64* partials, int* ticket,
                          int pending, int nty, int ntx, BandArgs bd) {
  if (!BAND) {   // constants: everything that depends on them folds away
    bd.sums = nullptr;
    bd.my_sums = nullptr;
    bd.n_bands = 0;
    bd.total_tiles = 0;
    bd.ty_mode = 0;
    bd.multi = nullptr;
    bd.n_multi = 0;
  } else {
    bd.pack_S = 0;
  }
  constexpr int C = 2;
  constexpr int TW = 64;         // columns -1 .. kSX of the tile
  constexpr int kRows = 4;       // rows per thread
  __shared__ float xt[C][(kSY + 2) * TW];
  __shared__ float lds[kNP * kBlock];
  __shared__ int s_last;
  int block = blockIdx.x;   // block index within the band
  if (bd.xcd_map) {
    // consecutive workgroups go to different XCDs (own L2 each), neighbouring
    // tiles share halo rows / the cache lines at their column seams: give every
    // XCD one contiguous run of tiles
    const int nb = static_cast<int>(gridDim.x), per = nb >> 3, rem = nb & 7;
    const int xcd = block & 7, idx = block >> 3;
    block = xcd * per + min(xcd, rem) + idx;
  }
  const BandDev* band = nullptr;   // multi-band launch: this workgroup's band
  if (bd.multi) {
    int bi = 0;
    while (bi + 1 < bd.n_multi && block >= bd.multi[bi + 1].base[bd.ty_mode]) ++bi;
    const BandDev& bb = bd.multi[bi];
    block -= bb.base[bd.ty_mode];
    x_in = bb.in[0];
    v_in = bb.in[1];
    a_in = bb.in[2];
    x_out = bb.out[0];
    v_out = bb.out[1];
    a_out = bb.out[2];
    prev = bb.prev;
    partials = bb.partials;
    ticket = bb.ticket;
    scal_in = bb.scal_in;
    scal_out = bb.scal_out;
    bd.my_sums = bb.my_sums;
    bd.total_tiles = bb.tiles;
    bd.ty_a = bb.ty_a;
    bd.ty_b = bb.ty_b;
    nty = bb.nty;
    p.N = bb.N;
    p.Y = bb.Y;
    p.own_y0 = bb.own_y0;
    p.own_y1 = bb.own_y1;
    band = &bb;   // (the neighbour-row fields are read where they are used: held
                  // in registers through the kernel they cost 24 more SGPRs)
  }
  const unsigned epoch = static_cast<unsigned>(ticket[1]) + 1u;

  // tile of this workgroup (a split step enumerates a subset of the tile rows)
  int tile = block;
  if (bd.ty_mode) {
    const int n_edge = bd.ty_a == bd.ty_b ? 1 : 2;
    const int rows_here = bd.ty_mode == 1 ? n_edge : nty - n_edge;
    const int txi = block % ntx;
    const int j = (block / ntx) % rows_here;
    const int pl = block / (ntx * rows_here);
    int tyi;
    if (bd.ty_mode == 1) {
      tyi = j == 0 ? bd.ty_a : bd.ty_b;
    } else {
      tyi = j;
      if (tyi >= bd.ty_a) ++tyi;
      if (bd.ty_b != bd.ty_a && tyi >= bd.ty_b) ++tyi;
    }
    tile = (pl * nty + tyi) * ntx + txi;
  }
  const int total_tiles = bd.total_tiles ? bd.total_tiles : static_cast<int>(gridDim.x);

  Scalars s;
  if (p.fire) {
    s = *scal_in;
    if (pending == 3) {
      Scalars o;
      band_scalars(s, bd.sums, bd.n_bands, p, &o);
      s = o;
    } else if (!pending) {
      s.gate = 1.f;
      for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
    }
    // band mode: the scalars this step runs on are next step's starting point
    if (bd.sums && block == 0 && threadIdx.x == 0 && bd.ty_mode != 2) *scal_out = s;
  } else {
    s.dt = p.vv_dt;
    s.alpha = 0.f;
    s.cap = fixed_cap;
    s.gate = 1.f;
    for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
  }
  const float dt = s.dt, alpha = s.alpha, cap = s.cap;
  const float c2 = 0.5f * (dt * dt);
  const bool fix = p.fire && pending;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long long plane;   // b * Z + z
  int gx;            // this lane's column
  int gy0;           // first row of this lane's tile
  bool col_own;
  if (bd.pack_S) {
    const int n_wide = nty * bd.pack_ntxw;
    const int per_plane = n_wide + (nty + bd.pack_S - 1) / bd.pack_S;
    plane = tile / per_plane;
    const int b = tile - static_cast<int>(plane) * per_plane;
    if (b < n_wide) {
      const int ty = b / bd.pack_ntxw, tx = b - ty * bd.pack_ntxw;
      gx = tx * kSX + lane - 1;
      gy0 = ty * kSY;
      col_own = lane >= 1 && lane <= kSX;
    } else {
      const int seg = (lane * bd.pack_recip) >> 16;   // lane / segw
      const int ls = lane - seg * bd.pack_segw;
      const int ty = (b - n_wide) * bd.pack_S + seg;
      gx = bd.pack_ntxw * kSX + ls - 1;
      gy0 = min(ty, nty - 1) * kSY;
      col_own = seg < bd.pack_S && ty < nty && ls >= 1 && ls <= bd.pack_segw - 2;
    }
  } else {
    const int tx = tile % ntx;
    const int ty = (tile / ntx) % nty;
    plane = tile / (ntx * nty);
    gx = tx * kSX + lane - 1;
    gy0 = ty * kSY;
    col_own = lane >= 1 && lane <= kSX;
  }
  const long long base = plane * p.Y * p.X;
  const int gxc = min(max(gx, 0), p.X - 1);      // clamped for the loads
  col_own = col_own && gx < p.X;
  const int r0 = kRows * wave;                   // first tile row of this thread

  // Position of one node after the position update (FUSED) / as stored.
  auto advanced = [&](long long n, int c, float* v_keep, float* a_keep) -> float {
    float xv = x_in[c * p.N + n];
    if (!FUSED) return xv;
    float vv = v_in[c * p.N + n];
    const float aa = a_in[c * p.N + n];
    if (fix) {
      vv = vv * s.gate;
      if (p.remove_drift) {
        xv = xv - s.mx[c];
        vv = vv - s.mv[c];
      }
    }
    if (v_keep) *v_keep = vv;
    if (a_keep) *a_keep = aa;
    return xv + (dt * vv + c2 * aa);
  };

  // All loads up front, unconditional, from clamped coordinates.
  float x_own[kRows][C], v_own[kRows][C], a_own[kRows][C], pv_own[kRows][C];
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    const long long n = base + (long long)min(gy0 + r0 + k, p.Y - 1) * p.X + gxc;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      x_own[k][c] = advanced(n, c, &v_own[k][c], &a_own[k][c]);
      pv_own[k][c] = p.has_prev ? prev[c * p.N + n] : 0.f;
      if (!FUSED) {
        a_own[k][c] = a_in[c * p.N + n];
        v_own[k][c] = v_in[c * p.N + n];
      }
    }
  }
  // halo rows: the row above the tile (wave 0) and the row below it (wave 1)
  const int hrow = wave == 0 ? -1 : kSY;
  float x_halo[C] = {0.f, 0.f};
  if (wave < 2) {
    const long long n = base + (long long)min(max(gy0 + hrow, 0), p.Y - 1) * p.X + gxc;
#pragma unroll
    for (int c = 0; c < C; ++c) x_halo[c] = advanced(n, c, nullptr, nullptr);
  }
#pragma unroll
  for (int k = 0; k < kRows; ++k)
#pragma unroll
    for (int c = 0; c < C; ++c) xt[c][(r0 + k + 1) * TW + lane] = x_own[k][c];
  if (wave < 2) {
#pragma unroll
    for (int c = 0; c < C; ++c) xt[c][(hrow + 1) * TW + lane] = x_halo[c];
  }
  __syncthreads();

  float l0[4];
#pragma unroll
  for (int L = 0; L < 4; ++L) l0[L] = vec_len(p.rest[L], 2);
  // Near-side springs of rows r0 - 1 .. r0 + 3 of this column: ns[L][k + 1][c].
  // (Nodes or neighbours outside the mesh give garbage that the `ok` selects
  // below discard; columns beyond the tile are clamped, their values unused.)
  float ns[4][kRows + 1][C];
  const int lc = lane;
  const int lp = min(lane + 1, TW - 1), lm = max(lane - 1, 0);
#pragma unroll
  for (int k = -1; k < kRows; ++k) {
    const int row = (r0 + k + 1) * TW;  // LDS row of tile row r0 + k
    const float s0 = xt[0][row + lc], s1 = xt[1][row + lc];
    if (k >= 0)
      spring_xy<1, 0>(xt[0][row + lp] - s0 + p.rest[0][0], xt[1][row + lp] - s1 + p.rest[0][1],
                      l0[0], p.neg_k[0], p.prefer, ns[0][k + 1]);
    spring_xy<0, 1>(xt[0][row + TW + lc] - s0 + p.rest[1][0],
                    xt[1][row + TW + lc] - s1 + p.rest[1][1], l0[1], p.neg_k[1], p.prefer,
                    ns[1][k + 1]);
    spring_xy<1, 1>(xt[0][row + TW + lp] - s0 + p.rest[2][0],
                    xt[1][row + TW + lp] - s1 + p.rest[2][1], l0[2], p.neg_k[2], p.prefer,
                    ns[2][k + 1]);
    spring_xy<-1, 1>(xt[0][row + TW + lm] - s0 + p.rest[3][0],
                     xt[1][row + TW + lm] - s1 + p.rest[3][1], l0[3], p.neg_k[3], p.prefer,
                     ns[3][k + 1]);
  }

  const float hdtg = (0.5f * dt) * p.gamma;
  const float fact0 = 1.0f / (1.0f + hdtg);
  const float fact1 = 1.0f - hdtg;
  const float hdt = 0.5f * dt;
  float part[kNP];
  for (int i = 0; i < kNP; ++i) part[i] = 0.f;
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    const int gy = gy0 + r0 + k;
    // far-side terms: the near-side spring of the node at -dir (wave shifts
    // run on every lane, owners or not)
    float fs[4][C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      fs[0][c] = lane_left(ns[0][k + 1][c]);   // (x - 1, y)
      fs[1][c] = ns[1][k][c];                  // (x, y - 1)
      fs[2][c] = lane_left(ns[2][k][c]);       // (x - 1, y - 1)
      fs[3][c] = lane_right(ns[3][k][c]);      // (x + 1, y - 1)
    }
    // (band mode: halo rows belong to the neighbour band -- neither stored nor summed)
    const bool own = col_own && gy >= p.own_y0 && gy < p.own_y1;
    if (!own) continue;
    const bool xm = gx - 1 >= 0, xp = gx + 1 < p.X, ym = gy - 1 >= 0, yp = gy + 1 < p.Y;
    const bool okf[4] = {xm, ym, xm && ym, xp && ym};
    const bool okn[4] = {xp, yp, xp && yp, xm && yp};
    float f[C] = {0.f, 0.f}, vn[C];
#pragma unroll
    for (int L = 0; L < 4; ++L)
#pragma unroll
      for (int c = 0; c < C; ++c) f[c] = f[c] + (okf[L] ? fs[L][c] : 0.f);
#pragma unroll
    for (int L = 0; L < 4; ++L)
#pragma unroll
      for (int c = 0; c < C; ++c) f[c] = f[c] - (okn[L] ? ns[L][k + 1][c] : 0.f);
    const long long n = base + (long long)gy * p.X + gx;
    float a2 = 0.f, v2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float xv = x_own[k][c];
      if (p.has_prev) f[c] = f[c] + prev_pull(xv, pv_own[k][c], p.neg_k0, cap);
      const float a_old = a_own[k][c];
      const float v_old = v_own[k][c];
      vn[c] = fact0 * (v_old * fact1 + hdt * (a_old + f[c]));
      a_out[c * p.N + n] = f[c];
      if (FUSED) x_out[c * p.N + n] = xv;
      a2 = a2 + f[c] * f[c];
      v2 = v2 + vn[c] * vn[c];
      if (p.fire) {
        part[0] = part[0] + f[c] * vn[c];
        part[1 + c] = part[1 + c] + xv;
      }
    }
    if (p.fire) {
      const float a_norm = sqrtf(a2) + 1e-6f;
      const float v_norm = sqrtf(v2);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        vn[c] = vn[c] + alpha * (f[c] / a_norm * v_norm - vn[c]);
        part[4 + c] = part[4 + c] + vn[c];
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) v_out[c * p.N + n] = vn[c];
    // multi-band launch: my first / last owned row is the neighbour's halo row
    if (bd.multi) {
#pragma unroll
      for (int sd = 0; sd < 2; ++sd) {
        if (gy == (sd == 0 ? p.own_y0 : p.own_y1 - 1) && band->nb[sd][0]) {
          float* const dx = band->nb[sd][0];
          float* const dv = band->nb[sd][1];
          float* const da = band->nb[sd][2];
          const long long nn = band->nb_n[sd];
          const long long m0 = plane * band->nb_plane[sd] + band->nb_off[sd] + gx;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const long long m = c * nn + m0;
            dx[m] = x_own[k][c];
            dv[m] = vn[c];
            da[m] = f[c];
          }
        }
      }
    }
  }
  if (!p.fire) return;
  block_sum(part, 7, lds);
  // (hand-off to the last workgroup: see the header)
  if (threadIdx.x == 0) {
    for (int i = 0; i < 7; ++i)
      __hip_atomic_store(&partials[tile * kNP + i],
                         (static_cast<u64>(epoch) << 32) | __float_as_uint(part[i]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged before the ticket
    s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) == total_tiles - 1;
  }
  __syncthreads();
  if (!s_last) return;
  float acc[kNP];
  tile_tail_gather(partials, total_tiles, p.remove_drift ? 7 : 1, epoch, acc);
  block_sum(acc, 7, lds);
  if (threadIdx.x == 0) {
    if (bd.sums) {
      for (int i = 0; i < kNP; ++i) bd.my_sums[i] = i < 7 ? acc[i] : 0.f;
    } else {
      Scalars in = *scal_in, o;
      scalars_from_sums(in, acc, p, &o);
      *scal_out = o;
    }
    ticket[1] = static_cast<int>(epoch);
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Reference quirk Q3 (mesh.py:496-497): the drift means are taken over axes
// (1, 2, 3); for a 5-D state [3, N, z, y, x] that is a mean per x COLUMN.
// colsum[c][xi] = mean of x, colsum[3 + c][xi] = mean of v over the column, summed
// in a fixed order: thread (g, xi) of chunk k adds rows k * rows_per + g, + G, ...
// (consecutive threads read consecutive addresses), the G row groups of a chunk
// are added in order, and the last workgroup of a component (ticket) adds the
// chunks in order and divides.  (One workgroup per column striding through the
// rows took 14.7 us on [3,64,12,12,12]; this takes ~4.)
constexpr int kColChunksMax = 64;
constexpr int kColBatch = 16;

template <int C>
__global__ void __launch_bounds__(kBlock)
drift_cols_kernel(const float* __restrict__ x, const float* __restrict__ v,
                  MeshParams p, float* __restrict__ colsum, float* __restrict__ col_part,
                  int* __restrict__ col_ticket, int rows_per) {
  __shared__ float lds[2][kBlock];
  __shared__ int s_last;
  const int c = blockIdx.y, chunk = blockIdx.x, n_chunks = gridDim.x;
  const long long rows = p.N / p.X;
  const int G = p.X <= kBlock ? kBlock / p.X : 1;
  const long long r0 = (long long)chunk * rows_per;
  const long long r1 = min(rows, r0 + rows_per);
  // columns beyond the workgroup width are walked in passes (X > 256: G == 1)
  for (int x0 = 0; x0 < p.X; x0 += kBlock) {
    const int g = threadIdx.x / p.X, xi = x0 + (G > 1 ? threadIdx.x % p.X : threadIdx.x);
    float acc[2] = {0.f, 0.f};
    if (g < G && xi < p.X)
      // batches of independent loads (the adds stay in row order): a rolled loop
      // waits for every row's pair of loads in turn
      for (long long rb = r0 + g; rb < r1; rb += (long long)G * kColBatch) {
        float bx[kColBatch], bv[kColBatch];
#pragma unroll
        for (int k = 0; k < kColBatch; ++k) {
          const long long r = min(rb + (long long)k * G, rows - 1);
          bx[k] = x[c * p.N + r * p.X + xi];
          bv[k] = v[c * p.N + r * p.X + xi];
        }
#pragma unroll
        for (int k = 0; k < kColBatch; ++k)
          if (rb + (long long)k * G < r1) {
            acc[0] = acc[0] + bx[k];
            acc[1] = acc[1] + bv[k];
          }
      }
    lds[0][threadIdx.x] = acc[0];
    lds[1][threadIdx.x] = acc[1];
    __syncthreads();
    if (threadIdx.x < min(p.X - x0, kBlock)) {
      float s0 = lds[0][threadIdx.x], s1 = lds[1][threadIdx.x];
      for (int k = 1; k < G; ++k) {
        s0 = s0 + lds[0][threadIdx.x + k * p.X];
        s1 = s1 + lds[1][threadIdx.x + k * p.X];
      }
      float* dst = col_part + (((long long)c * n_chunks + chunk) * 2) * p.X + x0 + threadIdx.x;
      __builtin_nontemporal_store(s0, dst);
      __builtin_nontemporal_store(s1, dst + p.X);
    }
    __syncthreads();
  }
  __threadfence();
  if (threadIdx.x == 0)
    s_last = atomicAdd(&col_ticket[c], 1) == n_chunks - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  for (int xi = threadIdx.x; xi < p.X; xi += kBlock) {
    float s0 = 0.f, s1 = 0.f;
    for (int kb = 0; kb < n_chunks; kb += kColBatch) {
      float b0[kColBatch], b1[kColBatch];
#pragma unroll
      for (int k = 0; k < kColBatch; ++k) {
        const float* src =
            col_part + (((long long)c * n_chunks + min(kb + k, n_chunks - 1)) * 2) * p.X + xi;
        b0[k] = __builtin_nontemporal_load(src);
        b1[k] = __builtin_nontemporal_load(src + p.X);
      }
#pragma unroll
      for (int k = 0; k < kColBatch; ++k)
        if (kb + k < n_chunks) {
          s0 = s0 + b0[k];
          s1 = s1 + b1[k];
        }
    }
    colsum[c * p.X + xi] = s0 / p.n_col;
    colsum[(3 + c) * p.X + xi] = s1 / p.n_col;
  }
  if (threadIdx.x == 0) col_ticket[c] = 0;   // ready for the next step's launch
}

// ---------------------------------------------------------------------------
// Volumetric montage (BASELINE configs[4]: [3, n_tiles, 12, 12, 12] with the native
// target mesh as prev_fn, elastic_mesh_3d, FIRE, per-column drift means): the whole
// chunk of steps in ONE launch.
//
// The multi-launch step of such a mesh is advance_kernel + target_mesh_kernel +
// integrate_kernel<3> + drift_cols_kernel: four launches of 7-16 us each for 110 k nodes,
// every one of them a launch ramp, two or three dependent round trips and a tail.  Here
// the SAME blocks run the SAME code -- a workgroup of this launch is block b of every one
// of those kernels: 256 consecutive nodes, one node per thread (x, v, a of the node stay
// in registers from step to step), its row of partial sums, the chunk sums of
// drift_cols_kernel in that kernel's decomposition -- and what separated the launches
// becomes a grid barrier (one atomic per workgroup and phase; the launch is as wide as
// the chip holds at once, checked by the caller):
//   A  FIRE scalars from the partial rows of the previous step (every workgroup reduces
//      them, like advance_kernel), per-column drift means from the chunk sums, position
//      update of the own node                                               -- barrier 1
//   B  target of the own node, sampled from the neighbour tiles' positions (a wave lies
//      inside ONE tile: 64 divides the tile's node count), kept in registers
//   C  spring force from the neighbours' positions, pull towards the target, velocity
//      update, FIRE mixing; the block's row of partial sums                 -- barrier 2
//   D  (per-column drift) chunk sums of x and v, drift_cols_kernel's code    -- barrier 3
// Same float operations in the same order as the four kernels: x, v, a, the FIRE
// scalars and the statistics are bit-identical to the multi-launch path, which is the
// default; this form is opt-in (SFM_MESH_PERSIST3D=1) because it is SLOWER (r6, [3,64,12,12,12],
// per step): 247 us with agent-scope release / acquire fences around the barriers (a release
// writes back the XCD's whole L2, an acquire per wave invalidates it: 1728 invalidates per
// barrier), 129 us with written-through agent-scope stores instead of the release, 95 us with
// one invalidate per workgroup -- against 47 us for the four launches.  Phase ticks
// (SFM_P3_TIMING): A 10, B + C 20-25, D 6 us of work, and 14-30 us in EACH barrier: 432
// device-scope atomics on one address are served one after the other at the memory side,
// and after every invalidate the neighbours' positions, the partial rows and the chunk sums
// come from memory instead of the L2 the four kernels find them in.  Eight XCDs with
// private L2s make a grid barrier a cache flush; what a single launch needs is a tile per
// workgroup (a 12^3 tile's springs never leave the workgroup) with only the overlap strips
// and the sums crossing -- the layout of the in-plane persistent kernel, not this one.
// A barrier that times out leaves the caller's state untouched (persist_commit_kernel).
// ---------------------------------------------------------------------------
struct Persist3dArgs {
  float* x;               // working copies of the caller's state
  float* v;
  float* a;
  Scalars* scal;          // [2]; scal[0] holds the chunk's start values
  float* partials;        // [grid, kNP]
  float* colsum;          // [6][X]
  float* col_part;        // [3][col_chunks][2][X]
  unsigned* bar;          // grid barrier counter (zero at launch)
  int* abort;
  int num_iters;
  float cap0;
  int col_chunks, col_rows_per;
  SfmTargetMeshDesc t;
};

constexpr int kP3MaxX = 256;

template <typename T>
__device__ __forceinline__ void st_agent(T* ptr, T v) {
  __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool grid_barrier(unsigned* bar, int* abort, unsigned target,
                                             int* ok_lds) {
  // What other workgroups read was stored with agent-scope stores (written through: a
  // release fence at agent scope would write back this XCD's whole L2, 432 times per phase --
  // measured 80 us per barrier); they only have to have left the wave.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    long long spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(16);   // (~0.5 us: 432 pollers of one address are a hot spot)
      if (++spins > (1LL << 21) ||
          __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    *ok_lds = ok;
  }
  // drop what this CU / XCD cached of the others' data: ONE wave's invalidate serves the
  // workgroup (the vector L1 is the CU's, the L2 the XCD's; every wave doing it was 1728
  // invalidates per barrier)
  if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  asm volatile("" ::: "memory");
  return *ok_lds != 0;
}

template <bool COLS>
__global__ void __launch_bounds__(kBlock, 2)
mesh_persist3d_kernel(MeshParams p, Persist3dArgs q) {
  using namespace sfm_target;
  __shared__ float lds[kNP * kBlock];
  __shared__ float cs_lds[6 * kP3MaxX];
  __shared__ NbEntry s_e[kBlock / 64][4];
  __shared__ int ok_lds;
  __shared__ float col_lds[2][kBlock];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long n = blockIdx.x * (long long)kBlock + tid;
  const bool live = n < p.N;
  const long long nn = live ? n : p.N - 1;
  const SfmTargetMeshDesc& d = q.t;
  const int mz = d.mesh_shape[0], my = d.mesh_shape[1], mx = d.mesh_shape[2];
  const long long mn = (long long)mz * my * mx;
  // the tile of this wave (64 divides mn) and the node's place in it
  const long long wn = __builtin_amdgcn_readfirstlane(
      static_cast<int>(min(blockIdx.x * (long long)kBlock + 64LL * wave, p.N - 1) / mn));
  const int tile = static_cast<int>(wn);
  const int in_tile = static_cast<int>(nn - (long long)tile * mn);
  const int tx = in_tile % mx, ty = (in_tile / mx) % my, tz = in_tile / (mx * my);
  if (lane < 4) s_e[wave][lane] = make_entry(d, tile, lane);
  __syncthreads();
  const int xi = static_cast<int>(nn % p.X);
  unsigned phase = 0;
  int cur = 0;
#ifdef SFM_P3_TIMING
  long long tk[6] = {0, 0, 0, 0, 0, 0}, tc = wall_clock64();
#define P3TICK(i) { const long long tn = wall_clock64(); tk[i] += tn - tc; tc = tn; }
#else
#define P3TICK(i)
#endif
  for (int it = 0; it < q.num_iters; ++it) {
    const int pending = it > 0;
    // ---- A: scalars, drift means, position update (advance_kernel) ----------------------
    Scalars s, s_in;
    {
      // (written by workgroup 0 one step ago: read from the L2, not through a scalar load)
      const int* src = reinterpret_cast<const int*>(&q.scal[cur]);
      int* dst = reinterpret_cast<int*>(&s_in);
      for (int i = 0; i < static_cast<int>(sizeof(Scalars) / 4); ++i)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (pending) {
      update_scalars(s_in, q.partials, static_cast<int>(gridDim.x), p, lds, &s);
    } else {
      s = s_in;
      s.gate = 1.f;
      for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
    }
    if (blockIdx.x == 0 && tid == 0) {
      const int* src = reinterpret_cast<const int*>(&s);
      int* dst = reinterpret_cast<int*>(&q.scal[cur ^ 1]);
      for (int i = 0; i < static_cast<int>(sizeof(Scalars) / 4); ++i) st_agent(dst + i, src[i]);
    }
    cur ^= 1;
    if (COLS && pending) {
      // the chunk sums in chunk order, then the mean (drift_cols_kernel's last workgroup)
      for (int t = tid; t < 3 * p.X; t += kBlock) {
        const int c = t / p.X, xc = t - c * p.X;
        float s0 = 0.f, s1 = 0.f;
        for (int kb = 0; kb < q.col_chunks; kb += kColBatch) {
          float b0[kColBatch], b1[kColBatch];
#pragma unroll
          for (int k = 0; k < kColBatch; ++k) {
            const float* src = q.col_part +
                (((long long)c * q.col_chunks + min(kb + k, q.col_chunks - 1)) * 2) * p.X + xc;
            b0[k] = src[0];
            b1[k] = src[p.X];
          }
#pragma unroll
          for (int k = 0; k < kColBatch; ++k)
            if (kb + k < q.col_chunks) {
              s0 = s0 + b0[k];
              s1 = s1 + b1[k];
            }
        }
        cs_lds[c * p.X + xc] = s0 / p.n_col;
        cs_lds[(3 + c) * p.X + xc] = s1 / p.n_col;
        if (blockIdx.x == 0) {
          q.colsum[c * p.X + xc] = s0 / p.n_col;
          q.colsum[(3 + c) * p.X + xc] = s1 / p.n_col;
        }
      }
      __syncthreads();
    }
    {
      const float dt = s.dt;
      const float c2 = 0.5f * (dt * dt);
      if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float xv = q.x[c * p.N + n];
          float vv = q.v[c * p.N + n];
          if (pending) {
            vv = vv * s.gate;
            if (COLS) {
              xv = xv - cs_lds[c * p.X + xi];
              vv = vv - cs_lds[(3 + c) * p.X + xi] * s.gate;
            } else if (p.remove_drift) {
              xv = xv - s.mx[c];
              vv = vv - s.mv[c];
            }
            st_agent(&q.v[c * p.N + n], vv);
          }
          st_agent(&q.x[c * p.N + n], xv + (dt * vv + c2 * q.a[c * p.N + n]));
        }
      }
    }
    P3TICK(0)
    if (!grid_barrier(q.bar, q.abort, ++phase * gridDim.x, &ok_lds)) return;
    P3TICK(1)
    // ---- B: the target of the own node (target_mesh_kernel<0>) --------------------------
    float prev[3] = {NAN, NAN, NAN};
    {
      auto plane_of = [&](int c, int nb_i) {
        return plain(q.x + ((long long)c * d.n_tiles + nb_i) * mn);
      };
      target_node<0>(d, s_e[wave], tz, ty, tx, plane_of, &prev[0], &prev[1], &prev[2]);
    }
    // ---- C: force, velocity, partial sums (integrate_kernel<3>) -------------------------
    {
      const float dt = s.dt, alpha = s.alpha, cap = s.cap;
      const float hdtg = (0.5f * dt) * p.gamma;
      const float fact0 = 1.0f / (1.0f + hdtg);
      const float fact1 = 1.0f - hdtg;
      const float hdt = 0.5f * dt;
      float part[kNP];
      for (int i = 0; i < kNP; ++i) part[i] = 0.f;
      if (live) {
        float f[3], vn[3];
        node_force<3>(q.x, p, n, f);
        float a2 = 0.f, v2 = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float xv = q.x[c * p.N + n];
          f[c] = f[c] + prev_pull(xv, prev[c], p.neg_k0, cap);
          const float a_old = q.a[c * p.N + n];
          vn[c] = fact0 * (q.v[c * p.N + n] * fact1 + hdt * (a_old + f[c]));
          q.a[c * p.N + n] = f[c];
          a2 = a2 + f[c] * f[c];
          v2 = v2 + vn[c] * vn[c];
          part[0] = part[0] + f[c] * vn[c];
          part[1 + c] = part[1 + c] + xv;
        }
        const float a_norm = sqrtf(a2) + 1e-6f;
        const float v_norm = sqrtf(v2);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vn[c] = vn[c] + alpha * (f[c] / a_norm * v_norm - vn[c]);
          part[4 + c] = part[4 + c] + vn[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) st_agent(&q.v[c * p.N + n], vn[c]);
      }
      block_sum(part, 7, lds);
      if (tid == 0)
        for (int i = 0; i < kNP; ++i) st_agent(&q.partials[blockIdx.x * kNP + i], part[i]);
    }
    P3TICK(2)
    if (!grid_barrier(q.bar, q.abort, ++phase * gridDim.x, &ok_lds)) return;
    P3TICK(3)
    // ---- D: chunk sums of the columns (drift_cols_kernel, blocks (chunk, c)) ------------
    if (COLS) {
      const long long rows = p.N / p.X;
      const int G = kBlock / p.X;
      for (int vb = blockIdx.x; vb < 3 * q.col_chunks; vb += gridDim.x) {
        const int c = vb / q.col_chunks, chunk = vb - c * q.col_chunks;
        const long long r0 = (long long)chunk * q.col_rows_per;
        const long long r1 = min(rows, r0 + q.col_rows_per);
        const int g = tid / p.X, xc = tid % p.X;
        float acc[2] = {0.f, 0.f};
        if (g < G)
          for (long long rb = r0 + g; rb < r1; rb += (long long)G * kColBatch) {
            float bx[kColBatch], bv[kColBatch];
#pragma unroll
            for (int k = 0; k < kColBatch; ++k) {
              const long long r = min(rb + (long long)k * G, rows - 1);
              bx[k] = q.x[c * p.N + r * p.X + xc];
              bv[k] = q.v[c * p.N + r * p.X + xc];
            }
#pragma unroll
            for (int k = 0; k < kColBatch; ++k)
              if (rb + (long long)k * G < r1) {
                acc[0] = acc[0] + bx[k];
                acc[1] = acc[1] + bv[k];
              }
          }
        col_lds[0][tid] = acc[0];
        col_lds[1][tid] = acc[1];
        __syncthreads();
        if (tid < p.X) {
          float s0 = col_lds[0][tid], s1 = col_lds[1][tid];
          for (int k = 1; k < G; ++k) {
            s0 = s0 + col_lds[0][tid + k * p.X];
            s1 = s1 + col_lds[1][tid + k * p.X];
          }
          float* dst = q.col_part + (((long long)c * q.col_chunks + chunk) * 2) * p.X + tid;
          st_agent(dst, s0);
          st_agent(dst + p.X, s1);
        }
        __syncthreads();
      }
      P3TICK(4)
      if (!grid_barrier(q.bar, q.abort, ++phase * gridDim.x, &ok_lds)) return;
      P3TICK(5)
    }
  }
#ifdef SFM_P3_TIMING
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 200 || blockIdx.x == 431))
    printf("P3 block %d: per step (10 ns ticks) A %lld bar1 %lld BC %lld bar2 %lld D %lld bar3 %lld\n",
           blockIdx.x, tk[0] / q.num_iters, tk[1] / q.num_iters, tk[2] / q.num_iters,
           tk[3] / q.num_iters, tk[4] / q.num_iters, tk[5] / q.num_iters);
#endif
  // the drift means the last step leaves pending (read by finish_kernel)
  if (COLS && blockIdx.x == 0 && q.num_iters > 0) {
    for (int t = tid; t < 3 * p.X; t += kBlock) {
      const int c = t / p.X, xc = t - c * p.X;
      float s0 = 0.f, s1 = 0.f;
      for (int kb = 0; kb < q.col_chunks; kb += kColBatch) {
        float b0[kColBatch], b1[kColBatch];
#pragma unroll
        for (int k = 0; k < kColBatch; ++k) {
          const float* src = q.col_part +
              (((long long)c * q.col_chunks + min(kb + k, q.col_chunks - 1)) * 2) * p.X + xc;
          b0[k] = src[0];
          b1[k] = src[p.X];
        }
#pragma unroll
        for (int k = 0; k < kColBatch; ++k)
          if (kb + k < q.col_chunks) {
            s0 = s0 + b0[k];
            s1 = s1 + b1[k];
          }
      }
      q.colsum[c * p.X + xc] = s0 / p.n_col;
      q.colsum[(3 + c) * p.X + xc] = s1 / p.n_col;
    }
  }
}

// Applies the pending gate / drift of the last step and emits the per-block
// kinetic-energy partials (mesh.py:492-497, 584-586).
template <int C>
__global__ void __launch_bounds__(kBlock)
finish_kernel(float* __restrict__ x, float* __restrict__ v, MeshParams p,
              const Scalars* __restrict__ scal_in, Scalars* __restrict__ scal_out,
              const float* __restrict__ partials, int n_part_rows, int pending,
              float* __restrict__ stat_partials, const float* __restrict__ colsum) {
  __shared__ float lds[kNP * kBlock];
  Scalars s;
  s.gate = 1.f;
  for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
  if (p.fire) {
    if (pending == 1) {
      update_scalars(*scal_in, partials, n_part_rows, p, lds, &s);
    } else if (pending == 2) {
      s = *scal_in;
    } else {
      s = *scal_in;
      s.gate = 1.f;
      for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *scal_out = s;
  }
  float ek = 0.f, vmax2 = 0.f;
  for (long long n = blockIdx.x * (long long)kBlock + threadIdx.x; n < p.N;
       n += (long long)gridDim.x * kBlock) {
    float v2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float vv = v[c * p.N + n];
      if (p.fire && pending) {
        vv = vv * s.gate;
        if (p.drift_cols) {
          const int xi = static_cast<int>(n % p.X);
          x[c * p.N + n] = x[c * p.N + n] - colsum[c * p.X + xi];
          vv = vv - colsum[(3 + c) * p.X + xi] * s.gate;
        } else if (p.remove_drift) {
          x[c * p.N + n] = x[c * p.N + n] - s.mx[c];
          vv = vv - s.mv[c];
        }
        v[c * p.N + n] = vv;
      }
      v2 = v2 + vv * vv;
    }
    const int yrow = static_cast<int>((n / p.X) % p.Y);
    if (yrow >= p.own_y0 && yrow < p.own_y1) {
      ek = ek + v2;
      vmax2 = fmaxf(vmax2, v2);
    }
  }
  lds[threadIdx.x] = ek;
  lds[kBlock + threadIdx.x] = vmax2;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + st];
      lds[kBlock + threadIdx.x] =
          fmaxf(lds[kBlock + threadIdx.x], lds[kBlock + threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stat_partials[blockIdx.x * 2] = lds[0];
    stat_partials[blockIdx.x * 2 + 1] = lds[kBlock];
  }
}

__global__ void __launch_bounds__(kBlock)
stats_kernel(const float* __restrict__ stat_partials, int rows,
             float* __restrict__ out) {
  __shared__ float lds[2 * kBlock];
  float ek = 0.f, vm = 0.f;
  for (int r = threadIdx.x; r < rows; r += kBlock) {
    ek = ek + stat_partials[r * 2];
    vm = fmaxf(vm, stat_partials[r * 2 + 1]);
  }
  lds[threadIdx.x] = ek;
  lds[kBlock + threadIdx.x] = vm;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + st];
      lds[kBlock + threadIdx.x] =
          fmaxf(lds[kBlock + threadIdx.x], lds[kBlock + threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = lds[0];
    out[1] = sqrtf(lds[kBlock]);
  }
}


// ---------------------------------------------------------------------------
// Persistent single-launch integrator for in-plane meshes that fit the chip
// (one 32 x 32 node tile per workgroup, one workgroup per CU, one node per
// thread, node state in registers).
//
// The multi-launch path above pays two kernel boundaries per step (~20 us per
// step for the 205 x 205 mesh of an 8192^2 section).  Here the whole chunk of
// num_iters steps is one launch; per step every workgroup publishes
//   - (x, v, a) of its 124 perimeter nodes, and
//   - its partial sums (power, drift sums)
// as 8-byte {epoch, value} granules written with write-through agent-scope
// stores, and polls the granules of its 8 neighbour tiles and the partials of
// all workgroups with relaxed agent-scope loads: the data is its own flag, no
// cache fence and no grid barrier is needed (MI355X hand-off recipe R2).
// Because a neighbour's (x, v, a) is known, a tile advances its halo nodes
// itself, so there is ONE exchange per step.  Every workgroup reduces the
// partials of all workgroups in the same fixed order, so the FIRE scalars
// (dt, alpha, n_pos, cap, gate) are computed redundantly and identically.
// Granule slots are double buffered by epoch parity: a workgroup can be at
// most one step ahead of the slowest one because it needs everyone's partials.
// Spins are bounded; on a timeout the kernel raises an abort flag, writes
// nothing back, and the host re-runs the chunk on the multi-launch path.
// ---------------------------------------------------------------------------
constexpr int kNodeGran = 6;   // x0 x1 v0 v1 a0 a1
constexpr int kPartGran = 8;   // power, sum x[3], sum v[3], pad
constexpr int kMaxWg = 256;
constexpr int kSpinLimit = 1 << 21;
// Replicas of a workgroup's partial power (speculative kernel).  EVERY workgroup
// reads every partial every step; readers of one cache line are served one
// after the other on the memory side (~12 ns each): 168 readers of one granule
// took ~2 us, the longest wait of the step.  A workgroup writes its partial
// kPartRep times, 128 bytes apart, and reader w takes replica w % kPartRep.
constexpr int kPartRep = 8;
constexpr int kPartPitch = 16;   // u64 per (replica, workgroup): 8 step places + padding


template <int T>
struct Tile {
  static constexpr int kThreads = T * T;
  static constexpr int kWavesT = kThreads / 64;
  static constexpr int kPerim = 4 * T - 4;
  static constexpr int kHalo = 4 * T + 4;
  static constexpr int kSlot = kPerim * kNodeGran + kPartGran;  // granules
  static constexpr int kHaloPolls = (kHalo * kNodeGran + kThreads - 1) / kThreads;
  static constexpr int kPartPolls = (kMaxWg * 7 + kThreads - 1) / kThreads;
  static constexpr int kPolls = kHaloPolls + kPartPolls;
};

struct PersistArgs {
  u64* comm;            // [nWG][2][slot] granules, zeroed before launch
  u64* part;            // [kPartRep][kMaxWg][kPartPitch] partial-power replicas
  int* abort;           // set on timeout
  const Scalars* scal_in;
  Scalars* scal_out;
  float* stat_partials; // [nWG][2]
  int num_iters;
  float cap0;
  int nty, ntx, n_wg;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits
// for every outstanding global access of the wave (s_waitcnt vmcnt(0)): in the
// persistent kernels that put the round trip of the write-through granule
// stores and of loads issued ahead of their use (the partial powers) inside the
// step -- about half of what the step phase cost.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void put_granule(u64* g, unsigned epoch, float v) {
  __hip_atomic_store(g, (static_cast<u64>(epoch) << 32) | __float_as_uint(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Polls N granules concurrently: all loads of a round are in flight together,
// so a collect costs one round trip, not one per granule.
template <int N>
__device__ __forceinline__ bool poll_granules(const u64* const* g, float* const* d,
                                              unsigned epoch, int* abort) {
  unsigned need = 0;
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (g[i]) need |= 1u << i;
  for (int spin = 0; need && spin < kSpinLimit; ++spin) {
    u64 x[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
      x[i] = (need >> i) & 1u
                 ? __hip_atomic_load(g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                 : 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (((need >> i) & 1u) && static_cast<unsigned>(x[i] >> 32) == epoch) {
        *d[i] = __uint_as_float(static_cast<unsigned>(x[i]));
        need &= ~(1u << i);
      }
    if (!need) break;
    if ((spin & 1023) == 1023 &&
        __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return false;
    __builtin_amdgcn_s_sleep(1);
  }
  if (need) {
    __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  return true;
}

// The same for PAIRS of granules that lie side by side in one aligned 16-byte
// unit, fetched with one 16-byte agent-scope (sc1) buffer load each: half the
// requests of two 8-byte polls.  Each half carries its own tag and is accepted on
// its own terms, so nothing here assumes that a 16-byte access is single-copy
// atomic (8-byte halves are).  off[i]: byte offset of pair i in the exchange area
// or -1; d[i]: where the two values go.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u load_pair(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);   // aux 16: sc1
}
// takes the halves of `x` that carry `epoch`; returns the mask of halves taken
__device__ __forceinline__ unsigned take_pair(const v4u& x, unsigned epoch, float* d) {
  unsigned got = 0;
  if (x.y == epoch) {
    d[0] = __uint_as_float(x.x);
    got |= 1u;
  }
  if (x.w == epoch) {
    d[1] = __uint_as_float(x.z);
    got |= 2u;
  }
  return got;
}
template <int N>
__device__ __forceinline__ bool poll_pairs(__amdgpu_buffer_rsrc_t rs, const int* off,
                                           float* const* d, unsigned* have, unsigned epoch,
                                           int* abort) {
  bool need = false;
#pragma unroll
  for (int i = 0; i < N; ++i) need = need || (off[i] >= 0 && have[i] != 3u);
  for (int spin = 0; need && spin < kSpinLimit; ++spin) {
    v4u x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = load_pair(rs, off[i] >= 0 ? off[i] : 0);
    need = false;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (off[i] >= 0 && have[i] != 3u) {
        float t[2];
        const unsigned got = take_pair(x[i], epoch, t) & ~have[i];
        if (got & 1u) d[i][0] = t[0];
        if (got & 2u) d[i][1] = t[1];
        have[i] |= got;
        need = need || have[i] != 3u;
      }
    if (!need) break;
    if ((spin & 1023) == 1023 &&
        __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return false;
    __builtin_amdgcn_s_sleep(1);
  }
  if (need) {
    __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  return true;
}

// Fixed-order wave sum on the DPP network + row broadcasts (result valid in
// lane 63).
#define SFM_DPP_F32(x, ctrl, rmask, bc)                                         \
  __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, \
                                             0xf, bc))
__device__ __forceinline__ float wave_sum63(float v) {
  v = v + SFM_DPP_F32(v, 0x111, 0xf, true);   // row_shr:1
  v = v + SFM_DPP_F32(v, 0x112, 0xf, true);   // row_shr:2
  v = v + SFM_DPP_F32(v, 0x114, 0xf, true);   // row_shr:4
  v = v + SFM_DPP_F32(v, 0x118, 0xf, true);   // row_shr:8
  v = v + SFM_DPP_F32(v, 0x142, 0xa, false);  // row_bcast:15
  v = v + SFM_DPP_F32(v, 0x143, 0xc, false);  // row_bcast:31
  return v;
}

template <int T>
__device__ __forceinline__ int perim_index(int ly, int lx) {
  if (ly == 0) return lx;
  if (ly == T - 1) return T + lx;
  if (lx == 0) return 2 * T + (ly - 1);
  if (lx == T - 1) return 2 * T + (T - 2) + (ly - 1);
  return -1;
}

template <int T>
__device__ __forceinline__ void halo_coord(int h, int* hy, int* hx) {
  if (h < T + 2) { *hy = 0; *hx = h; }
  else if (h < 2 * (T + 2)) { *hy = T + 1; *hx = h - (T + 2); }
  else if (h < 2 * (T + 2) + T) { *hx = 0; *hy = h - 2 * (T + 2) + 1; }
  else { *hx = T + 1; *hy = h - 2 * (T + 2) - T + 1; }
}

template <int T>
__global__ void __launch_bounds__(T * T)
mesh_persist2d_kernel(MeshParams p, const float* __restrict__ xg,
                      const float* __restrict__ vg, float* __restrict__ xo,
                      float* __restrict__ vo, float* __restrict__ ao,
                      const float* __restrict__ prevg, PersistArgs q) {
  using TL = Tile<T>;
  constexpr int NT = TL::kThreads;
  __shared__ float xt[2][T + 2][T + 3];
  __shared__ float hval[TL::kHalo][kNodeGran];
  __shared__ float part_all[kMaxWg][kPartGran];
  __shared__ float wred[TL::kWavesT][kPartGran];
  __shared__ float sums[kPartGran];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ly = tid / T, lx = tid % T;
  const int wg = blockIdx.x;
  const int tx_i = wg % q.ntx;
  const int ty_i = (wg / q.ntx) % q.nty;
  const int slice = wg / (q.ntx * q.nty);
  const int yi = ty_i * T + ly, xi = tx_i * T + lx;
  const bool active = yi < p.Y && xi < p.X;
  const long long plane = (long long)p.Y * p.X;
  const long long n = slice * plane + (long long)yi * p.X + xi;
  const int pidx = perim_index<T>(ly, lx);
  const int nred = p.remove_drift ? 7 : 1;  // values exchanged per workgroup

  // Halo slot owned by this thread (threads < kHalo): position + source node.
  int hy = 0, hx = 0;
  long long h_n = -1;
  if (tid < TL::kHalo) {
    halo_coord<T>(tid, &hy, &hx);
    const int gy = ty_i * T + hy - 1, gx = tx_i * T + hx - 1;
    if (gy >= 0 && gy < p.Y && gx >= 0 && gx < p.X)
      h_n = slice * plane + (long long)gy * p.X + gx;
  }
  // Granules this thread fetches in a collect: halo (slot relative offsets).
  long long h_off[TL::kHaloPolls];
  float* h_dst[TL::kHaloPolls];
#pragma unroll
  for (int u = 0; u < TL::kHaloPolls; ++u) {
    h_off[u] = -1;
    h_dst[u] = nullptr;
    const int t = tid + u * NT;
    if (t < TL::kHalo * kNodeGran) {
      const int h = t / kNodeGran, j = t - h * kNodeGran;
      int qy, qx;
      halo_coord<T>(h, &qy, &qx);
      const int gy = ty_i * T + qy - 1, gx = tx_i * T + qx - 1;
      if (gy >= 0 && gy < p.Y && gx >= 0 && gx < p.X) {
        const int oty = gy / T, otx = gx / T;
        const int owg = (slice * q.nty + oty) * q.ntx + otx;
        h_off[u] = (long long)owg * 2 * TL::kSlot +
                   perim_index<T>(gy - oty * T, gx - otx * T) * kNodeGran + j;
        h_dst[u] = &hval[h][j];
      }
    }
  }

  float x0 = 0.f, x1 = 0.f, v0 = 0.f, v1 = 0.f, a0 = 0.f, a1 = 0.f;
  float pr0 = 0.f, pr1 = 0.f;
  if (active) {
    x0 = xg[n];
    x1 = xg[p.N + n];
    v0 = vg[n];
    v1 = vg[p.N + n];
    if (p.has_prev) {
      pr0 = prevg[n];
      pr1 = prevg[p.N + n];
    }
  }
  Scalars s = *q.scal_in;
  s.gate = 1.f;
  for (int c = 0; c < 3; ++c) s.mx[c] = s.mv[c] = 0.f;
  if (!p.fire) s.dt = p.vv_dt;
  const float fixed_cap = q.cap0;
  float l0[4];
#pragma unroll
  for (int L = 0; L < 4; ++L) l0[L] = vec_len(p.rest[L], 2);

  auto tile_force = [&](float* out) {
    // identical operation order to node_force<2> with order2d; the four link
    // families of build_params unrolled with compile-time directions
    const float s0 = xt[0][ly + 1][lx + 1], s1 = xt[1][ly + 1][lx + 1];
    float acc0 = 0.f, acc1 = 0.f;
    // (branch free, see node_force_tile2d)
#define SFM_FAR(L, DX, DY)                                                          \
    {                                                                               \
      const bool ok = xi - (DX) >= 0 && xi - (DX) < p.X && yi - (DY) >= 0 &&        \
                      yi - (DY) < p.Y;                                              \
      float f[2];                                                                   \
      spring_xy<DX, DY>(s0 - xt[0][ly + 1 - (DY)][lx + 1 - (DX)] + p.rest[L][0],     \
                        s1 - xt[1][ly + 1 - (DY)][lx + 1 - (DX)] + p.rest[L][1],     \
                        l0[L], p.neg_k[L], p.prefer, f);                            \
      acc0 = acc0 + (ok ? f[0] : 0.f);                                              \
      acc1 = acc1 + (ok ? f[1] : 0.f);                                              \
    }
#define SFM_NEAR(L, DX, DY)                                                         \
    {                                                                               \
      const bool ok = xi + (DX) >= 0 && xi + (DX) < p.X && yi + (DY) >= 0 &&        \
                      yi + (DY) < p.Y;                                              \
      float f[2];                                                                   \
      spring_xy<DX, DY>(xt[0][ly + 1 + (DY)][lx + 1 + (DX)] - s0 + p.rest[L][0],     \
                        xt[1][ly + 1 + (DY)][lx + 1 + (DX)] - s1 + p.rest[L][1],     \
                        l0[L], p.neg_k[L], p.prefer, f);                            \
      acc0 = acc0 - (ok ? f[0] : 0.f);                                              \
      acc1 = acc1 - (ok ? f[1] : 0.f);                                              \
    }
    SFM_FAR(0, 1, 0) SFM_FAR(1, 0, 1) SFM_FAR(2, 1, 1) SFM_FAR(3, -1, 1)
    SFM_NEAR(0, 1, 0) SFM_NEAR(1, 0, 1) SFM_NEAR(2, 1, 1) SFM_NEAR(3, -1, 1)
#undef SFM_FAR
#undef SFM_NEAR
    out[0] = acc0;
    out[1] = acc1;
  };

  // ---- a = F(x) + pull at the initial positions ------------------------------
  xt[0][ly + 1][lx + 1] = x0;
  xt[1][ly + 1][lx + 1] = x1;
  if (h_n >= 0) {
    xt[0][hy][hx] = xg[h_n];
    xt[1][hy][hx] = xg[p.N + h_n];
  }
  __syncthreads();
  if (active) {
    float f[2];
    tile_force(f);
    if (p.has_prev) {
      f[0] = f[0] + prev_pull(x0, pr0, p.neg_k0, p.fire ? s.cap : fixed_cap);
      f[1] = f[1] + prev_pull(x1, pr1, p.neg_k0, p.fire ? s.cap : fixed_cap);
    }
    a0 = f[0];
    a1 = f[1];
  }
  float my_part = 0.f;  // threads < nred: this workgroup's partial sum #tid

#ifdef SFM_MESH_TIMING
  long long mt[6] = {0, 0, 0, 0, 0, 0}, mtc = clock64();
#define MTICK(i) { const long long tn = clock64(); mt[i] += tn - mtc; mtc = tn; }
#else
#define MTICK(i)
#endif
  bool ok = true;
  for (int k = 1; k <= q.num_iters + 1; ++k) {
    MTICK(5)
    const unsigned epoch = static_cast<unsigned>(k);
    const bool last = k == q.num_iters + 1;
    const long long slot_off = (long long)(k & 1) * TL::kSlot;
    u64* my_slot = q.comm + (long long)wg * 2 * TL::kSlot + slot_off;
    // ---- publish the state after k - 1 steps ----------------------------------
    if (!last && active && pidx >= 0) {
      u64* g = my_slot + pidx * kNodeGran;
      put_granule(g + 0, epoch, x0);
      put_granule(g + 1, epoch, x1);
      put_granule(g + 2, epoch, v0);
      put_granule(g + 3, epoch, v1);
      put_granule(g + 4, epoch, a0);
      put_granule(g + 5, epoch, a1);
    }
    const bool reduce_now = p.fire && k > 1;
    if (reduce_now && tid < nred)
      put_granule(my_slot + TL::kPerim * kNodeGran + tid, epoch, my_part);
    // ---- collect ------------------------------------------------------------------
    const u64* g[TL::kPolls];
    float* d[TL::kPolls];
#pragma unroll
    for (int u = 0; u < TL::kHaloPolls; ++u) {
      const bool want = !last && h_off[u] >= 0;
      g[u] = want ? q.comm + h_off[u] + slot_off : nullptr;
      d[u] = h_dst[u];
    }
#pragma unroll
    for (int u = 0; u < TL::kPartPolls; ++u) {
      const int t = tid + u * NT;
      const int w2 = t / nred, j = t - w2 * nred;
      const bool want = reduce_now && w2 < q.n_wg && w2 != wg;
      g[TL::kHaloPolls + u] =
          want ? q.comm + (long long)w2 * 2 * TL::kSlot + slot_off +
                     TL::kPerim * kNodeGran + j
               : nullptr;
      d[TL::kHaloPolls + u] = want ? &part_all[w2][j] : nullptr;
    }
    if (reduce_now && tid < nred) part_all[wg][tid] = my_part;  // own: no round trip
    const bool mine_ok = poll_granules<TL::kPolls>(g, d, epoch, q.abort);
    if (!__syncthreads_and(mine_ok ? 1 : 0)) {
      ok = false;
      break;
    }
    MTICK(0)
    // ---- FIRE scalars from the partials of step k - 1 -------------------------------
    if (reduce_now) {
      // value i is summed over workgroups by wave i mod #waves (fixed order:
      // strided per lane, then the DPP tree): identical in every workgroup.
      for (int i = wave; i < nred; i += TL::kWavesT) {
        float t = 0.f;
        for (int w2 = lane; w2 < q.n_wg; w2 += 64) t = t + part_all[w2][i];
        t = wave_sum63(t);
        if (lane == 63) sums[i] = t;
      }
      __syncthreads();
      const float power = sums[0];
      Scalars t = s;
      const bool downhill = power >= 0.f;
      t.n_pos = downhill ? s.n_pos + 1 : 0;
      if (downhill) {
        if (t.n_pos > p.n_min) {
          t.dt = fminf(s.dt * p.f_inc, p.dt_cap);
          t.alpha = s.alpha * p.f_alpha;
        }
        if (t.n_pos > 0 && (t.n_pos % p.cap_every) == 0) t.cap = p.cap_scale * s.cap;
      } else {
        t.dt = s.dt * p.f_dec;
        t.alpha = p.alpha0;
      }
      t.cap = fminf(t.cap, p.final_cap);
      t.gate = downhill ? 1.f : 0.f;
      for (int c = 0; c < 3; ++c) {
        t.mx[c] = p.remove_drift ? sums[1 + c] / p.n_f : 0.f;
        t.mv[c] = p.remove_drift ? (sums[4 + c] / p.n_f) * t.gate : 0.f;
      }
      s = t;
      // pending gate / drift of step k - 1 on the node's own state
      v0 = v0 * s.gate;
      v1 = v1 * s.gate;
      if (p.remove_drift) {
        x0 = x0 - s.mx[0];
        x1 = x1 - s.mx[1];
        v0 = v0 - s.mv[0];
        v1 = v1 - s.mv[1];
      }
    }
    MTICK(1)
    if (last) break;

    // ---- advance: x += dt v + dt^2/2 a for own and halo nodes --------------------
    const float dt = s.dt;
    const float c2 = 0.5f * (dt * dt);
    x0 = x0 + (dt * v0 + c2 * a0);
    x1 = x1 + (dt * v1 + c2 * a1);
    xt[0][ly + 1][lx + 1] = x0;
    xt[1][ly + 1][lx + 1] = x1;
    if (h_n >= 0) {
      float hx0 = hval[tid][0], hx1 = hval[tid][1];
      float hv0 = hval[tid][2], hv1 = hval[tid][3];
      if (reduce_now) {
        hv0 = hv0 * s.gate;
        hv1 = hv1 * s.gate;
        if (p.remove_drift) {
          hx0 = hx0 - s.mx[0];
          hx1 = hx1 - s.mx[1];
          hv0 = hv0 - s.mv[0];
          hv1 = hv1 - s.mv[1];
        }
      }
      xt[0][hy][hx] = hx0 + (dt * hv0 + c2 * hval[tid][4]);
      xt[1][hy][hx] = hx1 + (dt * hv1 + c2 * hval[tid][5]);
    }
    __syncthreads();
    MTICK(2)

    // ---- integrate ------------------------------------------------------------------
    float part[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
      const float cap = p.fire ? s.cap : fixed_cap;
      const float hdtg = (0.5f * dt) * p.gamma;
      const float fact0 = 1.0f / (1.0f + hdtg);
      const float fact1 = 1.0f - hdtg;
      const float hdt = 0.5f * dt;
      float f[2];
      tile_force(f);
      if (p.has_prev) {
        f[0] = f[0] + prev_pull(x0, pr0, p.neg_k0, cap);
        f[1] = f[1] + prev_pull(x1, pr1, p.neg_k0, cap);
      }
      float n0 = fact0 * (v0 * fact1 + hdt * (a0 + f[0]));
      float n1 = fact0 * (v1 * fact1 + hdt * (a1 + f[1]));
      a0 = f[0];
      a1 = f[1];
      if (p.fire) {
        float a2 = 0.f, v2 = 0.f, pw = 0.f;
        a2 = a2 + f[0] * f[0];
        v2 = v2 + n0 * n0;
        pw = pw + f[0] * n0;
        a2 = a2 + f[1] * f[1];
        v2 = v2 + n1 * n1;
        pw = pw + f[1] * n1;
        part[0] = pw;
        part[1] = x0;
        part[2] = x1;
        const float a_norm = sqrtf(a2) + 1e-6f;
        const float v_norm = sqrtf(v2);
        n0 = n0 + s.alpha * (f[0] / a_norm * v_norm - n0);
        n1 = n1 + s.alpha * (f[1] / a_norm * v_norm - n1);
        part[4] = n0;
        part[5] = n1;
      }
      v0 = n0;
      v1 = n1;
    }
    MTICK(3)
    if (p.fire) {
      // block sums in a fixed order: DPP tree per wave, then waves in order
      for (int i = 0; i < nred; ++i) {
        const float t = wave_sum63(part[i]);
        if (lane == 63) wred[wave][i] = t;
      }
      __syncthreads();  // also: everyone is done with xt / hval / part_all
      if (tid < nred) {
        float t = 0.f;
        for (int w2 = 0; w2 < TL::kWavesT; ++w2) t = t + wred[w2][tid];
        my_part = t;
      }
    } else {
      __syncthreads();
    }
    MTICK(4)
  }
#ifdef SFM_MESH_TIMING
  if (wg == 0 && tid == 0)
    printf("MESH wg0 per step: exchange %lld scalars %lld advance %lld force %lld sums %lld other %lld\n",
           mt[0] / q.num_iters, mt[1] / q.num_iters, mt[2] / q.num_iters, mt[3] / q.num_iters,
           mt[4] / q.num_iters, mt[5] / q.num_iters);
#endif

  if (!ok) return;  // timed out (the abort flag is set)
  // ---- write back, chunk statistics ------------------------------------------------
  // The result goes to a staging set (xo, vo, ao); persist_commit_kernel copies
  // it over the caller's state only when NO workgroup raised the abort flag, so
  // a timeout leaves the caller's x, v, a untouched as a whole.
  float ek = 0.f, vm2 = 0.f;
  if (active) {
    xo[n] = x0;
    xo[p.N + n] = x1;
    vo[n] = v0;
    vo[p.N + n] = v1;
    ao[n] = a0;
    ao[p.N + n] = a1;
    ek = v0 * v0 + v1 * v1;
    vm2 = ek;
  }
#pragma unroll
  for (int dd = 32; dd > 0; dd >>= 1) {
    ek = ek + __shfl_xor(ek, dd, 64);
    vm2 = fmaxf(vm2, __shfl_xor(vm2, dd, 64));
  }
  __syncthreads();
  if (lane == 0) {
    wred[wave][0] = ek;
    wred[wave][1] = vm2;
  }
  __syncthreads();
  if (tid == 0) {
    float e = 0.f, m = 0.f;
    for (int w2 = 0; w2 < TL::kWavesT; ++w2) {
      e = e + wred[w2][0];
      m = fmaxf(m, wred[w2][1]);
    }
    q.stat_partials[wg * 2] = e;
    q.stat_partials[wg * 2 + 1] = m;
    if (wg == 0) *q.scal_out = s;
  }
}

// ---------------------------------------------------------------------------
// Speculative FIRE variant of the persistent kernel (no drift removal).
//
// In mesh_persist2d_kernel every step waits for the partial `power` sums of ALL
// workgroups before it can advance (FIRE's dt, alpha, cap and the velocity gate
// of step k depend on the sign of power(k-1), mesh.py:455-492): a 169-way
// all-gather on the critical path of every step, ~60 % of the step time.  But
// power >= 0 (downhill) is by far the common case, and the update is a pure
// function of (previous scalars, sign).  So this kernel
//   1. publishes its halo and its partial of step k-1 and waits for the HALO
//      of its 8 neighbours only (one-to-one hand-offs),
//   2. runs step k with the scalars of the downhill branch,
//   3. then collects the partials of step k-1 -- published a whole step ago --
//      reduces them in the same fixed order as always, and
//   4. if the power was negative after all, restores the state saved before
//      the step and redoes it with the uphill scalars (dt *= f_dec, v = 0).
// Every workgroup sees the same sums, takes the same decision and redoes the
// same steps: the trajectory is bit-identical to the non-speculative kernel.
//
// T = 16 (r4): every spring ONCE.  The step is a chain of dependent latencies on
// one wave per SIMD (2.0 of the 4.6 us of a step are the force evaluation), and
// a node evaluated each of its eight springs itself although the far-side term
// of node n for link L is the near-side term of node n - dir(L), bit for bit
// (see integrate_shared2d_kernel).  Now a node evaluates its four near-side
// springs only and leaves them in LDS; the 94 near-side springs of HALO nodes
// that tile nodes need (left column for links 0 and 2, top row for 1, 2, 3,
// right column for 3) are evaluated by two extra waves, one spring per lane,
// with the run-time-direction form of the same spring; after a barrier every
// node adds far sides (from LDS) and near sides (its registers) in the order it
// always did.  Same floats in the same order: bit-identical trajectories.
// ---------------------------------------------------------------------------
template <int T>
constexpr int spec_threads() { return T == 16 ? T * T + 128 : T * T; }

template <int T>
__global__ void __launch_bounds__(spec_threads<T>())
mesh_persist2d_spec_kernel(MeshParams p, const float* __restrict__ xg,
                           const float* __restrict__ vg, float* __restrict__ xo,
                           float* __restrict__ vo, float* __restrict__ ao,
                           const float* __restrict__ prevg, PersistArgs q) {
  using TL = Tile<T>;
  constexpr bool SH = T == 16;            // every spring once, two halo waves
  constexpr int NM = TL::kThreads;        // threads that own a node
  constexpr int NT = spec_threads<T>();   // all threads (polls, reductions)
  // What a neighbour needs of a perimeter node is its POSITION after the next
  // position update, and that update is a function of this node's (x, v, a) and of
  // one of two sets of scalars everybody knows: the downhill ones (speculative
  // step, gate 1) or the uphill ones (redone step, gate 0).  So the owner publishes
  // the two candidate positions per component -- the expression the reader used to
  // evaluate, operation for operation: the same bits -- instead of (x, v, a): four
  // granules per node, of which a reader polls TWO per step (the downhill pair; the
  // other pair only when a step is redone).  Measured sensitivity before the change
  // (every granule published and polled twice, same trajectory): 3.8 -> 4.7 us per
  // step -- the hand-off is priced by the granules in flight, not only by its
  // round trip.
  constexpr int kGranS = 4;   // A0 A1 (downhill) B0 B1 (uphill)
  constexpr int kSlotS = TL::kPerim * kGranS + kPartGran;
  static_assert(kSlotS <= Tile<32>::kSlot, "exchange area is sized for Tile<32>");
  constexpr int kHaloPollsS = (TL::kHalo + NT - 1) / NT;   // one 16-byte pair per halo node
  constexpr int kPartPolls1 = (kMaxWg + NT - 1) / NT;  // one value per workgroup
  __shared__ float xt[2][T + 2][T + 3];
  // near-side spring forces of every frame node, by link and component (SH)
  __shared__ float nf[SH ? 4 : 1][2][SH ? T + 2 : 1][SH ? T + 3 : 1];
  __shared__ float hval[TL::kHalo][kGranS];
  __shared__ float part_all[kMaxWg];
  __shared__ float wred[TL::kWavesT];
  __shared__ float s_power;
  __shared__ int s_fail;   // a poll of this workgroup timed out (wg_and)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_fail = 0;
  // "did every thread's poll succeed?" through LDS: __syncthreads_and() would
  // also wait for the global loads that are in flight on purpose
  auto wg_and = [&](bool mine) {
    if (!mine) s_fail = 1;
    lds_barrier();
    return s_fail == 0;
  };
  const bool owner = tid < NM;            // (the halo waves own no node)
  const int ly = owner ? tid / T : 0, lx = owner ? tid % T : 0;
  const int wg = blockIdx.x;
  const int tx_i = wg % q.ntx;
  const int ty_i = (wg / q.ntx) % q.nty;
  const int slice = wg / (q.ntx * q.nty);
  const int yi = ty_i * T + ly, xi = tx_i * T + lx;
  const bool active = owner && yi < p.Y && xi < p.X;
  const long long plane = (long long)p.Y * p.X;
  const long long n = slice * plane + (long long)yi * p.X + xi;
  const int pidx = owner ? perim_index<T>(ly, lx) : -1;

  int hy = 0, hx = 0;
  long long h_n = -1;
  if (tid < TL::kHalo) {
    halo_coord<T>(tid, &hy, &hx);
    const int gy = ty_i * T + hy - 1, gx = tx_i * T + hx - 1;
    if (gy >= 0 && gy < p.Y && gx >= 0 && gx < p.X)
      h_n = slice * plane + (long long)gy * p.X + gx;
  }
  // byte offset (slot parity 0) of the downhill pair of halo node tid + u NT in the
  // exchange area, or -1; the uphill pair follows it
  const __amdgpu_buffer_rsrc_t comm_rs = __builtin_amdgcn_make_buffer_rsrc(
      q.comm, 0, static_cast<int>((long long)q.n_wg * 2 * kSlotS * sizeof(u64)), 0x00020000);
  int h_off[kHaloPollsS];
  float* h_dst[kHaloPollsS];
#pragma unroll
  for (int u = 0; u < kHaloPollsS; ++u) {
    h_off[u] = -1;
    h_dst[u] = nullptr;
    const int h = tid + u * NT;
    if (h < TL::kHalo) {
      int qy, qx;
      halo_coord<T>(h, &qy, &qx);
      const int gy = ty_i * T + qy - 1, gx = tx_i * T + qx - 1;
      if (gy >= 0 && gy < p.Y && gx >= 0 && gx < p.X) {
        const int oty = gy / T, otx = gx / T;
        const int owg = (slice * q.nty + oty) * q.ntx + otx;
        h_off[u] = (owg * 2 * kSlotS + perim_index<T>(gy - oty * T, gx - otx * T) * kGranS) *
                   static_cast<int>(sizeof(u64));
        h_dst[u] = &hval[h][0];
      }
    }
  }

  float x0 = 0.f, x1 = 0.f, v0 = 0.f, v1 = 0.f, a0 = 0.f, a1 = 0.f;
  float pr0 = 0.f, pr1 = 0.f;
  if (active) {
    x0 = xg[n];
    x1 = xg[p.N + n];
    v0 = vg[n];
    v1 = vg[p.N + n];
    if (p.has_prev) {
      pr0 = prevg[n];
      pr1 = prevg[p.N + n];
    }
  }
  Scalars s = *q.scal_in;
  s.gate = 1.f;
  float l0[4];
#pragma unroll
  for (int L = 0; L < 4; ++L) l0[L] = vec_len(p.rest[L], 2);

  // Halo waves (SH): lane -> (frame node, link) of one near-side spring that a
  // tile node needs as its far side.  Frame coordinates: tile nodes at 1 .. T.
  //   link 0 (1, 0):  (1..T, 0)                              T springs
  //   link 1 (0, 1):  (0, 1..T)                              T
  //   link 2 (1, 1):  (0, 0..T-1), (1..T-1, 0)               2 T - 1
  //   link 3 (-1, 1): (0, 2..T+1), (1..T-1, T+1)             2 T - 1
  int hs_y = 0, hs_x = 0, hs_L = -1, hs_dx = 0, hs_dy = 0;
  if (SH && !owner) {
    const int i = tid - NM;
    if (i < T) { hs_L = 0; hs_y = 1 + i; hs_x = 0; }
    else if (i < 2 * T) { hs_L = 1; hs_y = 0; hs_x = 1 + (i - T); }
    else if (i < 3 * T) { hs_L = 2; hs_y = 0; hs_x = i - 2 * T; }
    else if (i < 4 * T - 1) { hs_L = 2; hs_y = 1 + (i - 3 * T); hs_x = 0; }
    else if (i < 5 * T - 1) { hs_L = 3; hs_y = 0; hs_x = 2 + (i - (4 * T - 1)); }
    else if (i < 6 * T - 2) { hs_L = 3; hs_y = 1 + (i - (5 * T - 1)); hs_x = T + 1; }
  }
  // the lane's link constants in registers (a run-time index into the kernel
  // argument would be a global load inside every step)
  float hs_r0 = 0.f, hs_r1 = 0.f, hs_l0 = 0.f, hs_k = 0.f;
#pragma unroll
  for (int L = 0; L < 4; ++L)
    if (hs_L == L) {
      hs_dx = p.dir[L][0];
      hs_dy = p.dir[L][1];
      hs_r0 = p.rest[L][0];
      hs_r1 = p.rest[L][1];
      hs_l0 = l0[L];
      hs_k = p.neg_k[L];
    }
  static_assert(!SH || 6 * T - 2 <= NT - NM, "one halo spring per halo-wave lane");

  // The partial powers of the previous step, one per workgroup: requested at the
  // head of a step (`issue_early`; they left their workgroups most of an
  // iteration ago) and taken in at its end (`take_early`), BEFORE the next
  // halo is requested: the verification then touches LDS only and no wait of
  // this iteration covers more than one round trip.  Stragglers are polled for
  // in the verification.
  const u64* pg[kPartPolls1];
  float* pd[kPartPolls1];
  u64 early[kPartPolls1];
  bool early_pending = false;
  unsigned early_tag = 0;
  // The neighbours' state after this step, requested right behind this
  // workgroup's own publication (they publish at about the same time, and a
  // request needs half a round trip to get there): the other half of the round
  // trip hides behind the verification instead of opening the next step.
  // (a second request behind the verification's polls bought nothing once a step's
  // hand-off was two granules per node: 3.13 us per step with either one, 3.38 with
  // neither)
  v4u hearly[kHaloPollsS];
#pragma unroll
  for (int u = 0; u < kHaloPollsS; ++u) hearly[u] = v4u{0, 0, 0, 0};
  auto request_halo = [&](int step, v4u* dst) {
    const int next_off = ((step + 1) & 1) * kSlotS * static_cast<int>(sizeof(u64));
#pragma unroll
    for (int u = 0; u < kHaloPollsS; ++u)
      dst[u] = load_pair(comm_rs, h_off[u] >= 0 ? h_off[u] + next_off : 0);
  };
  auto issue_early = [&]() {
    if (!early_pending) return;
    early_pending = false;
    // (unconditional loads: a load under a per-lane condition gets its own basic
    // block and is waited for on the spot; lanes without a granule read the
    // first word of the exchange area and ignore it)
#pragma unroll
    for (int u = 0; u < kPartPolls1; ++u)
      early[u] = __hip_atomic_load(pg[u] ? pg[u] : q.comm, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
  };
  auto take_early = [&]() {
#pragma unroll
    for (int u = 0; u < kPartPolls1; ++u)
      if (pg[u] && static_cast<unsigned>(early[u] >> 32) == early_tag) {
        *pd[u] = __uint_as_float(static_cast<unsigned>(early[u]));
        pg[u] = nullptr;  // arrived: nothing left to poll
      }
  };

  // Net spring force on this thread's node from the positions in xt.  Contains
  // a barrier when SH: called by ALL threads.
  auto tile_force = [&](float* out) {
    const float s0 = xt[0][ly + 1][lx + 1], s1 = xt[1][ly + 1][lx + 1];
    float acc0 = 0.f, acc1 = 0.f;
    if constexpr (SH) {
      float nr[4][2];
      if (owner) {
#define SFM_NEAR_EVAL(L, DX, DY)                                                    \
        spring_xy<DX, DY>(xt[0][ly + 1 + (DY)][lx + 1 + (DX)] - s0 + p.rest[L][0],   \
                          xt[1][ly + 1 + (DY)][lx + 1 + (DX)] - s1 + p.rest[L][1],   \
                          l0[L], p.neg_k[L], p.prefer, nr[L]);                      \
        nf[L][0][ly + 1][lx + 1] = nr[L][0];                                        \
        nf[L][1][ly + 1][lx + 1] = nr[L][1];
        SFM_NEAR_EVAL(0, 1, 0) SFM_NEAR_EVAL(1, 0, 1) SFM_NEAR_EVAL(2, 1, 1)
        SFM_NEAR_EVAL(3, -1, 1)
#undef SFM_NEAR_EVAL
      } else if (hs_L >= 0) {
        float f[2];
        spring_xy_rt(xt[0][hs_y + hs_dy][hs_x + hs_dx] - xt[0][hs_y][hs_x] + hs_r0,
                     xt[1][hs_y + hs_dy][hs_x + hs_dx] - xt[1][hs_y][hs_x] + hs_r1,
                     hs_l0, hs_k, p.prefer, hs_dx, hs_dy, f);
        nf[hs_L][0][hs_y][hs_x] = f[0];
        nf[hs_L][1][hs_y][hs_x] = f[1];
      }
      lds_barrier();
#define SFM_FAR_ADD(L, DX, DY)                                                      \
      {                                                                             \
        const bool ok = xi - (DX) >= 0 && xi - (DX) < p.X && yi - (DY) >= 0 &&      \
                        yi - (DY) < p.Y;                                            \
        const float f0 = nf[L][0][ly + 1 - (DY)][lx + 1 - (DX)];                    \
        const float f1 = nf[L][1][ly + 1 - (DY)][lx + 1 - (DX)];                    \
        acc0 = acc0 + (ok ? f0 : 0.f);                                              \
        acc1 = acc1 + (ok ? f1 : 0.f);                                              \
      }
#define SFM_NEAR_SUB(L, DX, DY)                                                     \
      {                                                                             \
        const bool ok = xi + (DX) >= 0 && xi + (DX) < p.X && yi + (DY) >= 0 &&      \
                        yi + (DY) < p.Y;                                            \
        acc0 = acc0 - (ok ? nr[L][0] : 0.f);                                        \
        acc1 = acc1 - (ok ? nr[L][1] : 0.f);                                        \
      }
      if (owner) {
        SFM_FAR_ADD(0, 1, 0) SFM_FAR_ADD(1, 0, 1) SFM_FAR_ADD(2, 1, 1) SFM_FAR_ADD(3, -1, 1)
        SFM_NEAR_SUB(0, 1, 0) SFM_NEAR_SUB(1, 0, 1) SFM_NEAR_SUB(2, 1, 1) SFM_NEAR_SUB(3, -1, 1)
      }
#undef SFM_FAR_ADD
#undef SFM_NEAR_SUB
    } else {
#define SFM_FAR(L, DX, DY)                                                          \
    {                                                                               \
      const bool ok = xi - (DX) >= 0 && xi - (DX) < p.X && yi - (DY) >= 0 &&        \
                      yi - (DY) < p.Y;                                              \
      float f[2];                                                                   \
      spring_xy<DX, DY>(s0 - xt[0][ly + 1 - (DY)][lx + 1 - (DX)] + p.rest[L][0],     \
                        s1 - xt[1][ly + 1 - (DY)][lx + 1 - (DX)] + p.rest[L][1],     \
                        l0[L], p.neg_k[L], p.prefer, f);                            \
      acc0 = acc0 + (ok ? f[0] : 0.f);                                              \
      acc1 = acc1 + (ok ? f[1] : 0.f);                                              \
    }
#define SFM_NEAR(L, DX, DY)                                                         \
    {                                                                               \
      const bool ok = xi + (DX) >= 0 && xi + (DX) < p.X && yi + (DY) >= 0 &&        \
                      yi + (DY) < p.Y;                                              \
      float f[2];                                                                   \
      spring_xy<DX, DY>(xt[0][ly + 1 + (DY)][lx + 1 + (DX)] - s0 + p.rest[L][0],     \
                        xt[1][ly + 1 + (DY)][lx + 1 + (DX)] - s1 + p.rest[L][1],     \
                        l0[L], p.neg_k[L], p.prefer, f);                            \
      acc0 = acc0 - (ok ? f[0] : 0.f);                                              \
      acc1 = acc1 - (ok ? f[1] : 0.f);                                              \
    }
    SFM_FAR(0, 1, 0) SFM_FAR(1, 0, 1) SFM_FAR(2, 1, 1) SFM_FAR(3, -1, 1)
    SFM_NEAR(0, 1, 0) SFM_NEAR(1, 0, 1) SFM_NEAR(2, 1, 1) SFM_NEAR(3, -1, 1)
#undef SFM_FAR
#undef SFM_NEAR
    }
    out[0] = acc0;
    out[1] = acc1;
  };

  // FIRE scalar update of mesh.py:459-490 for a known sign of the power.
  auto next_scalars = [&](const Scalars& in, bool downhill) -> Scalars {
    Scalars t = in;
    t.n_pos = downhill ? in.n_pos + 1 : 0;
    if (downhill) {
      if (t.n_pos > p.n_min) {
        t.dt = fminf(in.dt * p.f_inc, p.dt_cap);
        t.alpha = in.alpha * p.f_alpha;
      }
      if (t.n_pos > 0 && (t.n_pos % p.cap_every) == 0) t.cap = p.cap_scale * in.cap;
    } else {
      t.dt = in.dt * p.f_dec;
      t.alpha = p.alpha0;
    }
    t.cap = fminf(t.cap, p.final_cap);
    t.gate = downhill ? 1.f : 0.f;
    return t;
  };

  // a = F(x) + pull at the initial positions
  if (owner) {
    xt[0][ly + 1][lx + 1] = x0;
    xt[1][ly + 1][lx + 1] = x1;
  }
  if (h_n >= 0) {
    xt[0][hy][hx] = xg[h_n];
    xt[1][hy][hx] = xg[p.N + h_n];
  }
  __syncthreads();
  {
    float f[2] = {0.f, 0.f};
    if (SH || active) tile_force(f);
    if (active) {
    if (p.has_prev) {
      f[0] = f[0] + prev_pull(x0, pr0, p.neg_k0, s.cap);
      f[1] = f[1] + prev_pull(x1, pr1, p.neg_k0, s.cap);
    }
    a0 = f[0];
    a1 = f[1];
    }
  }
  float my_part = 0.f;  // thread 0: this workgroup's partial power of the last step

  // Where and how the state after `step` steps and that step's partial power are
  // published.  Node granules: slot of parity (step + 1) & 1 (its previous
  // content, the state after step - 2 steps, was consumed by every neighbour
  // before it published the state this workgroup needed for step `step`).  The
  // partial: eight places (parity x four) -- it is read a whole iteration later,
  // by workgroups that no hand-off orders against this one.  tag = step + 1,
  // bit 24 set when the step was redone on the uphill branch: readers know
  // which of the two they need, because everyone takes the same decisions.
  // `sk`: the scalars the published state was stepped with; the next step runs on
  // next_scalars(sk, true) with the gate open, or, redone, on next_scalars(sk, false)
  // with v gated to zero (`first`: the initial state, stepped with `sk` itself).
  // The two expressions are do_step's position update of an own node.
  auto publish = [&](int step, unsigned tag, const Scalars& sk, bool first) {
    u64* slot = q.comm + (long long)wg * 2 * kSlotS + (long long)((step + 1) & 1) * kSlotS;
    if (active && pidx >= 0) {
      const Scalars sd = first ? sk : next_scalars(sk, true);
      const float dtd = sd.dt, c2d = 0.5f * (dtd * dtd);
      const float vd0 = v0 * 1.f, vd1 = v1 * 1.f;
      u64* g = slot + pidx * kGranS;
      put_granule(g + 0, tag, x0 + (dtd * vd0 + c2d * a0));
      put_granule(g + 1, tag, x1 + (dtd * vd1 + c2d * a1));
    }
  };
  // The uphill candidates of the state (xs, vs, as) after `step` steps, published
  // only when the step behind it is redone (everyone redoes it: everyone publishes).
  auto publish_uphill = [&](int step, unsigned tag, const Scalars& sk, float xs0, float xs1,
                            float vs0, float vs1, float as0, float as1) {
    u64* slot = q.comm + (long long)wg * 2 * kSlotS + (long long)((step + 1) & 1) * kSlotS;
    if (active && pidx >= 0) {
      const Scalars su = next_scalars(sk, false);
      const float dtu = su.dt, c2u = 0.5f * (dtu * dtu);
      const float vu0 = vs0 * 0.f, vu1 = vs1 * 0.f;
      u64* g = slot + pidx * kGranS;
      put_granule(g + 2, tag, xs0 + (dtu * vu0 + c2u * as0));
      put_granule(g + 3, tag, xs1 + (dtu * vu1 + c2u * as1));
    }
  };
  // replica `rep` of workgroup w2's partial of step `step` (eight step places)
  auto part_place = [&](int w2, int step, int rep) {
    return q.part + ((long long)rep * kMaxWg + w2) * kPartPitch + ((step + 1) & 7);
  };

#ifdef SFM_MESH_TIMING
  long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stc = clock64();
#define STICK(i) { const long long tn = clock64(); st[i] += tn - stc; stc = tn; }
#else
#define STICK(i)
#endif
  // One step with scalars `sk`; `gate` (0 / 1) is the pending velocity gate of
  // the previous step, applied to the node's own and the halo velocities.  The
  // new state and the step's partial power are published right away (`tag`):
  // the neighbours' next step and everyone's verification do not wait for this
  // workgroup's own verification.
  auto do_step = [&](const Scalars& sk, float gate, int step, unsigned tag) {
    v0 = v0 * gate;
    v1 = v1 * gate;
    const float dt = sk.dt;
    const float c2 = 0.5f * (dt * dt);
    x0 = x0 + (dt * v0 + c2 * a0);
    x1 = x1 + (dt * v1 + c2 * a1);
    if (owner) {
      xt[0][ly + 1][lx + 1] = x0;
      xt[1][ly + 1][lx + 1] = x1;
    }
    if (h_n >= 0) {   // (the owner evaluated this update: see publish)
      const int cand = gate == 0.f ? 2 : 0;
      xt[0][hy][hx] = hval[tid][cand];
      xt[1][hy][hx] = hval[tid][cand + 1];
    }
    lds_barrier();
    issue_early();
    STICK(5)
    float pw = 0.f;
    float f[2] = {0.f, 0.f};
    if (SH || active) tile_force(f);
    STICK(6)
    if (active) {
      const float hdtg = (0.5f * dt) * p.gamma;
      const float fact0 = 1.0f / (1.0f + hdtg);
      const float fact1 = 1.0f - hdtg;
      const float hdt = 0.5f * dt;
      if (p.has_prev) {
        f[0] = f[0] + prev_pull(x0, pr0, p.neg_k0, sk.cap);
        f[1] = f[1] + prev_pull(x1, pr1, p.neg_k0, sk.cap);
      }
      float n0 = fact0 * (v0 * fact1 + hdt * (a0 + f[0]));
      float n1 = fact0 * (v1 * fact1 + hdt * (a1 + f[1]));
      a0 = f[0];
      a1 = f[1];
      float a2 = 0.f, v2 = 0.f;
      a2 = a2 + f[0] * f[0];
      v2 = v2 + n0 * n0;
      pw = pw + f[0] * n0;
      a2 = a2 + f[1] * f[1];
      v2 = v2 + n1 * n1;
      pw = pw + f[1] * n1;
      const float a_norm = sqrtf(a2) + 1e-6f;
      const float v_norm = sqrtf(v2);
      n0 = n0 + sk.alpha * (f[0] / a_norm * v_norm - n0);
      n1 = n1 + sk.alpha * (f[1] / a_norm * v_norm - n1);
      v0 = n0;
      v1 = n1;
    }
    STICK(7)
    publish(step, tag, sk, false);   // the state is final: out before the power reduction
    STICK(8)
    const float t = wave_sum63(pw);
    if (lane == 63 && wave < TL::kWavesT) wred[wave] = t;
    lds_barrier();  // also: everyone is done with xt / hval / nf
    STICK(9)
    if (tid < kPartRep) {   // (every writer lane forms the same sum)
      float acc = 0.f;
      for (int w2 = 0; w2 < TL::kWavesT; ++w2) acc = acc + wred[w2];
      my_part = acc;
      put_granule(part_place(wg, step, tid), tag, acc);
    }
    // (after the last step: nobody's halo).  Two requests: behind the wait for
    // the partials and behind the verification's polls (a third one right after
    // the publication never found the neighbours' state: measured, removed).
    take_early();
    if (step < q.num_iters) request_halo(step, hearly);
  };

  bool ok = true;
  constexpr unsigned kRedoBit = 1u << 24;
  publish(0, 1u, s, true);   // the initial state, for everyone's first step
  unsigned redo_prev = 0;  // kRedoBit if the previous iteration redid its step
  for (int k = 1; k <= q.num_iters + 1; ++k) {
    STICK(4)
    const bool last = k == q.num_iters + 1;
    // the state after k - 1 steps / the partial power of step k - 1 carry this tag
    const unsigned want_tag = static_cast<unsigned>(k) | redo_prev;
    const int slot_off = (k & 1) * kSlotS * static_cast<int>(sizeof(u64));
    STICK(0)
    // ---- the halo of the 8 neighbours: the only wait in front of the step --------
    if (!last) {
      int off[kHaloPollsS];
      unsigned have[kHaloPollsS];
#pragma unroll
      for (int u = 0; u < kHaloPollsS; ++u) {
        off[u] = h_off[u] >= 0 ? h_off[u] + slot_off : -1;
        have[u] = 0;
        if (off[u] >= 0) {   // requested at the end of the previous step, arrived since?
          float t[2];
          have[u] = take_pair(hearly[u], want_tag, t);
          if (have[u] & 1u) h_dst[u][0] = t[0];
          if (have[u] & 2u) h_dst[u][1] = t[1];
        }
      }
      const bool mine_ok = poll_pairs<kHaloPollsS>(comm_rs, off, h_dst, have, want_tag, q.abort);
      if (!wg_and(mine_ok)) {
        ok = false;
        break;
      }
    }
    STICK(1)
    // ---- speculative step k on the downhill branch --------------------------------
    const Scalars s_before = s;  // scalars of step k - 1 (verified)
    const float bx0 = x0, bx1 = x1, bv0 = v0, bv1 = v1, ba0 = a0, ba1 = a1;
    const float part_before = my_part;
    // The partial powers of step k - 1 were published at the end of that step by
    // everyone: requested in the middle of this step, looked at after it.
#pragma unroll
    for (int u = 0; u < kPartPolls1; ++u) {
      const int w2 = tid + u * NT;
      const bool want = k > 1 && w2 < q.n_wg && w2 != wg;
      pg[u] = want ? part_place(w2, k - 1, wg % kPartRep) : nullptr;
      pd[u] = want ? &part_all[w2] : nullptr;
      early[u] = 0;
    }
    early_pending = k > 1;
    early_tag = want_tag;
    Scalars s_try = s;
    if (!last) {
      if (k > 1) s_try = next_scalars(s_before, true);
      do_step(s_try, 1.f, k, static_cast<unsigned>(k + 1));
    }
    issue_early();   // (last iteration: there was no step)
    take_early();
    STICK(2)
    redo_prev = 0;
    if (k == 1) continue;  // no power yet: nothing to verify
    // ---- the partial powers of step k - 1: verify the speculation -------------------
    {
      if (tid == 0) part_all[wg] = part_before;  // own: no round trip
      const bool mine_ok = poll_granules<kPartPolls1>(pg, pd, want_tag, q.abort);
      if (!wg_and(mine_ok)) {
        ok = false;
        break;
      }
      if (wave == 0) {  // fixed order: strided per lane, then the DPP tree
        float t = 0.f;
        for (int w2 = lane; w2 < q.n_wg; w2 += 64) t = t + part_all[w2];
        t = wave_sum63(t);
        if (lane == 63) s_power = t;
      }
      lds_barrier();
    }
    STICK(3)
    const bool downhill = s_power >= 0.f;
    if (last) {
      s = next_scalars(s_before, downhill);
      v0 = v0 * s.gate;
      v1 = v1 * s.gate;
      break;
    }
    if (downhill) {
      s = s_try;
    } else {
      // misprediction: back to the state before the step, uphill scalars, v = 0;
      // the redone step replaces what the speculative one published
      x0 = bx0;
      x1 = bx1;
      v0 = bv0;
      v1 = bv1;
      a0 = ba0;
      a1 = ba1;
      s = next_scalars(s_before, false);
      publish_uphill(k - 1, want_tag, s_before, x0, x1, v0, v1, a0, a1);
      {
        // the neighbours' uphill candidates of the state after k - 1 steps (same
        // slot and tag as the downhill ones this iteration started with)
        int off[kHaloPollsS];
        unsigned have[kHaloPollsS];
        float* d[kHaloPollsS];
#pragma unroll
        for (int u = 0; u < kHaloPollsS; ++u) {
          off[u] = h_off[u] >= 0 ? h_off[u] + slot_off + 2 * static_cast<int>(sizeof(u64)) : -1;
          have[u] = 0;
          d[u] = h_dst[u] ? h_dst[u] + 2 : nullptr;
        }
        const bool mine_ok = poll_pairs<kHaloPollsS>(comm_rs, off, d, have, want_tag, q.abort);
        if (!wg_and(mine_ok)) {
          ok = false;
          break;
        }
      }
      do_step(s, 0.f, k, static_cast<unsigned>(k + 1) | kRedoBit);
      redo_prev = kRedoBit;
    }
  }

#ifdef SFM_MESH_TIMING
  if ((wg == 0 || wg == q.n_wg / 2) && tid == 0)
    printf("SPEC wg %d per step: publish %lld halo-wait %lld step-tail %lld verify %lld other %lld | in step: "
           "advance+S1 %lld force(S2) %lld mix %lld publish %lld reduce+S3 %lld\n", wg,
           st[0] / q.num_iters, st[1] / q.num_iters, st[2] / q.num_iters, st[3] / q.num_iters,
           st[4] / q.num_iters, st[5] / q.num_iters, st[6] / q.num_iters, st[7] / q.num_iters,
           st[8] / q.num_iters, st[9] / q.num_iters);
#endif
  if (!ok) return;  // timed out (the abort flag is set)
  float ek = 0.f, vm2 = 0.f;
  if (active) {
    xo[n] = x0;
    xo[p.N + n] = x1;
    vo[n] = v0;
    vo[p.N + n] = v1;
    ao[n] = a0;
    ao[p.N + n] = a1;
    ek = v0 * v0 + v1 * v1;
    vm2 = ek;
  }
#pragma unroll
  for (int dd = 32; dd > 0; dd >>= 1) {
    ek = ek + __shfl_xor(ek, dd, 64);
    vm2 = fmaxf(vm2, __shfl_xor(vm2, dd, 64));
  }
  __shared__ float fin[TL::kWavesT][2];
  __syncthreads();
  if (lane == 0 && wave < TL::kWavesT) {
    fin[wave][0] = ek;
    fin[wave][1] = vm2;
  }
  __syncthreads();
  if (tid == 0) {
    float e = 0.f, m = 0.f;
    for (int w2 = 0; w2 < TL::kWavesT; ++w2) {
      e = e + fin[w2][0];
      m = fmaxf(m, fin[w2][1]);
    }
    q.stat_partials[wg * 2] = e;
    q.stat_partials[wg * 2 + 1] = m;
    if (wg == 0) *q.scal_out = s;
  }
}

// All-or-nothing hand-over of the persistent kernel's result.
__global__ void __launch_bounds__(kBlock)
persist_commit_kernel(const int* __restrict__ abort, const float* __restrict__ xs,
                      const float* __restrict__ vs, const float* __restrict__ as,
                      float* __restrict__ x, float* __restrict__ v,
                      float* __restrict__ a, long long n) {
  if (*abort) return;
  for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n;
       i += (long long)gridDim.x * kBlock) {
    x[i] = xs[i];
    v[i] = vs[i];
    a[i] = as[i];
  }
}

const int kDefaultLinks[13][3] = {
    {1, 0, 0}, {0, 1, 0},  {0, 0, 1},  {1, 1, 0},  {-1, 1, 0},
    {1, 0, 1}, {-1, 0, 1}, {0, 1, 1},  {0, -1, 1}, {1, 1, 1},
    {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}};

int build_params(const SfmMeshDesc* d, MeshParams* p) {
  if (!d) return sfm::fail(SFM_ERR_INVALID, "desc is NULL");
  if (d->ncomp != 2 && d->ncomp != 3)
    return sfm::fail(SFM_ERR_INVALID, "ncomp must be 2 or 3, got %d", d->ncomp);
  for (int i = 0; i < 4; ++i)
    if (d->shape[i] < 1)
      return sfm::fail(SFM_ERR_INVALID, "shape[%d] = %d", i, d->shape[i]);
  std::memset(p, 0, sizeof(*p));
  p->ncomp = d->ncomp;
  p->B = d->shape[0];
  p->Z = d->shape[1];
  p->Y = d->shape[2];
  p->X = d->shape[3];
  p->N = (long long)p->B * p->Z * p->Y * p->X;
  p->prefer = d->prefer_orig_order;
  p->neg_k0 = static_cast<float>(-d->k0);
  p->has_prev = d->prev != nullptr || d->target != nullptr || d->prev_cb != nullptr;
  p->force_kind = d->force_kind;
  if (d->force_kind < SFM_FORCE_SPRINGS || d->force_kind > SFM_FORCE_EXTERNAL)
    return sfm::fail(SFM_ERR_INVALID, "force_kind %d", d->force_kind);
  if (d->force_kind == SFM_FORCE_TILE_MESH) {
    if (!d->cx || !d->cy)
      return sfm::fail(SFM_ERR_INVALID, "tile mesh force needs cx and cy");
    p->cx = d->cx;
    p->cy = d->cy;
  }
  if (d->force_kind == SFM_FORCE_EXTERNAL) {
    if (!d->ext_force || !d->force_cb)
      return sfm::fail(SFM_ERR_INVALID, "external force needs ext_force and force_cb");
    p->ext = d->ext_force;
  }
  if (d->force_kind != SFM_FORCE_SPRINGS) {
    // no link stencil: sections (batch and z) are independent planes
    p->Z = 1;
    p->B = d->shape[0] * d->shape[1];
  } else if (d->ncomp == 2) {
    // Batch and z are both independent slices for the in-plane force; fold
    // them so the stencil never crosses a slice (Z extent of the stencil = 1).
    p->Z = 1;
    p->B = d->shape[0] * d->shape[1];
    const int dirs[4][3] = {{1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {-1, 1, 0}};
    p->n_links = 4;
    p->order2d = 1;
    const float kf = static_cast<float>(d->k);
    const float k2 = kf / sqrtf(2.0f);  // k / jnp.sqrt(2.0)  (mesh.py:137)
    for (int L = 0; L < 4; ++L) {
      for (int c = 0; c < 3; ++c) p->dir[L][c] = dirs[L][c];
      p->rest[L][0] = static_cast<float>(dirs[L][0] * d->stride[0]);
      p->rest[L][1] = static_cast<float>(dirs[L][1] * d->stride[1]);
      p->rest[L][2] = 0.f;
      p->neg_k[L] = L < 2 ? static_cast<float>(-d->k) : -k2;
    }
  } else {
    p->n_links = d->n_links > 0 ? d->n_links : 13;
    // (the unrolled default-link stencil and the z-march kernel use 32-bit byte offsets
    // over the three component planes; a mesh beyond 4 GB takes the generic link loop)
    p->default_links = d->n_links == 0 && (unsigned long long)p->N * 12ull < (1ull << 32);
    if (p->n_links > SFM_MESH_MAX_LINKS)
      return sfm::fail(SFM_ERR_INVALID, "too many links: %d", p->n_links);
    for (int L = 0; L < p->n_links; ++L) {
      float r2 = 0.f;
      for (int c = 0; c < 3; ++c) {
        const int v = d->n_links > 0 ? d->links[L][c] : kDefaultLinks[L][c];
        if (v < -1 || v > 1)
          return sfm::fail(SFM_ERR_INVALID,
                           "Only |v| <= 1 values supported within links.");
        p->dir[L][c] = v;
        p->rest[L][c] = static_cast<float>(d->stride[c] * v);
      }
      r2 = p->rest[L][0] * p->rest[L][0] + p->rest[L][1] * p->rest[L][1];
      r2 = r2 + p->rest[L][2] * p->rest[L][2];
      const float l0 = sqrtf(r2);
      // k_eff = k * stride_x / |l0|   (mesh.py:259)
      p->neg_k[L] = static_cast<float>(-(d->k * d->stride[0] / (double)l0));
    }
  }
  p->fire = d->fire;
  p->remove_drift = d->remove_drift;
  p->drift_cols = d->remove_drift == 2;
  p->n_col = static_cast<float>(p->N / p->X);
  p->gamma = static_cast<float>(d->gamma);
  p->vv_dt = static_cast<float>(d->dt);
  p->f_alpha = static_cast<float>(d->f_alpha);
  p->f_inc = static_cast<float>(d->f_inc);
  p->f_dec = static_cast<float>(d->f_dec);
  p->alpha0 = static_cast<float>(d->alpha0);
  p->n_min = d->n_min;
  p->dt_cap = static_cast<float>(d->dt_max * d->dt);
  p->final_cap = static_cast<float>(d->final_cap);
  p->cap_scale = static_cast<float>(d->cap_scale);
  p->cap_every = d->cap_upscale_every > 0 ? d->cap_upscale_every : 1;
  p->n_f = static_cast<float>(p->N);
  p->own_y0 = 0;
  p->own_y1 = p->Y;
  return SFM_OK;
}

bool small_enabled() {
  const char* e = sfm::option("SFM_MESH_SMALL");  // "0": the launch-per-kernel path
  return !(e && e[0] == '0');
}

bool persist3d_enabled() {
  // opt-in ("1"): built in round 6, bit-identical, and measured SLOWER than the four-launch
  // step it replaces (95-130 us per step against 47 on [3,64,12,12,12]: see the kernel)
  const char* e = sfm::option("SFM_MESH_PERSIST3D");
  return e && e[0] == '1';
}

// Plan of integrate_march3d_kernel for a mesh: the thread tile (halo included) that
// wastes the fewest thread slots, and enough chunks of planes to fill the CUs.
struct March3dPlan {
  March3dArgs g;
  int T = 0;        // threads per workgroup (0: not applicable)
  int grid = 0;
  size_t lds = 0;
};

int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

March3dPlan plan_march3d(const MeshParams& p) {
  March3dPlan best;
  // SFM_MESH_MARCH3D: "0" off, "1" on for every default-link volume, else by size
  const char* e = sfm::option("SFM_MESH_MARCH3D");
  if (e && e[0] == '0') return best;
  if (p.ncomp != 3 || !p.default_links || p.force_kind != SFM_FORCE_SPRINGS) return best;
  const bool forced = e && e[0] == '1';
  if (!forced && p.N < kMarch3dMinNodes) return best;
  const char* te = sfm::option("SFM_MESH_MARCH3D_T");
  const int t_only = te ? atoi(te) : 0;
  if ((unsigned long long)p.N * 12ull >= (1ull << 32)) return best;  // 32-bit byte offsets
  long long best_cost = 0;
  for (int T : {1024, 512, 256}) {
    if (t_only && T != t_only) continue;
    for (int ntx = 1; ntx <= p.X; ++ntx) {
      const int cxw = (p.X + ntx - 1) / ntx;
      const int txh = cxw + 2;
      if (txh * 3 > T) continue;
      const int rows = T / txh;  // >= 3; one halo row (on top) per tile
      const int nty = (p.Y + rows - 2) / (rows - 1);
      const int tyh = (p.Y + nty - 1) / nty + 1;
      // thread slots per plane; 16 waves in step at every barrier cost ~15 % against
      // two workgroups of 8 (measured on [3,4,100^3]: 180 us with 12 288 slots per plane
      // against 176 with 13 824)
      const long long cost = (long long)ntx * nty * T * (T == 1024 ? 115 : 100);
      if (best.T == 0 || cost < best_cost) {
        best_cost = cost;
        best.T = T;
        best.g.txh = txh;
        best.g.tyh = tyh;
        best.g.ntx = ntx;
        best.g.nty = nty;
      }
      if (cxw <= 8) break;  // narrower tiles only add halo
    }
  }
  if (best.T == 0) return best;
  best.lds = (size_t)36 * best.T * sizeof(float);
  const int per_cu = std::max<int>(1, static_cast<int>(160 * 1024 / best.lds));
  const long long cols = (long long)p.B * best.g.ntx * best.g.nty;
  if (cols > 0x7fffffffLL / std::max(p.Z, 1)) {
    best.T = 0;
    return best;
  }
  const long long slots = std::min<long long>((long long)device_cus() * per_cu, kMaxBlocks);
  const long long planes = cols * p.Z;
  // every workgroup the same number of planes (a run that crosses into the next column
  // pays one more start-up plane); not below 8 planes per workgroup
  long long run = std::max<long long>((planes + slots - 1) / slots, std::min<long long>(8, p.Z));
  const char* ze = sfm::option("SFM_MESH_MARCH3D_ZC");
  if (ze && atoi(ze) > 0) run = atoi(ze);
  if ((planes + run - 1) / run > kMaxBlocks) run = (planes + kMaxBlocks - 1) / kMaxBlocks;
  best.g.cols = static_cast<int>(cols);
  best.g.run = static_cast<int>(run);
  best.grid = static_cast<int>((planes + run - 1) / run);
  return best;
}

template <int T, bool PREFER>
int launch_march3d_t(const March3dPlan& m, hipStream_t st, const float* x, float* v, float* a,
                     const float* prev, const MeshParams& p, const Scalars* scal, float cap,
                     float* partials) {
  static bool attr_set = false;
  if (!attr_set) {
    SFM_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&integrate_march3d_kernel<T, PREFER>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 36 * T * static_cast<int>(sizeof(float))));
    attr_set = true;
  }
  hipLaunchKernelGGL((integrate_march3d_kernel<T, PREFER>), dim3(m.grid), dim3(T), m.lds, st, x,
                     v, a, prev, p, scal, cap, partials, m.g);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int launch_march3d(const March3dPlan& m, hipStream_t st, const float* x, float* v, float* a,
                   const float* prev, const MeshParams& p, const Scalars* scal, float cap,
                   float* partials) {
#define SFM_MARCH(T)                                                                      \
  return p.prefer ? launch_march3d_t<T, true>(m, st, x, v, a, prev, p, scal, cap, partials) \
                  : launch_march3d_t<T, false>(m, st, x, v, a, prev, p, scal, cap, partials)
  switch (m.T) {
    case 1024: SFM_MARCH(1024);
    case 512: SFM_MARCH(512);
    default: SFM_MARCH(256);
  }
#undef SFM_MARCH
}

bool persistent_enabled() {
  const char* e = sfm::option("SFM_MESH_PERSISTENT");
  return !(e && e[0] == '0');
}

int grid_for(long long n) {
  long long g = (n + kBlock - 1) / kBlock;
  if (g > kMaxBlocks) g = kMaxBlocks;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

struct MeshWorkspace {
  Scalars* scal;       // [2]
  float* partials;     // [kMaxBlocks * kNP]
  float* stat_part;    // [kMaxBlocks * 2]
  float* stats;        // [2]
  u64* comm;           // persistent path: [256][2][slot] granules
  int* abort;          // persistent path: timeout flag
  size_t comm_bytes;
  float* prev_buf;     // native prev_fn: prev = target_mesh(x), [ncomp * N]
  float* alt[3];       // tiled path: second (x, v, a) set, [ncomp * N] each
  u64* tile_part;      // tiled path: [n_tiles * kNP] {epoch, value} granules
  int* ticket;         // tiled path: last-workgroup counter
  float* colsum;       // per-column drift means [6][X] (remove_drift == 2)
  float* col_part;     // drift_cols_kernel: [3][kColChunksMax][2][X] chunk sums
  int* col_ticket;     // [3], zero between launches
  int* target_list;    // native prev_fn, in-plane: node blocks on the overlap strips
  size_t bytes;
};

// Tile shape of the tiled in-plane integrator (integrate_shared2d_kernel):
// tx = 0: not applicable.
struct TilePlan {
  int tx = 0, ty = 0, nty = 0, ntx = 0;
  long long tiles = 0;
  // packed remainder column (BandArgs::pack_*), un-split launches only
  int pack_S = 0, pack_segw = 0, pack_ntxw = 0;
  long long packed_tiles = 0;   // workgroups of a packed launch (== tiles without packing)
};

bool pack_enabled() {
  const char* e = sfm::option("SFM_MESH_PACK");   // "0": every tile row of the last column its own workgroup
  return !(e && e[0] == '0');
}

bool fuse_target_enabled() {
  const char* e = sfm::option("SFM_MESH_FUSE_TARGET");  // "0": advance + target + integrate
  return !(e && e[0] == '0');
}

bool tiled_enabled() {
  const char* e = sfm::option("SFM_MESH_TILED");
  return !(e && e[0] == '0');
}

TilePlan plan_tiles(int ncomp, long long planes, int Y, int X) {
  TilePlan best;
  // every spring once (integrate_shared2d_kernel): 16 x 62 tiles, a lane per
  // column; pays off unless most of a 64-lane row would hang over the mesh
  // (narrower meshes take the advance / integrate pair)
  if (ncomp != 2 || !tiled_enabled() || X < 40 || Y < 4) return best;
  best.ty = kSY;
  best.tx = kSX;
  best.nty = (Y + kSY - 1) / kSY;
  best.ntx = (X + kSX - 1) / kSX;
  best.tiles = planes * best.nty * best.ntx;
  if (best.tiles > 0x7fffffffLL / kNP) best = TilePlan();
  best.packed_tiles = best.tiles;
  const int rem = X % kSX;
  if (best.tiles && rem && rem + 2 <= 32 && best.nty >= 2) {
    best.pack_segw = rem + 2;
    best.pack_S = std::min(64 / best.pack_segw, best.nty);
    best.pack_ntxw = best.ntx - 1;
    best.packed_tiles =
        planes * ((long long)best.nty * best.pack_ntxw + (best.nty + best.pack_S - 1) / best.pack_S);
  }
  return best;
}

// BandArgs of an un-split launch of the tiled step.
BandArgs plain_band_args(const TilePlan& t, int xcd_map, bool pack) {
  BandArgs b{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0, xcd_map, 0, 0, 0, 0};
  if (pack && t.pack_S >= 2) {
    b.pack_S = t.pack_S;
    b.pack_segw = t.pack_segw;
    b.pack_ntxw = t.pack_ntxw;
    b.pack_recip = (65536 + t.pack_segw - 1) / t.pack_segw;
  }
  return b;
}

MeshWorkspace carve(void* ws, size_t prev_floats, size_t alt_floats, long long tiles,
                    int ncols, size_t list_ints = 0) {
  sfm::Carver c(ws);
  MeshWorkspace w;
  w.target_list = list_ints ? c.take<int>(list_ints) : nullptr;
  w.prev_buf = prev_floats ? c.take<float>(prev_floats) : nullptr;
  for (int i = 0; i < 3; ++i) w.alt[i] = alt_floats ? c.take<float>(alt_floats) : nullptr;
  w.tile_part = tiles ? c.take<u64>((size_t)tiles * kNP) : nullptr;
  w.ticket = c.take<int>(4);
  w.colsum = c.take<float>(6 * (size_t)(ncols > 0 ? ncols : 1));
  w.col_part = c.take<float>(ncols > 0 ? (size_t)3 * kColChunksMax * 2 * ncols : 1);
  w.col_ticket = c.take<int>(4);
  w.scal = c.take<Scalars>(2);
  w.partials = c.take<float>(kMaxBlocks * kNP);
  w.stat_part = c.take<float>(kMaxBlocks * 2);
  w.stats = c.take<float>(2);
  const size_t slot_max = Tile<32>::kSlot;
  const size_t part_u64 = (size_t)kPartRep * kMaxWg * kPartPitch;
  w.comm = c.take<u64>((size_t)kMaxWg * 2 * slot_max + 8 + part_u64);
  w.abort = reinterpret_cast<int*>(w.comm + (size_t)kMaxWg * 2 * slot_max);
  w.comm_bytes = ((size_t)kMaxWg * 2 * slot_max + 8 + part_u64) * sizeof(u64);
  w.bytes = c.total();
  return w;
}

MeshWorkspace carve_for(const SfmMeshDesc* d, void* ws, TilePlan* plan) {
  const size_t n = (size_t)d->shape[0] * d->shape[1] * d->shape[2] * d->shape[3];
  TilePlan t = d->force_kind == SFM_FORCE_SPRINGS
                   ? plan_tiles(d->ncomp, (long long)d->shape[0] * d->shape[1],
                                d->shape[2], d->shape[3])
                   : TilePlan();
  if (plan) *plan = t;
  const size_t cn = (size_t)d->ncomp * n;
  // second (x, v, a) set: ping-pong of the fused tiled step, staging of the
  // persistent kernel's result
  return carve(ws, d->target ? cn : 0,
               ((d->ncomp == 2 && !d->prev_cb && d->force_kind == SFM_FORCE_SPRINGS) ||
                (d->ncomp == 3 && d->target))   // (volumetric montage: mesh_persist3d_kernel)
                   ? cn : 0,
               t.tiles,
               d->shape[3], d->target ? sfm::target_list_ints(d->target) : 0);
}

}  // namespace

extern "C" {

size_t sfm_mesh_workspace_bytes(const SfmMeshDesc* d) {
  if (!d) return 0;
  return carve_for(d, nullptr, nullptr).bytes;
}

int sfm_mesh_force(const SfmMeshDesc* d, float* out) {
  MeshParams p;
  if (int rc = build_params(d, &p)) return rc;
  if (!d->x || !out) return sfm::fail(SFM_ERR_INVALID, "x/out is NULL");
  if (p.force_kind == SFM_FORCE_EXTERNAL)
    return sfm::fail(SFM_ERR_INVALID, "sfm_mesh_force: the external force is the caller's");
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int grid = grid_for(p.N);
  if (p.ncomp == 2)
    hipLaunchKernelGGL(force_kernel<2>, dim3(grid), dim3(kBlock), 0, st, d->x,
                       nullptr, out, p, 0.f, 0);
  else
    hipLaunchKernelGGL(force_kernel<3>, dim3(grid), dim3(kBlock), 0, st, d->x,
                       nullptr, out, p, 0.f, 0);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int sfm_mesh_relax_chunk(const SfmMeshDesc* d, SfmFireState* fire,
                         SfmChunkStats* stats) {
  MeshParams p;
  if (int rc = build_params(d, &p)) return rc;
  if (!d->x || !d->v || !d->a)
    return sfm::fail(SFM_ERR_INVALID, "x/v/a must be device pointers");
  if (!fire || !stats) return sfm::fail(SFM_ERR_INVALID, "fire/stats is NULL");
  if (d->num_iters < 0) return sfm::fail(SFM_ERR_INVALID, "num_iters < 0");
  if (d->remove_drift < 0 || d->remove_drift > 2 ||
      (d->remove_drift == 2 && d->ncomp != 3))
    return sfm::fail(SFM_ERR_INVALID,
                     "remove_drift: 0 none, 1 global mean, 2 per x column (3-D only)");
  if ((d->target || d->prev_cb) && d->prev)
    return sfm::fail(SFM_ERR_INVALID,
                     "Only one of: \"prev\" and \"prev_fn\" can be specified.");
  if (d->target && d->prev_cb)
    return sfm::fail(SFM_ERR_INVALID, "prev_fn: either the native target mesh or a callback");
  if (d->prev_cb && !d->ext_prev)
    return sfm::fail(SFM_ERR_INVALID, "prev_cb needs the ext_prev buffer");
  TilePlan tiles;
  MeshWorkspace w = carve_for(d, d->workspace, &tiles);
  if (!d->workspace || d->workspace_bytes < w.bytes)
    return sfm::fail(SFM_ERR_WORKSPACE, "mesh workspace needs %zu bytes, got %zu",
                     w.bytes, d->workspace_bytes);
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int grid = grid_for(p.N);
  const float* prev_ptr = d->target ? w.prev_buf : d->prev_cb ? d->ext_prev : d->prev;
  // prev = prev_fn(x) (mesh.py:429-430): the native target mesh, or the caller's
  // callable through prev_cb, re-evaluated in front of every force evaluation
  const bool dyn_prev = d->target || d->prev_cb;
  auto eval_prev = [&](hipStream_t s_) -> int {
    if (d->target) return sfm::launch_target_mesh(d->target, d->x, w.prev_buf, s_);
    if (d->prev_cb && d->prev_cb(d->prev_user) != 0)
      return sfm::fail(SFM_ERR_INVALID, "mesh: the prev_fn callback failed");
    return SFM_OK;
  };

  Scalars s0;
  std::memset(&s0, 0, sizeof(s0));
  s0.dt = fire->dt;
  s0.alpha = fire->alpha;
  s0.n_pos = 0;  // restarts every call (mesh.py:513)
  s0.cap = fire->cap;
  s0.gate = 1.f;
  SFM_HIP_CHECK(hipMemcpyAsync(&w.scal[0], &s0, sizeof(s0),
                               hipMemcpyHostToDevice, st));

  const float cap0 = fire->cap;

  // Persistent single-launch path for in-plane meshes that fit the chip.
  if (persistent_enabled() && p.ncomp == 2 && d->num_iters >= 1 && !dyn_prev &&
      p.force_kind == SFM_FORCE_SPRINGS) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    }
    // Small tiles spread the (latency bound) step over more CUs; fall back to
    // 32 x 32 tiles when there would be more workgroups than CUs.
    int tile = 0;
    const char* force_tile = sfm::option("SFM_MESH_TILE");  // experiment: 16 or 32
    for (int t : {16, 32}) {
      if (force_tile && atoi(force_tile) != t) continue;
      const long long nw = (long long)p.B * ((p.Y + t - 1) / t) * ((p.X + t - 1) / t);
      if (nw <= kMaxWg && nw <= cus) {
        tile = t;
        break;
      }
    }
    if (tile) {
      const int nty = (p.Y + tile - 1) / tile, ntx = (p.X + tile - 1) / tile;
      const long long n_wg = (long long)p.B * nty * ntx;
      PersistArgs q;
      q.comm = w.comm;
      q.part = w.comm + (size_t)kMaxWg * 2 * Tile<32>::kSlot + 8;
      q.abort = w.abort;
      q.scal_in = &w.scal[0];
      q.scal_out = &w.scal[1];
      q.stat_partials = w.stat_part;
      q.num_iters = d->num_iters;
      q.cap0 = cap0;
      q.nty = nty;
      q.ntx = ntx;
      q.n_wg = static_cast<int>(n_wg);
      SFM_HIP_CHECK(hipMemsetAsync(w.comm, 0, w.comm_bytes, st));
      sfm::prof_begin(sfm::kProfMesh, st);
      const char* spec_env = sfm::option("SFM_MESH_SPECULATE");
      const bool spec = p.fire && !p.remove_drift && !(spec_env && spec_env[0] == '0');
      if (spec && tile == 16)
        hipLaunchKernelGGL(mesh_persist2d_spec_kernel<16>, dim3(q.n_wg), dim3(spec_threads<16>()), 0, st,
                           p, d->x, d->v, w.alt[0], w.alt[1], w.alt[2], d->prev, q);
      else if (spec)
        hipLaunchKernelGGL(mesh_persist2d_spec_kernel<32>, dim3(q.n_wg), dim3(1024), 0, st,
                           p, d->x, d->v, w.alt[0], w.alt[1], w.alt[2], d->prev, q);
      else if (tile == 16)
        hipLaunchKernelGGL(mesh_persist2d_kernel<16>, dim3(q.n_wg), dim3(256), 0, st,
                           p, d->x, d->v, w.alt[0], w.alt[1], w.alt[2], d->prev, q);
      else
        hipLaunchKernelGGL(mesh_persist2d_kernel<32>, dim3(q.n_wg), dim3(1024), 0, st,
                           p, d->x, d->v, w.alt[0], w.alt[1], w.alt[2], d->prev, q);
      sfm::prof_end(sfm::kProfMesh, st);
      SFM_LAUNCH_CHECK();
      hipLaunchKernelGGL(persist_commit_kernel, dim3(grid_for((long long)p.ncomp * p.N)),
                         dim3(kBlock), 0, st, w.abort, w.alt[0], w.alt[1], w.alt[2], d->x,
                         d->v, d->a, (long long)p.ncomp * p.N);
      SFM_LAUNCH_CHECK();
      hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(kBlock), 0, st, w.stat_part,
                         q.n_wg, w.stats);
      SFM_LAUNCH_CHECK();
      Scalars s1;
      float hs[2];
      int aborted = 0;
      SFM_HIP_CHECK(hipMemcpyAsync(&s1, &w.scal[1], sizeof(s1),
                                   hipMemcpyDeviceToHost, st));
      SFM_HIP_CHECK(hipMemcpyAsync(hs, w.stats, sizeof(hs), hipMemcpyDeviceToHost, st));
      SFM_HIP_CHECK(hipMemcpyAsync(&aborted, w.abort, sizeof(int),
                                   hipMemcpyDeviceToHost, st));
      SFM_HIP_CHECK(hipStreamSynchronize(st));
      if (!aborted) {
        if (p.fire) {
          fire->dt = s1.dt;
          fire->alpha = s1.alpha;
          fire->n_pos = s1.n_pos;
          fire->cap = s1.cap;
        }
        stats->e_kin = hs[0];
        stats->v_max = hs[1];
        return SFM_OK;
      }
      // Timed out (workgroups not co-resident?): the commit kernel saw the
      // abort flag and left x, v, a untouched; fall through to the
      // multi-launch path.
      SFM_HIP_CHECK(hipMemcpyAsync(&w.scal[0], &s0, sizeof(s0),
                                   hipMemcpyHostToDevice, st));
    }
  }
#define SFM_MESH_DISPATCH(KERNEL, ...)                                       \
  do {                                                                       \
    if (p.ncomp == 2)                                                        \
      hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(kBlock), 0, st,         \
                         __VA_ARGS__);                                       \
    else                                                                     \
      hipLaunchKernelGGL(KERNEL<3>, dim3(grid), dim3(kBlock), 0, st,         \
                         __VA_ARGS__);                                       \
    SFM_LAUNCH_CHECK();                                                      \
  } while (0)

  // The caller's mesh_force (SFM_FORCE_EXTERNAL) is evaluated on the host's
  // initiative right before the kernel that consumes it.
  auto external_force = [&]() -> int {
    if (p.force_kind != SFM_FORCE_EXTERNAL) return SFM_OK;
    if (d->force_cb(d->force_user) != 0)
      return sfm::fail(SFM_ERR_INVALID, "mesh: the external force callback failed");
    return SFM_OK;
  };

  // a = F(x) + pull(prev, cap)   (mesh.py:501); prev = prev_fn(x) if native
  if (int rc = eval_prev(st)) return rc;
  if (int rc = external_force()) return rc;
  SFM_MESH_DISPATCH(force_kernel, d->x, prev_ptr, d->a, p, cap0, p.has_prev);

  int cur = 0;
  int finish_mode = d->num_iters > 0 ? 1 : 0;
  // One workgroup's worth of nodes: every step in one launch (mesh_small_kernel).
  const bool small = small_enabled() && grid == 1 && d->num_iters > 0 && !dyn_prev &&
                     p.force_kind != SFM_FORCE_EXTERNAL && !p.drift_cols &&
                     p.own_y0 <= 0 && p.own_y1 >= p.Y;
  if (small) {
    sfm::prof_begin(sfm::kProfMesh, st);
    if (p.ncomp == 2)
      hipLaunchKernelGGL(mesh_small_kernel<2>, dim3(1), dim3(kBlock), 0, st, d->x, d->v, d->a,
                         prev_ptr, p, w.scal, cap0, w.partials, d->num_iters);
    else
      hipLaunchKernelGGL(mesh_small_kernel<3>, dim3(1), dim3(kBlock), 0, st, d->x, d->v, d->a,
                         prev_ptr, p, w.scal, cap0, w.partials, d->num_iters);
    sfm::prof_end(sfm::kProfMesh, st);
    SFM_LAUNCH_CHECK();
  }
  const bool tiled = tiles.tx && d->num_iters > 0;
  float* bufs[2][3] = {{d->x, d->v, d->a}, {w.alt[0], w.alt[1], w.alt[2]}};
  int in = 0;
  // Native prev_fn on a tiled in-plane mesh: the target mesh is sampled from the
  // positions AFTER the position update, which the target kernel forms itself
  // from (x, v, a) on the overlap strips (sfm::AdvanceView) -- so the step is
  // target mesh (strips only) + the fused integrator instead of advance + target
  // mesh (all nodes) + integrate.  Same float operations: bit-identical.
  const bool fuse_target = tiled && d->target && w.alt[0] && fuse_target_enabled();
  const bool fused = tiled && (!dyn_prev || fuse_target);
  if (fuse_target && w.target_list)
    if (int rc = sfm::build_target_list(d->target, w.target_list, st)) return rc;
  const bool pack = tiles.pack_S >= 2 && pack_enabled();
  const int tgrid = static_cast<int>(pack ? tiles.packed_tiles : tiles.tiles);
  // XCD-contiguous tile order: on once there are several rounds of workgroups
  // (measured after the SGPR spills were gone: [2,4,2048^2] 270 -> 261 us,
  // [2,64,204^2] 62.2 -> 58.6; [2,1,1000^2], one round of 1071 tiles: 31.1 -> 31.5).
  // SFM_MESH_XCD=0 / 1: off / on for any grid of at least 64 tiles.
  const char* xcd_opt = sfm::option("SFM_MESH_XCD");
  const int xcd_map = xcd_opt && xcd_opt[0] == '0'   ? 0
                      : xcd_opt && xcd_opt[0] == '1' ? (tgrid >= 64 ? 1 : 0)
                                                     : (tgrid >= 2048 ? 1 : 0);
  if (tiled) {
    // LDS-tiled integrator (2-D): one launch per step, or advance + prev_fn +
    // integrate when the spring targets depend on the advanced positions.
    SFM_HIP_CHECK(hipMemsetAsync(w.ticket, 0, 2 * sizeof(int), st));
    SFM_HIP_CHECK(hipMemsetAsync(w.tile_part, 0, (size_t)tiles.tiles * kNP * sizeof(u64), st));
    finish_mode = 2;
  }
  // default-link volumes: every spring once (integrate_march3d_kernel)
  const March3dPlan march = (!tiled && !small) ? plan_march3d(p) : March3dPlan();
  const int part_rows = march.T ? march.grid : grid;
  // One integration step, enqueued on `ls`.
  hipStream_t ls = st;
#define SFM_STEP_DISPATCH(KERNEL, ...)                                       \
  do {                                                                       \
    if (p.ncomp == 2)                                                        \
      hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(kBlock), 0, ls,         \
                         __VA_ARGS__);                                       \
    else                                                                     \
      hipLaunchKernelGGL(KERNEL<3>, dim3(grid), dim3(kBlock), 0, ls,         \
                         __VA_ARGS__);                                       \
    SFM_LAUNCH_CHECK();                                                      \
  } while (0)
  // column means of the state in (xs, vs): ~16 rows per thread and chunk
  const long long col_rows = p.N / p.X;
  const int col_groups = p.X <= kBlock ? kBlock / p.X : 1;
  const int col_chunks = static_cast<int>(std::max<long long>(
      1, std::min<long long>(kColChunksMax, (col_rows + col_groups * 16LL - 1) / (col_groups * 16LL))));
  const int col_rows_per = static_cast<int>((col_rows + col_chunks - 1) / col_chunks);
  if (p.fire && p.drift_cols) SFM_HIP_CHECK(hipMemsetAsync(w.col_ticket, 0, 4 * sizeof(int), st));
  auto column_means = [&](const float* xs, const float* vs) {
    hipLaunchKernelGGL(drift_cols_kernel<3>, dim3(col_chunks, 3), dim3(kBlock), 0, ls, xs, vs, p,
                       w.colsum, w.col_part, w.col_ticket, col_rows_per);
  };
  bool persist3d_done = false;

  // Volumetric montage: every step of the chunk in one launch (mesh_persist3d_kernel).
  if (persist3d_enabled() && p.ncomp == 3 && d->target && !d->prev_cb && p.fire &&
      d->num_iters >= 1 && p.force_kind == SFM_FORCE_SPRINGS && p.has_prev && w.alt[0] &&
      p.own_y0 <= 0 && p.own_y1 >= p.Y && p.X <= kP3MaxX && d->target->ncomp == 3 &&
      d->target->n_eval == 0 && (long long)grid * kBlock >= p.N) {
    const SfmTargetMeshDesc& t = *d->target;
    const long long mn = (long long)t.mesh_shape[0] * t.mesh_shape[1] * t.mesh_shape[2];
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    const void* kfn = p.drift_cols ? reinterpret_cast<const void*>(&mesh_persist3d_kernel<true>)
                                   : reinterpret_cast<const void*>(&mesh_persist3d_kernel<false>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kBlock, 0) != hipSuccess)
      per_cu = 0;
    // every workgroup has to be resident at once (grid barriers); a wave inside one tile
    if (mn % 64 == 0 && mn * t.n_tiles == p.N && (long long)per_cu * cus >= grid) {
      const size_t bytes = (size_t)3 * p.N * sizeof(float);
      SFM_HIP_CHECK(hipMemcpyAsync(w.alt[0], d->x, bytes, hipMemcpyDeviceToDevice, st));
      SFM_HIP_CHECK(hipMemcpyAsync(w.alt[1], d->v, bytes, hipMemcpyDeviceToDevice, st));
      SFM_HIP_CHECK(hipMemcpyAsync(w.alt[2], d->a, bytes, hipMemcpyDeviceToDevice, st));
      SFM_HIP_CHECK(hipMemsetAsync(w.comm, 0, 64, st));
      SFM_HIP_CHECK(hipMemsetAsync(w.abort, 0, sizeof(int), st));
      Persist3dArgs q;
      q.x = w.alt[0];
      q.v = w.alt[1];
      q.a = w.alt[2];
      q.scal = w.scal;
      q.partials = w.partials;
      q.colsum = w.colsum;
      q.col_part = w.col_part;
      q.bar = reinterpret_cast<unsigned*>(w.comm);
      q.abort = w.abort;
      q.num_iters = d->num_iters;
      q.cap0 = cap0;
      q.col_chunks = col_chunks;
      q.col_rows_per = col_rows_per;
      q.t = t;
      sfm::prof_begin(sfm::kProfMesh, st);
      if (p.drift_cols)
        hipLaunchKernelGGL(mesh_persist3d_kernel<true>, dim3(grid), dim3(kBlock), 0, st, p, q);
      else
        hipLaunchKernelGGL(mesh_persist3d_kernel<false>, dim3(grid), dim3(kBlock), 0, st, p, q);
      sfm::prof_end(sfm::kProfMesh, st);
      SFM_LAUNCH_CHECK();
      hipLaunchKernelGGL(persist_commit_kernel, dim3(grid_for(3 * p.N)), dim3(kBlock), 0, st,
                         w.abort, w.alt[0], w.alt[1], w.alt[2], d->x, d->v, d->a, 3 * p.N);
      SFM_LAUNCH_CHECK();
      int aborted = 0;
      SFM_HIP_CHECK(hipMemcpyAsync(&aborted, w.abort, sizeof(int), hipMemcpyDeviceToHost, st));
      SFM_HIP_CHECK(hipStreamSynchronize(st));
      if (!aborted) {
        persist3d_done = true;
        cur = d->num_iters & 1;
      } else {
        // a barrier timed out (workgroups not co-resident): x, v, a are untouched; start over
        SFM_HIP_CHECK(hipMemcpyAsync(&w.scal[0], &s0, sizeof(s0), hipMemcpyHostToDevice, st));
      }
    }
  }
  auto step = [&](int pending) -> int {
    if (fused) {
      float** bi = bufs[in];
      float** bo = bufs[in ^ 1];
      if (fuse_target) {
        const sfm::AdvanceView av{bi[1], bi[2], &w.scal[cur], p.fire, pending,
                                  p.remove_drift, p.vv_dt, nullptr};
        if (int rc = sfm::launch_target_mesh(d->target, bi[0], w.prev_buf, ls, &av, true,
                                             w.target_list))
          return rc;
      }
      sfm::prof_begin(sfm::kProfMesh, ls);
      hipLaunchKernelGGL(integrate_shared2d_kernel<true>, dim3(tgrid), dim3(kBlock), 0, ls,
                         bi[0], bi[1], bi[2], prev_ptr, bo[0], bo[1], bo[2], p, &w.scal[cur],
                         &w.scal[cur ^ 1], cap0, w.tile_part, w.ticket, pending, tiles.nty,
                         tiles.ntx, plain_band_args(tiles, xcd_map, pack));
      sfm::prof_end(sfm::kProfMesh, ls);
      SFM_LAUNCH_CHECK();
      in ^= 1;
      if (p.fire) cur ^= 1;
    } else if (tiled) {
      SFM_STEP_DISPATCH(advance_kernel, d->x, d->v, d->a, p, &w.scal[cur],
                        &w.scal[cur ^ 1], w.partials, grid, pending ? 2 : 0, w.colsum);
      cur ^= 1;
      if (int rc = eval_prev(ls)) return rc;
      sfm::prof_begin(sfm::kProfMesh, ls);
      hipLaunchKernelGGL(integrate_shared2d_kernel<false>, dim3(tgrid), dim3(kBlock), 0, ls,
                         d->x, d->v, d->a, prev_ptr, d->x, d->v, d->a, p, &w.scal[cur],
                         &w.scal[cur ^ 1], cap0, w.tile_part, w.ticket, 1, tiles.nty,
                         tiles.ntx, plain_band_args(tiles, xcd_map, pack));
      sfm::prof_end(sfm::kProfMesh, ls);
      SFM_LAUNCH_CHECK();
      if (p.fire) cur ^= 1;
    } else {
      SFM_STEP_DISPATCH(advance_kernel, d->x, d->v, d->a, p, &w.scal[cur],
                        &w.scal[cur ^ 1], w.partials, part_rows, pending, w.colsum);
      cur ^= 1;
      if (int rc = eval_prev(ls)) return rc;
      if (int rc = external_force()) return rc;
      sfm::prof_begin(sfm::kProfMesh, ls);
      if (march.T) {
        if (int rc = launch_march3d(march, ls, d->x, d->v, d->a, prev_ptr, p, &w.scal[cur], cap0,
                                    w.partials))
          return rc;
      } else {
        SFM_STEP_DISPATCH(integrate_kernel, d->x, d->v, d->a, prev_ptr, p,
                          &w.scal[cur], cap0, w.partials);
      }
      sfm::prof_end(sfm::kProfMesh, ls);
      if (p.fire && p.drift_cols) {
        column_means(d->x, d->v);
        SFM_LAUNCH_CHECK();
      }
    }
    return SFM_OK;
  };
#undef SFM_STEP_DISPATCH

  int it = (small || persist3d_done) ? d->num_iters : 0;
  if (!small && !persist3d_done && d->num_iters > 0) {
    if (int rc = step(0)) return rc;
    it = 1;
  }
  for (; it < d->num_iters; ++it)
    if (int rc = step(1)) return rc;
  if (in == 1) {
    const size_t bytes = (size_t)p.ncomp * p.N * sizeof(float);
    SFM_HIP_CHECK(hipMemcpyAsync(d->x, w.alt[0], bytes, hipMemcpyDeviceToDevice, st));
    SFM_HIP_CHECK(hipMemcpyAsync(d->v, w.alt[1], bytes, hipMemcpyDeviceToDevice, st));
    SFM_HIP_CHECK(hipMemcpyAsync(d->a, w.alt[2], bytes, hipMemcpyDeviceToDevice, st));
  }
  SFM_MESH_DISPATCH(finish_kernel, d->x, d->v, p, &w.scal[cur],
                    &w.scal[cur ^ 1], w.partials, finish_mode == 1 ? part_rows : grid,
                    finish_mode, w.stat_part, w.colsum);
  cur ^= 1;
#undef SFM_MESH_DISPATCH
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(kBlock), 0, st, w.stat_part,
                     grid, w.stats);
  SFM_LAUNCH_CHECK();

  Scalars s1;
  float hs[2];
  SFM_HIP_CHECK(hipMemcpyAsync(&s1, &w.scal[cur], sizeof(s1),
                               hipMemcpyDeviceToHost, st));
  SFM_HIP_CHECK(hipMemcpyAsync(hs, w.stats, sizeof(hs), hipMemcpyDeviceToHost, st));
  SFM_HIP_CHECK(hipStreamSynchronize(st));
  if (p.fire) {
    fire->dt = s1.dt;
    fire->alpha = s1.alpha;
    fire->n_pos = s1.n_pos;
    fire->cap = s1.cap;
  }
  stats->e_kin = hs[0];
  stats->v_max = hs[1];
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// One mesh across GPUs: the multi-launch step split at its exchange points.
// A rank holds a band of rows [own_y0, own_y1) of every section plus the halo
// rows next to it.  Per step the caller (sofima_amd/dist.py)
//   1. exchanges (x, v, a) of the boundary rows with the neighbour bands and
//      all-gathers the per-band partial sums of the previous step into `sums`,
//   2. sfm_mesh_shard_advance: every band reduces `sums` in rank order -> the
//      same FIRE scalars everywhere; pending gate / drift; x += dt v + dt^2/2 a
//      on the owned AND the halo rows,
//   3. sfm_mesh_shard_integrate: force + velocity update; partial sums of the
//      owned rows -> `my_sums`.
// All of it is enqueued on desc->stream; nothing synchronises until
// sfm_mesh_shard_finish.
// ---------------------------------------------------------------------------
namespace {

// One row of sums from the per-block partials of integrate_kernel (fixed order).
__global__ void __launch_bounds__(kBlock)
shard_sums_kernel(const float* __restrict__ partials, int rows, float* __restrict__ out) {
  __shared__ float lds[kNP * kBlock];
  float acc[kNP];
  for (int i = 0; i < kNP; ++i) acc[i] = 0.f;
  for (int r = threadIdx.x; r < rows; r += kBlock)
    for (int i = 0; i < 7; ++i) acc[i] = acc[i] + partials[r * kNP + i];
  block_sum(acc, 7, lds);
  if (threadIdx.x == 0)
    for (int i = 0; i < kNP; ++i) out[i] = i < 7 ? acc[i] : 0.f;
}

int shard_setup(const SfmMeshDesc* d, const SfmMeshShard* sh, MeshParams* p,
                MeshWorkspace* w) {
  if (int rc = build_params(d, p)) return rc;
  if (!sh) return sfm::fail(SFM_ERR_INVALID, "shard is NULL");
  if (!d->x || !d->v || !d->a)
    return sfm::fail(SFM_ERR_INVALID, "x/v/a must be device pointers");
  if (d->target || d->prev_cb || d->force_kind == SFM_FORCE_EXTERNAL)
    return sfm::fail(SFM_ERR_INVALID, "band shards: prev_fn / external forces unsupported");
  if (d->remove_drift == 2)
    return sfm::fail(SFM_ERR_INVALID, "band shards: per-column drift removal unsupported");
  if (sh->own_y0 < 0 || sh->own_y1 > p->Y || sh->own_y0 >= sh->own_y1)
    return sfm::fail(SFM_ERR_INVALID, "band shards: owned rows [%d, %d) of %d",
                     sh->own_y0, sh->own_y1, p->Y);
  if (sh->n_ranks < 1 || sh->n_ranks > kMaxBlocks || !sh->sums || !sh->my_sums)
    return sfm::fail(SFM_ERR_INVALID, "band shards: sums buffers / n_ranks");
  if (sh->global_nodes < p->N / ((long long)p->Y) * (sh->own_y1 - sh->own_y0))
    return sfm::fail(SFM_ERR_INVALID, "band shards: global_nodes too small");
  p->own_y0 = sh->own_y0;
  p->own_y1 = sh->own_y1;
  p->n_f = static_cast<float>(sh->global_nodes);
  *w = carve_for(d, d->workspace, nullptr);
  if (!d->workspace || d->workspace_bytes < w->bytes)
    return sfm::fail(SFM_ERR_WORKSPACE, "mesh workspace needs %zu bytes, got %zu",
                     w->bytes, d->workspace_bytes);
  return SFM_OK;
}

#define SFM_SHARD_DISPATCH(KERNEL, ...)                                      \
  do {                                                                       \
    if (p.ncomp == 2)                                                        \
      hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(kBlock), 0, st,         \
                         __VA_ARGS__);                                       \
    else                                                                     \
      hipLaunchKernelGGL(KERNEL<3>, dim3(grid), dim3(kBlock), 0, st,         \
                         __VA_ARGS__);                                       \
    SFM_LAUNCH_CHECK();                                                      \
  } while (0)

}  // namespace

int sfm_mesh_shard_begin(const SfmMeshDesc* d, SfmMeshShard* sh,
                         const SfmFireState* fire) {
  MeshParams p;
  MeshWorkspace w;
  if (int rc = shard_setup(d, sh, &p, &w)) return rc;
  if (!fire) return sfm::fail(SFM_ERR_INVALID, "fire is NULL");
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int grid = grid_for(p.N);
  Scalars s0;
  std::memset(&s0, 0, sizeof(s0));
  s0.dt = fire->dt;
  s0.alpha = fire->alpha;
  s0.n_pos = 0;
  s0.cap = fire->cap;
  s0.gate = 1.f;
  SFM_HIP_CHECK(hipMemcpyAsync(&w.scal[0], &s0, sizeof(s0), hipMemcpyHostToDevice, st));
  sh->phase = 0;
  sh->cap0 = fire->cap;
  // a = F(x) + pull(prev, cap) on every local row; the halo rows' values are
  // replaced by the owners' at the first exchange
  SFM_SHARD_DISPATCH(force_kernel, d->x, d->prev, d->a, p, fire->cap, p.has_prev);
  return SFM_OK;
}

int sfm_mesh_shard_advance(const SfmMeshDesc* d, SfmMeshShard* sh) {
  MeshParams p;
  MeshWorkspace w;
  if (int rc = shard_setup(d, sh, &p, &w)) return rc;
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int grid = grid_for(p.N);
  const int cur = sh->phase & 1;
  const int pending = sh->phase > 0 ? 1 : 0;
  SFM_SHARD_DISPATCH(advance_kernel, d->x, d->v, d->a, p, &w.scal[cur], &w.scal[cur ^ 1],
                     sh->sums, sh->n_ranks, pending, w.colsum);
  sh->phase += 1;
  return SFM_OK;
}

int sfm_mesh_shard_integrate(const SfmMeshDesc* d, SfmMeshShard* sh) {
  MeshParams p;
  MeshWorkspace w;
  if (int rc = shard_setup(d, sh, &p, &w)) return rc;
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int grid = grid_for(p.N);
  const int cur = sh->phase & 1;
  sfm::prof_begin(sfm::kProfMesh, st);
  SFM_SHARD_DISPATCH(integrate_kernel, d->x, d->v, d->a, d->prev, p, &w.scal[cur],
                     sh->cap0, w.partials);
  sfm::prof_end(sfm::kProfMesh, st);
  if (p.fire) {
    hipLaunchKernelGGL(shard_sums_kernel, dim3(1), dim3(kBlock), 0, st, w.partials, grid,
                       sh->my_sums);
    SFM_LAUNCH_CHECK();
  }
  return SFM_OK;
}

int sfm_mesh_shard_finish(const SfmMeshDesc* d, SfmMeshShard* sh, SfmFireState* fire,
                          SfmChunkStats* stats) {
  MeshParams p;
  MeshWorkspace w;
  if (int rc = shard_setup(d, sh, &p, &w)) return rc;
  if (!fire || !stats) return sfm::fail(SFM_ERR_INVALID, "fire/stats is NULL");
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int grid = grid_for(p.N);
  int cur = sh->phase & 1;
  const int pending = sh->phase > 0 ? 1 : 0;
  SFM_SHARD_DISPATCH(finish_kernel, d->x, d->v, p, &w.scal[cur], &w.scal[cur ^ 1],
                     sh->sums, sh->n_ranks, pending, w.stat_part, w.colsum);
  cur ^= 1;
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(kBlock), 0, st, w.stat_part, grid,
                     w.stats);
  SFM_LAUNCH_CHECK();
  Scalars s1;
  float hs[2];
  SFM_HIP_CHECK(hipMemcpyAsync(&s1, &w.scal[cur], sizeof(s1), hipMemcpyDeviceToHost, st));
  SFM_HIP_CHECK(hipMemcpyAsync(hs, w.stats, sizeof(hs), hipMemcpyDeviceToHost, st));
  SFM_HIP_CHECK(hipStreamSynchronize(st));
  if (p.fire) {
    fire->dt = s1.dt;
    fire->alpha = s1.alpha;
    fire->n_pos = s1.n_pos;
    fire->cap = s1.cap;
  }
  stats->e_kin = hs[0];   // of the owned rows: the caller adds the bands up
  stats->v_max = hs[1];   // max over the owned rows
  return SFM_OK;
}
#undef SFM_SHARD_DISPATCH

}  // extern "C"

// ---------------------------------------------------------------------------
// sfm_mesh_relax_banded: ONE mesh as bands of rows, the whole chunk of steps in
// one C call (the step loop of sofima_amd/dist.py moved into the library).
//
// Every rank holds `n_local` consecutive bands; band g = rank * n_local + i.  A
// band's arrays are its owned rows plus one halo row per existing neighbour.
// Per step (mesh.py:436-499 split at its exchange points):
//
//   main stream                           comm stream
//   edge tile rows of every local band
//   -- event ---------------------------> rows at the band edges: device copies
//   interior tile rows (last workgroup       between local bands, pack +
//     of a band leaves its partial sums)     grouped RCCL send/recv + unpack
//   -- event --------------------------->    between ranks, into the halo rows
//                                          all-gather of the partial sums
//   <------------------------------------ event
//
// i.e. the rows a neighbour needs leave while the interior is still being
// integrated; only the 8-float all-gather (FIRE's `power` decides dt, alpha and
// the velocity gate of the next step: mesh.py:455-492) is exposed.  In-plane
// spring meshes run the fused tiled kernel (integrate_shared2d_kernel, state
// ping-pong); everything else takes the advance / integrate pair on one stream.
// ---------------------------------------------------------------------------
namespace {

constexpr int kMaxLocalBands = 16;
constexpr int kMaxRowJobs = 4 * kMaxLocalBands;

// Copies (x, v, a) of one mesh row between arrays of different row counts (a
// packed buffer is an array with one row per plane).
struct RowJob {
  const float* src[3];
  float* dst[3];
  long long src_n, dst_n;      // component stride (floats)
  long long src_plane, dst_plane;
  long long src_off, dst_off;  // row * X
};
struct RowJobs {
  int n;
  int ncomp, planes, X;
  RowJob job[kMaxRowJobs];
};

__global__ void __launch_bounds__(kBlock) band_rows_kernel(RowJobs jobs) {
  const RowJob& j = jobs.job[blockIdx.z];
  const int arr = blockIdx.y;
  const long long per = (long long)jobs.planes * jobs.X;
  for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < per * jobs.ncomp;
       i += (long long)gridDim.x * kBlock) {
    const int c = static_cast<int>(i / per);
    const long long r = i % per;
    const long long pl = r / jobs.X, xi = r % jobs.X;
    j.dst[arr][c * j.dst_n + pl * j.dst_plane + j.dst_off + xi] =
        j.src[arr][c * j.src_n + pl * j.src_plane + j.src_off + xi];
  }
}

struct BandState {
  MeshParams p;
  MeshWorkspace w;
  TilePlan tiles;
  float* set[2][3];     // (x, v, a) ping-pong sets; set[0] = the caller's arrays
  int grid;             // advance / integrate / finish grid
  bool has_lo, has_hi;  // neighbours (any rank)
  float* buf[4];        // packed rows: send lo, send hi, recv lo, recv hi
};

void launch_rows(const RowJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return;
  const long long per = (long long)jobs.planes * jobs.X * jobs.ncomp;
  const int gx = static_cast<int>(std::min<long long>((per + kBlock - 1) / kBlock, 64));
  hipLaunchKernelGGL(band_rows_kernel, dim3(gx, 3, jobs.n), dim3(kBlock), 0, st, jobs);
}

}  // namespace

extern "C" {

size_t sfm_mesh_banded_scratch_bytes(const SfmBandedDesc* b) {
  if (!b || !b->bands || b->n_local < 1) return 0;
  const SfmMeshDesc& d = b->bands[0];
  const size_t row = 3 * (size_t)d.ncomp * d.shape[0] * d.shape[1] * d.shape[3];
  sfm::Carver c(nullptr);
  const size_t total = (size_t)std::max(b->n_ranks, 1) * b->n_local;
  c.take<float>(2 * total * kNP);      // sums of all bands, double buffered by step parity
  c.take<float>(total * 2);            // e_kin, v_max of all bands
  c.take<BandDev>(2 * (size_t)b->n_local);  // multi-band launch tables (two parities)
  for (int i = 0; i < b->n_local; ++i)
    for (int k = 0; k < 4; ++k) c.take<float>(row);
  return c.total();
}

int sfm_mesh_relax_banded(const SfmBandedDesc* b, SfmFireState* fire, SfmChunkStats* stats) {
  if (!b || !b->bands || !b->shards || !fire || !stats)
    return sfm::fail(SFM_ERR_INVALID, "banded: NULL argument");
  const int nl = b->n_local;
  const int n_ranks = std::max(b->n_ranks, 1);
  if (nl < 1 || nl > kMaxLocalBands)
    return sfm::fail(SFM_ERR_INVALID, "banded: 1..%d bands per rank", kMaxLocalBands);
  if (b->rank < 0 || b->rank >= n_ranks) return sfm::fail(SFM_ERR_INVALID, "banded: rank");
  // host-staged transport: the inter-rank branch without RCCL peers
  const bool host_x = !b->comm && n_ranks > 1 && b->host_halo && b->host_allgather;
  if (n_ranks > 1 && !b->comm && !host_x)
    return sfm::fail(SFM_ERR_INVALID,
                     "banded: %d ranks need a communicator or the host_halo / host_allgather pair",
                     n_ranks);
  const bool transport = b->comm != nullptr || host_x;
  const bool loopback = (b->flags & SFM_BANDED_LOOPBACK) != 0;
  if (loopback && !b->comm)
    return sfm::fail(SFM_ERR_INVALID, "banded: loop-back needs a communicator");
  if (b->comm && (sfm::comm_size(b->comm) != n_ranks || sfm::comm_rank(b->comm) != b->rank))
    return sfm::fail(SFM_ERR_INVALID, "banded: communicator is rank %d of %d, desc says %d of %d",
                     sfm::comm_rank(b->comm), sfm::comm_size(b->comm), b->rank, n_ranks);
  const int total = n_ranks * nl;
  if (total > kMaxBlocks) return sfm::fail(SFM_ERR_INVALID, "banded: too many bands");
  const SfmMeshDesc& d0 = b->bands[0];
  hipStream_t st = static_cast<hipStream_t>(d0.stream);
  hipStream_t xs = b->comm_stream ? static_cast<hipStream_t>(b->comm_stream) : st;
  const int iters = d0.num_iters;
  if (iters < 0) return sfm::fail(SFM_ERR_INVALID, "num_iters < 0");

  // scratch
  const size_t need = sfm_mesh_banded_scratch_bytes(b);
  if (!b->scratch || b->scratch_bytes < need)
    return sfm::fail(SFM_ERR_WORKSPACE, "banded scratch needs %zu bytes, got %zu", need,
                     b->scratch_bytes);
  sfm::Carver carve_s(b->scratch);
  // Partial sums of all bands, one buffer per step parity: a band's kernel of
  // step k leaves its sums while a later band's kernel of the same step still
  // reads the sums of step k - 1.
  float* sums_buf = carve_s.take<float>((size_t)2 * total * kNP);
  auto sums_of = [&](int step) { return sums_buf + (size_t)(step & 1) * total * kNP; };
  float* stats_all = carve_s.take<float>((size_t)total * 2);
  BandDev* band_dev_all = carve_s.take<BandDev>(2 * (size_t)nl);
  BandDev* band_dev[2] = {band_dev_all, band_dev_all + nl};
  const size_t row_floats = (size_t)d0.ncomp * d0.shape[0] * d0.shape[1] * d0.shape[3];

  BandState bs[kMaxLocalBands];
  bool fused = true;
  for (int i = 0; i < nl; ++i) {
    const SfmMeshDesc& d = b->bands[i];
    SfmMeshShard& sh = b->shards[i];
    if (d.stream != d0.stream || d.num_iters != iters || d.ncomp != d0.ncomp ||
        d.shape[0] != d0.shape[0] || d.shape[1] != d0.shape[1] || d.shape[3] != d0.shape[3] ||
        d.fire != d0.fire || d.remove_drift != d0.remove_drift)
      return sfm::fail(SFM_ERR_INVALID, "banded: the bands of a mesh share shape and config");
    const int g = b->rank * nl + i;
    sh.n_ranks = total;
    sh.sums = sums_buf;
    sh.my_sums = sums_buf + (size_t)g * kNP;
    BandState& s = bs[i];
    if (int rc = shard_setup(&d, &sh, &s.p, &s.w)) return rc;
    s.w = carve_for(&d, d.workspace, &s.tiles);
    s.grid = grid_for(s.p.N);
    s.has_lo = g > 0;
    s.has_hi = g < total - 1;
    if (s.has_lo != (sh.own_y0 > 0) || s.has_hi != (sh.own_y1 < s.p.Y) ||
        sh.own_y0 > 1 || s.p.Y - sh.own_y1 > 1)
      return sfm::fail(SFM_ERR_INVALID,
                       "banded: band %d of %d owns rows [%d, %d) of %d: one halo row per "
                       "existing neighbour", g, total, sh.own_y0, sh.own_y1, s.p.Y);
    for (int k = 0; k < 4; ++k) s.buf[k] = carve_s.take<float>(3 * row_floats);
    s.set[0][0] = d.x;
    s.set[0][1] = d.v;
    s.set[0][2] = d.a;
    for (int k = 0; k < 3; ++k) s.set[1][k] = s.w.alt[k];
    fused = fused && s.p.ncomp == 2 && s.p.force_kind == SFM_FORCE_SPRINGS &&
            s.tiles.tx == kSX && s.w.alt[0] != nullptr;
  }
  // The second stream pays for itself only when edge rows really travel (RCCL):
  // between bands of one process the exchange is one small copy kernel.
  const bool overlap = fused && xs != st && transport && !(b->flags & SFM_BANDED_NO_OVERLAP);
  const int C = d0.ncomp;
  const int planes = d0.shape[0] * d0.shape[1], X = d0.shape[3];

  // Row jobs of one exchange on buffer set `q`: `before` runs ahead of the RCCL
  // group (local copies + packing), `after` behind it (unpacking).
  auto array_side = [&](RowJob* j, bool src, const BandState& s, int q, int row) {
    for (int k = 0; k < 3; ++k) {
      if (src) j->src[k] = s.set[q][k]; else j->dst[k] = s.set[q][k];
    }
    const long long n = s.p.N, pl = (long long)s.p.Y * X, off = (long long)row * X;
    if (src) { j->src_n = n; j->src_plane = pl; j->src_off = off; }
    else { j->dst_n = n; j->dst_plane = pl; j->dst_off = off; }
  };
  auto buffer_side = [&](RowJob* j, bool src, float* buf) {
    for (int k = 0; k < 3; ++k) {
      float* base = buf + (size_t)k * row_floats;
      if (src) j->src[k] = base; else j->dst[k] = base;
    }
    const long long n = (long long)planes * X;
    if (src) { j->src_n = n; j->src_plane = X; j->src_off = 0; }
    else { j->dst_n = n; j->dst_plane = X; j->dst_off = 0; }
  };
  auto build_jobs = [&](int q, RowJobs* before, RowJobs* after) {
    before->n = after->n = 0;
    before->ncomp = after->ncomp = C;
    before->planes = after->planes = planes;
    before->X = after->X = X;
    for (int i = 0; i < nl; ++i) {
      BandState& s = bs[i];
      const SfmMeshShard& sh = b->shards[i];
      const bool lo_remote = s.has_lo && (i == 0 || loopback);
      const bool hi_remote = s.has_hi && (i == nl - 1 || loopback);
      if (s.has_hi && !hi_remote) {
        // local pair (i, i + 1): my last owned row -> its low halo row and back
        BandState& t = bs[i + 1];
        RowJob* j = &before->job[before->n++];
        array_side(j, true, s, q, sh.own_y1 - 1);
        array_side(j, false, t, q, b->shards[i + 1].own_y0 - 1);
        j = &before->job[before->n++];
        array_side(j, true, t, q, b->shards[i + 1].own_y0);
        array_side(j, false, s, q, sh.own_y1);
      }
      if (lo_remote) {
        RowJob* j = &before->job[before->n++];
        array_side(j, true, s, q, sh.own_y0);
        buffer_side(j, false, s.buf[0]);
        j = &after->job[after->n++];
        buffer_side(j, true, s.buf[2]);
        array_side(j, false, s, q, sh.own_y0 - 1);
      }
      if (hi_remote) {
        RowJob* j = &before->job[before->n++];
        array_side(j, true, s, q, sh.own_y1 - 1);
        buffer_side(j, false, s.buf[1]);
        j = &after->job[after->n++];
        buffer_side(j, true, s.buf[3]);
        array_side(j, false, s, q, sh.own_y1);
      }
    }
  };
  RowJobs before[2], after[2];
  build_jobs(0, &before[0], &after[0]);
  if (fused) build_jobs(1, &before[1], &after[1]);

  // Multi-band launch tables of the fused step, one per parity q of the input
  // set (the scalars and the sums buffers alternate with it).
  int grid_mode[3] = {0, 0, 0};
  BandDev host[2][kMaxLocalBands];   // (function scope: source of async copies)
  if (fused) {
    std::memset(host, 0, sizeof(host));
    int base[3] = {0, 0, 0};
    for (int i = 0; i < nl; ++i) {
      const BandState& sb = bs[i];
      const SfmMeshShard& sh = b->shards[i];
      const int ty_a = sh.own_y0 / kSY, ty_b = (sh.own_y1 - 1) / kSY;
      const int n_edge = ty_a == ty_b ? 1 : 2;
      const int count[3] = {static_cast<int>(sb.tiles.tiles),
                            planes * std::min(n_edge, sb.tiles.nty) * sb.tiles.ntx,
                            planes * std::max(sb.tiles.nty - n_edge, 0) * sb.tiles.ntx};
      const int g = b->rank * nl + i;
      for (int q = 0; q < 2; ++q) {
        BandDev& h = host[q][i];
        for (int k = 0; k < 3; ++k) {
          h.in[k] = sb.set[q][k];
          h.out[k] = sb.set[q ^ 1][k];
        }
        h.prev = b->bands[i].prev;
        h.partials = sb.w.tile_part;
        h.ticket = sb.w.ticket;
        h.my_sums = sums_buf + (size_t)q * total * kNP + (size_t)g * kNP;
        h.scal_in = &sb.w.scal[q];
        h.scal_out = &sb.w.scal[q ^ 1];
        // lo side: my first owned row -> the high halo row of the band below
        for (int sd = 0; sd < 2; ++sd) {
          const bool exists = sd == 0 ? sb.has_lo : sb.has_hi;
          if (!exists) continue;
          const bool local = !loopback && (sd == 0 ? i > 0 : i < nl - 1);
          if (local) {
            const int j = sd == 0 ? i - 1 : i + 1;
            const int row = sd == 0 ? b->shards[j].own_y1 : b->shards[j].own_y0 - 1;
            for (int k = 0; k < 3; ++k) h.nb[sd][k] = bs[j].set[q ^ 1][k];
            h.nb_n[sd] = bs[j].p.N;
            h.nb_plane[sd] = (long long)bs[j].p.Y * X;
            h.nb_off[sd] = (long long)row * X;
          } else {
            for (int k = 0; k < 3; ++k) h.nb[sd][k] = sb.buf[sd] + (size_t)k * row_floats;
            h.nb_n[sd] = (long long)planes * X;
            h.nb_plane[sd] = X;
            h.nb_off[sd] = 0;
          }
        }
        h.N = sb.p.N;
        h.Y = sb.p.Y;
        h.own_y0 = sh.own_y0;
        h.own_y1 = sh.own_y1;
        h.nty = sb.tiles.nty;
        h.tiles = static_cast<int>(sb.tiles.tiles);
        h.ty_a = ty_a;
        h.ty_b = ty_b;
        for (int m = 0; m < 3; ++m) h.base[m] = base[m];
      }
      for (int m = 0; m < 3; ++m) base[m] += count[m];
    }
    for (int m = 0; m < 3; ++m) grid_mode[m] = base[m];
    SFM_HIP_CHECK(hipMemcpyAsync(band_dev[0], host[0], sizeof(BandDev) * nl,
                                 hipMemcpyHostToDevice, st));
    SFM_HIP_CHECK(hipMemcpyAsync(band_dev[1], host[1], sizeof(BandDev) * nl,
                                 hipMemcpyHostToDevice, st));
  }

  // The grouped point-to-point part of one exchange.  Between ranks: the first
  // band's low edge <-> rank - 1, the last band's high edge <-> rank + 1.  Loop
  // back (tests on one GPU): every edge between local bands travels through a
  // self send / recv -- sends and receives to the same peer match in order.
  const size_t cnt = 3 * row_floats;
  // Host staging of the transport callbacks (pageable memory: the copies return
  // when they are done).  Parts: send lo, send hi, recv lo, recv hi.
  std::vector<float> stage_rows, stage_gather;
  if (host_x) {
    stage_rows.resize(4 * cnt);
    stage_gather.resize((size_t)total * std::max<size_t>(kNP, 2));
  }
  auto host_p2p = [&](hipStream_t s_) -> int {
    const bool lo = bs[0].has_lo && b->rank > 0;
    const bool hi = bs[nl - 1].has_hi && b->rank < n_ranks - 1;
    if (!lo && !hi) return SFM_OK;
    float* h = stage_rows.data();
    if (lo) SFM_HIP_CHECK(hipMemcpyAsync(h, bs[0].buf[0], cnt * sizeof(float), hipMemcpyDeviceToHost, s_));
    if (hi) SFM_HIP_CHECK(hipMemcpyAsync(h + cnt, bs[nl - 1].buf[1], cnt * sizeof(float), hipMemcpyDeviceToHost, s_));
    SFM_HIP_CHECK(hipStreamSynchronize(s_));
    if (b->host_halo(b->host_user, lo ? b->rank - 1 : -1, lo ? h : nullptr, lo ? h + 2 * cnt : nullptr,
                     hi ? b->rank + 1 : -1, hi ? h + cnt : nullptr, hi ? h + 3 * cnt : nullptr, cnt) != 0)
      return sfm::fail(SFM_ERR_INVALID, "banded: host_halo callback failed");
    if (lo) SFM_HIP_CHECK(hipMemcpyAsync(bs[0].buf[2], h + 2 * cnt, cnt * sizeof(float), hipMemcpyHostToDevice, s_));
    if (hi) SFM_HIP_CHECK(hipMemcpyAsync(bs[nl - 1].buf[3], h + 3 * cnt, cnt * sizeof(float), hipMemcpyHostToDevice, s_));
    SFM_HIP_CHECK(hipStreamSynchronize(s_));   // the staging area is reused
    return SFM_OK;
  };
  // recv[r * count ..) = the `count` floats at recv + rank * count of rank r (in place)
  auto allgather = [&](float* recv, size_t count, hipStream_t s_) -> int {
    if (b->comm) return sfm_comm_allgather(b->comm, recv + (size_t)b->rank * count, recv, count, s_);
    float* h = stage_gather.data();
    SFM_HIP_CHECK(hipMemcpyAsync(h + (size_t)b->rank * count, recv + (size_t)b->rank * count,
                                 count * sizeof(float), hipMemcpyDeviceToHost, s_));
    SFM_HIP_CHECK(hipStreamSynchronize(s_));
    if (b->host_allgather(b->host_user, h + (size_t)b->rank * count, h, count) != 0)
      return sfm::fail(SFM_ERR_INVALID, "banded: host_allgather callback failed");
    SFM_HIP_CHECK(hipMemcpyAsync(recv, h, (size_t)n_ranks * count * sizeof(float), hipMemcpyHostToDevice, s_));
    SFM_HIP_CHECK(hipStreamSynchronize(s_));
    return SFM_OK;
  };
  auto p2p = [&](hipStream_t s_) -> int {
    if (host_x) return host_p2p(s_);
    if (!b->comm) return SFM_OK;
    bool any = false;
    for (int i = 0; i < nl; ++i)
      any = any || (bs[i].has_lo && (i == 0 || loopback)) || (bs[i].has_hi && (i == nl - 1 || loopback));
    if (!any) return SFM_OK;
    if (int rc = sfm::comm_group_begin(b->comm)) return rc;
    int rc = SFM_OK;
    auto note = [&](int r) { if (!rc) rc = r; };
    if (bs[0].has_lo && b->rank > 0) {
      note(sfm::comm_send(b->comm, bs[0].buf[0], cnt, b->rank - 1, s_));
      note(sfm::comm_recv(b->comm, bs[0].buf[2], cnt, b->rank - 1, s_));
    }
    if (bs[nl - 1].has_hi && b->rank < n_ranks - 1) {
      note(sfm::comm_send(b->comm, bs[nl - 1].buf[1], cnt, b->rank + 1, s_));
      note(sfm::comm_recv(b->comm, bs[nl - 1].buf[3], cnt, b->rank + 1, s_));
    }
    if (loopback) {
      for (int i = 0; i + 1 < nl; ++i) {
        // band i's high edge -> band i + 1's low halo, and the reverse
        note(sfm::comm_send(b->comm, bs[i].buf[1], cnt, b->rank, s_));
        note(sfm::comm_recv(b->comm, bs[i + 1].buf[2], cnt, b->rank, s_));
        note(sfm::comm_send(b->comm, bs[i + 1].buf[0], cnt, b->rank, s_));
        note(sfm::comm_recv(b->comm, bs[i].buf[3], cnt, b->rank, s_));
      }
    }
    return sfm::comm_group_end(b->comm, rc);
  };
  auto exchange = [&](int q, hipStream_t s_) -> int {
    launch_rows(before[q], s_);
    SFM_LAUNCH_CHECK();
    if (int rc = p2p(s_)) return rc;
    launch_rows(after[q], s_);
    SFM_LAUNCH_CHECK();
    return SFM_OK;
  };
  auto gather_sums = [&](int step, hipStream_t s_) -> int {
    if (n_ranks == 1 && !loopback) return SFM_OK;   // my_sums are rows of the buffer already
    // in place: this rank's rows sit at their final position
    return allgather(sums_of(step), (size_t)nl * kNP, s_);
  };

  hipEvent_t ev_edge = nullptr, ev_int = nullptr, ev_x = nullptr;
  struct EventGuard {
    hipEvent_t* e[3];
    ~EventGuard() { for (auto p : e) if (*p) (void)hipEventDestroy(*p); }
  } guard{{&ev_edge, &ev_int, &ev_x}};
  if (overlap) {
    SFM_HIP_CHECK(hipEventCreateWithFlags(&ev_edge, hipEventDisableTiming));
    SFM_HIP_CHECK(hipEventCreateWithFlags(&ev_int, hipEventDisableTiming));
    SFM_HIP_CHECK(hipEventCreateWithFlags(&ev_x, hipEventDisableTiming));
  }
  // An error return between a fork (the exchange stream waits for the main
  // stream) and its join must not leave the exchange stream running behind the
  // caller's back: the guard joins it into the main stream on the way out.
  struct JoinGuard {
    hipStream_t st, xs;
    hipEvent_t* ev;
    bool forked;
    ~JoinGuard() {
      if (forked && *ev && hipEventRecord(*ev, xs) == hipSuccess) (void)hipStreamWaitEvent(st, *ev, 0);
    }
  } join{st, xs, &ev_x, false};

  // -- begin: scalars, a = F(x) + pull on the local rows of every band --------
  Scalars s0;
  std::memset(&s0, 0, sizeof(s0));
  s0.dt = fire->dt;
  s0.alpha = fire->alpha;
  s0.n_pos = 0;
  s0.cap = fire->cap;
  s0.gate = 1.f;
  const float cap0 = fire->cap;
  for (int i = 0; i < nl; ++i) {
    BandState& s = bs[i];
    const SfmMeshDesc& d = b->bands[i];
    SFM_HIP_CHECK(hipMemcpyAsync(&s.w.scal[0], &s0, sizeof(s0), hipMemcpyHostToDevice, st));
    if (fused) {
      SFM_HIP_CHECK(hipMemsetAsync(s.w.ticket, 0, 2 * sizeof(int), st));
      SFM_HIP_CHECK(hipMemsetAsync(s.w.tile_part, 0, (size_t)s.tiles.tiles * kNP * sizeof(u64), st));
    }
    if (s.p.ncomp == 2)
      hipLaunchKernelGGL(force_kernel<2>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.prev, d.a,
                         s.p, cap0, s.p.has_prev);
    else
      hipLaunchKernelGGL(force_kernel<3>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.prev, d.a,
                         s.p, cap0, s.p.has_prev);
    SFM_LAUNCH_CHECK();
  }
  // Everything that touches the communicator runs on the comm stream when there
  // is one (fork from / join into the main stream around `fn`).
  auto on_comm_stream = [&](auto fn) -> int {
    if (!overlap) return fn(st);
    SFM_HIP_CHECK(hipEventRecord(ev_int, st));
    SFM_HIP_CHECK(hipStreamWaitEvent(xs, ev_int, 0));
    join.forked = true;
    if (int rc = fn(xs)) return rc;
    SFM_HIP_CHECK(hipEventRecord(ev_x, xs));
    SFM_HIP_CHECK(hipStreamWaitEvent(st, ev_x, 0));
    join.forked = false;
    return SFM_OK;
  };
  // the halo rows' a (and, from the second chunk on, nothing else) is stale
  if (int rc = on_comm_stream([&](hipStream_t s_) { return exchange(0, s_); })) return rc;

  int xcd_opt = -1;   // SFM_MESH_XCD: 0 off, 1 from 64 tiles on, default from 2048 on
  {
    const std::string xo = sfm::option_str("SFM_MESH_XCD");
    if (!xo.empty() && (xo[0] == '0' || xo[0] == '1')) xcd_opt = xo[0] - '0';
  }
  int in = 0, cur = 0;
  for (int k = 0; k < iters; ++k) {
    const int pending = k > 0 ? 1 : 0;
    if (fused) {
      const int out = in ^ 1;
      // every local band in ONE launch (BandDev table of this parity); the rows at
      // the band edges land in the neighbours' halo rows / the send buffers
      auto launch = [&](int mode, int grid) {
        if (grid <= 0) return;
        // (XCD-contiguous tile order like the un-split step: the band of a block is
        // looked up after the remap, so a run may span bands)
        const int xcd = xcd_opt == 0 ? 0 : (xcd_opt == 1 ? grid >= 64 : grid >= 2048);
        BandArgs ba{sums_of(k - 1), nullptr, total, 0, mode, 0, 0, band_dev[in], nl, xcd};
        hipLaunchKernelGGL((integrate_shared2d_kernel<true, true>), dim3(grid), dim3(kBlock), 0, st,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, bs[0].p,
                           nullptr, nullptr, cap0, nullptr, nullptr, pending ? 3 : 0, 0,
                           bs[0].tiles.ntx, ba);
      };
      auto exchange_fused = [&](hipStream_t s_) -> int {
        if (int rc = p2p(s_)) return rc;
        launch_rows(after[out], s_);
        SFM_LAUNCH_CHECK();
        return SFM_OK;
      };
      sfm::prof_begin(sfm::kProfMesh, st);
      if (overlap) {
        launch(1, grid_mode[1]);
        SFM_LAUNCH_CHECK();
        SFM_HIP_CHECK(hipEventRecord(ev_edge, st));
        SFM_HIP_CHECK(hipStreamWaitEvent(xs, ev_edge, 0));
        join.forked = true;
        if (int rc = exchange_fused(xs)) return rc;
        launch(2, grid_mode[2]);
        SFM_LAUNCH_CHECK();
        sfm::prof_end(sfm::kProfMesh, st);
        SFM_HIP_CHECK(hipEventRecord(ev_int, st));
        SFM_HIP_CHECK(hipStreamWaitEvent(xs, ev_int, 0));
        if (d0.fire)
          if (int rc = gather_sums(k, xs)) return rc;
        SFM_HIP_CHECK(hipEventRecord(ev_x, xs));
        SFM_HIP_CHECK(hipStreamWaitEvent(st, ev_x, 0));
        join.forked = false;
      } else {
        launch(0, grid_mode[0]);
        SFM_LAUNCH_CHECK();
        sfm::prof_end(sfm::kProfMesh, st);
        if (int rc = exchange_fused(st)) return rc;
        if (d0.fire)
          if (int rc = gather_sums(k, st)) return rc;
      }
      in = out;
      if (d0.fire) cur ^= 1;
    } else {
      // advance / integrate pair in place; the rows were exchanged after the
      // previous integrate (or by `begin`)
      for (int i = 0; i < nl; ++i) {
        BandState& s = bs[i];
        const SfmMeshDesc& d = b->bands[i];
        if (s.p.ncomp == 2)
          hipLaunchKernelGGL(advance_kernel<2>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.v, d.a,
                             s.p, &s.w.scal[cur], &s.w.scal[cur ^ 1], sums_of(k - 1), total,
                             pending, s.w.colsum);
        else
          hipLaunchKernelGGL(advance_kernel<3>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.v, d.a,
                             s.p, &s.w.scal[cur], &s.w.scal[cur ^ 1], sums_of(k - 1), total,
                             pending, s.w.colsum);
        SFM_LAUNCH_CHECK();
      }
      cur ^= 1;
      sfm::prof_begin(sfm::kProfMesh, st);
      for (int i = 0; i < nl; ++i) {
        BandState& s = bs[i];
        const SfmMeshDesc& d = b->bands[i];
        if (s.p.ncomp == 2)
          hipLaunchKernelGGL(integrate_kernel<2>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.v, d.a,
                             d.prev, s.p, &s.w.scal[cur], cap0, s.w.partials);
        else
          hipLaunchKernelGGL(integrate_kernel<3>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.v, d.a,
                             d.prev, s.p, &s.w.scal[cur], cap0, s.w.partials);
        SFM_LAUNCH_CHECK();
        if (s.p.fire) {
          hipLaunchKernelGGL(shard_sums_kernel, dim3(1), dim3(kBlock), 0, st, s.w.partials,
                             s.grid, sums_of(k) + (size_t)(b->rank * nl + i) * kNP);
          SFM_LAUNCH_CHECK();
        }
      }
      sfm::prof_end(sfm::kProfMesh, st);
      if (int rc = exchange(0, st)) return rc;
      if (d0.fire)
        if (int rc = gather_sums(k, st)) return rc;
    }
  }
  const float* sums_last = sums_of(iters - 1);

  // -- finish: pending gate / drift of the last step, statistics ---------------
  const Scalars* final_scal = nullptr;
  for (int i = 0; i < nl; ++i) {
    BandState& s = bs[i];
    const SfmMeshDesc& d = b->bands[i];
    if (in == 1) {
      const size_t bytes = (size_t)s.p.ncomp * s.p.N * sizeof(float);
      for (int k = 0; k < 3; ++k)
        SFM_HIP_CHECK(hipMemcpyAsync(s.set[0][k], s.set[1][k], bytes, hipMemcpyDeviceToDevice, st));
    }
    int mode = iters > 0 ? 1 : 0;
    int c = cur;
    if (fused && iters > 0 && s.p.fire) {
      // same (band-order) reduction as the step kernels', then finish with the
      // scalars as they are
      hipLaunchKernelGGL(band_scalars_kernel, dim3(1), dim3(64), 0, st, &s.w.scal[c],
                         &s.w.scal[c ^ 1], sums_last, total, s.p);
      SFM_LAUNCH_CHECK();
      c ^= 1;
      mode = 2;
    }
    if (s.p.ncomp == 2)
      hipLaunchKernelGGL(finish_kernel<2>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.v, s.p,
                         &s.w.scal[c], &s.w.scal[c ^ 1], sums_last, total, mode, s.w.stat_part,
                         s.w.colsum);
    else
      hipLaunchKernelGGL(finish_kernel<3>, dim3(s.grid), dim3(kBlock), 0, st, d.x, d.v, s.p,
                         &s.w.scal[c], &s.w.scal[c ^ 1], sums_last, total, mode, s.w.stat_part,
                         s.w.colsum);
    SFM_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(kBlock), 0, st, s.w.stat_part, s.grid,
                       stats_all + (size_t)(b->rank * nl + i) * 2);
    SFM_LAUNCH_CHECK();
    // scal[c ^ 1]: the chunk's final scalars (identical in every band)
    if (i == 0) final_scal = &s.w.scal[c ^ 1];
  }
  if (n_ranks > 1)
    if (int rc = on_comm_stream([&](hipStream_t s_) {
          return allgather(stats_all, (size_t)nl * 2, s_);
        }))
      return rc;
  Scalars s1;
  float hs[2 * kMaxBlocks];
  SFM_HIP_CHECK(hipMemcpyAsync(&s1, final_scal, sizeof(s1), hipMemcpyDeviceToHost, st));
  SFM_HIP_CHECK(hipMemcpyAsync(hs, stats_all, sizeof(float) * 2 * total, hipMemcpyDeviceToHost, st));
  SFM_HIP_CHECK(hipStreamSynchronize(st));
  if (d0.fire) {
    fire->dt = s1.dt;
    fire->alpha = s1.alpha;
    fire->n_pos = s1.n_pos;
    fire->cap = s1.cap;
  }
  float ek = 0.f, vm = 0.f;
  for (int g = 0; g < total; ++g) {   // band order, float32 like the device sums
    ek = ek + hs[2 * g];
    vm = std::max(vm, hs[2 * g + 1]);
  }
  stats->e_kin = ek;
  stats->v_max = vm;
  return SFM_OK;
}

}  // extern "C"
