// Device code shared by the target-mesh kernel (sfm_maps.hip) and the fused
// volumetric montage step (sfm_mesh.hip): JAX's order-1 map_coordinates, the
// per-tile neighbour entries of stitch_elastic.compute_target_mesh
// (stitch_elastic.py:456-676) and the target of ONE node.  Both translation
// units are compiled with -ffp-contract=off: the same float operations.
#ifndef SFM_TARGET_H_
#define SFM_TARGET_H_

#include "sfm_common.h"

namespace sfm_target {

// order-1 map_coordinates of JAX: per axis lower = floor(q), weights
// (1 - t, t); corners visited lower-before-upper with the first axis
// outermost; products of weights times the corner value summed left to right;
// mode constant: a corner with any index out of range contributes cval = NaN.
struct Axis {
  int lo;
  float w_lo, w_hi;
};

__device__ __forceinline__ Axis make_axis(float q) {
  Axis a;
  const float f = floorf(q);
  a.w_hi = q - f;
  a.w_lo = 1.0f - a.w_hi;
  a.lo = static_cast<int>(f);
  return a;
}

// One component plane of a mesh as the sampler sees it: the stored values, or
// the positions after the running step's position update (sfm::AdvanceView:
// the expression of advance_kernel in sfm_mesh.hip, operation for operation).
struct Plane {
  const float* x;
  const float* v;   // ADV only
  const float* a;
  float dt, c2, gate, mx, mv;   // no pending gate / drift: gate = 1, mx = mv = 0
  // per-x-column drift means of the previous step (5-D states, mesh.py:496-497)
  // of x / v of this component, or nullptr
  const float* cs_x;
  const float* cs_v;
  // MODE 0: stored values; 1: advanced positions; 2: advanced, per-column drift
  // means pending.  A compile-time choice: a load under a run-time condition is
  // waited for on its own (the in-plane montage step lost 11 us to one such
  // branch).  Within a mode the expression is branch free: v * 1, x - 0 and
  // v - 0 are exact, so one form serves "nothing pending" too.
  template <int MODE>
  __device__ __forceinline__ float at(long long i, int xi) const {
    if (MODE == 0) return x[i];
    if (MODE == 2) {   // advance_kernel with drift_cols, operation for operation
      const float vv = v[i] * gate;
      const float xv = x[i] - cs_x[xi];
      const float vw = vv - cs_v[xi] * gate;
      return xv + (dt * vw + c2 * a[i]);
    }
    const float xv = x[i] - mx;
    const float vv = v[i] * gate - mv;
    return xv + (dt * vv + c2 * a[i]);
  }
};

__device__ __forceinline__ Plane plain(const float* m) {
  return Plane{m, nullptr, nullptr, 0.f, 0.f, 1.f, 0.f, 0.f, nullptr, nullptr};
}

// Bilinear sample of plane `m` [ny, nx] + ref (ref = offset + index * step
// along `ref_axis`) at (qy, qx).
template <int MODE = 0>
__device__ inline float sample2(const Plane& m, int ny, int nx, float qy,
                         float qx, bool constant, int ref_axis, float ref_off,
                         float ref_step) {
  if (isnan(qy) || isnan(qx)) return NAN;
  const Axis ay = make_axis(qy), ax = make_axis(qx);
  float sum = 0.f;
  bool first = true;
#pragma unroll
  for (int cy = 0; cy < 2; ++cy)
#pragma unroll
    for (int cx = 0; cx < 2; ++cx) {
      int iy = ay.lo + cy, ix = ax.lo + cx;
      const float w = (cy ? ay.w_hi : ay.w_lo) * (cx ? ax.w_hi : ax.w_lo);
      bool valid = iy >= 0 && iy < ny && ix >= 0 && ix < nx;
      iy = min(max(iy, 0), ny - 1);
      ix = min(max(ix, 0), nx - 1);
      float v = m.template at<MODE>((long long)iy * nx + ix, ix) +
                (ref_off + static_cast<float>(ref_axis == 0 ? iy : ix)) * ref_step;
      if (constant && !valid) v = NAN;
      const float t = w * v;
      sum = first ? t : sum + t;
      first = false;
    }
  return sum;
}

template <int MODE = 0>
__device__ inline float sample3(const Plane& m, int nz, int ny, int nx,
                         float qz, float qy, float qx, bool constant, int ref_axis,
                         float ref_off, float ref_step) {
  if (isnan(qz) || isnan(qy) || isnan(qx)) return NAN;
  const Axis az = make_axis(qz), ay = make_axis(qy), ax = make_axis(qx);
  float sum = 0.f;
  bool first = true;
#pragma unroll
  for (int cz = 0; cz < 2; ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        int iz = az.lo + cz, iy = ay.lo + cy, ix = ax.lo + cx;
        const float w = ((cz ? az.w_hi : az.w_lo) * (cy ? ay.w_hi : ay.w_lo)) *
                        (cx ? ax.w_hi : ax.w_lo);
        bool valid = iz >= 0 && iz < nz && iy >= 0 && iy < ny && ix >= 0 && ix < nx;
        iz = min(max(iz, 0), nz - 1);
        iy = min(max(iy, 0), ny - 1);
        ix = min(max(ix, 0), nx - 1);
        const int ri = ref_axis == 0 ? iz : (ref_axis == 1 ? iy : ix);
        float v = m.template at<MODE>(((long long)iz * ny + iy) * nx + ix, ix) +
                  (ref_off + static_cast<float>(ri)) * ref_step;
        if (constant && !valid) v = NAN;
        const float t = w * v;
        sum = first ? t : sum + t;
        first = false;
      }
  return sum;
}

// NeighborInfo field indices (stitch_elastic.py:43-72).
enum { kNbor = 0, kFlow = 1, kOffOrtho = 2, kSizeOrtho = 3, kSizeOverlap = 4,
       kFineX = 5, kFineY = 6, kDim = 7, kOffZ = 8, kSizeZ = 9, kFineZ = 10 };

// One thread per (tile, node).  The reference pastes the four neighbour
// updates in order into a NaN canvas, keeping the previous value where the
// update is NaN (per component); the last non-NaN update wins.  In-plane
// montages (ncomp 2, one section) and volumetric ones (ncomp 3: z start /
// target offsets and the z fine offset of stitch_elastic.py:509-518, 544-561).
// Everything about one neighbour entry that does not depend on the node,
// derived once per workgroup (a workgroup works on ONE tile) and held in
// scalar registers.
struct NbEntry {
  int valid, mult, dim;
  int tg[3], st[3], fsz[3];  // zyx: paste origin, compose start, flow size
  int n_f, fi, nb_i;
  int fine[3];               // x, y, z fine offsets times mult
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Everything about neighbour entry j of `tile` that does not depend on the node.
__device__ inline NbEntry make_entry(const SfmTargetMeshDesc& d, int tile, int j) {
  const int nc = d.ncomp;
  const int mz = d.mesh_shape[0], my = d.mesh_shape[1], mx = d.mesh_shape[2];
  const int* nb = d.nbors + ((long long)tile * 4 + j) * d.nbor_fields;
    NbEntry e;
    const int nbor = nb[kNbor];
    e.valid = nbor != -1;
    const int flow_idx = nb[kFlow];
    e.dim = nb[kDim] == 0 ? 0 : 1;
    e.mult = nbor == flow_idx ? 1 : -1;
    const int off_ortho = nb[kOffOrtho];
    const int f_ortho = nb[kSizeOrtho], f_overlap = nb[kSizeOverlap];
    const int* fshape = e.dim == 0 ? d.fx_shape : d.fy_shape;
    e.n_f = e.dim == 0 ? d.n_fx : d.n_fy;
    for (int k = 0; k < 3; ++k) e.fsz[k] = fshape[k];
    // size of the neighbour mesh along / across the overlap direction
    const int par_n = e.dim == 0 ? mx : my;
    const int ortho_n = e.dim == 0 ? my : mx;
    const int start_par = e.mult == 1 ? par_n - f_overlap : 0;
    const bool s_hi = (e.mult == 1 && off_ortho > 0) || (e.mult == -1 && off_ortho < 0);
    const int start_ortho = s_hi ? ortho_n - f_ortho : 0;
    e.st[1] = e.dim == 0 ? start_ortho : start_par;
    e.st[2] = e.dim == 0 ? start_par : start_ortho;
    const int tg_par = e.mult == 1 ? 0 : par_n - f_overlap;
    const bool t_hi = (e.mult == 1 && off_ortho < 0) || (e.mult == -1 && off_ortho > 0);
    const int tg_ortho = t_hi ? ortho_n - f_ortho : 0;
    e.tg[1] = e.dim == 0 ? tg_ortho : tg_par;
    e.tg[2] = e.dim == 0 ? tg_par : tg_ortho;
    e.st[0] = e.tg[0] = 0;
    e.fine[2] = 0;
    if (nc == 3) {
      const int off_z = nb[kOffZ], f_z = nb[kSizeZ];
      const bool sz_hi = (e.mult == 1 && off_z > 0) || (e.mult == -1 && off_z < 0);
      const bool tz_hi = (e.mult == 1 && off_z < 0) || (e.mult == -1 && off_z > 0);
      e.st[0] = sz_hi ? mz - f_z : 0;
      e.tg[0] = tz_hi ? mz - f_z : 0;
      e.fine[2] = e.mult * nb[kFineZ];
    }
    e.fine[0] = e.mult * nb[kFineX];
    e.fine[1] = e.mult * nb[kFineY];
    // jax clamps the dynamic indices; valid data never needs it
    e.fi = min(max(flow_idx, 0), e.n_f - 1);
    e.nb_i = min(max(nbor, 0), d.n_tiles - 1);
    return e;
}

// Position source of the neighbour meshes for target_node: plane_of(c, tile).
// Target of node (tz, ty, tx) of the tile whose entries are e[0..3] (workgroup
// uniform: a workgroup works on ONE tile): the four neighbour updates pasted in
// order, the last non-NaN one wins per component.  Returns whether the node lies
// in any paste region; *r stay NaN outside.
template <int MODE, int UNROLL_J = 4, typename PlaneOf>
__device__ __forceinline__ bool target_node(const SfmTargetMeshDesc& d, const NbEntry* s_e, int tz,
                                            int ty, int tx, PlaneOf plane_of, float* rx_out,
                                            float* ry_out, float* rz_out) {
  const int nc = d.ncomp;
  const int mz = d.mesh_shape[0], my = d.mesh_shape[1], mx = d.mesh_shape[2];
  const float sz = d.stride[0], sy = d.stride[1], sx = d.stride[2];
  float rx = NAN, ry = NAN, rz = NAN;
  bool in_region = false;
  // UNROLL_J = 1 keeps one copy of the sampler in a kernel that is short of registers
#pragma unroll UNROLL_J
  for (int j = 0; j < 4; ++j) {
    if (!uniform(s_e[j].valid)) continue;
    const int uz = tz - uniform(s_e[j].tg[0]);
    const int uy = ty - uniform(s_e[j].tg[1]);
    const int ux = tx - uniform(s_e[j].tg[2]);
    const int fz_n = uniform(s_e[j].fsz[0]), fy_n = uniform(s_e[j].fsz[1]),
              fx_n = uniform(s_e[j].fsz[2]);
    if (uz < 0 || uz >= fz_n || uy < 0 || uy >= fy_n || ux < 0 || ux >= fx_n) continue;
    in_region = true;
    const int dim = uniform(s_e[j].dim), n_f = uniform(s_e[j].n_f);
    const float* farr = dim == 0 ? d.fx : d.fy;
    const long long fvol = (long long)fz_n * fy_n * fx_n;
    const long long fo = (long long)uniform(s_e[j].fi) * fvol +
                         ((long long)uz * fy_n + uy) * fx_n + ux;
    const float fm = static_cast<float>(uniform(s_e[j].mult));
    const float m1x = fm * farr[fo];
    const float m1y = fm * farr[(long long)n_f * fvol + fo];
    // compose_maps_fast(flow @ start, neighbour mesh @ 0, mode constant)
    const float ref1x =
        (static_cast<float>(ux) + static_cast<float>(uniform(s_e[j].st[2]))) * sx;
    const float ref1y =
        (static_cast<float>(uy) + static_cast<float>(uniform(s_e[j].st[1]))) * sy;
    const float qx = (ref1x + m1x) / sx;
    const float qy = (ref1y + m1y) / sy;
    const int nb_i = uniform(s_e[j].nb_i);
    const Plane nx0 = plane_of(0, nb_i);
    const Plane nx1 = plane_of(1, nb_i);
    float ux_v, uy_v, uz_v = NAN;
    if (nc == 2) {
      ux_v = sample2<MODE>(nx0, my, mx, qy, qx, true, 1, 0.f, sx) - ref1x;
      uy_v = sample2<MODE>(nx1, my, mx, qy, qx, true, 0, 0.f, sy) - ref1y;
    } else {
      const float m1z = fm * farr[2LL * n_f * fvol + fo];
      const float ref1z =
          (static_cast<float>(uz) + static_cast<float>(uniform(s_e[j].st[0]))) * sz;
      const float qz = (ref1z + m1z) / sz;
      const Plane nx2 = plane_of(2, nb_i);
      ux_v = sample3<MODE>(nx0, mz, my, mx, qz, qy, qx, true, 2, 0.f, sx) - ref1x;
      uy_v = sample3<MODE>(nx1, mz, my, mx, qz, qy, qx, true, 1, 0.f, sy) - ref1y;
      uz_v = sample3<MODE>(nx2, mz, my, mx, qz, qy, qx, true, 0, 0.f, sz) - ref1z;
      uz_v = uz_v + static_cast<float>(uniform(s_e[j].fine[2]));
    }
    ux_v = ux_v + static_cast<float>(uniform(s_e[j].fine[0]));
    uy_v = uy_v + static_cast<float>(uniform(s_e[j].fine[1]));
    if (!isnan(ux_v)) rx = ux_v;
    if (!isnan(uy_v)) ry = uy_v;
    if (!isnan(uz_v)) rz = uz_v;
  }
  *rx_out = rx;
  *ry_out = ry;
  *rz_out = rz;
  return in_region;
}

}  // namespace sfm_target

#endif  // SFM_TARGET_H_
