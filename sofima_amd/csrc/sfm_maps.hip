// Coordinate-map composition and the montage target mesh for gfx950.
//
//   compose_kernel      <-> map_utils.compose_maps_fast (map_utils.py:616-734),
//                           i.e. jax.scipy.ndimage.map_coordinates(order=1)
//   target_mesh_kernel  <-> stitch_elastic.compute_target_mesh
//                           (stitch_elastic.py:456-676) for all tiles
//
// Both are gathers with a handful of flops per output: HBM/latency bound, one
// thread per output node, coalesced along x.
#include "sfm_common.h"
#include "sfm_target.h"

#include <algorithm>
#include <cmath>

namespace {

using namespace sfm_target;

constexpr int kBlock = 256;

struct ComposeArgs {
  int ncomp, constant;
  int s1[3], s2[3];
  float off1[3], off2[3];  // start - origin, zyx
  float st1[3], st2[3];
  const float* m1;
  const float* m2;
  float* out;
};

__global__ void __launch_bounds__(kBlock) compose_kernel(ComposeArgs a) {
  const long long n1 = (long long)a.s1[0] * a.s1[1] * a.s1[2];
  const long long n2 = (long long)a.s2[0] * a.s2[1] * a.s2[2];
  const long long plane2 = (long long)a.s2[1] * a.s2[2];
  for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n1;
       i += (long long)gridDim.x * kBlock) {
    const int x = static_cast<int>(i % a.s1[2]);
    const long long r = i / a.s1[2];
    const int y = static_cast<int>(r % a.s1[1]);
    const int z = static_cast<int>(r / a.s1[1]);
    const float ref1x = (static_cast<float>(x) + a.off1[2]) * a.st1[2];
    const float ref1y = (static_cast<float>(y) + a.off1[1]) * a.st1[1];
    const float qx = (ref1x + a.m1[i]) / a.st2[2];
    const float qy = (ref1y + a.m1[n1 + i]) / a.st2[1];
    if (a.ncomp == 2) {
      // every z section on its own (map_utils.py:666-695); a section missing
      // in map2 cannot occur: the reference indexes map2[:, z] directly.
      const float* p0 = a.m2 + (long long)z * plane2;
      const float* p1 = a.m2 + n2 + (long long)z * plane2;
      a.out[i] = sample2(plain(p0), a.s2[1], a.s2[2], qy, qx, a.constant, 1, a.off2[2],
                         a.st2[2]) - ref1x;
      a.out[n1 + i] = sample2(plain(p1), a.s2[1], a.s2[2], qy, qx, a.constant, 0,
                              a.off2[1], a.st2[1]) - ref1y;
    } else {
      const float ref1z = (static_cast<float>(z) + a.off1[0]) * a.st1[0];
      const float qz = (ref1z + a.m1[2 * n1 + i]) / a.st2[0];
      a.out[i] = sample3(plain(a.m2), a.s2[0], a.s2[1], a.s2[2], qz, qy, qx, a.constant,
                         2, a.off2[2], a.st2[2]) - ref1x;
      a.out[n1 + i] = sample3(plain(a.m2 + n2), a.s2[0], a.s2[1], a.s2[2], qz, qy, qx,
                              a.constant, 1, a.off2[1], a.st2[1]) - ref1y;
      a.out[2 * n1 + i] = sample3(plain(a.m2 + 2 * n2), a.s2[0], a.s2[1], a.s2[2], qz, qy,
                                  qx, a.constant, 0, a.off2[0], a.st2[0]) - ref1z;
    }
  }
}

struct TargetArgs {
  SfmTargetMeshDesc d;
  const float* x;
  float* out;
  sfm::AdvanceView adv;   // adv.v == nullptr: x holds the positions to sample
  int strips_only;        // leave nodes outside every paste region untouched
  const int* list;        // strips-only: blocks to evaluate (target_list_kernel)
  const int* count;
};

// Which 16 x 16 node blocks of which tiles touch a paste region (in-plane
// montages): built once per chunk, so that the per-step launches of the
// strips-only evaluation spend their workgroups on the overlap strips only.
// list[i] = tile * blocks_per_tile + block; *count = entries.
__global__ void __launch_bounds__(kBlock)
target_list_kernel(SfmTargetMeshDesc d, int* __restrict__ list, int* __restrict__ count) {
  const int my = d.mesh_shape[1], mx = d.mesh_shape[2];
  const int n_bx = (mx + 15) >> 4, n_by = (my + 15) >> 4;
  const int tile = blockIdx.y;
  const int blk = blockIdx.x * kBlock + threadIdx.x;
  if (blk >= n_bx * n_by) return;
  const int bx = blk % n_bx, by = blk / n_bx;
  bool any = false;
  for (int j = 0; j < 4; ++j) {
    const NbEntry e = make_entry(d, tile, j);
    if (!e.valid) continue;
    const int y0 = e.tg[1], x0 = e.tg[2], y1 = y0 + e.fsz[1], x1 = x0 + e.fsz[2];
    any = any || (by * 16 < y1 && by * 16 + 16 > y0 && bx * 16 < x1 && bx * 16 + 16 > x0);
  }
  if (any) list[atomicAdd(count, 1)] = tile * (n_bx * n_by) + blk;
}

// MODE: sfm_target::Plane::at (0 stored positions, 1 advanced, 2 advanced with
// per-column drift means pending)
template <int MODE>
__global__ void __launch_bounds__(kBlock) target_mesh_kernel(TargetArgs a) {
  constexpr bool ADV = MODE != 0;
  const SfmTargetMeshDesc& d = a.d;
  const int nc = d.ncomp;
  const int mz = d.mesh_shape[0], my = d.mesh_shape[1], mx = d.mesh_shape[2];
  const long long mn = (long long)mz * my * mx;
  int tile = blockIdx.y, blk = blockIdx.x;
  if (a.list) {   // strips-only launch over the block list of target_list_kernel
    if (static_cast<int>(blockIdx.x) >= *a.count) return;
    const int bpt = ((mx + 15) >> 4) * ((my + 15) >> 4);
    const int code = a.list[blockIdx.x];
    tile = code / bpt;
    blk = code - tile * bpt;
  }
  __shared__ NbEntry s_e[4];
  if (threadIdx.x < 4) {
    const NbEntry e = make_entry(d, tile, threadIdx.x);
    s_e[threadIdx.x] = e;
  }
  __syncthreads();
  // the step's position update, if the caller has not applied it yet; the
  // pending gate / drift of the previous step folded into neutral values
  float a_dt = 0.f, a_c2 = 0.f, a_gate = 1.f, a_mx[3] = {0.f, 0.f, 0.f},
        a_mv[3] = {0.f, 0.f, 0.f};
  constexpr bool cols = MODE == 2;   // per-column drift means pending instead of the global ones
  if (ADV) {
    a_dt = a.adv.fire ? a.adv.scal->dt : a.adv.vv_dt;
    a_c2 = 0.5f * (a_dt * a_dt);
    if (a.adv.fire && a.adv.pending) {
      a_gate = a.adv.scal->gate;
      if (a.adv.remove_drift && !cols)
        for (int c = 0; c < 3; ++c) {
          a_mx[c] = a.adv.scal->mx[c];
          a_mv[c] = a.adv.scal->mv[c];
        }
    }
  }
  auto plane_of = [&](int c, int nb_i) {
    const long long off = ((long long)c * d.n_tiles + nb_i) * mn;
    if (!ADV) return plain(a.x + off);
    Plane pl{a.x + off, a.adv.v + off, a.adv.a + off, a_dt, a_c2, a_gate, a_mx[c], a_mv[c],
             nullptr, nullptr};
    if (cols) {
      pl.cs_x = a.adv.colmean + c * mx;
      pl.cs_v = a.adv.colmean + (3 + c) * mx;
    }
    return pl;
  };
  // A wave covers 16 columns x 4 rows, a workgroup 16 x 16 nodes: waves are
  // either inside an overlap strip or outside (with one row of 64 nodes per
  // wave, every wave crossing a 20-node wide left / right strip ran the
  // sampling path with 10 % of its lanes).
  const int n_bx = (mx + 15) >> 4;
  const int bx = blk % n_bx, by = blk / n_bx;
  if (a.strips_only && nc == 2 && !a.list) {
    // does this 16 x 16 block of nodes touch a paste region at all?
    bool any = false;
    for (int j = 0; j < 4; ++j) {
      if (!uniform(s_e[j].valid)) continue;
      const int y0 = uniform(s_e[j].tg[1]), x0 = uniform(s_e[j].tg[2]);
      const int y1 = y0 + uniform(s_e[j].fsz[1]), x1 = x0 + uniform(s_e[j].fsz[2]);
      any = any || (by * 16 < y1 && by * 16 + 16 > y0 && bx * 16 < x1 && bx * 16 + 16 > x0);
    }
    if (!any) return;
  }
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = bx * 16 + (lane & 15);
    const int row = by * 16 + wave * 4 + (lane >> 4);  // tz * my + ty
    if (tx >= mx || row >= mz * my) return;
    const int tz = mz == 1 ? 0 : row / my;
    const int ty = row - tz * my;
    const int node = row * mx + tx;
    float rx, ry, rz;
    const bool in_region = target_node<MODE>(d, s_e, tz, ty, tx, plane_of, &rx, &ry, &rz);
    if (a.strips_only && !in_region) return;   // NaN since the first full evaluation
    // out holds the evaluated tiles only: [ncomp, n_eval, *mesh]
    const long long n_out = d.n_eval > 0 ? d.n_eval : d.n_tiles;
    a.out[(long long)tile * mn + node] = rx;
    a.out[(n_out + tile) * mn + node] = ry;
    if (nc == 3) a.out[(2 * n_out + tile) * mn + node] = rz;
  }
}

int grid_for(long long n) {
  long long g = (n + kBlock - 1) / kBlock;
  return static_cast<int>(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

namespace sfm {

size_t target_list_ints(const SfmTargetMeshDesc* d) {
  if (!d || d->ncomp != 2) return 0;
  const long long bpt = (long long)((d->mesh_shape[2] + 15) / 16) * ((d->mesh_shape[1] + 15) / 16);
  return static_cast<size_t>(bpt * d->n_tiles + 4);
}

int build_target_list(const SfmTargetMeshDesc* d, int* list, hipStream_t st) {
  if (!d || !list || d->ncomp != 2) return fail(SFM_ERR_INVALID, "target list: in-plane only");
  const int bpt = ((d->mesh_shape[2] + 15) / 16) * ((d->mesh_shape[1] + 15) / 16);
  SFM_HIP_CHECK(hipMemsetAsync(list, 0, sizeof(int), st));   // list[0]: the count
  hipLaunchKernelGGL(target_list_kernel, dim3((bpt + kBlock - 1) / kBlock, d->n_tiles),
                     dim3(kBlock), 0, st, *d, list + 4, list);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int launch_target_mesh(const SfmTargetMeshDesc* d, const float* x, float* out,
                       hipStream_t st, const AdvanceView* adv, bool strips_only,
                       const int* block_list) {
  if (!d || !x || !out) return fail(SFM_ERR_INVALID, "target mesh: NULL argument");
  if (d->ncomp != 2 && d->ncomp != 3)
    return fail(SFM_ERR_INVALID, "target mesh: ncomp must be 2 or 3");
  if (d->ncomp == 2 && d->mesh_shape[0] != 1)
    return fail(SFM_ERR_INVALID, "target mesh: in-plane montages have one section");
  if (d->nbor_fields < (d->ncomp == 3 ? 11 : 8) || !d->nbors || !d->fx || !d->fy ||
      d->n_tiles < 1 || d->n_eval < 0)
    return fail(SFM_ERR_INVALID, "target mesh: bad neighbour / flow arrays");
  if (d->n_eval > 0 && out == x)
    return fail(SFM_ERR_INVALID, "target mesh: partial evaluation needs its own output");
  TargetArgs a;
  a.d = *d;
  a.x = x;
  a.out = out;
  a.adv = adv ? *adv : AdvanceView{nullptr, nullptr, nullptr, 0, 0, 0, 0.f, nullptr};
  a.strips_only = strips_only ? 1 : 0;
  a.list = strips_only && block_list && d->ncomp == 2 ? block_list + 4 : nullptr;
  a.count = block_list;
  const long long per_tile =
      (long long)d->mesh_shape[0] * d->mesh_shape[1] * d->mesh_shape[2];
  if (d->nbor_fields > 11)
    return fail(SFM_ERR_INVALID, "target mesh: at most 11 neighbour fields");
  if (per_tile > 0x7fffffffLL) return fail(SFM_ERR_INVALID, "target mesh: tile too large");
  const long long gx = (long long)((d->mesh_shape[2] + 15) / 16) *
                       (((long long)d->mesh_shape[0] * d->mesh_shape[1] + 15) / 16);
  dim3 grid(static_cast<unsigned>(gx), d->n_eval > 0 ? d->n_eval : d->n_tiles);
  if (a.list) {
    if (d->n_eval > 0) return fail(SFM_ERR_INVALID, "target mesh: block list with n_eval");
    grid = dim3(static_cast<unsigned>(gx * d->n_tiles), 1);   // upper bound; extra blocks exit
  }
  if (a.adv.v && a.adv.colmean && a.adv.fire && a.adv.pending)
    hipLaunchKernelGGL(target_mesh_kernel<2>, grid, dim3(kBlock), 0, st, a);
  else if (a.adv.v)
    hipLaunchKernelGGL(target_mesh_kernel<1>, grid, dim3(kBlock), 0, st, a);
  else
    hipLaunchKernelGGL(target_mesh_kernel<0>, grid, dim3(kBlock), 0, st, a);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

}  // namespace sfm

extern "C" {

int sfm_compose_maps(const SfmComposeDesc* d, float* out) {
  if (!d || !out || !d->map1 || !d->map2)
    return sfm::fail(SFM_ERR_INVALID, "compose: NULL argument");
  if (d->ncomp != 2 && d->ncomp != 3)
    return sfm::fail(SFM_ERR_INVALID, "compose: ncomp must be 2 or 3");
  ComposeArgs a;
  a.ncomp = d->ncomp;
  a.constant = d->mode == SFM_INTERP_CONSTANT;
  for (int i = 0; i < 3; ++i) {
    if (d->shape1[i] < 1 || d->shape2[i] < 1)
      return sfm::fail(SFM_ERR_INVALID, "compose: bad shape");
    a.s1[i] = d->shape1[i];
    a.s2[i] = d->shape2[i];
    const float origin = fminf(d->start1[i], d->start2[i]);
    a.off1[i] = d->start1[i] - origin;
    a.off2[i] = d->start2[i] - origin;
    a.st1[i] = d->stride1[i];
    a.st2[i] = d->stride2[i];
  }
  if (d->ncomp == 2 && d->shape1[0] > d->shape2[0])
    return sfm::fail(SFM_ERR_INVALID, "compose: map2 has fewer sections than map1");
  a.m1 = d->map1;
  a.m2 = d->map2;
  a.out = out;
  const long long n1 = (long long)a.s1[0] * a.s1[1] * a.s1[2];
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  hipLaunchKernelGGL(compose_kernel, dim3(grid_for(n1)), dim3(kBlock), 0, st, a);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int sfm_target_mesh(const SfmTargetMeshDesc* d, const float* x, float* out,
                    void* stream) {
  return sfm::launch_target_mesh(d, x, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
