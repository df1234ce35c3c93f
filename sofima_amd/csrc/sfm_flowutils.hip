// Flow-field clean-up on the device: the
// quality filter that runs between flow estimation and mesh relaxation.
//
//   sfm_clean_flow      <->  flow_utils.clean_flow (flow_utils.py:37-78)
//   sfm_mask_irregular  <->  map_utils.mask_irregular (map_utils.py:737-786)
//   sfm_range_mask      <->  the dynamic-range mask of stitch_rigid._estimate_offset
//                            (stitch_rigid.py:47-60)
//   sfm_flow_starts     <->  the start-coordinate / targeting arithmetic of
//                            flow_field() (flow_field.py:620-680)
//   sfm_flow_scatter    <->  the result scatter of flow_field() (:701-709)
//
// One thread per vector.  The field is small (one vector per patch), so the
// point of the kernel is that the flow can stay in HBM from the correlation
// peaks to the mesh relaxation; it is bound by launch latency.
#include "sfm_common.h"

#include <hip/hip_runtime.h>

namespace {

constexpr int kBlock = 256;

struct CleanArgs {
  const float* flow;
  float* out;
  int dim, channels;
  int Z, Y, X;
  float min_ratio, min_sharp, max_mag, max_dev;
};

// np.nan_to_num for float32: NaN -> 0, +-inf -> +-FLT_MAX.
__device__ __forceinline__ float nan_to_num(float v) {
  if (isnan(v)) return 0.f;
  if (isinf(v)) return v > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

// scipy.ndimage "reflect" boundary (d c b a | a b c d | d c b a) for a
// one-element overhang.
__device__ __forceinline__ int reflect(int i, int n) {
  if (i < 0) return -i - 1 < n ? -i - 1 : n - 1;
  if (i >= n) return 2 * n - 1 - i >= 0 ? 2 * n - 1 - i : 0;
  return i;
}

// Median of the (1|3) x 3 x 3 window of nan_to_num(comp) around (z, y, x).
__device__ float window_median(const float* comp, const CleanArgs& a, int z, int y,
                               int x) {
  float w[27];
  int n = 0;
  const int zr = a.dim == 3 ? 1 : 0;
  for (int dz = -zr; dz <= zr; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int zz = reflect(z + dz, a.Z), yy = reflect(y + dy, a.Y),
                  xx = reflect(x + dx, a.X);
        const float v = nan_to_num(comp[((long long)zz * a.Y + yy) * a.X + xx]);
        int k = n++;  // insertion sort
        while (k > 0 && w[k - 1] > v) {
          w[k] = w[k - 1];
          --k;
        }
        w[k] = v;
      }
  return w[n / 2];
}

__global__ void __launch_bounds__(kBlock) clean_flow_kernel(CleanArgs a) {
  const long long n_vec = (long long)a.Z * a.Y * a.X;
  const long long i = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (i >= n_vec) return;
  const int x = static_cast<int>(i % a.X);
  const int y = static_cast<int>((i / a.X) % a.Y);
  const int z = static_cast<int>(i / ((long long)a.X * a.Y));
  bool bad = false;
  if (a.channels == a.dim + 2) {
    // comparisons with NaN are false, as in NumPy
    bad = fabsf(a.flow[a.dim * n_vec + i]) < a.min_sharp;
    const float pr = fabsf(a.flow[(a.dim + 1) * n_vec + i]);
    bad = bad || (pr > 0.f && pr < a.min_ratio);
  }
  float v[3];
  bool any_nan = false;
  for (int c = 0; c < a.dim; ++c) {
    v[c] = a.flow[c * n_vec + i];
    any_nan = any_nan || isnan(v[c]);
  }
  // np.max over the components propagates NaN, and NaN > t is false
  if (!any_nan) {
    if (a.max_mag > 0.f) {
      float m = 0.f;
      for (int c = 0; c < a.dim; ++c) m = fmaxf(m, fabsf(v[c]));
      bad = bad || m > a.max_mag;
    }
    if (a.max_dev > 0.f) {
      float m = 0.f;
      for (int c = 0; c < a.dim; ++c) {
        const float med = window_median(a.flow + c * n_vec, a, z, y, x);
        m = fmaxf(m, fabsf(med - v[c]));
      }
      bad = bad || m > a.max_dev;
    }
  }
  for (int c = 0; c < a.dim; ++c) a.out[c * n_vec + i] = bad ? NAN : v[c];
}

struct IrregArgs {
  float* map;          // [2, Y, X]
  unsigned char* bad;  // [Y, X]
  int Y, X, iters;
  float sx, sy, lo_x, lo_y, hi_x, hi_y;
};

// bad before dilation (map_utils.py:768-775); out-of-range nodes are not bad.
__device__ __forceinline__ bool irregular_at(const IrregArgs& a, int y, int x) {
  if (y < 0 || y >= a.Y || x < 0 || x >= a.X) return false;
  const long long n = (long long)a.Y * a.X, i = (long long)y * a.X + x;
  // np.diff padded with 0 at the far edge, then + stride
  const float dx = (x + 1 < a.X ? a.map[i + 1] - a.map[i] : 0.f) + a.sx;
  const float dy = (y + 1 < a.Y ? a.map[n + i + a.X] - a.map[n + i] : 0.f) + a.sy;
  return dx < a.lo_x || dy < a.lo_y || dx > a.hi_x || dy > a.hi_y;
}

// `iters` dilations with the full 3 x 3 structure == OR over the square window
// of radius `iters` (border value 0).
__global__ void __launch_bounds__(kBlock) irregular_mark_kernel(IrregArgs a) {
  const long long n = (long long)a.Y * a.X;
  const long long i = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (i >= n) return;
  const int x = static_cast<int>(i % a.X), y = static_cast<int>(i / a.X);
  bool bad = false;
  for (int dy = -a.iters; dy <= a.iters && !bad; ++dy)
    for (int dx = -a.iters; dx <= a.iters; ++dx)
      if (irregular_at(a, y + dy, x + dx)) {
        bad = true;
        break;
      }
  a.bad[i] = bad ? 1 : 0;
}

// Second launch: the NaN fill must not feed back into the differences above.
__global__ void __launch_bounds__(kBlock) irregular_fill_kernel(IrregArgs a) {
  const long long n = (long long)a.Y * a.X;
  const long long i = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (i >= n || !a.bad[i]) return;
  a.map[i] = NAN;
  a.map[n + i] = NAN;
}

}  // namespace

extern "C" int sfm_mask_irregular(const SfmMaskIrregularDesc* d, float* coord_map,
                                  uint8_t* bad) {
  if (!d || !coord_map || !bad)
    return sfm::fail(SFM_ERR_INVALID, "mask_irregular: NULL argument");
  if (d->shape[0] < 1 || d->shape[1] < 1 || d->dilation_iters < 0)
    return sfm::fail(SFM_ERR_INVALID, "mask_irregular: bad shape / iterations");
  IrregArgs a;
  a.map = coord_map;
  a.bad = bad;
  a.Y = d->shape[0];
  a.X = d->shape[1];
  a.iters = d->dilation_iters;
  a.sx = d->stride[0];
  a.sy = d->stride[1];
  a.lo_x = d->frac * a.sx;
  a.lo_y = d->frac * a.sy;
  a.hi_x = d->max_frac * a.sx;
  a.hi_y = d->max_frac * a.sy;
  const long long n = (long long)a.Y * a.X;
  const unsigned grid = static_cast<unsigned>((n + kBlock - 1) / kBlock);
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  hipLaunchKernelGGL(irregular_mark_kernel, dim3(grid), dim3(kBlock), 0, st, a);
  hipLaunchKernelGGL(irregular_fill_kernel, dim3(grid), dim3(kBlock), 0, st, a);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// Dynamic-range mask (stitch_rigid.py:47-60):
//   (maximum_filter(img, size) - minimum_filter(img, size)) < range_limit  [| extra]
// scipy.ndimage filters, mode "reflect", origin 0: the window of pixel i covers
// [i - size / 2, i - size / 2 + size - 1] per axis.  uint8 images subtract in
// uint8 (max >= min: no wrap) and compare as integers against the limit;
// float images subtract and compare in float32.
// ---------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int reflect_any(int i, int n) {
  // whole-period reflection: valid for any overhang
  if (n == 1) return 0;
  const int period = 2 * n;
  int m = i % period;
  if (m < 0) m += period;
  return m < n ? m : period - 1 - m;
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
range_mask_kernel(const T* __restrict__ img, const uint8_t* __restrict__ extra,
                  uint8_t* __restrict__ out, int Y, int X, int size, double limit) {
  const long long n = (long long)Y * X;
  const long long i = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (i >= n) return;
  const int y = static_cast<int>(i / X), x = static_cast<int>(i % X);
  const int lo = size / 2;
  T mx = img[i], mn = img[i];
  for (int dy = 0; dy < size; ++dy) {
    const int yy = reflect_any(y - lo + dy, Y);
    for (int dx = 0; dx < size; ++dx) {
      const int xx = reflect_any(x - lo + dx, X);
      const T v = img[(long long)yy * X + xx];
      mx = v > mx ? v : mx;
      mn = v < mn ? v : mn;
    }
  }
  // NumPy compares the (uint8 | float32) difference with the Python scalar in
  // double precision for floats' sake; both are exact in double.
  const T diff = static_cast<T>(mx - mn);
  uint8_t m = static_cast<double>(diff) < limit ? 1 : 0;
  if (extra) m |= extra[i] != 0;
  out[i] = m;
}

}  // namespace

extern "C" int sfm_range_mask(const SfmRangeMaskDesc* d, uint8_t* out) {
  if (!d || !d->image || !out)
    return sfm::fail(SFM_ERR_INVALID, "range_mask: NULL argument");
  if (d->shape[0] < 1 || d->shape[1] < 1 || d->filter_size < 1)
    return sfm::fail(SFM_ERR_INVALID, "range_mask: bad shape / filter size");
  const long long n = (long long)d->shape[0] * d->shape[1];
  const long long grid = (n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffLL) return sfm::fail(SFM_ERR_INVALID, "range_mask: too large");
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  if (d->dtype == SFM_DTYPE_U8)
    hipLaunchKernelGGL(range_mask_kernel<uint8_t>, dim3(static_cast<unsigned>(grid)),
                       dim3(kBlock), 0, st, static_cast<const uint8_t*>(d->image),
                       d->extra_mask, out, d->shape[0], d->shape[1], d->filter_size,
                       d->range_limit);
  else if (d->dtype == SFM_DTYPE_F32)
    hipLaunchKernelGGL(range_mask_kernel<float>, dim3(static_cast<unsigned>(grid)),
                       dim3(kBlock), 0, st, static_cast<const float*>(d->image),
                       d->extra_mask, out, d->shape[0], d->shape[1], d->filter_size,
                       d->range_limit);
  else
    return sfm::fail(SFM_ERR_INVALID, "range_mask: dtype %d", d->dtype);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// Host loop of JAXMaskedXCorrWithStatsCalculator.flow_field on the device.
// ---------------------------------------------------------------------------
namespace {

struct StartsArgs {
  const int* pos;       // [n, nd] grid positions, [z]yx
  int n, nd;
  int step[3], patch[3], post_patch[3], pre_shape[3], post_shape[3];
  const float* tf[2];   // targeting fields [nd, *tshape] (xy[z] components) or NULL
  int tshape[2][3];
  int tstep[2][3];
  int* pre_starts;      // [n, nd]
  int* post_starts;
  int* tg[2];           // [n, nd] offsets applied to the pre / post starts ([z]yx)
};

// np.round (half to even) of a double
__device__ __forceinline__ long long round_even(double v) {
  return static_cast<long long>(rint(v));
}

// Integer patch shift from a targeting field, kept inside the image
// (flow_field.py:626-649 / :652-677).  `starts` is [z]yx.
__device__ void target_offset(const StartsArgs& a, int side, const int* starts,
                              const int* psize, const int* img_shape, int* off) {
  int q[3];
  long long cell = 0;
  long long plane = 1;
  for (int i = 0; i < a.nd; ++i) plane *= a.tshape[side][i];
  for (int i = 0; i < a.nd; ++i) {
    const double c = static_cast<double>(starts[i] + psize[i] / 2) /
                     static_cast<double>(a.tstep[side][i]);
    long long v = round_even(c);
    v = v < 0 ? 0 : (v > a.tshape[side][i] - 1 ? a.tshape[side][i] - 1 : v);
    q[i] = static_cast<int>(v);
    cell = cell * a.tshape[side][i] + q[i];
  }
  for (int i = 0; i < a.nd; ++i) {
    // field component for image axis i: components are x, y[, z] = reversed axes
    float f = a.tf[side][(long long)(a.nd - 1 - i) * plane + cell];
    if (isnan(f)) f = 0.f;
    if (isinf(f)) f = f > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    // .astype(int): truncation; out-of-range values are INT64_MIN in NumPy --
    // fields hold pixel offsets, far inside the int range
    int o = static_cast<int>(f);
    const int ns = starts[i] + o;
    o = o - (ns < 0 ? ns : 0);
    const int hi = ns + psize[i];
    const int over = (hi > img_shape[i] ? hi : img_shape[i]) - img_shape[i];
    off[i] = o - over;
  }
}

__global__ void __launch_bounds__(kBlock) flow_starts_kernel(StartsArgs a) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= a.n) return;
  int post[3], pre[3], off[3];
  for (int i = 0; i < a.nd; ++i) {
    post[i] = a.pos[b * a.nd + i] * a.step[i];
    // NumPy floor division (flow_field.py:620): -1 // 2 == -1, not 0
    const int diff = a.patch[i] - a.post_patch[i];
    const int p = post[i] - (diff >= 0 ? diff / 2 : -((-diff + 1) / 2));
    pre[i] = p < 0 ? 0 : p;
  }
  if (a.tf[0]) {
    target_offset(a, 0, pre, a.patch, a.pre_shape, off);
    for (int i = 0; i < a.nd; ++i) {
      pre[i] += off[i];
      a.tg[0][b * a.nd + i] = off[i];
    }
  }
  if (a.tf[1]) {
    target_offset(a, 1, post, a.post_patch, a.post_shape, off);
    for (int i = 0; i < a.nd; ++i) {
      post[i] += off[i];
      a.tg[1][b * a.nd + i] = off[i];
    }
  }
  for (int i = 0; i < a.nd; ++i) {
    a.pre_starts[b * a.nd + i] = pre[i] < 0 ? 0 : pre[i];
    a.post_starts[b * a.nd + i] = post[i] < 0 ? 0 : post[i];
  }
}

struct ScatterArgs {
  const float* peaks;   // [n, nd + 2]
  const int* pos;       // [n, nd]
  const int* tg[2];     // or NULL
  float* out;           // [nd + 2, *grid], NaN filled by the caller
  int n, nd;
  int grid[3];
};

__global__ void __launch_bounds__(kBlock) flow_scatter_kernel(ScatterArgs a) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= a.n) return;
  long long cell = 0, cells = 1;
  for (int i = 0; i < a.nd; ++i) {
    cell = cell * a.grid[i] + a.pos[b * a.nd + i];
    cells *= a.grid[i];
  }
  for (int c = 0; c < a.nd + 2; ++c) {
    float v = a.peaks[b * (a.nd + 2) + c];
    if (c < a.nd) {  // vector component c <-> image axis nd - 1 - c
      if (a.tg[0]) v = v + static_cast<float>(a.tg[0][b * a.nd + a.nd - 1 - c]);
      if (a.tg[1]) v = v - static_cast<float>(a.tg[1][b * a.nd + a.nd - 1 - c]);
    }
    a.out[c * cells + cell] = v;
  }
}

}  // namespace

extern "C" int sfm_flow_starts(const SfmFlowStartsDesc* d) {
  if (!d || !d->positions || !d->pre_starts || !d->post_starts)
    return sfm::fail(SFM_ERR_INVALID, "flow_starts: NULL argument");
  if (d->ndim != 2 && d->ndim != 3) return sfm::fail(SFM_ERR_INVALID, "flow_starts: ndim");
  if (d->n < 0) return sfm::fail(SFM_ERR_INVALID, "flow_starts: n");
  if (d->n == 0) return SFM_OK;
  StartsArgs a;
  a.pos = d->positions;
  a.n = d->n;
  a.nd = d->ndim;
  const int o = 3 - d->ndim;  // descriptors pad [z]yx to 3 entries in front
  for (int i = 0; i < d->ndim; ++i) {
    a.step[i] = d->step[o + i];
    a.patch[i] = d->patch[o + i];
    a.post_patch[i] = d->post_patch[o + i];
    a.pre_shape[i] = d->pre_shape[o + i];
    a.post_shape[i] = d->post_shape[o + i];
    if (a.step[i] < 1) return sfm::fail(SFM_ERR_INVALID, "flow_starts: step");
  }
  const float* tf[2] = {d->pre_targeting_field, d->post_targeting_field};
  const int32_t* tsh[2] = {d->pre_targeting_shape, d->post_targeting_shape};
  const int32_t* tst[2] = {d->pre_targeting_step, d->post_targeting_step};
  int* tg[2] = {d->pre_offsets, d->post_offsets};
  for (int s2 = 0; s2 < 2; ++s2) {
    a.tf[s2] = tf[s2];
    a.tg[s2] = tg[s2];
    if (tf[s2] && !tg[s2])
      return sfm::fail(SFM_ERR_INVALID, "flow_starts: offsets buffer missing");
    for (int i = 0; i < d->ndim; ++i) {
      a.tshape[s2][i] = tsh[s2][o + i];
      a.tstep[s2][i] = tst[s2][o + i];
      if (tf[s2] && (a.tshape[s2][i] < 1 || a.tstep[s2][i] < 1))
        return sfm::fail(SFM_ERR_INVALID, "flow_starts: targeting shape / step");
    }
  }
  a.pre_starts = d->pre_starts;
  a.post_starts = d->post_starts;
  hipLaunchKernelGGL(flow_starts_kernel, dim3((d->n + kBlock - 1) / kBlock), dim3(kBlock),
                     0, static_cast<hipStream_t>(d->stream), a);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

extern "C" int sfm_flow_scatter(const SfmFlowScatterDesc* d) {
  if (!d || !d->peaks || !d->positions || !d->out)
    return sfm::fail(SFM_ERR_INVALID, "flow_scatter: NULL argument");
  if (d->ndim != 2 && d->ndim != 3) return sfm::fail(SFM_ERR_INVALID, "flow_scatter: ndim");
  if (d->n <= 0) return d->n == 0 ? SFM_OK : sfm::fail(SFM_ERR_INVALID, "flow_scatter: n");
  ScatterArgs a;
  a.peaks = d->peaks;
  a.pos = d->positions;
  a.tg[0] = d->pre_offsets;
  a.tg[1] = d->post_offsets;
  a.out = d->out;
  a.n = d->n;
  a.nd = d->ndim;
  for (int i = 0; i < d->ndim; ++i) a.grid[i] = d->grid[3 - d->ndim + i];
  hipLaunchKernelGGL(flow_scatter_kernel, dim3((d->n + kBlock - 1) / kBlock), dim3(kBlock),
                     0, static_cast<hipStream_t>(d->stream), a);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

extern "C" int sfm_clean_flow(const SfmCleanFlowDesc* d, float* out) {
  if (!d || !d->flow || !out)
    return sfm::fail(SFM_ERR_INVALID, "clean_flow: NULL argument");
  if (d->dim != 2 && d->dim != 3)
    return sfm::fail(SFM_ERR_INVALID, "clean_flow: dim must be 2 or 3");
  if (d->channels < d->dim || d->channels > d->dim + 2)
    return sfm::fail(SFM_ERR_INVALID, "clean_flow: %d channels for dim %d",
                     d->channels, d->dim);
  for (int i = 0; i < 3; ++i)
    if (d->shape[i] < 1) return sfm::fail(SFM_ERR_INVALID, "clean_flow: bad shape");
  CleanArgs a;
  a.flow = d->flow;
  a.out = out;
  a.dim = d->dim;
  a.channels = d->channels;
  a.Z = d->shape[0];
  a.Y = d->shape[1];
  a.X = d->shape[2];
  a.min_ratio = d->min_peak_ratio;
  a.min_sharp = d->min_peak_sharpness;
  a.max_mag = d->max_magnitude;
  a.max_dev = d->max_deviation;
  const long long n = (long long)a.Z * a.Y * a.X;
  const long long grid = (n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffLL) return sfm::fail(SFM_ERR_INVALID, "clean_flow: too large");
  hipLaunchKernelGGL(clean_flow_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock),
                     0, static_cast<hipStream_t>(d->stream), a);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}
