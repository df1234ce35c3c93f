// Flow-field clean-up on the device: the
// quality filter that runs between flow estimation and mesh relaxation.
//
//   sfm_clean_flow      <->  flow_utils.clean_flow (flow_utils.py:37-78)
//   sfm_mask_irregular  <->  map_utils.mask_irregular (map_utils.py:737-786)
//   sfm_range_mask      <->  the dynamic-range mask of stitch_rigid._estimate_offset
//                            (stitch_rigid.py:47-60)
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
