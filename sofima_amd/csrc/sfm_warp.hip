// Image warping by an inverse coordinate map on the device (SURVEY.md 8f, rank 4:
// the rendering step that follows mesh relaxation).
//
//   sfm_warp_section  <->  warp.warp_subvolume._warp_section (warp.py:145-167)
//
// The reference does, per section: (1) scipy RegularGridInterpolator (linear,
// extrapolating) of the node coordinates to every output pixel, in float64,
// cast to float32; (2) cv2.convertMaps to the fixed-point CV_16SC2 format
// (1/32 pixel); (3) cv2.remap with constant (0) border.  This kernel fuses the
// three steps: one thread per output pixel, the dense coordinate maps never
// exist in memory.  Interpolation weights come from caller-built tables (the
// host side builds them like OpenCV's initInterTab2D: 32 x 32 sub-pixel phases,
// ksize x ksize taps; 15-bit fixed point for 8-bit images).
#include "sfm_common.h"

#include <cmath>

namespace {

constexpr int kBlock = 256;
constexpr int kTabBits = 5, kTabSize = 1 << kTabBits;

struct WarpArgs {
  const void* image;
  const void* map;      // [2, my, mx] absolute source coordinates (x, y), float or double
  int map_f64;
  void* out;
  const void* tab;      // [32 * 32, ks * ks] short (u8 images) or float
  int dtype, nearest, ks;
  int iy, ix, my, mx, oy, ox;
  double org_y, org_x, stride;
};

// saturate_cast<short>
__device__ __forceinline__ int sat_short(long long v) {
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : static_cast<int>(v));
}

// cvRound: round half to even; NaN / out of range -> INT_MIN like cvtss2si
__device__ __forceinline__ long long cv_round(double v) {
  if (!(v > -2147483648.0 && v < 2147483647.0)) return -2147483648LL;
  return static_cast<long long>(rint(v));
}

// Linear interpolation (with extrapolation) of one map component at grid
// position (gy, gx) in node units (RegularGridInterpolator, method "linear",
// fill_value None).
template <typename M>
__device__ __forceinline__ double map_at(const M* m, int my, int mx, double gy,
                                         double gx) {
  int i = static_cast<int>(floor(gy)), j = static_cast<int>(floor(gx));
  i = i < 0 ? 0 : (i > my - 2 ? my - 2 : i);
  j = j < 0 ? 0 : (j > mx - 2 ? mx - 2 : j);
  const double t = gy - i, u = gx - j;
  const double v00 = m[i * mx + j], v01 = m[i * mx + j + 1];
  const double v10 = m[(i + 1) * mx + j], v11 = m[(i + 1) * mx + j + 1];
  return (1.0 - t) * (1.0 - u) * v00 + (1.0 - t) * u * v01 + t * (1.0 - u) * v10 +
         t * u * v11;
}

template <typename T>
__device__ __forceinline__ T pixel(const T* img, int iy, int ix, int y, int x) {
  return (y >= 0 && y < iy && x >= 0 && x < ix) ? img[(long long)y * ix + x] : T(0);
}

template <typename T, bool FIXED>
__global__ void __launch_bounds__(kBlock) warp_kernel(WarpArgs a) {
  const long long n = (long long)a.oy * a.ox;
  const long long idx = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (idx >= n) return;
  const int py = static_cast<int>(idx / a.ox), px = static_cast<int>(idx % a.ox);
  const double gy = (py - a.org_y) / a.stride, gx = (px - a.org_x) / a.stride;
  const long long plane = (long long)a.my * a.mx;
  float sx, sy;   // dense coordinates: interpolated in the map's dtype, cast to float32
  if (a.map_f64) {
    const double* m = static_cast<const double*>(a.map);
    sx = static_cast<float>(map_at(m, a.my, a.mx, gy, gx));
    sy = static_cast<float>(map_at(m + plane, a.my, a.mx, gy, gx));
  } else {
    const float* m = static_cast<const float*>(a.map);
    sx = static_cast<float>(map_at(m, a.my, a.mx, gy, gx));
    sy = static_cast<float>(map_at(m + plane, a.my, a.mx, gy, gx));
  }
  const T* img = static_cast<const T*>(a.image);
  T* out = static_cast<T*>(a.out);
  if (a.nearest) {
    const int x = sat_short(cv_round(sx)), y = sat_short(cv_round(sy));
    out[idx] = pixel(img, a.iy, a.ix, y, x);
    return;
  }
  // fixed-point map: integer part and 5-bit fraction
  const long long fx = cv_round(static_cast<double>(sx) * kTabSize);
  const long long fy = cv_round(static_cast<double>(sy) * kTabSize);
  const int x0 = sat_short(fx >> kTabBits), y0 = sat_short(fy >> kTabBits);
  const int phase = static_cast<int>(fy & (kTabSize - 1)) * kTabSize +
                    static_cast<int>(fx & (kTabSize - 1));
  const int ks = a.ks, ofs = ks / 2 - 1;
  if (FIXED) {
    const short* w = static_cast<const short*>(a.tab) + (long long)phase * ks * ks;
    int acc = 0;
    for (int k1 = 0; k1 < ks; ++k1)
      for (int k2 = 0; k2 < ks; ++k2)
        acc += static_cast<int>(w[k1 * ks + k2]) *
               static_cast<int>(pixel(img, a.iy, a.ix, y0 + k1 - ofs, x0 + k2 - ofs));
    int v = (acc + (1 << 14)) >> 15;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[idx] = static_cast<T>(v);
  } else {
    const float* w = static_cast<const float*>(a.tab) + (long long)phase * ks * ks;
    float acc = 0.f;
    for (int k1 = 0; k1 < ks; ++k1)
      for (int k2 = 0; k2 < ks; ++k2)
        acc += w[k1 * ks + k2] *
               static_cast<float>(pixel(img, a.iy, a.ix, y0 + k1 - ofs, x0 + k2 - ofs));
    if (sizeof(T) == 2) {  // uint16: saturate_cast<ushort>(cvRound)
      const long long r = cv_round(acc);
      out[idx] = static_cast<T>(r < 0 ? 0 : (r > 65535 ? 65535 : r));
    } else {
      out[idx] = static_cast<T>(acc);
    }
  }
}

}  // namespace

extern "C" int sfm_warp_section(const SfmWarpDesc* d) {
  if (!d || !d->image || !d->coord_map || !d->out)
    return sfm::fail(SFM_ERR_INVALID, "warp: NULL argument");
  if (d->map_shape[0] < 2 || d->map_shape[1] < 2)
    return sfm::fail(SFM_ERR_INVALID, "warp: the coordinate map needs 2 x 2 nodes");
  for (int i = 0; i < 2; ++i)
    if (d->image_shape[i] < 1 || d->out_shape[i] < 1)
      return sfm::fail(SFM_ERR_INVALID, "warp: bad shape");
  const bool nearest = d->interpolation == SFM_WARP_NEAREST;
  if (!nearest && (!d->weights || (d->ksize != 2 && d->ksize != 4 && d->ksize != 8)))
    return sfm::fail(SFM_ERR_INVALID, "warp: weight table / ksize");
  if (!(d->stride > 0.f)) return sfm::fail(SFM_ERR_INVALID, "warp: stride");
  WarpArgs a;
  a.image = d->image;
  a.map = d->coord_map;
  a.map_f64 = d->coord_map_f64 ? 1 : 0;
  a.out = d->out;
  a.tab = d->weights;
  a.dtype = d->dtype;
  a.nearest = nearest ? 1 : 0;
  a.ks = d->ksize;
  a.iy = d->image_shape[0];
  a.ix = d->image_shape[1];
  a.my = d->map_shape[0];
  a.mx = d->map_shape[1];
  a.oy = d->out_shape[0];
  a.ox = d->out_shape[1];
  a.org_y = d->map_origin[0];
  a.org_x = d->map_origin[1];
  a.stride = d->stride;
  const long long n = (long long)a.oy * a.ox;
  const long long grid = (n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffLL) return sfm::fail(SFM_ERR_INVALID, "warp: too large");
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const dim3 g(static_cast<unsigned>(grid)), b(kBlock);
  switch (d->dtype) {
    case SFM_DTYPE_U8:
      hipLaunchKernelGGL((warp_kernel<unsigned char, true>), g, b, 0, st, a);
      break;
    case SFM_DTYPE_U16:
      hipLaunchKernelGGL((warp_kernel<unsigned short, false>), g, b, 0, st, a);
      break;
    case SFM_DTYPE_F32:
      hipLaunchKernelGGL((warp_kernel<float, false>), g, b, 0, st, a);
      break;
    case SFM_DTYPE_I32:
      if (!nearest) return sfm::fail(SFM_ERR_INVALID, "warp: int32 labels are nearest only");
      hipLaunchKernelGGL((warp_kernel<int, false>), g, b, 0, st, a);
      break;
    default:
      return sfm::fail(SFM_ERR_INVALID, "warp: dtype %d", d->dtype);
  }
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// warp.ndimage_warp (warp.py:189-335): scipy.ndimage.map_coordinates twice --
// order 1 on the float64 source map at the node-space position of the output
// voxel, then order 0 / 1 on the image at the resulting coordinates.  Double
// arithmetic in SciPy's order (this file is built with -ffp-contract=off).
// ---------------------------------------------------------------------------
namespace {

struct NdWarpArgs {
  const void* image;
  const double* map;
  void* out;
  int order;
  int ishape[3], mshape[3], oshape[3];
  double stride[3], offset[3];
};

// scipy NI_GeometricTransform, order 1, mode "constant", cval 0: outside
// [0, len - 1] on any axis -> 0; taps beyond the last sample carry weight 0 and
// are read from the last sample.
template <int DIM, typename T>
__device__ __forceinline__ double linear_sample(const T* __restrict__ a, const int* shape,
                                                const double* c) {
  long long base[DIM];
  double w[DIM][2];
  long long pitch = 1;
  long long step[DIM];
#pragma unroll
  for (int d = DIM - 1; d >= 0; --d) {
    step[d] = pitch;
    pitch *= shape[3 - DIM + d];
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const int len = shape[3 - DIM + d];
    if (c[d] < 0.0 || c[d] > static_cast<double>(len - 1)) return 0.0;
    const double fl = floor(c[d]);
    const double t = c[d] - fl;
    w[d][0] = 1.0 - t;
    w[d][1] = 1.0 - w[d][0];
    base[d] = static_cast<long long>(fl);
  }
  double acc = 0.0;
#pragma unroll
  for (int tap = 0; tap < (1 << DIM); ++tap) {
    long long idx = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const int bit = (tap >> (DIM - 1 - d)) & 1;
      const int len = shape[3 - DIM + d];
      long long i = base[d] + bit;
      i = i > len - 1 ? len - 1 : i;
      idx += i * step[d];
    }
    double coeff = static_cast<double>(a[idx]);
#pragma unroll
    for (int d = 0; d < DIM; ++d) coeff = coeff * w[d][(tap >> (DIM - 1 - d)) & 1];
    acc = acc + coeff;
  }
  return acc;
}

template <int DIM, typename T>
__device__ __forceinline__ double nearest_sample(const T* __restrict__ a, const int* shape,
                                                 const double* c) {
  long long idx = 0, pitch = 1;
#pragma unroll
  for (int d = DIM - 1; d >= 0; --d) {
    const int len = shape[3 - DIM + d];
    if (c[d] < 0.0 || c[d] > static_cast<double>(len - 1)) return 0.0;
    long long i = static_cast<long long>(floor(c[d] + 0.5));
    i = i > len - 1 ? len - 1 : i;
    idx += i * pitch;
    pitch *= len;
  }
  return static_cast<double>(a[idx]);
}

// double -> output type like NI's CASE_INTERP_OUT_*: unsigned types add 0.5 to
// positive values, clip and truncate; floats are narrowed.
template <typename T>
__device__ __forceinline__ T nd_convert(double v, double vmax) {
  v = v > 0.0 ? v + 0.5 : 0.0;
  v = v > vmax ? vmax : v;
  return static_cast<T>(v);
}
template <>
__device__ __forceinline__ float nd_convert<float>(double v, double) {
  return static_cast<float>(v);
}

template <int DIM, typename T>
__global__ void __launch_bounds__(kBlock) ndimage_warp_kernel(NdWarpArgs a) {
  const long long n = (long long)a.oshape[0] * a.oshape[1] * a.oshape[2];
  const long long i = blockIdx.x * (long long)kBlock + threadIdx.x;
  if (i >= n) return;
  int o[3];
  o[2] = static_cast<int>(i % a.oshape[2]);
  o[1] = static_cast<int>((i / a.oshape[2]) % a.oshape[1]);
  o[0] = static_cast<int>(i / ((long long)a.oshape[2] * a.oshape[1]));
  // node-space position (warp.py:300-301)
  double pos[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const int ax = 3 - DIM + d;
    pos[d] = (static_cast<double>(o[ax]) - a.offset[ax]) / a.stride[ax];
  }
  // dense coordinates z, y, x = channels DIM - 1 ... 0 of the map (warp.py:304-307)
  const long long nodes = (long long)a.mshape[0] * a.mshape[1] * a.mshape[2];
  double dense[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d)
    dense[d] = linear_sample<DIM, double>(a.map + (long long)(DIM - 1 - d) * nodes, a.mshape, pos);
  const T* img = static_cast<const T*>(a.image);
  const double v = a.order == 0 ? nearest_sample<DIM, T>(img, a.ishape, dense)
                                : linear_sample<DIM, T>(img, a.ishape, dense);
  constexpr double vmax = sizeof(T) == 1 ? 255.0 : 65535.0;
  static_cast<T*>(a.out)[i] = nd_convert<T>(v, vmax);
}

}  // namespace

extern "C" int sfm_ndimage_warp(const SfmNdWarpDesc* d) {
  if (!d || !d->image || !d->src_map || !d->out)
    return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: NULL argument");
  if (d->ndim != 2 && d->ndim != 3) return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: ndim %d", d->ndim);
  if (d->order != 0 && d->order != 1)
    return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: interpolation order %d (0 and 1 are built)",
                     d->order);
  NdWarpArgs a;
  a.image = d->image;
  a.map = d->src_map;
  a.out = d->out;
  a.order = d->order;
  long long n = 1;
  for (int k = 0; k < 3; ++k) {
    if (d->image_shape[k] < 1 || d->map_shape[k] < 1 || d->out_shape[k] < 1)
      return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: bad shape");
    if (d->ndim == 2 && k == 0 &&
        (d->image_shape[0] != 1 || d->map_shape[0] != 1 || d->out_shape[0] != 1))
      return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: 2-d arrays have shape[0] = 1");
    if (k >= 3 - d->ndim && !(d->stride[k] != 0.0))
      return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: stride");
    a.ishape[k] = d->image_shape[k];
    a.mshape[k] = d->map_shape[k];
    a.oshape[k] = d->out_shape[k];
    a.stride[k] = d->stride[k];
    a.offset[k] = d->offset[k];
    n *= d->out_shape[k];
  }
  const long long grid = (n + kBlock - 1) / kBlock;
  if (grid > 0x7fffffffLL) return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: too large");
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const dim3 g(static_cast<unsigned>(grid)), b(kBlock);
#define SFM_NDW(T)                                                                 \
  do {                                                                             \
    if (d->ndim == 2)                                                              \
      hipLaunchKernelGGL((ndimage_warp_kernel<2, T>), g, b, 0, st, a);             \
    else                                                                           \
      hipLaunchKernelGGL((ndimage_warp_kernel<3, T>), g, b, 0, st, a);             \
  } while (0)
  switch (d->dtype) {
    case SFM_DTYPE_U8: SFM_NDW(unsigned char); break;
    case SFM_DTYPE_U16: SFM_NDW(unsigned short); break;
    case SFM_DTYPE_F32: SFM_NDW(float); break;
    default: return sfm::fail(SFM_ERR_INVALID, "ndimage_warp: dtype %d", d->dtype);
  }
#undef SFM_NDW
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}
