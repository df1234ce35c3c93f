// Patch cross-correlation + peak statistics for gfx950: batch driver, patch
// gather, the general shift-by-shift correlation kernel and the peak kernels.
//
// Device replacement for flow_field.py of the reference:
//   _batched_xcorr   (flow_field.py:278-371)  -> gather_kernel (clamped patch
//                     gather, per-patch (masked) mean, validity planes)
//   masked_xcorr     (flow_field.py:36-156)   -> corr_direct_kernel (+ the
//                     Padfield assembly and masked_finalize_kernel), or the
//                     int8 MFMA kernel in sfm_xcorr_mfma.hip for uint8 data
//   _batched_peaks / _peak_stats (flow_field.py:178-275)
//                                             -> peaks_first_kernel,
//                                                peaks_second_kernel
//
// The reference evaluates the full linear correlation with zero-padded FFTs;
// here every output shift is summed directly:
//     out[k] = sum_i A[i + k - (Q-1)] * B[i]      (zero shift at k = Q-1)
// which is the same quantity (oracle/flow_oracle.py checks both forms against
// the reference's own output).  The masked (Padfield) surface is assembled
// from six such sums exactly as SURVEY.md section 8a derives.
#include "sfm_common.h"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace sfm {
// Implemented in sfm_xcorr_mfma.hip.
bool mfma_i8_eligible(const SfmXcorrDesc* d);
size_t mfma_i8_workspace_bytes(const SfmXcorrDesc* d);
// Writes a surface padded to whole 16 x 16 tiles: [B, rows, pitch].
int mfma_i8_surface(const SfmXcorrDesc* d, void* ws, float* surface,
                    const FusedPeaks* fused);
void mfma_i8_padded_dims(const SfmXcorrDesc* d, int* rows, int* pitch);
// sfm_xcorr_fft.hip
bool fft_preferred(const SfmXcorrDesc* d);
int fft_check(const SfmXcorrDesc* d);
size_t fft_workspace_bytes(const SfmXcorrDesc* d);
int fft_correlate(const SfmXcorrDesc* d, const float* a0, const float* b0,
                  const float* va, const float* vb, float* surface, float* den,
                  float* ov, unsigned int* maxima, void* ws, unsigned int* smax);
// Masked correlation: the normalised surface, padded to whole tiles; `smax`
// (optional, zeroed by the caller) receives the ordered bits of every surface
// maximum for the peak search.
int mfma_i8_masked(const SfmXcorrDesc* d, void* ws, float* surface,
                   unsigned int* maxima, unsigned int* smax, float* blkmax);
// rows per block of `blkmax` (the assembly workgroups take 4 waves x 8 rows)
constexpr int kMaskedBlkRows = 32;
}  // namespace sfm

namespace {

constexpr int kBlock = 256;
constexpr int kCandCap = 2048;  // per-patch candidate list capacity
constexpr int kHotCap = 4096;   // per-patch hot list of the fused MFMA path
constexpr float kEps = 1.1920928955078125e-07f;  // float32 eps

struct Geo {
  int nd;
  int P[3], Q[3], S[3];  // [z]yx; S = P + Q - 1
  long long Pn, Qn, Sn;
};

int make_geo(const SfmXcorrDesc* d, Geo* g) {
  if (d->ndim != 2 && d->ndim != 3)
    return sfm::fail(SFM_ERR_INVALID, "ndim must be 2 or 3, got %d", d->ndim);
  g->nd = d->ndim;
  g->Pn = g->Qn = g->Sn = 1;
  for (int i = 0; i < 3; ++i) {
    g->P[i] = d->patch[i];
    g->Q[i] = d->post_patch[i];
    if (g->P[i] < 1 || g->Q[i] < 1)
      return sfm::fail(SFM_ERR_INVALID, "patch sizes must be >= 1");
    if (d->ndim == 2 && i == 0 && (g->P[0] != 1 || g->Q[0] != 1))
      return sfm::fail(SFM_ERR_INVALID, "2-D descriptors need patch[0] == 1");
    if (g->P[i] > d->pre_shape[i] || g->Q[i] > d->post_shape[i])
      return sfm::fail(SFM_ERR_INVALID, "patch larger than image on axis %d", i);
    g->S[i] = g->P[i] + g->Q[i] - 1;
    g->Pn *= g->P[i];
    g->Qn *= g->Q[i];
    g->Sn *= g->S[i];
  }
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// gather: clamped patch extraction, mean subtraction, mask zeroing
// ---------------------------------------------------------------------------
struct GatherArgs {
  const void* img;
  const unsigned char* mask;
  int ishape[3], mshape[3], psz[3];
  const int* starts;
  int nd;
  int use_mean;
  float mean;
  float* out;    // [B, psz]
  float* valid;  // [B, psz] or null
  long long pn;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) gather_kernel(GatherArgs g0, GatherArgs g1) {
  const GatherArgs& g = blockIdx.y == 0 ? g0 : g1;
  const int b = blockIdx.x;
  __shared__ double s_sum[kBlock];
  __shared__ double s_cnt[kBlock];
  int st[3] = {0, 0, 0}, ms[3] = {0, 0, 0};
  for (int i = 0; i < g.nd; ++i) {
    const int v = g.starts[b * g.nd + i];
    const int ax = 3 - g.nd + i;
    // lax.dynamic_slice clamps the start so the slice stays in bounds; image
    // and mask are clamped against their own shapes.
    st[ax] = min(max(v, 0), g.ishape[ax] - g.psz[ax]);
    ms[ax] = min(max(v, 0), g.mshape[ax] - g.psz[ax]);
  }
  const T* img = static_cast<const T*>(g.img);
  const int py = g.psz[1], px = g.psz[2];
  const int pn = static_cast<int>(g.pn);  // < 2^16 here (larger patches: gather_big_kernel)
  // Eight elements per round, their loads issued as one straight-line group
  // (elements past the end repeat the last one and are ignored; the mask test
  // is hoisted: a branch between the loads would make each wait on its own).
  constexpr int kRound = 8;
  auto offsets = [&](int i, long long* off, long long* mo) {
    const int x = i % px;
    const int r = i / px;
    const int y = r % py;
    const int z = r / py;
    *off = ((long long)(st[0] + z) * g.ishape[1] + (st[1] + y)) * g.ishape[2] + (st[2] + x);
    *mo = ((long long)(ms[0] + z) * g.mshape[1] + (ms[1] + y)) * g.mshape[2] + (ms[2] + x);
  };
  auto fetch = [&](int i0, float* val, bool* masked) {
    if (g.mask) {
#pragma unroll
      for (int u = 0; u < kRound; ++u) {
        long long off, mo;
        offsets(min(i0 + u * kBlock, pn - 1), &off, &mo);
        val[u] = static_cast<float>(img[off]);
        masked[u] = g.mask[mo] != 0;
      }
    } else {
#pragma unroll
      for (int u = 0; u < kRound; ++u) {
        long long off, mo;
        offsets(min(i0 + u * kBlock, pn - 1), &off, &mo);
        val[u] = static_cast<float>(img[off]);
        masked[u] = false;
      }
    }
  };
  float mu = g.mean;
  if (!g.use_mean) {
    double s = 0.0, c = 0.0;
    for (int i0 = threadIdx.x; i0 < pn; i0 += kBlock * kRound) {
      float v[kRound];
      bool m[kRound];
      fetch(i0, v, m);
#pragma unroll
      for (int u = 0; u < kRound; ++u)
        if (i0 + u * kBlock < pn && !m[u]) {  // same order of additions as element by element
          s += v[u];
          c += 1.0;
        }
    }
    s_sum[threadIdx.x] = s;
    s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) {
      if (threadIdx.x < k) {
        s_sum[threadIdx.x] += s_sum[threadIdx.x + k];
        s_cnt[threadIdx.x] += s_cnt[threadIdx.x + k];
      }
      __syncthreads();
    }
    mu = static_cast<float>(s_sum[0] / s_cnt[0]);  // NaN when all masked
  }
  for (int i0 = threadIdx.x; i0 < pn; i0 += kBlock * kRound) {
    float v[kRound];
    bool m[kRound];
    fetch(i0, v, m);
#pragma unroll
    for (int u = 0; u < kRound; ++u) {
      const int i = i0 + u * kBlock;
      if (i < pn) {
        g.out[b * g.pn + i] = m[u] ? 0.f : v[u] - mu;
        if (g.valid) g.valid[b * g.pn + i] = m[u] ? 0.f : 1.f;
      }
    }
  }
}

// Large patches (3-D, whole overlaps): kGatherChunks workgroups per patch.
// PHASE 0 leaves per-chunk (sum, count) partials; PHASE 1 adds them in chunk
// order (every workgroup gets the identical mean) and writes its chunk.
constexpr int kGatherChunks = 32;

template <typename T, int PHASE>
__global__ void __launch_bounds__(kBlock)
gather_big_kernel(GatherArgs g0, GatherArgs g1, double* __restrict__ partial) {
  const GatherArgs& g = blockIdx.z == 0 ? g0 : g1;
  const int b = blockIdx.y, chunk = blockIdx.x;
  __shared__ double s_sum[kBlock];
  __shared__ double s_cnt[kBlock];
  int st[3] = {0, 0, 0}, ms[3] = {0, 0, 0};
  for (int i = 0; i < g.nd; ++i) {
    const int v = g.starts[b * g.nd + i];
    const int ax = 3 - g.nd + i;
    st[ax] = min(max(v, 0), g.ishape[ax] - g.psz[ax]);
    ms[ax] = min(max(v, 0), g.mshape[ax] - g.psz[ax]);
  }
  const T* img = static_cast<const T*>(g.img);
  const int py = g.psz[1], px = g.psz[2];
  const int rows = g.psz[0] * py;
  const int per = (rows + kGatherChunks - 1) / kGatherChunks;
  const int r0 = min(rows, chunk * per), r1 = min(rows, r0 + per);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* part = partial + ((long long)(blockIdx.z * gridDim.y + b) * kGatherChunks) * 2;
  float mu = g.mean;
  if (PHASE == 1 && !g.use_mean) {
    double s = 0.0, c = 0.0;
    for (int k = 0; k < kGatherChunks; ++k) {
      s += part[2 * k];
      c += part[2 * k + 1];
    }
    mu = static_cast<float>(s / c);  // NaN when all masked
  }
  double s = 0.0, c = 0.0;
  // Four rows x two 64-wide column groups per round, loads issued as one
  // straight-line group (clamped rows / columns, mask test hoisted); the sums
  // are still added in (row, column group) order.
  constexpr int kR = 4, kG = 2, kStep = kBlock / 64;
  for (int rb = r0 + wave; rb < r1; rb += kStep * kR) {
    for (int xb = 0; xb < px; xb += 64 * kG) {
      float v[kR][kG];
      bool m[kR][kG];
      long long io[kR], mo[kR];
#pragma unroll
      for (int j = 0; j < kR; ++j) {
        const int r = min(rb + j * kStep, r1 - 1);
        const int z = r / py, y = r - z * py;
        io[j] = ((long long)(st[0] + z) * g.ishape[1] + (st[1] + y)) * g.ishape[2] + st[2];
        mo[j] = ((long long)(ms[0] + z) * g.mshape[1] + (ms[1] + y)) * g.mshape[2] + ms[2];
      }
      if (g.mask) {
#pragma unroll
        for (int j = 0; j < kR; ++j)
#pragma unroll
          for (int k = 0; k < kG; ++k) {
            const int x = min(xb + lane + 64 * k, px - 1);
            v[j][k] = static_cast<float>(img[io[j] + x]);
            m[j][k] = g.mask[mo[j] + x] != 0;
          }
      } else {
#pragma unroll
        for (int j = 0; j < kR; ++j)
#pragma unroll
          for (int k = 0; k < kG; ++k) {
            v[j][k] = static_cast<float>(img[io[j] + min(xb + lane + 64 * k, px - 1)]);
            m[j][k] = false;
          }
      }
#pragma unroll
      for (int j = 0; j < kR; ++j)
#pragma unroll
        for (int k = 0; k < kG; ++k) {
          const int r = rb + j * kStep, x = xb + lane + 64 * k;
          if (r >= r1 || x >= px) continue;
          if (PHASE == 0) {
            if (!m[j][k]) {
              s += v[j][k];
              c += 1.0;
            }
          } else {
            const long long o = b * g.pn + (long long)r * px + x;
            g.out[o] = m[j][k] ? 0.f : v[j][k] - mu;
            if (g.valid) g.valid[o] = m[j][k] ? 0.f : 1.f;
          }
        }
    }
  }
  if (PHASE == 0) {
    s_sum[threadIdx.x] = s;
    s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) {
      if (threadIdx.x < k) {
        s_sum[threadIdx.x] += s_sum[threadIdx.x + k];
        s_cnt[threadIdx.x] += s_cnt[threadIdx.x + k];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      part[2 * chunk] = s_sum[0];
      part[2 * chunk + 1] = s_cnt[0];
    }
  }
}

// ---------------------------------------------------------------------------
// general direct correlation (any dim / dtype / masks)
// ---------------------------------------------------------------------------
struct CorrArgs {
  const float* a;   // [B, P] centred, masked pixels zeroed
  const float* b;   // [B, Q]
  const float* va;  // validity planes or null
  const float* vb;
  Geo g;
  float* out;       // unmasked: surface; masked: numerator
  float* den;       // masked only
  float* ov;        // masked only
  unsigned int* maxima;  // masked only: bits of max |den|, max overlap
};

template <bool MASKED>
__global__ void __launch_bounds__(kBlock) corr_direct_kernel(CorrArgs c) {
  const Geo& g = c.g;
  const int kx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ky = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int kz = blockIdx.z % g.S[0];
  const int bi = blockIdx.z / g.S[0];
  const bool live = kx < g.S[2] && ky < g.S[1];
  float xc = 0.f, sa = 0.f, sb = 0.f, nov = 0.f, qa = 0.f, qb = 0.f;
  if (live) {
    const int dz = kz - (g.Q[0] - 1), dy = ky - (g.Q[1] - 1),
              dx = kx - (g.Q[2] - 1);
    const int z0 = max(0, -dz), z1 = min(g.Q[0], g.P[0] - dz);
    const int y0 = max(0, -dy), y1 = min(g.Q[1], g.P[1] - dy);
    const int x0 = max(0, -dx), x1 = min(g.Q[2], g.P[2] - dx);
    const float* A = c.a + bi * g.Pn;
    const float* Bp = c.b + bi * g.Qn;
    const float* VA = MASKED ? c.va + bi * g.Pn : nullptr;
    const float* VB = MASKED ? c.vb + bi * g.Qn : nullptr;
    for (int z = z0; z < z1; ++z)
      for (int y = y0; y < y1; ++y) {
        const long long ao =
            ((long long)(z + dz) * g.P[1] + (y + dy)) * g.P[2] + dx;
        const long long bo = ((long long)z * g.Q[1] + y) * g.Q[2];
        for (int x = x0; x < x1; ++x) {
          const float av = A[ao + x];
          const float bv = Bp[bo + x];
          xc = fmaf(av, bv, xc);
          if (MASKED) {
            const float wa = VA[ao + x];
            const float wb = VB[bo + x];
            sa = fmaf(av, wb, sa);
            sb = fmaf(wa, bv, sb);
            nov = fmaf(wa, wb, nov);
            qa = fmaf(av * av, wb, qa);
            qb = fmaf(wa, bv * bv, qb);
          }
        }
      }
  }
  if (!live) return;
  const long long o =
      bi * g.Sn + ((long long)kz * g.S[1] + ky) * g.S[2] + kx;
  if (!MASKED) {
    c.out[o] = xc;
    return;
  }
  // Padfield assembly (flow_field.py:113-131).
  float ovv = fmaxf(rintf(nov), kEps);
  const float inv = 1.0f / ovv;
  const float num = xc - sa * sb * inv;
  const float pd = fmaxf(qa - sa * sa * inv, 0.f);
  const float cd = fmaxf(qb - sb * sb * inv, 0.f);
  const float den = sqrtf(pd * cd);
  c.out[o] = num;
  c.den[o] = den;
  c.ov[o] = ovv;
  // Batch-global maxima (flow_field.py:137, 151); values are >= 0 so the
  // integer order of the bit patterns is the float order.
  // (same-address atomics serialise: only the elements that raise a maximum)
  if (__float_as_uint(fabsf(den)) > __atomic_load_n(&c.maxima[0], __ATOMIC_RELAXED))
    atomicMax(&c.maxima[0], __float_as_uint(fabsf(den)));
  if (__float_as_uint(ovv) > __atomic_load_n(&c.maxima[1], __ATOMIC_RELAXED))
    atomicMax(&c.maxima[1], __float_as_uint(ovv));
}

__global__ void __launch_bounds__(kBlock)
masked_finalize_kernel(float* __restrict__ out, const float* __restrict__ den,
                       const float* __restrict__ ov,
                       const unsigned int* __restrict__ maxima, long long n) {
  const float tol = 1e3f * kEps * __uint_as_float(maxima[0]);
  const float px_thr = 0.3f * __uint_as_float(maxima[1]);
  for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n;
       i += (long long)gridDim.x * kBlock) {
    const float dn = den[i];
    float v = dn > tol ? out[i] / dn : 0.f;
    v = fminf(fmaxf(v, -1.f), 1.f);
    if (ov[i] < px_thr) v = 0.f;
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------
// patch selection: masked-pixel count per grid patch
// ---------------------------------------------------------------------------
struct MaskCountArgs {
  const unsigned char* mask;
  int S[3], P[3], T[3], O[3];  // mask shape, patch, step, output grid
  int* out;
};

__global__ void __launch_bounds__(kBlock) mask_count_kernel(MaskCountArgs a) {
  __shared__ int red[kBlock];
  const long long o = blockIdx.x;
  const int ox = static_cast<int>(o % a.O[2]);
  const int oy = static_cast<int>((o / a.O[2]) % a.O[1]);
  const int oz = static_cast<int>(o / ((long long)a.O[2] * a.O[1]));
  const long long base =
      ((long long)oz * a.T[0] * a.S[1] + (long long)oy * a.T[1]) * a.S[2] +
      (long long)ox * a.T[2];
  const long long rows = (long long)a.P[0] * a.P[1];
  int cnt = 0;
  for (long long i = threadIdx.x; i < rows * a.P[2]; i += kBlock) {
    const int x = static_cast<int>(i % a.P[2]);
    const long long r = i / a.P[2];
    const int y = static_cast<int>(r % a.P[1]);
    const int z = static_cast<int>(r / a.P[1]);
    cnt += a.mask[base + ((long long)z * a.S[1] + y) * a.S[2] + x] != 0;
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int k = kBlock / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.out[o] = red[0];
}

// Same counts for patch rows that are not too wide: one workgroup per (tile of
// x outputs, oy, oz) adds the mask rows of the window column-wise (16 columns
// per thread, one 16-byte load per row), prefix-sums the column sums in LDS and
// writes every output of the tile as a difference of two prefix values.  Each
// mask byte is read P_y / T_y times instead of P_y P_x / (T_y T_x) times.
constexpr int kMcCols = 16 * kBlock;  // columns per tile

// 1 in every byte of w that is not zero
__device__ __forceinline__ unsigned nonzero_bytes(unsigned w) {
  return ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u) >> 7;
}

// `splits` > 1: few outputs with tall windows (whole-overlap patches): the window
// rows are divided among `splits` workgroups per tile, which add their partial
// counts atomically into the zeroed output (integers: order does not matter).
__global__ void __launch_bounds__(kBlock) mask_count_rows_kernel(MaskCountArgs a, int ot,
                                                                 int splits) {
  __shared__ int pref[kMcCols + 1];
  __shared__ int wsum[kBlock / 64];
  const int oy = blockIdx.y, oz = blockIdx.z;
  const int tile = blockIdx.x / splits, split = blockIdx.x - tile * splits;
  const int ox0 = tile * ot;
  const int nout = min(ot, a.O[2] - ox0);
  const int x0 = ox0 * a.T[2];
  const int ncol = (nout - 1) * a.T[2] + a.P[2];  // <= kMcCols
  const int c0 = 16 * threadIdx.x;
  const long long rows = (long long)a.P[0] * a.P[1];
  const long long per = (rows + splits - 1) / splits;
  const long long r_lo = min(rows, split * per), r_hi = min(rows, r_lo + per);
  int cs[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) cs[j] = 0;
  unsigned pk[4] = {0, 0, 0, 0};  // packed byte counters, flushed before they overflow
  int in_pk = 0;
  const bool whole = c0 + 16 <= ncol;  // x0 + ncol <= S[2] by construction
  for (long long r = r_lo; r < r_hi; ++r) {
    const int z = static_cast<int>(r / a.P[1]), y = static_cast<int>(r - (long long)z * a.P[1]);
    const unsigned char* row =
        a.mask + (((long long)oz * a.T[0] + z) * a.S[1] + (long long)oy * a.T[1] + y) * a.S[2] +
        x0 + c0;
    unsigned w[4] = {0, 0, 0, 0};
    if (whole) {
      __builtin_memcpy(w, row, 16);
    } else {
      for (int j = 0; j < 16; ++j)
        if (c0 + j < ncol) w[j >> 2] |= static_cast<unsigned>(row[j]) << (8 * (j & 3));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] += nonzero_bytes(w[k]);
    if (++in_pk == 255 || r + 1 == r_hi) {
#pragma unroll
      for (int j = 0; j < 16; ++j) cs[j] += (pk[j >> 2] >> (8 * (j & 3))) & 0xffu;
      pk[0] = pk[1] = pk[2] = pk[3] = 0;
      in_pk = 0;
    }
  }
  // inclusive prefix over the tile's columns
  int run = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    run += cs[j];
    cs[j] = run;
  }
  int incl = run;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - run;
  for (int k = 0; k < wave; ++k) base += wsum[k];
#pragma unroll
  for (int j = 0; j < 16; ++j) pref[c0 + j + 1] = base + cs[j];
  if (threadIdx.x == 0) pref[0] = 0;
  __syncthreads();
  for (int o = threadIdx.x; o < nout; o += kBlock) {
    int* dst = &a.out[((long long)oz * a.O[1] + oy) * a.O[2] + ox0 + o];
    const int cnt = pref[o * a.T[2] + a.P[2]] - pref[o * a.T[2]];
    if (splits > 1)
      atomicAdd(dst, cnt);
    else
      *dst = cnt;
  }
}

// Patches wider than a tile (whole-overlap strips): every output is cut into
// column segments of kMcCols and row ranges; a workgroup counts one piece and
// adds it atomically to the zeroed output.
__global__ void __launch_bounds__(kBlock) mask_count_wide_kernel(MaskCountArgs a, int nseg,
                                                                 int splits) {
  __shared__ int red[kBlock];
  const long long o = blockIdx.y;
  const int ox = static_cast<int>(o % a.O[2]);
  const int oy = static_cast<int>((o / a.O[2]) % a.O[1]);
  const int oz = static_cast<int>(o / ((long long)a.O[2] * a.O[1]));
  const int seg = blockIdx.x / splits, split = blockIdx.x - seg * splits;
  const int c0 = seg * kMcCols + 16 * threadIdx.x;  // first column of this thread
  const long long rows = (long long)a.P[0] * a.P[1];
  const long long per = (rows + splits - 1) / splits;
  const long long r_lo = min(rows, split * per), r_hi = min(rows, r_lo + per);
  const int nb = min(16, a.P[2] - c0);  // bytes of this thread inside the window
  int cnt = 0;
  if (nb > 0) {
    for (long long r = r_lo; r < r_hi; ++r) {
      const int z = static_cast<int>(r / a.P[1]), y = static_cast<int>(r - (long long)z * a.P[1]);
      const unsigned char* row =
          a.mask + (((long long)oz * a.T[0] + z) * a.S[1] + (long long)oy * a.T[1] + y) * a.S[2] +
          (long long)ox * a.T[2] + c0;
      unsigned w[4] = {0, 0, 0, 0};
      if (nb == 16) {
        __builtin_memcpy(w, row, 16);
      } else {
        for (int j = 0; j < nb; ++j) w[j >> 2] |= static_cast<unsigned>(row[j]) << (8 * (j & 3));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cnt += __builtin_popcount(nonzero_bytes(w[k]));
    }
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int k = kBlock / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0]) atomicAdd(&a.out[o], red[0]);
}

// ---------------------------------------------------------------------------
// peaks
// ---------------------------------------------------------------------------
struct PeakArgs {
  const float* surf;  // [B, S] with row pitch / per-surface stride below
  int nd;
  int S[3];
  long long Sn;           // logical elements per surface
  int pitch;              // floats between consecutive x-rows (>= S[2])
  long long bstride;      // floats between consecutive surfaces
  int batch;
  float center[3];  // [z]yx
  int min_distance;
  float threshold_rel;
  int radius[3];
  // workspace
  int* idx1;               // [B]
  float* v1;               // [B]
  int* zero_is_peak;       // [B]
  int* cand_count;         // [B]
  float* cand_val;         // [B, kCandCap]
  int* cand_idx;           // [B, kCandCap]
  unsigned int* bitmap;    // [n_groups, bitmap_words]
  int group;               // rows per coupling group
  int bitmap_words;
  unsigned int* smax;        // [B] ordered bits of the surface maximum (large surfaces)
  unsigned long long* best;  // [B] packed (value, index) arg-max (large surfaces)
  float* out;              // [B, nd + 2]
  const int* skipmask;     // [B] or NULL: 16-row tiles of the surface that were never stored
  // masked matrix-core path: rows whose largest possible overlap ny * Qx is below
  // 0.3 x the batch maximum of the overlap are zero (flow_field.py:151-155); the
  // large-surface sweeps leave them out
  const unsigned int* live_ovmax;  // [groups, 2] (+1): float bits of the batch maximum of the overlap, or NULL
  // masked matrix-core path: maximum of every block of blk_rows surface rows, or NULL
  const float* blkmax;
  int blk_rows, blk_n;
  int live_py, live_qy, live_qx;
};

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
  return v > bv || (v == bv && i < bi);
}

// (value, index) arg-max with first-index tie break across the block.
// (`better` is a total order on (value, -index): any reduction tree gives the same
// winner.  Butterfly over the wave on the shuffle network, then the waves' winners
// through LDS: two barriers instead of the nine of an LDS tree -- the peak kernels
// run one workgroup per surface and are priced by their chains of round trips.)
__device__ void block_argmax(float* v, int* i, float* lv, int* li) {
  float bv = *v;
  int bi = *i;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float ov = __shfl_xor(bv, d, 64);
    const int oi = __shfl_xor(bi, d, 64);
    if (better(ov, oi, bv, bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    lv[threadIdx.x >> 6] = bv;
    li[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  bv = lv[0];
  bi = li[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w)
    if (better(lv[w], li[w], bv, bi)) {
      bv = lv[w];
      bi = li[w];
    }
  *v = bv;
  *i = bi;
  __syncthreads();
}

// Element (z, y, x) of a surface with row pitch.
__device__ __forceinline__ float surf_at(const float* s, const PeakArgs& p, int z,
                                         int y, int x) {
  return s[((long long)z * p.S[1] + y) * p.pitch + x];
}

// img == maxfilter(img) (zero 'same' padding) for an element already known to
// exceed the threshold (flow_field.py:238-254).
// FAST: the 5 x 5 window of the default min_distance as 25 loads in flight (costs
// registers: not for kernels where this is the rare path).
template <bool FAST>
__device__ bool is_window_max(const float* s, const PeakArgs& p, int z, int y,
                              int x, float v) {
  const int m = p.min_distance;
  const int mz = p.nd == 3 ? m : 0;
  // window maximum from clamped addresses: unconditional loads (all in flight
  // together); a clamped position repeats an element of the window, which cannot
  // change the maximum
  float wm = -INFINITY;
  const bool outside = z - mz < 0 || z + mz >= p.S[0] || y - m < 0 || y + m >= p.S[1] ||
                       x - m < 0 || x + m >= p.S[2];
  if (FAST && m == 2) {  // the default min_distance: a plane's 25 loads in flight together
    for (int dz = -mz; dz <= mz; ++dz) {
      float w[25];
      const int zz = min(max(z + dz, 0), p.S[0] - 1);
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx)
          w[(dy + 2) * 5 + dx + 2] = surf_at(s, p, zz, min(max(y + dy, 0), p.S[1] - 1),
                                             min(max(x + dx, 0), p.S[2] - 1));
#pragma unroll
      for (int k = 0; k < 25; ++k) wm = fmaxf(wm, w[k]);
    }
  } else {
    for (int dz = -mz; dz <= mz; ++dz)
      for (int dy = -m; dy <= m; ++dy)
        for (int dx = -m; dx <= m; ++dx)
          wm = fmaxf(wm, surf_at(s, p, min(max(z + dz, 0), p.S[0] - 1),
                                 min(max(y + dy, 0), p.S[1] - 1),
                                 min(max(x + dx, 0), p.S[2] - 1)));
  }
  if (outside) wm = fmaxf(wm, 0.f);
  return v == wm;
}

// Row sweeps of the peak search: a wave takes whole rows and issues the loads of
// a piece -- G 64-wide column groups of R consecutive rows of its share, G R = 8
// -- before it looks at the first value (clamped addresses, no load under a
// per-lane condition: otherwise every load is waited for on its own).  G is the
// smallest of 1, 2, 4, 8 that covers the row, so narrow rows do not pay for
// column groups that only repeat their last element.
constexpr int kPiece = 8;

template <int G, typename Body>
__device__ __forceinline__ void sweep_rows(const float* s, const PeakArgs& p, int r0, int rows,
                                           Body body) {
  constexpr int R = kPiece / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = p.S[2];
  constexpr int kStep = kBlock / 64;  // rows are dealt to the waves round robin
  for (int rb = r0 + wave; rb < rows; rb += kStep * R) {
    for (int xb = 0; xb < w; xb += 64 * G) {
      float v[R][G];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const float* row = s + (long long)min(rb + j * kStep, rows - 1) * p.pitch;
#pragma unroll
        for (int k = 0; k < G; ++k) v[j][k] = row[min(xb + lane + 64 * k, w - 1)];
      }
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const int r = rb + j * kStep, x = xb + lane + 64 * k;
          if (r < rows && x < w) body(r, x, v[j][k]);
        }
    }
  }
}

template <typename Body>
__device__ __forceinline__ void sweep_rows_any(const float* s, const PeakArgs& p, int r0, int rows,
                                               Body body) {
  const int w = p.S[2];
  if (w <= 64)
    sweep_rows<1>(s, p, r0, rows, body);
  else if (w <= 128)
    sweep_rows<2>(s, p, r0, rows, body);
  else if (w <= 256)
    sweep_rows<4>(s, p, r0, rows, body);
  else
    sweep_rows<8>(s, p, r0, rows, body);
}

__device__ float surface_max(const float* s, const PeakArgs& p, float* lv,
                             int* li, int r0 = 0, int r1 = -1) {
  const int rows = r1 < 0 ? p.S[0] * p.S[1] : r1;
  float mx = -INFINITY;
  sweep_rows_any(s, p, r0, rows, [&](int, int, float v) { mx = fmaxf(mx, v); });
  int dummy = 0;
  block_argmax(&mx, &dummy, lv, li);
  return mx;
}

// Calls fn(flat_index, value) for every peak of the surface.
template <bool FAST = true, typename F>
__device__ void for_each_peak(const float* s, const PeakArgs& p, float thr, F fn,
                              int r0 = 0, int r1 = -1) {
  const int rows = r1 < 0 ? p.S[0] * p.S[1] : r1;
  const int w = p.S[2];
  sweep_rows_any(s, p, r0, rows, [&](int r, int x, float v) {
    if (v > thr) {
      const int z = r / p.S[1], y = r - z * p.S[1];
      if (is_window_max<FAST>(s, p, z, y, x, v)) fn(r * w + x, v);
    }
  });
}

__global__ void __launch_bounds__(kBlock) peaks_first_kernel(PeakArgs p) {
  __shared__ float lv[kBlock];
  __shared__ int li[kBlock];
  const int b = blockIdx.x;
  const float* s = p.surf + b * p.bstride;
  const float mx = surface_max(s, p, lv, li);
  const float thr = p.threshold_rel * mx;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for_each_peak(s, p, thr, [&](int i, float v) {
    if (better(v, i, bv, bi)) {
      bv = v;
      bi = i;
    }
    const int slot = atomicAdd(&p.cand_count[b], 1);
    if (slot < kCandCap) {
      p.cand_val[(long long)b * kCandCap + slot] = v;
      p.cand_idx[(long long)b * kCandCap + slot] = i;
    }
    if (i == 0) p.zero_is_peak[b] = 1;
  });
  block_argmax(&bv, &bi, lv, li);
  if (threadIdx.x == 0) {
    const int i1 = bv == -INFINITY ? 0 : bi;  // argmax of an all -inf row is 0
    p.idx1[b] = i1;
    p.v1[b] = bv;
    sfm::set_bit_once(&p.bitmap[(long long)(b / p.group) * p.bitmap_words + (i1 >> 5)],
                       1u << (i1 & 31));
  }
}

// ---- first pass over LARGE surfaces (3-D, whole-overlap patches): several
// workgroups per surface.  Floats are ordered through a monotonic uint map so
// that the surface maximum and the (value, lowest index) arg-max can be merged
// with integer atomics; the result is identical to peaks_first_kernel.
__device__ __forceinline__ unsigned ord_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_float(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ void chunk_rows(const PeakArgs& p, int b, int* r0, int* r1) {
  int lo = 0, hi = p.S[0] * p.S[1];
  if (p.live_ovmax && p.S[0] == 1) {
    // ny(ky) rises to min(Py, Qy) and falls again: the live rows are one range
    // (two maxima per reference batch: denominator, overlap)
    const float thr = 0.3f * __uint_as_float(p.live_ovmax[2 * (b / p.group)]);
    auto dead = [&](int ky) {
      const int dy = ky - (p.live_qy - 1);
      const int ny = min(p.live_py, p.live_qy + dy) - max(0, dy);
      return static_cast<float>(ny * p.live_qx) < thr;
    };
    while (lo < hi && dead(lo)) ++lo;
    while (hi > lo && dead(hi - 1)) --hi;
  }
  const int rows = hi - lo;
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  *r0 = lo + min(rows, static_cast<int>(blockIdx.x) * per);
  *r1 = lo + min(rows, static_cast<int>(blockIdx.x) * per + per);
}

__global__ void __launch_bounds__(kBlock) peaks_max_kernel(PeakArgs p) {
  __shared__ float lv[kBlock];
  __shared__ int li[kBlock];
  const int b = blockIdx.y;
  int r0, r1;
  chunk_rows(p, b, &r0, &r1);
  const float mx = surface_max(p.surf + b * p.bstride, p, lv, li, r0, r1);
  if (threadIdx.x == 0 && r1 > r0) atomicMax(&p.smax[b], ord_bits(mx));
}

__global__ void __launch_bounds__(kBlock) peaks_scan_kernel(PeakArgs p) {
  __shared__ float lv[kBlock];
  __shared__ int li[kBlock];
  // (with block maxima most surfaces cost a handful of loads: a workgroup walks
  // several of them, or the launch is bound by the workgroup dispatch rate)
  for (int b = blockIdx.y; b < p.batch; b += gridDim.y) {
  const float* s = p.surf + b * p.bstride;
  int r0, r1;
  chunk_rows(p, b, &r0, &r1);
  const float thr = p.threshold_rel * ord_float(p.smax[b]);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  auto found = [&](int i, float v) {
    if (better(v, i, bv, bi)) {
      bv = v;
      bi = i;
    }
    const int slot = atomicAdd(&p.cand_count[b], 1);
    if (slot < kCandCap) {
      p.cand_val[(long long)b * kCandCap + slot] = v;
      p.cand_idx[(long long)b * kCandCap + slot] = i;
    }
    if (i == 0) p.zero_is_peak[b] = 1;
  };
  if (p.blkmax) {
    // a peak exceeds thr: blocks of rows whose maximum does not are not read
    // (lane k of every wave looks at block k: one round trip for the whole list)
    const int lane = threadIdx.x & 63;
    const int kc = min(lane, p.blk_n - 1);
    const bool hot = p.blkmax[(long long)b * p.blk_n + kc] > thr && lane < p.blk_n &&
                     lane * p.blk_rows < r1 && (lane + 1) * p.blk_rows > r0;
    unsigned long long todo = __ballot(hot);
    while (todo) {
      const int k = __builtin_ctzll(todo);
      todo &= todo - 1;
      for_each_peak(s, p, thr, found, max(r0, k * p.blk_rows), min(r1, (k + 1) * p.blk_rows));
    }
  } else {
    for_each_peak(s, p, thr, found, r0, r1);
  }
  block_argmax(&bv, &bi, lv, li);
  if (threadIdx.x == 0 && bv != -INFINITY)
    atomicMax(&p.best[b], (static_cast<unsigned long long>(ord_bits(bv)) << 32) |
                              (0xffffffffu - static_cast<unsigned>(bi)));
  }
}

__global__ void __launch_bounds__(kBlock) peaks_first_finish_kernel(PeakArgs p) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= p.batch) return;
  const unsigned long long k = p.best[b];
  const float bv = k ? ord_float(static_cast<unsigned>(k >> 32)) : -INFINITY;
  const int i1 = k ? static_cast<int>(0xffffffffu - static_cast<unsigned>(k)) : 0;
  p.idx1[b] = i1;
  p.v1[b] = bv;
  sfm::set_bit_once(&p.bitmap[(long long)(b / p.group) * p.bitmap_words + (i1 >> 5)],
                       1u << (i1 & 31));
}

// Second peak + sharpness of surface b.  WAVE: by one wave (lanes instead of
// threads, reductions on the shuffle network, no barrier) -- the common case, a
// candidate list that did not overflow; otherwise by the whole workgroup.
template <bool WAVE>
__device__ void second_peak_body(const PeakArgs& p, int b, float* lv, int* li) {
  constexpr int NT = WAVE ? 64 : kBlock;
  const int tid = WAVE ? static_cast<int>(threadIdx.x & 63) : static_cast<int>(threadIdx.x);
  const float* s = p.surf + b * p.bstride;
  const float v1 = p.v1[b];
  const int i1 = p.idx1[b];
  const int w = p.nd + 2;
  const unsigned int* bitmap = p.bitmap + (long long)(b / p.group) * p.bitmap_words;
  if (v1 == -INFINITY) {
    if (tid < w) p.out[b * w + tid] = NAN;
    return;
  }
  // Sharpness window (flow_field.py:186-192): its position depends on the first
  // peak only, so its loads go out first and return behind the candidate list's
  // (one round trip less in this kernel's chain).
  int pos[3], start[3], size[3];
  {
    long long r = i1;
    pos[2] = static_cast<int>(r % p.S[2]);
    r /= p.S[2];
    pos[1] = static_cast<int>(r % p.S[1]);
    pos[0] = static_cast<int>(r / p.S[1]);
  }
  long long wn = 1;
  for (int a = 0; a < 3; ++a) {
    size[a] = (a == 0 && p.nd == 2) ? 1 : 2 * p.radius[a] + 1;
    size[a] = min(size[a], p.S[a]);
    start[a] = min(max(pos[a] - size[a] / 2, 0), p.S[a] - size[a]);
    wn *= size[a];
  }
  float mn = INFINITY;
  for (long long k = tid; k < wn; k += NT) {
    const int x = static_cast<int>(k % size[2]);
    const long long r = k / size[2];
    const int y = static_cast<int>(r % size[1]);
    const int z = static_cast<int>(r / size[1]);
    mn = fminf(mn, surf_at(s, p, start[0] + z, start[1] + y, start[2] + x));
  }
  const float peak_val = surf_at(s, p, pos[0], pos[1], pos[2]);
  // Second peak: best candidate whose flat index is not a first-peak index of
  // ANY surface in the batch (flow_field.py:263-265).
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  const int cnt = p.cand_count[b];
  if (WAVE || cnt <= kCandCap) {
    for (int k = tid; k < cnt; k += NT) {
      const float v = p.cand_val[(long long)b * kCandCap + k];
      const int i = p.cand_idx[(long long)b * kCandCap + k];
      if ((bitmap[i >> 5] >> (i & 31)) & 1u) continue;
      if (better(v, i, bv, bi)) {
        bv = v;
        bi = i;
      }
    }
  } else if constexpr (!WAVE) {
    // Candidate list overflowed (plateaus): rescan the surface.  Row tiles the
    // correlation kernel pruned were never stored (all their elements are below
    // the threshold): zeros for the sweep.
    const int skipped = p.skipmask ? p.skipmask[b] : 0;
    if (skipped) {
      float* sw = const_cast<float*>(s);
      for (int t = 0; t < 32; ++t) {
        if (!((skipped >> t) & 1)) continue;
        const int y1 = min(16 * t + 16, p.S[1]);
        for (int i = threadIdx.x; i < (y1 - 16 * t) * p.S[2]; i += kBlock)
          sw[(long long)(16 * t + i / p.S[2]) * p.pitch + i % p.S[2]] = 0.f;
      }
      __threadfence_block();
      __syncthreads();
    }
    const float mx = surface_max(s, p, lv, li);
    for_each_peak<false>(s, p, p.threshold_rel * mx, [&](int i, float v) {
      if (((bitmap[i >> 5] >> (i & 31)) & 1u) == 0 && better(v, i, bv, bi)) {
        bv = v;
        bi = i;
      }
    });
  }
  if constexpr (WAVE) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const float ov = __shfl_xor(bv, d, 64);
      const int oi = __shfl_xor(bi, d, 64);
      if (better(ov, oi, bv, bi)) {
        bv = ov;
        bi = oi;
      }
    }
  } else {
    block_argmax(&bv, &bi, lv, li);
  }
  // The value is read from the UN-suppressed array (flow_field.py:266-268):
  // with nothing left the arg-max is index 0, whose value is the surface value
  // there if index 0 is itself a peak.
  float v2 = bv;
  if (bv == -INFINITY && p.zero_is_peak[b]) v2 = s[0];

  // Sharpness: peak / min over the clamped window (loaded above).
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mn = fminf(mn, __shfl_xor(mn, d, 64));
  if constexpr (!WAVE) {
    if ((threadIdx.x & 63) == 0) lv[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
      for (int w2 = 0; w2 < kBlock / 64; ++w2) mn = fminf(mn, lv[w2]);
  }
  if (tid == 0) {
    float* o = p.out + b * w;
    // x, y[, z] = reversed axis order
    for (int a = 0; a < p.nd; ++a) {
      const int ax = 2 - a;
      o[a] = static_cast<float>(pos[ax]) - p.center[ax];
    }
    o[p.nd] = peak_val / mn;
    o[p.nd + 1] = v2 == -INFINITY ? 0.f : v1 / v2;
  }
}


// Four surfaces per workgroup, a wave each (40 401 workgroups of one short chain
// of round trips were priced by their dispatch); a workgroup in which any
// candidate list overflowed takes its surfaces one after the other instead.
__global__ void __launch_bounds__(kBlock) peaks_second_kernel(PeakArgs p) {
  __shared__ float lv[kBlock];
  __shared__ int li[kBlock];
  constexpr int kPer = kBlock / 64;
  const int b = blockIdx.x * kPer + static_cast<int>(threadIdx.x >> 6);
  const bool live = b < p.batch;
  const int cnt = live ? p.cand_count[b] : 0;
  if (!__syncthreads_or(cnt > kCandCap ? 1 : 0)) {
    if (live) second_peak_body<true>(p, b, nullptr, nullptr);
    return;
  }
  for (int w2 = 0; w2 < kPer; ++w2) {
    const int bb = blockIdx.x * kPer + w2;
    if (bb < p.batch) second_peak_body<false>(p, bb, lv, li);
    __syncthreads();
  }
}

// Copies a tile-padded surface [B, rows, pitch] into the compact [B, Sy, Sx].
__global__ void __launch_bounds__(kBlock)
compact_surface_kernel(const float* __restrict__ src, float* __restrict__ dst,
                       int sy, int sx, int rows, int pitch) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < sy * sx;
       i += gridDim.x * kBlock) {
    const int y = i / sx, x = i - y * sx;
    dst[(long long)b * sy * sx + i] = src[((long long)b * rows + y) * pitch + x];
  }
}

struct PeakWs {
  int* idx1;
  float* v1;
  int* zero_is_peak;
  int* cand_count;
  float* cand_val;
  int* cand_idx;
  unsigned int* bitmap;
  int group, bitmap_words;
  unsigned int* smax;
  float* blkmax;       // masked MFMA path only: [batch, kBlkMaxN]
  unsigned long long* best;
  int* hot_count;      // fused MFMA path only
  int* skipmask;       // fused MFMA path: pruned row tiles per surface
  float* hot_val;
  int* hot_idx;
  size_t zero_from, zero_bytes;  // region that must be cleared per batch
  size_t bytes;
};

constexpr int kBlkMaxN = 16;   // blocks of sfm::kMaskedBlkRows rows: surfaces up to 512 rows

PeakWs carve_peaks(sfm::Carver& c, int batch, long long sn, bool hot = false,
                   int group = 0, bool blk = false) {
  PeakWs w;
  w.blkmax = blk ? c.take<float>((size_t)batch * kBlkMaxN) : nullptr;
  w.group = group > 0 && group < batch ? group : batch;
  w.bitmap_words = static_cast<int>((sn + 31) / 32);
  w.idx1 = c.take<int>(batch);
  w.v1 = c.take<float>(batch);
  w.hot_val = hot ? c.take<float>((size_t)batch * kHotCap) : nullptr;
  w.hot_idx = hot ? c.take<int>((size_t)batch * kHotCap) : nullptr;
  w.cand_val = c.take<float>((size_t)batch * kCandCap);
  w.cand_idx = c.take<int>((size_t)batch * kCandCap);
  const size_t z0 = sfm::align_up(c.off, 256);
  w.zero_is_peak = c.take<int>(batch);
  w.cand_count = c.take<int>(batch);
  w.hot_count = c.take<int>(batch);
  w.skipmask = c.take<int>(batch);
  w.smax = c.take<unsigned int>(batch);
  w.best = c.take<unsigned long long>(batch);
  w.bitmap = c.take<unsigned int>(
      (size_t)((batch + w.group - 1) / w.group) * w.bitmap_words);
  w.zero_from = z0;
  w.zero_bytes = c.total() - z0;
  w.bytes = c.total();
  return w;
}

int run_peaks(const PeakWs& w, char* ws_base, const float* surf, int pitch,
              long long bstride, int nd, const int* S, long long sn, int batch,
              const float* center,
              int min_distance, float threshold_rel, const int* radius,
              float* out, hipStream_t st, bool first_pass_done = false,
              bool smax_done = false, const unsigned int* live_ovmax = nullptr,
              const int* live_geo = nullptr, bool use_blkmax = false) {
  PeakArgs p;
  p.blkmax = use_blkmax ? w.blkmax : nullptr;
  p.blk_rows = sfm::kMaskedBlkRows;
  p.blk_n = (S[1] + sfm::kMaskedBlkRows - 1) / sfm::kMaskedBlkRows;
  p.live_ovmax = live_ovmax;
  p.live_py = live_geo ? live_geo[0] : 0;
  p.live_qy = live_geo ? live_geo[1] : 0;
  p.live_qx = live_geo ? live_geo[2] : 0;
  p.surf = surf;
  p.nd = nd;
  for (int i = 0; i < 3; ++i) {
    p.S[i] = S[i];
    p.center[i] = center[i];
    p.radius[i] = radius[i];
  }
  p.Sn = sn;
  p.pitch = pitch;
  p.bstride = bstride;
  p.batch = batch;
  p.min_distance = min_distance;
  p.threshold_rel = threshold_rel;
  p.idx1 = w.idx1;
  p.v1 = w.v1;
  p.zero_is_peak = w.zero_is_peak;
  p.skipmask = w.skipmask;
  p.cand_count = w.cand_count;
  p.cand_val = w.cand_val;
  p.cand_idx = w.cand_idx;
  p.bitmap = w.bitmap;
  p.group = w.group;
  p.bitmap_words = w.bitmap_words;
  p.smax = w.smax;
  p.best = w.best;
  p.out = out;
  if (!first_pass_done) {
    // smax_done: the producer of the surfaces already left their maxima in
    // w.smax (and the per-batch state was cleared before it ran)
    if (!smax_done)
      SFM_HIP_CHECK(hipMemsetAsync(ws_base + w.zero_from, 0, w.zero_bytes, st));
    if (sn >= (1LL << 18) || smax_done) {
      // large surfaces: ~32 K elements per workgroup
      const long long rows = (long long)S[0] * S[1];
      const int chunks = static_cast<int>(std::min<long long>(
          std::min<long long>(rows, 1024), std::max<long long>(1, sn >> 15)));
      if (!smax_done)
        hipLaunchKernelGGL(peaks_max_kernel, dim3(chunks, batch), dim3(kBlock), 0, st, p);
      if (p.blkmax)   // one workgroup per surface, at most 4096 of them
        hipLaunchKernelGGL(peaks_scan_kernel, dim3(1, std::min(batch, 4096)), dim3(kBlock), 0, st, p);
      else
        hipLaunchKernelGGL(peaks_scan_kernel, dim3(chunks, batch), dim3(kBlock), 0, st, p);
      hipLaunchKernelGGL(peaks_first_finish_kernel, dim3((batch + kBlock - 1) / kBlock),
                         dim3(kBlock), 0, st, p);
    } else {
      hipLaunchKernelGGL(peaks_first_kernel, dim3(batch), dim3(kBlock), 0, st, p);
    }
    SFM_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(peaks_second_kernel, dim3((batch + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, p);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// batch driver
// ---------------------------------------------------------------------------
struct XcorrWs {
  float *a0, *b0, *va, *vb, *surface, *den, *ov;
  unsigned int* maxima;
  void* mfma;
  void* fft;
  double* gather_part;  // large patches: per-chunk (sum, count) partials
  PeakWs peaks;
  int srows, spitch;  // layout of `surface`: [B, srows, spitch]
  size_t bytes;
};

bool is_masked(const SfmXcorrDesc* d) { return d->pre_mask || d->post_mask; }

bool use_mfma(const SfmXcorrDesc* d) {
  if (d->method == SFM_XCORR_DIRECT || d->method == SFM_XCORR_FFT) return false;
  return sfm::mfma_i8_eligible(d);
}

// FFT form: on request, or automatically for the patches the matrix-core
// kernel does not take once they are large enough (3-D, float, wide).
bool use_fft(const SfmXcorrDesc* d) {
  if (d->method == SFM_XCORR_FFT) return true;
  if (d->method != SFM_XCORR_AUTO) return false;
  return !sfm::mfma_i8_eligible(d) && sfm::fft_preferred(d);
}

XcorrWs carve_xcorr(const SfmXcorrDesc* d, const Geo& g, bool with_surface,
                    bool with_peaks) {
  sfm::Carver c(d->workspace);
  XcorrWs w;
  std::memset(&w, 0, sizeof(w));
  const size_t B = d->batch;
  const bool masked = is_masked(d);
  if (use_mfma(d)) {
    const size_t n = sfm::mfma_i8_workspace_bytes(d);
    w.mfma = c.take<char>(n);
    if (masked) {  // two maxima per reference batch of the call
      const size_t rows = d->group > 0 && d->group < d->batch ? d->group : d->batch;
      w.maxima = c.take<unsigned int>(2 * ((B + rows - 1) / rows));
    }
  } else {
    w.a0 = c.take<float>(B * g.Pn);
    w.b0 = c.take<float>(B * g.Qn);
    if (masked) {
      w.va = c.take<float>(B * g.Pn);
      w.vb = c.take<float>(B * g.Qn);
      w.den = c.take<float>(B * g.Sn);
      w.ov = c.take<float>(B * g.Sn);
      w.maxima = c.take<unsigned int>(2);
    }
    if (use_fft(d)) w.fft = c.take<char>(sfm::fft_workspace_bytes(d));
    if (std::max(g.Pn, g.Qn) >= (1LL << 16))
      w.gather_part = c.take<double>(B * 2 * kGatherChunks * 2);
  }
  w.srows = g.S[0] * g.S[1];
  w.spitch = g.S[2];
  if (use_mfma(d)) {
    // The MFMA kernel stores whole 16 x 16 tiles: padded work surface.
    sfm::mfma_i8_padded_dims(d, &w.srows, &w.spitch);
    with_surface = true;
  }
  if (with_surface) w.surface = c.take<float>(B * (size_t)w.srows * w.spitch);
  if (with_peaks)
    w.peaks = carve_peaks(c, d->batch, g.Sn, use_mfma(d) && !masked, d->group,
                          use_mfma(d) && masked && (g.S[1] + sfm::kMaskedBlkRows - 1) / sfm::kMaskedBlkRows <= kBlkMaxN);
  w.bytes = c.total();
  return w;
}

int check_desc(const SfmXcorrDesc* d) {
  if (!d) return sfm::fail(SFM_ERR_INVALID, "desc is NULL");
  if (d->batch < 1) return sfm::fail(SFM_ERR_INVALID, "batch must be >= 1");
  if (d->group < 0) return sfm::fail(SFM_ERR_INVALID, "group must be >= 0");
  if (!d->pre_image || !d->post_image || !d->pre_starts || !d->post_starts)
    return sfm::fail(SFM_ERR_INVALID, "image / starts pointers must be set");
  if (d->dtype != SFM_DTYPE_U8 && d->dtype != SFM_DTYPE_F32)
    return sfm::fail(SFM_ERR_INVALID, "unsupported dtype tag %d", d->dtype);
  if (d->method < SFM_XCORR_AUTO || d->method > SFM_XCORR_FFT)
    return sfm::fail(SFM_ERR_INVALID, "unknown method %d", d->method);
  if (d->method == SFM_XCORR_MFMA_I8 && !sfm::mfma_i8_eligible(d))
    return sfm::fail(SFM_ERR_INVALID,
                     "MFMA_I8 needs uint8 2-D images, post patches up to 160 wide and (un-masked) pre patches up to 320 wide");
  return SFM_OK;
}

int compute_surface(const SfmXcorrDesc* d, const Geo& g, const XcorrWs& w,
                    float* surface, const sfm::FusedPeaks* fused = nullptr,
                    unsigned int* smax = nullptr) {
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const bool masked = is_masked(d);
  if (use_mfma(d) && masked) {
    // exact integer correlations on the matrix cores + Padfield assembly
    return sfm::mfma_i8_masked(d, w.mfma, surface, w.maxima, smax, w.peaks.blkmax);
  }
  if (use_mfma(d)) return sfm::mfma_i8_surface(d, w.mfma, surface, fused);
  GatherArgs ga[2];
  for (int k = 0; k < 2; ++k) {
    GatherArgs& a = ga[k];
    a.img = k == 0 ? d->pre_image : d->post_image;
    a.mask = k == 0 ? d->pre_mask : d->post_mask;
    for (int i = 0; i < 3; ++i) {
      a.ishape[i] = k == 0 ? d->pre_shape[i] : d->post_shape[i];
      a.mshape[i] = k == 0 ? d->pre_mask_shape[i] : d->post_mask_shape[i];
      a.psz[i] = k == 0 ? g.P[i] : g.Q[i];
      if (a.mask && a.mshape[i] < a.psz[i])
        return sfm::fail(SFM_ERR_INVALID, "mask smaller than patch on axis %d", i);
    }
    a.starts = k == 0 ? d->pre_starts : d->post_starts;
    a.nd = d->ndim;
    a.use_mean = d->use_mean;
    a.mean = d->mean;
    a.out = k == 0 ? w.a0 : w.b0;
    a.valid = masked ? (k == 0 ? w.va : w.vb) : nullptr;
    a.pn = k == 0 ? g.Pn : g.Qn;
  }
  if (w.gather_part) {
    const dim3 grid(kGatherChunks, d->batch, 2);
    if (d->dtype == SFM_DTYPE_U8) {
      if (!d->use_mean)
        hipLaunchKernelGGL((gather_big_kernel<unsigned char, 0>), grid, dim3(kBlock), 0,
                           st, ga[0], ga[1], w.gather_part);
      hipLaunchKernelGGL((gather_big_kernel<unsigned char, 1>), grid, dim3(kBlock), 0, st,
                         ga[0], ga[1], w.gather_part);
    } else {
      if (!d->use_mean)
        hipLaunchKernelGGL((gather_big_kernel<float, 0>), grid, dim3(kBlock), 0, st,
                           ga[0], ga[1], w.gather_part);
      hipLaunchKernelGGL((gather_big_kernel<float, 1>), grid, dim3(kBlock), 0, st, ga[0],
                         ga[1], w.gather_part);
    }
  } else if (d->dtype == SFM_DTYPE_U8) {
    hipLaunchKernelGGL(gather_kernel<unsigned char>, dim3(d->batch, 2),
                       dim3(kBlock), 0, st, ga[0], ga[1]);
  } else {
    hipLaunchKernelGGL(gather_kernel<float>, dim3(d->batch, 2), dim3(kBlock), 0,
                       st, ga[0], ga[1]);
  }
  SFM_LAUNCH_CHECK();

  CorrArgs c;
  c.a = w.a0;
  c.b = w.b0;
  c.va = w.va;
  c.vb = w.vb;
  c.g = g;
  c.out = surface;
  c.den = w.den;
  c.ov = w.ov;
  c.maxima = w.maxima;
  const long long gz = (long long)d->batch * g.S[0];
  if (gz > 65535LL * 32768)
    return sfm::fail(SFM_ERR_INVALID, "batch * S_z too large");
  dim3 grid((g.S[2] + 63) / 64, (g.S[1] + 3) / 4, (unsigned)gz);
  if (use_fft(d)) {
    if (masked) SFM_HIP_CHECK(hipMemsetAsync(w.maxima, 0, 2 * sizeof(unsigned int), st));
    sfm::prof_begin(sfm::kProfXcorr, st);
    const int rc = sfm::fft_correlate(d, w.a0, w.b0, w.va, w.vb, surface, w.den, w.ov,
                                      w.maxima, w.fft, masked ? nullptr : smax);
    sfm::prof_end(sfm::kProfXcorr, st);
    if (rc) return rc;
    if (masked) {
      const long long n = (long long)d->batch * g.Sn;
      const int fg = (int)((n + kBlock - 1) / kBlock > 4096 ? 4096
                                                             : (n + kBlock - 1) / kBlock);
      hipLaunchKernelGGL(masked_finalize_kernel, dim3(fg), dim3(kBlock), 0, st,
                         surface, w.den, w.ov, w.maxima, n);
      SFM_LAUNCH_CHECK();
    }
    return SFM_OK;
  }
  if (masked) {
    SFM_HIP_CHECK(hipMemsetAsync(w.maxima, 0, 2 * sizeof(unsigned int), st));
    hipLaunchKernelGGL(corr_direct_kernel<true>, grid, dim3(kBlock), 0, st, c);
    SFM_LAUNCH_CHECK();
    const long long n = (long long)d->batch * g.Sn;
    const int fg = (int)((n + kBlock - 1) / kBlock > 4096 ? 4096
                                                           : (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(masked_finalize_kernel, dim3(fg), dim3(kBlock), 0, st,
                       surface, w.den, w.ov, w.maxima, n);
    SFM_LAUNCH_CHECK();
  } else {
    sfm::prof_begin(sfm::kProfXcorr, st);
    hipLaunchKernelGGL(corr_direct_kernel<false>, grid, dim3(kBlock), 0, st, c);
    sfm::prof_end(sfm::kProfXcorr, st);
    SFM_LAUNCH_CHECK();
  }
  return SFM_OK;
}

// Rows that share the reference's batch-coupled behaviours.
int group_rows(const SfmXcorrDesc* d) {
  return d->group > 0 && d->group < d->batch ? d->group : d->batch;
}

// The fused MFMA path keeps the coupled state per group and takes any number
// of groups in one launch; the masked matrix-core path keeps its maxima per group
// and takes kMaskedGroups groups per round of launches (its product surfaces are
// 4 GB per 1024 patches of 160^2; SFM_MASKED_GROUPS); every other path runs group
// by group.
constexpr int kMaskedGroups = 8;

int masked_groups() {
  const char* e = sfm::option("SFM_MASKED_GROUPS");
  const int n = e ? std::atoi(e) : kMaskedGroups;
  return n < 1 ? 1 : n;
}

bool one_launch(const SfmXcorrDesc* d) {
  return group_rows(d) == d->batch || (use_mfma(d) && !is_masked(d));
}

// Rows of one round of launches of sfm_xcorr_peaks (a multiple of the group).
long long call_rows(const SfmXcorrDesc* d) {
  if (one_launch(d)) return d->batch;
  const long long rows = group_rows(d);
  if (use_mfma(d) && is_masked(d))
    return std::min<long long>(d->batch, rows * masked_groups());
  return rows;
}

SfmXcorrDesc sub_desc(const SfmXcorrDesc* d, int off, long long rows = 0) {
  SfmXcorrDesc sub = *d;
  if (rows <= 0) rows = group_rows(d);
  sub.batch = static_cast<int>(d->batch - off < rows ? d->batch - off : rows);
  sub.group = rows > group_rows(d) ? group_rows(d) : 0;
  sub.pre_starts = d->pre_starts + (long long)off * d->ndim;
  sub.post_starts = d->post_starts + (long long)off * d->ndim;
  return sub;
}

int surface_one(const SfmXcorrDesc* d, const Geo& g, float* surface) {
  XcorrWs w = carve_xcorr(d, g, false, false);
  if (!d->workspace || d->workspace_bytes < w.bytes)
    return sfm::fail(SFM_ERR_WORKSPACE, "xcorr workspace needs %zu bytes, got %zu",
                     w.bytes, d->workspace_bytes);
  if (use_mfma(d)) {
    if (int rc = compute_surface(d, g, w, w.surface)) return rc;
    hipStream_t st = static_cast<hipStream_t>(d->stream);
    hipLaunchKernelGGL(compact_surface_kernel, dim3(64, d->batch), dim3(kBlock), 0,
                       st, w.surface, surface, g.S[1], g.S[2], w.srows, w.spitch);
    SFM_LAUNCH_CHECK();
    return SFM_OK;
  }
  return compute_surface(d, g, w, surface);
}

// SFM_MASKED_DEADROWS=0: the peak sweeps of the masked path read every row.
bool live_rows_enabled() {
  const char* e = sfm::option("SFM_MASKED_DEADROWS");
  return !(e && e[0] == '0');
}

// SFM_MASKED_BLKMAX=0: the peak sweep of the masked path reads every live row.
bool blkmax_enabled() {
  const char* e = sfm::option("SFM_MASKED_BLKMAX");
  return !(e && e[0] == '0');
}

int peaks_one(const SfmXcorrDesc* d, const Geo& g, float* peaks) {
  XcorrWs w = carve_xcorr(d, g, true, true);
  if (!d->workspace || d->workspace_bytes < w.bytes)
    return sfm::fail(SFM_ERR_WORKSPACE, "xcorr workspace needs %zu bytes, got %zu",
                     w.bytes, d->workspace_bytes);
  // With the MFMA kernel the first peak pass runs inside it, per finished
  // surface; its per-batch state has to be cleared before the launch.
  const bool fuse = use_mfma(d) && !is_masked(d);
  sfm::FusedPeaks fp;
  // masked matrix-core path and un-masked FFT form: the surface maxima come with
  // the surfaces
  const bool smax_pre = (use_mfma(d) && is_masked(d)) || (!use_mfma(d) && use_fft(d) && !is_masked(d));
  if (fuse || smax_pre)
    SFM_HIP_CHECK(hipMemsetAsync(static_cast<char*>(d->workspace) + w.peaks.zero_from,
                                 0, w.peaks.zero_bytes,
                                 static_cast<hipStream_t>(d->stream)));
  if (fuse) {
    fp.cand_cap = kCandCap;
    fp.idx1 = w.peaks.idx1;
    fp.v1 = w.peaks.v1;
    fp.zero_is_peak = w.peaks.zero_is_peak;
    fp.cand_count = w.peaks.cand_count;
    fp.cand_val = w.peaks.cand_val;
    fp.cand_idx = w.peaks.cand_idx;
    fp.bitmap = w.peaks.bitmap;
    fp.group = w.peaks.group;
    fp.bitmap_words = w.peaks.bitmap_words;
    fp.hot_cap = kHotCap;
    fp.hot_count = w.peaks.hot_count;
    fp.skipmask = w.peaks.skipmask;
    fp.hot_val = w.peaks.hot_val;
    fp.hot_idx = w.peaks.hot_idx;
  }
  if (int rc = compute_surface(d, g, w, w.surface, fuse ? &fp : nullptr,
                               smax_pre ? w.peaks.smax : nullptr))
    return rc;
  float center[3];
  for (int i = 0; i < 3; ++i)
    center[i] = static_cast<float>((g.P[i] + g.Q[i]) / 2 - 1);
  // masked matrix-core surfaces: rows below the overlap threshold are zeros (the
  // assembly's second maximum, `maxima[1]`, is the batch maximum of the overlap)
  const bool masked_mfma = use_mfma(d) && is_masked(d);
  const int live_geo[3] = {g.P[1], g.Q[1], g.Q[2]};
  return run_peaks(w.peaks, static_cast<char*>(d->workspace), w.surface,
                   w.spitch, (long long)w.srows * w.spitch, d->ndim, g.S, g.Sn,
                   d->batch, center, d->min_distance,
                   d->threshold_rel, d->peak_radius, peaks,
                   static_cast<hipStream_t>(d->stream), fuse, smax_pre,
                   masked_mfma && live_rows_enabled() ? w.maxima + 1 : nullptr, live_geo,
                   masked_mfma && w.peaks.blkmax && blkmax_enabled());
}

}  // namespace

extern "C" {

size_t sfm_xcorr_workspace_bytes(const SfmXcorrDesc* d) {
  if (check_desc(d) != SFM_OK) return 0;
  Geo g;
  if (make_geo(d, &g) != SFM_OK) return 0;
  if (use_fft(d) && sfm::fft_check(d) != SFM_OK) return 0;   // message in sfm_last_error
  SfmXcorrDesc tmp = *d;
  tmp.workspace = nullptr;
  if (!one_launch(d)) {  // run group by group: scratch for one round of launches
    const long long rows = call_rows(d);
    tmp.batch = static_cast<int>(rows);
    tmp.group = rows > group_rows(d) ? group_rows(d) : 0;
  }
  return carve_xcorr(&tmp, g, true, true).bytes;
}

int sfm_xcorr_surface(const SfmXcorrDesc* d, float* surface) {
  if (int rc = check_desc(d)) return rc;
  if (!surface) return sfm::fail(SFM_ERR_INVALID, "surface is NULL");
  Geo g;
  if (int rc = make_geo(d, &g)) return rc;
  if (!is_masked(d) || group_rows(d) == d->batch) return surface_one(d, g, surface);
  // masked surfaces are normalised with maxima over their reference batch
  for (int off = 0; off < d->batch; off += group_rows(d)) {
    const SfmXcorrDesc sub = sub_desc(d, off);
    if (int rc = surface_one(&sub, g, surface + (long long)off * g.Sn)) return rc;
  }
  return SFM_OK;
}

int sfm_xcorr_peaks(const SfmXcorrDesc* d, float* peaks) {
  if (int rc = check_desc(d)) return rc;
  if (!peaks) return sfm::fail(SFM_ERR_INVALID, "peaks is NULL");
  Geo g;
  if (int rc = make_geo(d, &g)) return rc;
  if (one_launch(d)) return peaks_one(d, g, peaks);
  const long long rows = call_rows(d);
  for (long long off = 0; off < d->batch; off += rows) {
    const SfmXcorrDesc sub = sub_desc(d, static_cast<int>(off), rows);
    if (int rc = peaks_one(&sub, g, peaks + off * (d->ndim + 2))) return rc;
  }
  return SFM_OK;
}

int sfm_mask_patch_counts(const SfmMaskCountDesc* d, int32_t* counts) {
  if (!d || !d->mask || !counts)
    return sfm::fail(SFM_ERR_INVALID, "mask counts: NULL argument");
  if (d->ndim != 2 && d->ndim != 3)
    return sfm::fail(SFM_ERR_INVALID, "mask counts: ndim must be 2 or 3");
  MaskCountArgs a;
  a.mask = d->mask;
  a.out = counts;
  long long n = 1;
  for (int i = 0; i < 3; ++i) {
    a.S[i] = d->shape[i];
    a.P[i] = d->patch[i];
    a.T[i] = d->step[i];
    if (a.P[i] < 1 || a.T[i] < 1 || a.S[i] < a.P[i])
      return sfm::fail(SFM_ERR_INVALID, "mask counts: bad geometry on axis %d", i);
    a.O[i] = (a.S[i] - a.P[i]) / a.T[i] + 1;
    n *= a.O[i];
  }
  if (n > 0x7fffffffLL) return sfm::fail(SFM_ERR_INVALID, "mask counts: grid too large");
  if (a.P[2] <= kMcCols / 2 && a.O[1] <= 65535 && a.O[0] <= 65535) {
    const int ot = (kMcCols - a.P[2]) / a.T[2] + 1;  // x outputs per tile
    const int tiles = (a.O[2] + ot - 1) / ot;
    const long long wgs = (long long)tiles * a.O[1] * a.O[0];
    const long long rows = (long long)a.P[0] * a.P[1];
    int splits = 1;
    if (wgs < 512)  // not enough workgroups to fill the chip: divide the window rows
      splits = static_cast<int>(std::max<long long>(
          1, std::min<long long>((1024 + wgs - 1) / wgs, (rows + 31) / 32)));
    hipStream_t st = static_cast<hipStream_t>(d->stream);
    if (splits > 1) SFM_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n * sizeof(int32_t), st));
    const dim3 grid(tiles * splits, a.O[1], a.O[0]);
    hipLaunchKernelGGL(mask_count_rows_kernel, grid, dim3(kBlock), 0, st, a, ot, splits);
  } else if (n <= 65535) {
    hipStream_t st = static_cast<hipStream_t>(d->stream);
    const int nseg = (a.P[2] + kMcCols - 1) / kMcCols;
    const long long rows = (long long)a.P[0] * a.P[1];
    const long long wgs = n * nseg;
    const int splits = static_cast<int>(std::max<long long>(
        1, std::min<long long>((1024 + wgs - 1) / wgs, (rows + 31) / 32)));
    SFM_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n * sizeof(int32_t), st));
    hipLaunchKernelGGL(mask_count_wide_kernel, dim3(nseg * splits, static_cast<unsigned>(n)),
                       dim3(kBlock), 0, st, a, nseg, splits);
  } else {
    hipLaunchKernelGGL(mask_count_kernel, dim3(static_cast<unsigned>(n)), dim3(kBlock),
                       0, static_cast<hipStream_t>(d->stream), a);
  }
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

size_t sfm_peaks_workspace_bytes(const SfmPeaksDesc* d) {
  if (!d || d->batch < 1) return 0;
  long long sn = 1;
  for (int i = 0; i < 3; ++i) sn *= d->shape[i];
  sfm::Carver c(nullptr);
  return carve_peaks(c, d->batch, sn).bytes;
}

int sfm_peaks(const SfmPeaksDesc* d, float* peaks) {
  if (!d || !peaks || !d->surface)
    return sfm::fail(SFM_ERR_INVALID, "desc / surface / peaks is NULL");
  if (d->ndim != 2 && d->ndim != 3)
    return sfm::fail(SFM_ERR_INVALID, "ndim must be 2 or 3");
  if (d->batch < 1) return sfm::fail(SFM_ERR_INVALID, "batch must be >= 1");
  long long sn = 1;
  for (int i = 0; i < 3; ++i) {
    if (d->shape[i] < 1) return sfm::fail(SFM_ERR_INVALID, "bad surface shape");
    sn *= d->shape[i];
  }
  if (sn > 0x7fffffffLL)
    return sfm::fail(SFM_ERR_INVALID, "surface too large for int32 indices");
  sfm::Carver c(d->workspace);
  PeakWs w = carve_peaks(c, d->batch, sn);
  if (!d->workspace || d->workspace_bytes < w.bytes)
    return sfm::fail(SFM_ERR_WORKSPACE, "peaks workspace needs %zu bytes, got %zu",
                     w.bytes, d->workspace_bytes);
  return run_peaks(w, static_cast<char*>(d->workspace), d->surface, d->shape[2],
                   sn, d->ndim, d->shape, sn, d->batch, d->center_offset,
                   d->min_distance,
                   d->threshold_rel, d->peak_radius, peaks,
                   static_cast<hipStream_t>(d->stream));
}

}  // extern "C"
