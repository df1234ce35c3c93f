// FFT form of the patch cross-correlation for the regimes where summing every
// shift directly is the wrong algorithm: 3-D patches (direct work grows like
// P^6: 5.2e11 flop per 80^3 patch) and large / float 2-D patches that the
// int8 MFMA kernel does not take.  This is the reference's own formulation
// (flow_field.py:66-89: zero-padded rFFT, product with the conjugate spectrum,
// inverse rFFT; :91-131 for the masked terms) on the hand-written transforms of
// sfm_fft_own.hip (zero-skipping LDS Stockham passes; padding, spectrum product
// and crop fused into them; no library FFT); the six-term Padfield assembly of
// the masked form and the axis swap of strips whose LONG axis is x live here.
// The production uint8 2-D path never comes here.
//
//   corr[k] = sum_i a[i + k - (Q - 1)] b[i]
//           = irfft(rfft(a_pad) conj(rfft(b_pad)))[(k - (Q - 1)) mod F]
#include "sfm_common.h"

#include <algorithm>
#include <cstdint>

#include <map>
#include <mutex>
#include <tuple>

namespace sfm {

// sfm_fft_own.hip: the hand-written transforms
bool own_fft_supported(int rank, const int* F);
bool own_fft_fits_tile(int n);
// in-plane patches beyond one tile along both axes (full-complex path)
bool own_fft_big_supported(int rank, const int* F);
int own_fft_big_correlate(const int* P, const int* Q, const int* S, const int* F, int nb,
                          const float* a0, const float* b0, float2* sa, float2* sb, float2* tmp,
                          float* surface, unsigned int* smax, hipStream_t st);
int own_fft_big_forward(const int* R, const int* F, int nb, const float* src, int square,
                        float2* spec, float2* tmp, hipStream_t st);
int own_fft_big_inverse_product(const int* F, int nb, const float2* lhs, const float2* rhs,
                                float2* work, float2* tmp, float* real_out, hipStream_t st);
int own_fft_correlate(const int* P, const int* Q, const int* S, const int* F, int nb,
                      const float* a0, const float* b0, float2* sa, float2* sb, float* surface,
                      unsigned int* smax, hipStream_t st);
int own_fft_forward(const int* R, const int* F, int nb, const float* src, int square,
                    float2* spec, hipStream_t st);
int own_fft_inverse_product(const int* F, int nb, const float2* lhs, const float2* rhs,
                            float2* work, float* real_out, hipStream_t st);

namespace {

constexpr int kBlock = 256;
constexpr float kEps = 1.1920928955078125e-07f;  // float32 eps
constexpr size_t kBudget = 1536ull << 20;         // FFT scratch per call

struct FftGeo {
  int rank;
  int P[3], Q[3], S[3], F[3];
  long long Pn, Qn, Sn, Fn, Cn;  // Cn = complex elements of one half spectrum
};

// Smallest even 2^a 3^b 5^c >= n: the composites scipy.fftpack.next_fast_len
// (flow_field.py:67) picks from, and the radices of the hand-written transforms.
int fast_len(int n) {
  for (int f = n + (n & 1);; f += 2) {
    int m = f;
    for (int p : {2, 3, 5})
      while (m % p == 0) m /= p;
    if (m == 1) return f;
  }
}

// [nb, R, Cc] -> [nb, Cc, R] through a 32 x 33 LDS tile (strips whose long
// axis is x are correlated with their axes swapped: corr(a^T, b^T) = corr(a, b)^T,
// and the long axis of the transforms must be y, where the four-step split lives).
__global__ void __launch_bounds__(kBlock)
transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const float* s = src + b * (long long)R * Cc;
  float* d = dst + b * (long long)R * Cc;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = min(r0 + j, R - 1), c = min(c0 + tx, Cc - 1);
    tile[j][tx] = s[(long long)r * Cc + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < Cc && r < R) d[(long long)c * R + r] = tile[tx][j];
  }
}

void launch_transpose(const float* src, float* dst, int nb, int R, int Cc, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, nb), dim3(kBlock), 0,
                     st, src, dst, R, Cc);
}

struct CropArgs {
  const float* r[6];  // circular correlations [nb, Fn]: xc, sa, sb, nov, qa, qb
  float* out;         // [nb, Sn]
  float* den;
  float* ov;
  unsigned int* maxima;
  unsigned int* smax;  // [nb] or NULL: un-masked surfaces' maxima (ordered bits)
  int S[3], F[3], Q[3];
  long long Sn, Fn;
  float scale;        // 1 / Fn (the inverse transforms are unnormalised)
  long long total;    // nb * Sn
};

// A wave per surface row (b, kz, ky); see pad_kernel.
template <bool MASKED>
__global__ void __launch_bounds__(kBlock) crop_kernel(CropArgs c) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rows_per_patch = c.S[0] * c.S[1];
  float mden = 0.f, mov = 0.f;
  // grid.y = surface: the rows of a wave belong to one surface, so its maximum
  // stays in a register until the end; waves of a workgroup take adjacent rows
  const long long b = blockIdx.y;
  float smax_run = -INFINITY;
  for (long long row = b * rows_per_patch + blockIdx.x * (kBlock / 64) + wave;
       row < (b + 1) * rows_per_patch; row += (long long)gridDim.x * (kBlock / 64)) {
    const int r = static_cast<int>(row - b * rows_per_patch);
    const int kz = r / c.S[1], ky = r - kz * c.S[1];
    int dz = kz - (c.Q[0] - 1), dy = ky - (c.Q[1] - 1);
    if (dz < 0) dz += c.F[0];
    if (dy < 0) dy += c.F[1];
    const long long s0 = b * c.Fn + ((long long)dz * c.F[1] + dy) * c.F[2];
    const long long o0 = row * c.S[2];
    float rmax = -INFINITY;
    if (!MASKED) {
      // four 64-wide column groups per round, loads first (clamped columns)
      for (int k0 = 0; k0 < c.S[2]; k0 += 256) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int dx = min(k0 + lane + 64 * j, c.S[2] - 1) - (c.Q[2] - 1);
          if (dx < 0) dx += c.F[2];
          v[j] = c.r[0][s0 + dx] * c.scale;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kx = k0 + lane + 64 * j;
          if (kx < c.S[2]) {
            c.out[o0 + kx] = v[j];
            rmax = fmaxf(rmax, v[j]);
          }
        }
      }
      smax_run = fmaxf(smax_run, rmax);
      continue;
    }
    for (int kx = lane; kx < c.S[2]; kx += 64) {
      int dx = kx - (c.Q[2] - 1);
      if (dx < 0) dx += c.F[2];
      const long long s = s0 + dx;
      const float xc = c.r[0][s] * c.scale;
      const float sa = c.r[1][s] * c.scale, sb = c.r[2][s] * c.scale;
      const float nov = c.r[3][s] * c.scale;
      const float qa = c.r[4][s] * c.scale, qb = c.r[5][s] * c.scale;
      // Padfield assembly (flow_field.py:113-131), as in corr_direct_kernel
      const float ovv = fmaxf(rintf(nov), kEps);
      const float inv = 1.0f / ovv;
      const float num = xc - sa * sb * inv;
      const float pd = fmaxf(qa - sa * sa * inv, 0.f);
      const float cd = fmaxf(qb - sb * sb * inv, 0.f);
      const float den = sqrtf(pd * cd);
      c.out[o0 + kx] = num;
      c.den[o0 + kx] = den;
      c.ov[o0 + kx] = ovv;
      mden = fmaxf(mden, fabsf(den));
      mov = fmaxf(mov, ovv);
    }
    smax_run = fmaxf(smax_run, rmax);
  }
  if (!MASKED && c.smax) {
    // the surface maximum for the peak search (monotonic uint image of the float)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) smax_run = fmaxf(smax_run, __shfl_xor(smax_run, d, 64));
    if (lane == 0 && smax_run > -INFINITY) {
      const unsigned u = __float_as_uint(smax_run);
      const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      // (same-address atomics serialise: only the waves that raise the maximum)
      if (o > __atomic_load_n(&c.smax[b], __ATOMIC_RELAXED)) atomicMax(&c.smax[b], o);
    }
  }
  if (MASKED) {
    // one pair of atomics per wave, and only when it can raise a maximum
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mden = fmaxf(mden, __shfl_xor(mden, d, 64));
      mov = fmaxf(mov, __shfl_xor(mov, d, 64));
    }
    if (lane == 0) {
      if (__float_as_uint(mden) > __atomic_load_n(&c.maxima[0], __ATOMIC_RELAXED))
        atomicMax(&c.maxima[0], __float_as_uint(mden));
      if (__float_as_uint(mov) > __atomic_load_n(&c.maxima[1], __ATOMIC_RELAXED))
        atomicMax(&c.maxima[1], __float_as_uint(mov));
    }
  }
}

FftGeo make_fft_geo(const SfmXcorrDesc* d) {
  FftGeo g;
  g.rank = d->ndim;
  g.Pn = g.Qn = g.Sn = g.Fn = 1;
  for (int i = 0; i < 3; ++i) {
    g.P[i] = d->patch[i];
    g.Q[i] = d->post_patch[i];
    g.S[i] = g.P[i] + g.Q[i] - 1;
    g.F[i] = (d->ndim == 2 && i == 0) ? 1 : fast_len(g.S[i]);
    g.Pn *= g.P[i];
    g.Qn *= g.Q[i];
    g.Sn *= g.S[i];
    g.Fn *= g.F[i];
  }
  g.Cn = (long long)g.F[0] * g.F[1] * (g.F[2] / 2 + 1);
  return g;
}

// In-plane patches whose padded x extent does not fit one LDS tile while the y
// extent does: correlate the transposed patches.
bool needs_swap(const FftGeo& g) {
  return g.rank == 2 && !own_fft_fits_tile(g.F[2]) && own_fft_fits_tile(g.F[1]);
}

// ... and along both: the full-complex path.
bool needs_big(const FftGeo& g) {
  return g.rank == 2 && !own_fft_fits_tile(g.F[2]) && !own_fft_fits_tile(g.F[1]);
}

FftGeo swapped(FftGeo g) {
  std::swap(g.P[1], g.P[2]);
  std::swap(g.Q[1], g.Q[2]);
  std::swap(g.S[1], g.S[2]);
  std::swap(g.F[1], g.F[2]);
  g.Cn = (long long)g.F[0] * g.F[1] * (g.F[2] / 2 + 1);
  return g;
}

// Floats of scratch per patch of a sub-batch.
size_t floats_per_patch(const FftGeo& g, bool masked) {
  const size_t real = static_cast<size_t>(g.Fn), spec = 2 * static_cast<size_t>(g.Cn);
  // unmasked: 2 spectra (the product overwrites one, the inverse crops straight into
  // the surface); masked: 6 spectra + product + 6 real circular arrays
  size_t n = masked ? 7 * spec + 6 * real : 2 * spec;
  if (needs_big(g))   // full spectra [F1][F2] complex + one transposition buffer
    n = masked ? (7 + 1) * 2 * real + 6 * real : 3 * 2 * real;
  if (needs_swap(g))  // transposed inputs and outputs
    n += (masked ? 2 : 1) * static_cast<size_t>(g.Pn + g.Qn) + (masked ? 3 : 1) * static_cast<size_t>(g.Sn);
  return n;
}

int sub_batch(const FftGeo& g, bool masked, int batch) {
  const size_t per = floats_per_patch(g, masked) * sizeof(float);
  size_t nb = kBudget / per;
  if (nb < 1) nb = 1;
  if (nb > static_cast<size_t>(batch)) nb = batch;
  return static_cast<int>(nb);
}

}  // namespace

// The FFT path pays off once a patch pair has more than ~2^20 pixel pairs
// (32^2 x 32^2); below that the direct kernel is exact and as fast.
bool fft_preferred(const SfmXcorrDesc* d) {
  long long pn = 1, qn = 1;
  for (int i = 0; i < 3; ++i) {
    pn *= d->patch[i];
    qn *= d->post_patch[i];
  }
  return pn * qn >= (1LL << 20);
}

size_t fft_workspace_bytes(const SfmXcorrDesc* d) {
  const FftGeo g = make_fft_geo(d);
  const bool masked = d->pre_mask || d->post_mask;
  const int nb = sub_batch(g, masked, d->batch);
  return floats_per_patch(g, masked) * sizeof(float) * nb + 4096;
}

// a0 / b0: mean-subtracted, mask-zeroed patches [B, Pn] / [B, Qn]; va / vb:
// validity planes (masked only).  Outputs as corr_direct_kernel: `surface`
// (raw correlation, or the Padfield numerator), den, ov, maxima.
// Can the hand-written transforms take this descriptor's padded patch?  Called
// when the workspace is sized, so that an unsupported shape is an error with a
// message BEFORE anything is allocated (the library has no general FFT behind
// its own: INTEGRATION.md, limits).
int fft_check(const SfmXcorrDesc* d) {
  const FftGeo g0 = make_fft_geo(d);
  const FftGeo g = needs_swap(g0) ? swapped(g0) : g0;
  const bool big = needs_big(g0);
  if (big ? !own_fft_big_supported(g.rank, g.F) : !own_fft_supported(g.rank, g.F))
    return fail(SFM_ERR_INVALID,
                "FFT form: padded patch extent %d x %d x %d is beyond the hand-written "
                "transforms (in-plane axes <= 262144; volumes: every axis <= 1728)",
                g0.F[0], g0.F[1], g0.F[2]);
  return SFM_OK;
}

int fft_correlate(const SfmXcorrDesc* d, const float* a0, const float* b0,
                  const float* va, const float* vb, float* surface, float* den,
                  float* ov, unsigned int* maxima, void* ws, unsigned int* smax) {
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const FftGeo g0 = make_fft_geo(d);
  const bool masked = d->pre_mask || d->post_mask;
  const bool swap = needs_swap(g0);
  const bool big = needs_big(g0);
  const FftGeo g = swap ? swapped(g0) : g0;
  if (big ? !own_fft_big_supported(g.rank, g.F) : !own_fft_supported(g.rank, g.F))
    return fail(SFM_ERR_INVALID,
                "FFT form: padded patch extent %d x %d x %d is beyond the hand-written "
                "transforms (in-plane axes <= 262144; volumes: every axis <= 1728)",
                g0.F[0], g0.F[1], g0.F[2]);
  const int nb_max = sub_batch(g0, masked, d->batch);
  Carver c(ws);
  const int n_spec = masked ? 7 : 2;
  float2* spec[7];
  const size_t spec_n = big ? (size_t)g.Fn : (size_t)g.Cn;   // complex values per spectrum
  for (int i = 0; i < n_spec; ++i) spec[i] = c.take<float2>((size_t)nb_max * spec_n);
  float2* tmp = big ? c.take<float2>((size_t)nb_max * spec_n) : nullptr;
  float* real[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (masked)
    for (int i = 0; i < 6; ++i) real[i] = c.take<float>((size_t)nb_max * g.Fn);
  // axis swap: transposed copies of the inputs, transposed outputs
  float *ta = nullptr, *tb = nullptr, *tva = nullptr, *tvb = nullptr;
  float *ts = nullptr, *tden = nullptr, *tov = nullptr;
  if (swap) {
    ta = c.take<float>((size_t)nb_max * g.Pn);
    tb = c.take<float>((size_t)nb_max * g.Qn);
    ts = c.take<float>((size_t)nb_max * g.Sn);
    if (masked) {
      tva = c.take<float>((size_t)nb_max * g.Pn);
      tvb = c.take<float>((size_t)nb_max * g.Qn);
      tden = c.take<float>((size_t)nb_max * g.Sn);
      tov = c.take<float>((size_t)nb_max * g.Sn);
    }
  }

  for (int lo = 0; lo < d->batch; lo += nb_max) {
    const int nb = d->batch - lo < nb_max ? d->batch - lo : nb_max;
    const float* pa = a0 + (long long)lo * g.Pn;
    const float* pb = b0 + (long long)lo * g.Qn;
    const float* pva = va ? va + (long long)lo * g.Pn : nullptr;
    const float* pvb = vb ? vb + (long long)lo * g.Qn : nullptr;
    float* out_s = surface + (long long)lo * g.Sn;
    float* out_den = den ? den + (long long)lo * g.Sn : nullptr;
    float* out_ov = ov ? ov + (long long)lo * g.Sn : nullptr;
    if (swap) {
      launch_transpose(pa, ta, nb, g0.P[1], g0.P[2], st);
      launch_transpose(pb, tb, nb, g0.Q[1], g0.Q[2], st);
      pa = ta;
      pb = tb;
      if (masked) {
        launch_transpose(pva, tva, nb, g0.P[1], g0.P[2], st);
        launch_transpose(pvb, tvb, nb, g0.Q[1], g0.Q[2], st);
        pva = tva;
        pvb = tvb;
      }
      SFM_LAUNCH_CHECK();
    }
    float* w_s = swap ? ts : out_s;
    float* w_den = swap ? tden : out_den;
    float* w_ov = swap ? tov : out_ov;
    if (!masked) {
      // zero-skipping transforms with pad / product / crop fused in
      if (big) {
        if (int rc = own_fft_big_correlate(g.P, g.Q, g.S, g.F, nb, pa, pb, spec[0], spec[1], tmp,
                                           w_s, smax ? smax + lo : nullptr, st))
          return rc;
      } else if (int rc = own_fft_correlate(g.P, g.Q, g.S, g.F, nb, pa, pb, spec[0], spec[1], w_s,
                                            smax ? smax + lo : nullptr, st)) {
        return rc;
      }
    } else {
      float2 *FA = spec[0], *FVA = spec[1], *FA2 = spec[2];
      float2 *FB = spec[3], *FVB = spec[4], *FB2 = spec[5], *prod = spec[6];
      auto fwd = [&](const int* R, const float* src, int square, float2* out) -> int {
        return big ? own_fft_big_forward(R, g.F, nb, src, square, out, tmp, st)
                   : own_fft_forward(R, g.F, nb, src, square, out, st);
      };
      if (int rc = fwd(g.P, pa, 0, FA)) return rc;
      if (int rc = fwd(g.P, pva, 0, FVA)) return rc;
      if (int rc = fwd(g.P, pa, 1, FA2)) return rc;
      if (int rc = fwd(g.Q, pb, 0, FB)) return rc;
      if (int rc = fwd(g.Q, pvb, 0, FVB)) return rc;
      if (int rc = fwd(g.Q, pb, 1, FB2)) return rc;
      // xc, sa, sb, nov, qa, qb of corr_direct_kernel
      const float2* lhs[6] = {FA, FA, FVA, FVA, FA2, FVA};
      const float2* rhs[6] = {FB, FVB, FB, FVB, FVB, FB2};
      CropArgs cr;
      for (int i = 0; i < 3; ++i) {
        cr.S[i] = g.S[i];
        cr.F[i] = g.F[i];
        cr.Q[i] = g.Q[i];
      }
      cr.Sn = g.Sn;
      cr.Fn = g.Fn;
      cr.scale = 1.0f / static_cast<float>(g.Fn);
      cr.total = (long long)nb * g.Sn;
      cr.out = w_s;
      cr.den = w_den;
      cr.ov = w_ov;
      cr.maxima = maxima;
      cr.smax = nullptr;
      for (int i = 0; i < 6; ++i) {
        if (int rc = big ? own_fft_big_inverse_product(g.F, nb, lhs[i], rhs[i], prod, tmp, real[i], st)
                         : own_fft_inverse_product(g.F, nb, lhs[i], rhs[i], prod, real[i], st))
          return rc;
        cr.r[i] = real[i];
      }
      // (x: workgroups striding over the rows of one surface, y: surface; about 2048
      // workgroups in all: thousands of two-row workgroups are bound by the dispatch rate)
      const int crop_rows = g.S[0] * g.S[1];
      const dim3 crop_grid(std::max(1, std::min((crop_rows + 3) / 4, 2048 / std::max(nb, 1))), nb);
      hipLaunchKernelGGL(crop_kernel<true>, crop_grid, dim3(kBlock), 0, st, cr);
      SFM_LAUNCH_CHECK();
    }
    if (swap) {
      launch_transpose(ts, out_s, nb, g.S[1], g.S[2], st);
      if (masked) {
        launch_transpose(tden, out_den, nb, g.S[1], g.S[2], st);
        launch_transpose(tov, out_ov, nb, g.S[1], g.S[2], st);
      }
      SFM_LAUNCH_CHECK();
    }
  }
  return SFM_OK;
}

}  // namespace sfm
