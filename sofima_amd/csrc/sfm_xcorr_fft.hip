// FFT form of the patch cross-correlation for the regimes where summing every
// shift directly is the wrong algorithm: 3-D patches (direct work grows like
// P^6: 5.2e11 flop per 80^3 patch) and large / float 2-D patches that the
// int8 MFMA kernel does not take.  This is the reference's own formulation
// (flow_field.py:66-89: zero-padded rFFT, product with the conjugate spectrum,
// inverse rFFT; :91-131 for the masked terms) with the transforms done by
// hipFFT -- a plain library FFT, like the reference's jnp.fft -- and everything
// around them (padding, spectrum products, crop + Padfield assembly) in the
// kernels below.  The production uint8 2-D path never comes here.
//
//   corr[k] = sum_i a[i + k - (Q - 1)] b[i]
//           = irfft(rfft(a_pad) conj(rfft(b_pad)))[(k - (Q - 1)) mod F]
#include "sfm_common.h"

#include <algorithm>
#include <cstdint>
#include <hipfft/hipfft.h>

#include <map>
#include <mutex>
#include <tuple>

namespace sfm {

// sfm_fft_own.hip: hand-written transforms for un-masked volumetric patches
bool own_fft_supported(int rank, const int* F);
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

// Smallest 2^a 3^b 5^c 7^d >= n (even, so that R2C halves cleanly).
int fast_len(int n) {
  for (int f = n + (n & 1);; f += 2) {
    int m = f;
    for (int p : {2, 3, 5, 7})
      while (m % p == 0) m /= p;
    if (m == 1) return f;
  }
}

struct PadArgs {
  const float* src;  // [nb, Pn]
  float* dst;        // [nb, Fn]
  int P[3], F[3];
  long long Pn, Fn;
  int square;
  long long total;   // nb * Fn
};

// A wave per padded row (b, z, y): the row decomposition is wave-uniform
// (scalar divisions, once per row) and the lanes sweep x with coalesced stores,
// 16 bytes per lane when the row lengths allow it.
template <bool VEC4>
__global__ void __launch_bounds__(kBlock) pad_kernel(PadArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long n_rows = a.total / a.F[2];
  const int rows_per_patch = a.F[0] * a.F[1];
  for (long long row = blockIdx.x * (long long)(kBlock / 64) + wave; row < n_rows;
       row += (long long)gridDim.x * (kBlock / 64)) {
    const long long b = row / rows_per_patch;
    const int r = static_cast<int>(row - b * rows_per_patch);
    const int z = r / a.F[1], y = r - z * a.F[1];
    float* dst = a.dst + row * a.F[2];
    const bool inside = z < a.P[0] && y < a.P[1];
    const float* src = a.src + b * a.Pn + ((long long)(inside ? z : 0) * a.P[1] + (inside ? y : 0)) * a.P[2];
    if (VEC4) {
      for (int x = 4 * lane; x < a.F[2]; x += 256) {
        // unconditional load, clamped address
        float4 v = *reinterpret_cast<const float4*>(src + min(x, a.P[2] - 4));
        if (a.square) v = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
        if (!(inside && x < a.P[2])) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dst + x) = v;
      }
    } else {
      for (int x = lane; x < a.F[2]; x += 64) {
        float v = src[min(x, a.P[2] - 1)];
        if (a.square) v = v * v;
        dst[x] = (inside && x < a.P[2]) ? v : 0.f;
      }
    }
  }
}

// out = A conj(B)
__global__ void __launch_bounds__(kBlock)
cmul_conj_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                 float2* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n;
       i += (long long)gridDim.x * kBlock) {
    const float2 a = A[i], b = B[i];
    out[i] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
  }
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
  float scale;        // 1 / Fn (hipFFT's inverse is unnormalised)
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

int grid_for(long long n) {
  const long long g = (n + kBlock - 1) / kBlock;
  return static_cast<int>(g > 65535 * 4 ? 65535 * 4 : (g < 1 ? 1 : g));
}

// Plan cache: (device, stream, rank, F, batch, type) -> handle.  A plan owns its
// (auto-allocated) work area, so it must never serve two streams at once: the
// stream is part of the key and calls on the same stream are ordered by the
// stream itself.  The device is part of the key because a handle belongs to
// the device it was created on.  g_exec_mu serialises the host-side
// enqueueing (hipfftSetStream + Exec are not atomic).
std::mutex g_plan_mu;
std::mutex g_exec_mu;
typedef std::tuple<int, const void*, int, int, int, int, int, int> PlanKey;
std::map<PlanKey, hipfftHandle> g_plans;
constexpr size_t kMaxPlans = 96;

int get_plan(const FftGeo& g, int batch, hipfftType type, hipStream_t st,
             hipfftHandle* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(SFM_ERR_HIP, "hipGetDevice failed");
  const PlanKey key = std::make_tuple(dev, static_cast<const void*>(st), g.rank, g.F[0],
                                      g.F[1], g.F[2], batch, static_cast<int>(type));
  std::lock_guard<std::mutex> lk(g_plan_mu);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) {
    *out = it->second;
    return SFM_OK;
  }
  if (g_plans.size() >= kMaxPlans) {
    // Streams come and go (per-thread torch streams): drop everything rather
    // than grow without bound.  Callers hold g_exec_mu, and a destroyed plan's
    // enqueued work keeps its work area alive until the stream drains
    // (hipfftDestroy frees with stream-ordered semantics after a device sync).
    (void)hipDeviceSynchronize();
    for (auto& kv : g_plans) (void)hipfftDestroy(kv.second);
    g_plans.clear();
  }
  int n[3];
  for (int i = 0; i < g.rank; ++i) n[i] = g.F[3 - g.rank + i];
  hipfftHandle h;
  const hipfftResult rc =
      hipfftPlanMany(&h, g.rank, n, nullptr, 1, 0, nullptr, 1, 0, type, batch);
  if (rc != HIPFFT_SUCCESS)
    return fail(SFM_ERR_HIP, "hipfftPlanMany failed with %d", static_cast<int>(rc));
  if (hipfftSetStream(h, st) != HIPFFT_SUCCESS) {
    (void)hipfftDestroy(h);
    return fail(SFM_ERR_HIP, "hipfftSetStream failed");
  }
  g_plans[key] = h;
  *out = h;
  return SFM_OK;
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

// Floats of scratch per patch of a sub-batch.
size_t floats_per_patch(const FftGeo& g, bool masked) {
  const size_t real = static_cast<size_t>(g.Fn), spec = 2 * static_cast<size_t>(g.Cn);
  // unmasked: pad buffer + 2 spectra (the product overwrites one, the inverse
  // lands in the pad buffer); masked: pad buffer + 6 spectra + product + 6 real
  return masked ? real + 7 * spec + 6 * real : real + 2 * spec;
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
  return floats_per_patch(g, masked) * sizeof(float) * nb + 256;
}

// a0 / b0: mean-subtracted, mask-zeroed patches [B, Pn] / [B, Qn]; va / vb:
// validity planes (masked only).  Outputs as corr_direct_kernel: `surface`
// (raw correlation, or the Padfield numerator), den, ov, maxima.
int fft_correlate(const SfmXcorrDesc* d, const float* a0, const float* b0,
                  const float* va, const float* vb, float* surface, float* den,
                  float* ov, unsigned int* maxima, void* ws, unsigned int* smax) {
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  std::lock_guard<std::mutex> exec_lock(g_exec_mu);
  const FftGeo g = make_fft_geo(d);
  const bool masked = d->pre_mask || d->post_mask;
  const int nb_max = sub_batch(g, masked, d->batch);
  Carver c(ws);
  float* pad = c.take<float>((size_t)nb_max * g.Fn);
  const int n_spec = masked ? 7 : 2;
  float2* spec[7];
  for (int i = 0; i < n_spec; ++i) spec[i] = c.take<float2>((size_t)nb_max * g.Cn);
  float* real[6] = {pad, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (masked)
    for (int i = 0; i < 6; ++i) real[i] = c.take<float>((size_t)nb_max * g.Fn);

  const bool own_shape = own_fft_supported(g.rank, g.F);
  const bool own = !masked && own_shape;
  for (int lo = 0; lo < d->batch; lo += nb_max) {
    const int nb = d->batch - lo < nb_max ? d->batch - lo : nb_max;
    if (own) {
      // zero-skipping transforms with pad / product / crop fused in (no hipFFT)
      if (int rc = own_fft_correlate(g.P, g.Q, g.S, g.F, nb, a0 + (long long)lo * g.Pn,
                                     b0 + (long long)lo * g.Qn, spec[0], spec[1],
                                     surface + (long long)lo * g.Sn, smax ? smax + lo : nullptr, st))
        return rc;
      continue;
    }
    hipfftHandle fwd = 0, inv = 0;
    if (!own_shape) {
      if (int rc = get_plan(g, nb, HIPFFT_R2C, st, &fwd)) return rc;
      if (int rc = get_plan(g, nb, HIPFFT_C2R, st, &inv)) return rc;
    }

    auto forward = [&](const float* src, bool pre, int square, float2* out) -> int {
      if (own_shape)  // masked volumes: zero-skipping hand-written transforms
        return own_fft_forward(pre ? g.P : g.Q, g.F, nb,
                               src + (long long)lo * (pre ? g.Pn : g.Qn), square, out, st);
      PadArgs p;
      p.src = src + (long long)lo * (pre ? g.Pn : g.Qn);
      p.dst = pad;
      for (int i = 0; i < 3; ++i) {
        p.P[i] = pre ? g.P[i] : g.Q[i];
        p.F[i] = g.F[i];
      }
      p.Pn = pre ? g.Pn : g.Qn;
      p.Fn = g.Fn;
      p.square = square;
      p.total = (long long)nb * g.Fn;
      // 16-byte lanes need rows that start on 16-byte boundaries on both sides
      const bool vec4 = p.P[2] % 4 == 0 && p.F[2] % 4 == 0 && p.Pn % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(p.dst) & 15) == 0;
      if (vec4)
        hipLaunchKernelGGL(pad_kernel<true>, dim3(std::min(grid_for(p.total / g.F[2] * 64), 8192)),
                           dim3(kBlock), 0, st, p);
      else
        hipLaunchKernelGGL(pad_kernel<false>, dim3(std::min(grid_for(p.total / g.F[2] * 64), 8192)),
                           dim3(kBlock), 0, st, p);
      SFM_LAUNCH_CHECK();
      if (hipfftExecR2C(fwd, pad, reinterpret_cast<hipfftComplex*>(out)) != HIPFFT_SUCCESS)
        return fail(SFM_ERR_HIP, "hipfftExecR2C failed");
      return SFM_OK;
    };
    auto product = [&](const float2* A, const float2* B, float2* prod, float* out) -> int {
      if (own_shape) return own_fft_inverse_product(g.F, nb, A, B, prod, out, st);
      const long long n = (long long)nb * g.Cn;
      hipLaunchKernelGGL(cmul_conj_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, A, B,
                         prod, n);
      SFM_LAUNCH_CHECK();
      if (hipfftExecC2R(inv, reinterpret_cast<hipfftComplex*>(prod), out) != HIPFFT_SUCCESS)
        return fail(SFM_ERR_HIP, "hipfftExecC2R failed");
      return SFM_OK;
    };

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
    cr.out = surface + (long long)lo * g.Sn;
    cr.den = den ? den + (long long)lo * g.Sn : nullptr;
    cr.ov = ov ? ov + (long long)lo * g.Sn : nullptr;
    cr.maxima = maxima;
    // (x: workgroups striding over the rows of one surface, y: surface)
    const int crop_rows = g.S[0] * g.S[1];
    // (about 2048 workgroups in all: thousands of two-row workgroups are bound by
    // the dispatch rate, not by memory)
    const dim3 crop_grid(std::max(1, std::min((crop_rows + 3) / 4, 2048 / std::max(nb, 1))), nb);
    cr.smax = smax ? smax + lo : nullptr;
    if (!masked) {
      if (int rc = forward(a0, true, 0, spec[0])) return rc;
      if (int rc = forward(b0, false, 0, spec[1])) return rc;
      if (int rc = product(spec[0], spec[1], spec[0], pad)) return rc;
      for (int i = 0; i < 6; ++i) cr.r[i] = pad;
      hipLaunchKernelGGL(crop_kernel<false>, crop_grid,
                         dim3(kBlock), 0, st, cr);
      SFM_LAUNCH_CHECK();
    } else {
      float2 *FA = spec[0], *FVA = spec[1], *FA2 = spec[2];
      float2 *FB = spec[3], *FVB = spec[4], *FB2 = spec[5], *prod = spec[6];
      if (int rc = forward(a0, true, 0, FA)) return rc;
      if (int rc = forward(va, true, 0, FVA)) return rc;
      if (int rc = forward(a0, true, 1, FA2)) return rc;
      if (int rc = forward(b0, false, 0, FB)) return rc;
      if (int rc = forward(vb, false, 0, FVB)) return rc;
      if (int rc = forward(b0, false, 1, FB2)) return rc;
      // xc, sa, sb, nov, qa, qb of corr_direct_kernel
      const float2* lhs[6] = {FA, FA, FVA, FVA, FA2, FVA};
      const float2* rhs[6] = {FB, FVB, FB, FVB, FVB, FB2};
      for (int i = 0; i < 6; ++i) {
        if (int rc = product(lhs[i], rhs[i], prod, real[i])) return rc;
        cr.r[i] = real[i];
      }
      hipLaunchKernelGGL(crop_kernel<true>, crop_grid,
                         dim3(kBlock), 0, st, cr);
      SFM_LAUNCH_CHECK();
    }
  }
  return SFM_OK;
}

}  // namespace sfm
