// Hand-written FFT correlation: every transform of the library (no hipFFT).
//
//   corr[k] = irfft(rfft(a_pad) conj(rfft(b_pad)))[(k - (Q - 1)) mod F]   (flow_field.py:66-89)
//
// What a library FFT cannot know is that most of the padded input is zero (7/8
// for volumes, 3/4 in the plane) and that only the cropped surface is wanted.
// The transform is done as passes of 1-D FFTs that skip the zero parts and fuse
// their neighbours:
//
//   forward  x: rows (z < P0, y < P1) only; reads the un-padded patch, two real
//               rows per complex FFT (two-for-one), writes F2/2+1 bins per row
//            y: planes z < P0 only; P1 input samples, zero extended in LDS
//            z: P0 input samples; for the second operand the product with the
//               conjugate... (A conj(B)) is formed here  [in-plane patches: no z
//               passes, the product rides on the y pass]
//   inverse  z, y: full
//            x: two Hermitian rows per complex FFT; scales, crops with the
//               wrap-around index and stores the surface row directly; leaves
//               the surface maximum for the peak search
//
// One workgroup transforms KT pencils at a time in LDS with a Stockham auto-sort
// FFT (radices 2, 3, 4, 5; KT = 16 / 8 / 4 by length, lengths up to 1728),
// pencils laid out [n][t] so that global accesses are contiguous across the
// pencils of a tile (strided passes) or along the pencil (x passes).  A y axis
// beyond one tile (whole-overlap strips: 8192) takes two strided passes with a
// twiddle in between (four-step split; see PencilArgs).
#include "sfm_common.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace sfm {

namespace {

constexpr int kThreads = 256;
// Pencils per workgroup: 16, or 8 for lengths whose two LDS buffers would not
// fit with 16 (KT is a template parameter of the kernels; the LDS pitch KT + 1
// complex elements is odd: the x passes run along n).
constexpr int kMaxN = 1728;   // 4 pencils per workgroup: (2 n 5 + n) 8 bytes <= 150 KB
constexpr int kMaxStages = 12;
constexpr size_t kLdsLimit = 150 * 1024;

struct Plan {
  int N;
  int stages;
  int radix[kMaxStages];
};

bool make_plan(int n, Plan* p) {
  p->N = n;
  p->stages = 0;
  if (n < 2 || n > kMaxN) return false;
  int m = n;
  for (int r : {4, 2, 3, 5})
    while (m % r == 0) {
      if (p->stages == kMaxStages) return false;
      p->radix[p->stages++] = r;
      m /= r;
    }
  return m == 1;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 conjf2(float2 a) { return make_float2(a.x, -a.y); }
// multiplication by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a) {
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// r-point DFT in place, forward sign exp(-2 pi i / r), inverse: conjugate roots.
template <bool INV>
__device__ __forceinline__ void dft2(float2* v) {
  const float2 a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2* v) {
  const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
  const float2 s13 = cadd(v[1], v[3]), d13 = rot90<INV>(csub(v[1], v[3]));
  v[0] = cadd(s02, s13);
  v[2] = csub(s02, s13);
  v[1] = cadd(d02, d13);
  v[3] = csub(d02, d13);
}
template <bool INV>
__device__ __forceinline__ void dft3(float2* v) {
  constexpr float kC = -0.5f, kS = 0.86602540378443864676f;
  const float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  const float2 m = make_float2(v[0].x + kC * s.x, v[0].y + kC * s.y);
  // forward: -i kS d, inverse: +i kS d
  const float2 r = INV ? make_float2(-kS * d.y, kS * d.x) : make_float2(kS * d.y, -kS * d.x);
  v[0] = cadd(v[0], s);
  v[1] = cadd(m, r);
  v[2] = csub(m, r);
}
template <bool INV>
__device__ __forceinline__ void dft5(float2* v) {
  constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
  const float2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
  const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
  const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
  // t1 = s1 b1 + s2 b2, t2 = s2 b1 - s1 b2; forward multiplies them by -i
  const float2 t1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
  const float2 t2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
  const float2 r1 = rot90<INV>(t1), r2 = rot90<INV>(t2);
  v[0] = cadd(v[0], cadd(a1, a2));
  v[1] = cadd(m1, r1);
  v[4] = csub(m1, r1);
  v[2] = cadd(m2, r2);
  v[3] = csub(m2, r2);
}

// One Stockham stage of radix R over the kT pencils of a tile (compile-time R:
// the butterfly lives in registers; the sub-transform length ns is a power of
// two except after an odd radix, so k = j mod ns is a mask almost always).
template <int R, bool INV, int KT>
__device__ __forceinline__ void lds_stage(const float2* a, float2* b, const float2* tw, int N,
                                          int ns) {
  constexpr int kT = KT, kTP = KT + 1;
  const int nr = N / R;
  const int twstep = N / (ns * R);
  const bool pow2 = (ns & (ns - 1)) == 0;
  for (int it = threadIdx.x; it < nr * kT; it += kThreads) {
    const int t = it % kT, j = it / kT;
    const int k = pow2 ? (j & (ns - 1)) : (j % ns);
    float2 v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = a[(j + q * nr) * kTP + t];
    if (k) {  // W^(k q twstep): k q twstep < N
#pragma unroll
      for (int q = 1; q < R; ++q) {
        float2 w = tw[k * q * twstep];
        if (INV) w.y = -w.y;
        v[q] = cmul(v[q], w);
      }
    }
    if (R == 4) dft4<INV>(v);
    else if (R == 2) dft2<INV>(v);
    else if (R == 5) dft5<INV>(v);
    else dft3<INV>(v);
    const int j0 = (j - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) b[(j0 + q * ns) * kTP + t] = v[q];
  }
}

// Stockham auto-sort FFT of kT pencils held in LDS as buf[n * kTP + t]; the
// result is in the returned buffer (one of the two).  tw[k] = exp(-2 pi i k / N).
template <bool INV, int KT>
__device__ float2* lds_fft(float2* a, float2* b, const float2* tw, const Plan& pl) {
  const int N = pl.N;
  int ns = 1;
  for (int st = 0; st < pl.stages; ++st) {
    const int r = pl.radix[st];
    switch (r) {
      case 4: lds_stage<4, INV, KT>(a, b, tw, N, ns); break;
      case 2: lds_stage<2, INV, KT>(a, b, tw, N, ns); break;
      case 5: lds_stage<5, INV, KT>(a, b, tw, N, ns); break;
      default: lds_stage<3, INV, KT>(a, b, tw, N, ns); break;
    }
    __syncthreads();
    float2* tmp = a;
    a = b;
    b = tmp;
    ns *= r;
  }
  return a;
}

// ---------------------------------------------------------------------------
// strided pencils (y and z passes)
// ---------------------------------------------------------------------------
struct PencilArgs {
  const float2* in;
  float2* out;
  const float2* mul;   // optional: out = FFT(in) * conj(mul) ... see conj_out
  Plan plan;
  const float2* tw;
  int n_in;            // input samples per pencil (the rest is zero)
  long long stride;    // elements between consecutive samples of a pencil (in and out)
  int n_inner;         // pencils that are contiguous in memory
  int n_o0;            // first outer index (e.g. planes z < P0)
  long long s_o0;
  int n_o1;            // second outer index (batch)
  long long s_o1;
  int product;         // 1: out = mul * conj(FFT(in))  (A conj(B), mul = A);  2: FFT(in * conj(mul))
  // Four-step split of a long axis (N = N1 N2, sample n = N2 n1 + n2, bin k = k1 + N1 k2):
  // pass A transforms over n1 for every n2 (= o0), pass B over n2 for every k1 (= o0);
  // between them every element (k1, n2) is multiplied by W_N^(k1 n2) [conjugated on the
  // way back], which rides on the store of the pass in front of the multiplication.
  int nin_step;        // > 0: pencil o0 has ceil((n_in - o0) / nin_step) non-zero samples
  const float2* twl;   // table exp(-2 pi i j / N), j < N, or NULL
  int twl_conj;
};

template <bool INV, int KT>
__global__ void __launch_bounds__(kThreads) fft_pencil_kernel(PencilArgs g) {
  constexpr int kT = KT, kTP = KT + 1;
  extern __shared__ float2 fft_lds[];
  float2* bufa = fft_lds;
  float2* bufb = bufa + g.plan.N * kTP;
  float2* tw = bufb + g.plan.N * kTP;
  const int N = g.plan.N;
  for (int k = threadIdx.x; k < N; k += kThreads) tw[k] = g.tw[k];
  const int tiles_inner = (g.n_inner + kT - 1) / kT;
  long long tile = blockIdx.x;
  const int ti = static_cast<int>(tile % tiles_inner);
  tile /= tiles_inner;
  const int o0 = static_cast<int>(tile % g.n_o0);
  const int o1 = static_cast<int>(tile / g.n_o0);
  const long long base = o1 * g.s_o1 + o0 * g.s_o0 + (long long)ti * kT;
  const int t = threadIdx.x % kT, n0 = threadIdx.x / kT;
  constexpr int kRowsPerIt = kThreads / kT;
  const bool t_ok = ti * kT + t < g.n_inner;
  // load: rows n of the tile, kT consecutive pencils each; ten rows per round,
  // unconditional loads from clamped addresses (loads under per-lane conditions
  // would be waited for one by one), zeros for the padding
  const int tc = min(t, g.n_inner - 1 - ti * kT);
  const int n_in = g.nin_step ? max(0, (g.n_in - o0 + g.nin_step - 1) / g.nin_step) : g.n_in;
  const int n_in_c = max(n_in, 1);   // clamp of the load addresses
  constexpr int kLd = 10;  // N = 160: the whole pencil in one round
  for (int nb0 = n0; nb0 < N; nb0 += kLd * kRowsPerIt) {
    float2 v[kLd];
#pragma unroll
    for (int u = 0; u < kLd; ++u) {
      const int n = min(nb0 + u * kRowsPerIt, n_in_c - 1);
      v[u] = g.in[base + n * g.stride + tc];
    }
    if (g.product == 2) {  // (wave-uniform: a second group of loads)
      float2 m[kLd];
#pragma unroll
      for (int u = 0; u < kLd; ++u)
        m[u] = g.mul[base + min(nb0 + u * kRowsPerIt, n_in_c - 1) * g.stride + tc];
#pragma unroll
      for (int u = 0; u < kLd; ++u) v[u] = cmul(v[u], conjf2(m[u]));
    }
#pragma unroll
    for (int u = 0; u < kLd; ++u) {
      const int n = nb0 + u * kRowsPerIt;
      if (n < N) bufa[n * kTP + t] = (n < n_in && t_ok) ? v[u] : make_float2(0.f, 0.f);
    }
  }
  __syncthreads();
  float2* res = lds_fft<INV, KT>(bufa, bufb, tw, g.plan);
  if (g.product == 1) {
    for (int nb0 = n0; nb0 < N; nb0 += kLd * kRowsPerIt) {
      float2 m[kLd];
#pragma unroll
      for (int u = 0; u < kLd; ++u)
        m[u] = g.mul[base + min(nb0 + u * kRowsPerIt, N - 1) * g.stride + tc];
#pragma unroll
      for (int u = 0; u < kLd; ++u) {
        const int n = nb0 + u * kRowsPerIt;
        if (n < N && t_ok) g.out[base + n * g.stride + t] = cmul(m[u], conjf2(res[n * kTP + t]));
      }
    }
  } else if (g.twl) {
    for (int n = n0; n < N; n += kRowsPerIt) {
      float2 w = g.twl[(long long)o0 * n];   // o0 n < N1 N2
      if (g.twl_conj) w.y = -w.y;
      if (t_ok) g.out[base + n * g.stride + t] = cmul(res[n * kTP + t], w);
    }
  } else {
    for (int n = n0; n < N; n += kRowsPerIt)
      if (t_ok) g.out[base + n * g.stride + t] = res[n * kTP + t];
  }
}

// ---------------------------------------------------------------------------
// forward x pass: un-padded real rows, two per complex FFT
// ---------------------------------------------------------------------------
struct XFwdArgs {
  const float* src;   // [nb, P0, P1, P2] mean-subtracted patch values
  float2* out;        // [nb, F0, F1, C] half spectra (rows y < P1 of planes z < P0 written)
  Plan plan;          // N = F2
  const float2* tw;
  int P[3], F[3], C;
  int nb;
  int square;         // transform the squares of the values (masked terms)
};

template <int KT>
__global__ void __launch_bounds__(kThreads) fft_xfwd_kernel(XFwdArgs g) {
  constexpr int kT = KT, kTP = KT + 1;
  extern __shared__ float2 fft_lds[];
  float2* bufa = fft_lds;
  float2* bufb = bufa + g.plan.N * kTP;
  float2* tw = bufb + g.plan.N * kTP;
  const int N = g.plan.N;
  for (int k = threadIdx.x; k < N; k += kThreads) tw[k] = g.tw[k];
  // pencil = (b, z, row pair); a tile = kT consecutive pairs of one (b, z) plane
  const int pairs = (g.P[1] + 1) / 2;
  const int tiles = (pairs + kT - 1) / kT;
  long long tile = blockIdx.x;
  const int tp = static_cast<int>(tile % tiles);
  tile /= tiles;
  const int z = static_cast<int>(tile % g.P[0]);
  const int b = static_cast<int>(tile / g.P[0]);
  const float* plane = g.src + ((long long)b * g.P[0] + z) * g.P[1] * g.P[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // load: a wave takes pencils t = wave, wave + 4, ...; lanes run along x
  for (int t = wave; t < kT; t += kThreads / 64) {
    const int y = 2 * (tp * kT + t);
    const int y0c = min(y, g.P[1] - 1), y1c = min(y + 1, g.P[1] - 1);
    // four 64-sample groups per round (unconditional loads from clamped indices)
    for (int u0 = 0; u0 * 64 < N; u0 += 4) {
      float r0[4], r1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = min(lane + 64 * (u0 + u), g.P[2] - 1);
        r0[u] = plane[(long long)y0c * g.P[2] + n];
        r1[u] = plane[(long long)y1c * g.P[2] + n];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = lane + 64 * (u0 + u);
        if (n >= N) continue;
        float2 v = make_float2(0.f, 0.f);
        if (n < g.P[2]) {
          if (y < g.P[1]) v.x = g.square ? r0[u] * r0[u] : r0[u];
          if (y + 1 < g.P[1]) v.y = g.square ? r1[u] * r1[u] : r1[u];
        }
        bufa[n * kTP + t] = v;
      }
    }
  }
  __syncthreads();
  const float2* res = lds_fft<false, KT>(bufa, bufb, tw, g.plan);
  // split: X1[k] = (Z[k] + conj(Z[N-k])) / 2,  X2[k] = (Z[k] - conj(Z[N-k])) / (2 i)
  for (int t = wave; t < kT; t += kThreads / 64) {
    const int y = 2 * (tp * kT + t);
    if (y >= g.P[1]) continue;
    float2* row0 = g.out + (((long long)b * g.F[0] + z) * g.F[1] + y) * g.C;
    float2* row1 = row0 + g.C;
    for (int k = lane; k < g.C; k += 64) {
      const float2 zk = res[k * kTP + t];
      const float2 zn = conjf2(res[((N - k) % N) * kTP + t]);
      row0[k] = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
      if (y + 1 < g.P[1]) {
        const float2 d = csub(zk, zn);          // 2 i X2
        row1[k] = make_float2(0.5f * d.y, -0.5f * d.x);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// inverse x pass: two Hermitian rows per complex FFT, scale, crop, maximum
// ---------------------------------------------------------------------------
struct XInvArgs {
  const float2* in;   // [nb, F0, F1, C]
  float* out;         // [nb, S0, S1, S2] surface
  unsigned int* smax; // [nb] or NULL
  Plan plan;
  const float2* tw;
  int F[3], S[3], Q[3], C;
  float scale;
  int nb;
  int raw;            // 1: out = the un-scaled circular real array [nb, F0, F1, F2]
};

// surface index of circular index d along an axis, or -1 (the padding gap)
__device__ __forceinline__ int surf_index(int d, int F, int Q, int S) {
  // crop: d = (k - (Q - 1)) mod F  <=>  k = d + Q - 1 (d small) or d - F + Q - 1
  int k = d + Q - 1;
  if (k >= S) k = d - F + Q - 1;
  return (k >= 0 && k < S) ? k : -1;
}

template <int KT>
__global__ void __launch_bounds__(kThreads) fft_xinv_kernel(XInvArgs g) {
  constexpr int kT = KT, kTP = KT + 1;
  extern __shared__ float2 fft_lds[];
  float2* bufa = fft_lds;
  float2* bufb = bufa + g.plan.N * kTP;
  float2* tw = bufb + g.plan.N * kTP;
  const int N = g.plan.N;
  for (int k = threadIdx.x; k < N; k += kThreads) tw[k] = g.tw[k];
  const int pairs = g.F[1] / 2;  // F is even
  const int tiles = (pairs + kT - 1) / kT;
  long long tile = blockIdx.x;
  const int tp = static_cast<int>(tile % tiles);
  tile /= tiles;
  const int dz = static_cast<int>(tile % g.F[0]);
  const int b = static_cast<int>(tile / g.F[0]);
  const int kz = g.raw ? dz : surf_index(dz, g.F[0], g.Q[0], g.S[0]);
  if (kz < 0) return;  // whole workgroup: a plane of the gap
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // all loads of a round first: the wave's pencils x four 64-bin groups
  constexpr int kPW = kT / (kThreads / 64);  // pencils per wave
  for (int u0 = 0; u0 * 64 < N; u0 += 4) {
    float2 s1[kPW][4], s2[kPW][4];
#pragma unroll
    for (int p = 0; p < kPW; ++p) {
      const int dy = 2 * (tp * kT + wave + p * (kThreads / 64));
      const float2* row0 =
          g.in + (((long long)b * g.F[0] + dz) * g.F[1] + min(dy, g.F[1] - 2)) * g.C;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = min(lane + 64 * (u0 + u), N - 1);
        const int kk = k < g.C ? k : N - k;
        s1[p][u] = row0[kk];
        s2[p][u] = row0[g.C + kk];
      }
    }
#pragma unroll
    for (int p = 0; p < kPW; ++p) {
      const int t = wave + p * (kThreads / 64);
      const bool live = 2 * (tp * kT + t) < g.F[1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = lane + 64 * (u0 + u);
        if (k >= N) continue;
        // Z[k] = S1[k] + i S2[k]; the upper half from the Hermitian symmetry
        float2 a1 = s1[p][u], a2 = s2[p][u];
        if (k >= g.C) {
          a1 = conjf2(a1);
          a2 = conjf2(a2);
        }
        float2 v = make_float2(a1.x - a2.y, a1.y + a2.x);
        if (!live) v = make_float2(0.f, 0.f);
        bufa[k * kTP + t] = v;
      }
    }
  }
  __syncthreads();
  const float2* res = lds_fft<true, KT>(bufa, bufb, tw, g.plan);
  float mx = -INFINITY;
  for (int t = wave; t < kT; t += kThreads / 64) {
    const int dy = 2 * (tp * kT + t);
    if (dy >= g.F[1]) continue;
    if (g.raw) {
      float* o = g.out + (((long long)b * g.F[0] + dz) * g.F[1] + dy) * g.F[2];
      for (int x = lane; x < N; x += 64) {
        const float2 v = res[x * kTP + t];
        o[x] = v.x;
        o[g.F[2] + x] = v.y;
      }
      continue;
    }
    const int ky0 = surf_index(dy, g.F[1], g.Q[1], g.S[1]);
    const int ky1 = surf_index(dy + 1, g.F[1], g.Q[1], g.S[1]);
    float* o0 = ky0 >= 0 ? g.out + (((long long)b * g.S[0] + kz) * g.S[1] + ky0) * g.S[2] : nullptr;
    float* o1 = ky1 >= 0 ? g.out + (((long long)b * g.S[0] + kz) * g.S[1] + ky1) * g.S[2] : nullptr;
    for (int kx = lane; kx < g.S[2]; kx += 64) {
      int dx = kx - (g.Q[2] - 1);
      if (dx < 0) dx += g.F[2];
      const float2 v = res[dx * kTP + t];
      if (o0) {
        const float r = v.x * g.scale;
        o0[kx] = r;
        mx = fmaxf(mx, r);
      }
      if (o1) {
        const float r = v.y * g.scale;
        o1[kx] = r;
        mx = fmaxf(mx, r);
      }
    }
  }
  if (g.smax) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if (lane == 0 && mx > -INFINITY) {
      const unsigned u = __float_as_uint(mx);
      const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      if (o > __atomic_load_n(&g.smax[b], __ATOMIC_RELAXED)) atomicMax(&g.smax[b], o);
    }
  }
}

// twiddle tables, per device and length
std::mutex g_tw_mu;
std::map<std::pair<int, int>, float2*> g_tw;

const float2* twiddles(int n) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_tw_mu);
  auto it = g_tw.find({dev, n});
  if (it != g_tw.end()) return it->second;
  std::vector<float2> h(n);
  for (int k = 0; k < n; ++k) {
    const double ang = -2.0 * M_PI * k / n;
    h[k] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
  }
  float2* d = nullptr;
  if (hipMalloc(&d, n * sizeof(float2)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  g_tw[{dev, n}] = d;
  return d;
}

size_t lds_bytes(int n, int kt) { return (size_t)(2 * n * (kt + 1) + n) * sizeof(float2); }

// Pencils per workgroup for transforms of length n: 16 while the LDS tile fits.
int pick_kt(int n) {
  return lds_bytes(n, 16) <= kLdsLimit ? 16 : lds_bytes(n, 8) <= kLdsLimit ? 8 : 4;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024)
    SFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(bytes)));
  return SFM_OK;
}

// Launchers that pick the KT instantiation.
int launch_pencil(bool inv, const PencilArgs& a, long long tiles_outer, hipStream_t st) {
  const int kt = pick_kt(a.plan.N);
  const size_t lds = lds_bytes(a.plan.N, kt);
  const unsigned grid = (unsigned)(tiles_outer * ((a.n_inner + kt - 1) / kt));
#define SFM_PENCIL(INV, KT)                                                          \
  do {                                                                               \
    if (int rc = set_lds(&fft_pencil_kernel<INV, KT>, lds)) return rc;               \
    hipLaunchKernelGGL((fft_pencil_kernel<INV, KT>), dim3(grid), dim3(kThreads), lds, st, a); \
  } while (0)
  if (inv) {
    if (kt == 16) SFM_PENCIL(true, 16); else if (kt == 8) SFM_PENCIL(true, 8); else SFM_PENCIL(true, 4);
  } else {
    if (kt == 16) SFM_PENCIL(false, 16); else if (kt == 8) SFM_PENCIL(false, 8); else SFM_PENCIL(false, 4);
  }
#undef SFM_PENCIL
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int launch_xfwd(const XFwdArgs& x, hipStream_t st) {
  const int kt = pick_kt(x.plan.N);
  const size_t lds = lds_bytes(x.plan.N, kt);
  const int xt = ((x.P[1] + 1) / 2 + kt - 1) / kt;
  const unsigned grid = (unsigned)((long long)x.nb * x.P[0] * xt);
  if (kt == 16) {
    if (int rc = set_lds(&fft_xfwd_kernel<16>, lds)) return rc;
    hipLaunchKernelGGL(fft_xfwd_kernel<16>, dim3(grid), dim3(kThreads), lds, st, x);
  } else if (kt == 8) {
    if (int rc = set_lds(&fft_xfwd_kernel<8>, lds)) return rc;
    hipLaunchKernelGGL(fft_xfwd_kernel<8>, dim3(grid), dim3(kThreads), lds, st, x);
  } else {
    if (int rc = set_lds(&fft_xfwd_kernel<4>, lds)) return rc;
    hipLaunchKernelGGL(fft_xfwd_kernel<4>, dim3(grid), dim3(kThreads), lds, st, x);
  }
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int launch_xinv(const XInvArgs& x, hipStream_t st) {
  const int kt = pick_kt(x.plan.N);
  const size_t lds = lds_bytes(x.plan.N, kt);
  const int it = (x.F[1] / 2 + kt - 1) / kt;
  const unsigned grid = (unsigned)((long long)x.nb * x.F[0] * it);
  if (kt == 16) {
    if (int rc = set_lds(&fft_xinv_kernel<16>, lds)) return rc;
    hipLaunchKernelGGL(fft_xinv_kernel<16>, dim3(grid), dim3(kThreads), lds, st, x);
  } else if (kt == 8) {
    if (int rc = set_lds(&fft_xinv_kernel<8>, lds)) return rc;
    hipLaunchKernelGGL(fft_xinv_kernel<8>, dim3(grid), dim3(kThreads), lds, st, x);
  } else {
    if (int rc = set_lds(&fft_xinv_kernel<4>, lds)) return rc;
    hipLaunchKernelGGL(fft_xinv_kernel<4>, dim3(grid), dim3(kThreads), lds, st, x);
  }
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}


}  // namespace

namespace {

struct OwnGeo {
  Plan px, py, pz;
  const float2 *twx, *twy, *twz;
  int C;
  long long plane, vol;
  bool has_z;   // false: in-plane patches (F[0] == 1), the z passes are skipped
  // long y axis (in-plane only): F1 = n1 * n2, two pencil passes (py = plan of n1,
  // py2 = plan of n2, twl = table of F1)
  int n1, n2;
  Plan py2;
  const float2 *twy2, *twl;
};

// Balanced split N = n1 n2 of a length beyond one LDS tile into two that fit.
bool split_long(int n, int* n1, int* n2) {
  Plan p;
  int best = 0;
  for (int d = 2; (long long)d * d <= n; ++d)
    if (n % d == 0 && make_plan(d, &p) && make_plan(n / d, &p) &&
        lds_bytes(n / d, 16) <= kLdsLimit)
      best = d;
  if (!best) return false;
  *n1 = best;
  *n2 = n / best;
  return true;
}

bool fits_tile(int n) {
  Plan p;
  return make_plan(n, &p) && lds_bytes(n, 4) <= kLdsLimit;
}

int own_setup(const int* F, OwnGeo* o) {
  o->has_z = F[0] > 1;
  o->n1 = o->n2 = 0;
  o->twy2 = o->twl = nullptr;
  if (!o->has_z && !fits_tile(F[1])) {
    if (!split_long(F[1], &o->n1, &o->n2) || !make_plan(o->n1, &o->py) ||
        !make_plan(o->n2, &o->py2))
      return fail(SFM_ERR_INVALID, "own FFT: unsupported length %d", F[1]);
    o->twy2 = twiddles(o->n2);
    o->twl = twiddles(F[1]);
    if (!o->twy2 || !o->twl) return fail(SFM_ERR_HIP, "own FFT: twiddle table allocation failed");
  } else if (!make_plan(F[1], &o->py)) {
    return fail(SFM_ERR_INVALID, "own FFT: unsupported length %d", F[1]);
  }
  if (!make_plan(F[2], &o->px) || (o->has_z && !make_plan(F[0], &o->pz)))
    return fail(SFM_ERR_INVALID, "own FFT: unsupported length");
  o->twx = twiddles(F[2]);
  o->twy = twiddles(o->n1 ? o->n1 : F[1]);
  o->twz = o->has_z ? twiddles(F[0]) : o->twy;
  if (!o->twx || !o->twy || !o->twz)
    return fail(SFM_ERR_HIP, "own FFT: twiddle table allocation failed");
  o->C = F[2] / 2 + 1;
  o->plane = (long long)F[1] * o->C;
  o->vol = (long long)F[0] * o->plane;
  return SFM_OK;
}

// Half spectrum [nb, F0, F1, C] of the zero-padded patches src [nb, R0, R1, R2];
// mul != NULL: the product mul * conj(spectrum) instead (formed in the last pass).
int own_forward(const OwnGeo& o, const int* R, const int* F, int nb, const float* src,
                int square, float2* spec, const float2* mul, hipStream_t st) {
  XFwdArgs x;
  x.src = src;
  x.out = spec;
  x.plan = o.px;
  x.tw = o.twx;
  for (int i = 0; i < 3; ++i) {
    x.P[i] = R[i];
    x.F[i] = F[i];
  }
  x.C = o.C;
  x.nb = nb;
  x.square = square;
  if (int rc = launch_xfwd(x, st)) return rc;
  PencilArgs y = {};
  y.in = spec;
  y.out = spec;
  y.mul = nullptr;
  y.plan = o.py;
  y.tw = o.twy;
  y.n_in = R[1];
  y.stride = o.C;
  y.n_inner = o.C;
  y.n_o0 = R[0];
  y.s_o0 = o.plane;
  y.n_o1 = nb;
  y.s_o1 = o.vol;
  y.product = 0;
  y.nin_step = 0;
  y.twl = nullptr;
  y.twl_conj = 0;
  if (o.n1) {
    // long y axis: pass A over n1 for every n2 (rows N2 n1 + n2), twiddle on the
    // store; pass B over n2 for every k1 -> bin k1 + N1 k2 at row N2 k1 + k2
    PencilArgs pa = y;
    pa.stride = (long long)o.n2 * o.C;
    pa.n_o0 = o.n2;
    pa.s_o0 = o.C;
    pa.nin_step = o.n2;      // rows < R[1]: ceil((R1 - n2) / N2) samples
    pa.twl = o.twl;
    if (int rc = launch_pencil(false, pa, (long long)nb * o.n2, st)) return rc;
    PencilArgs pb = y;
    pb.plan = o.py2;
    pb.tw = o.twy2;
    pb.n_in = o.n2;
    pb.n_o0 = o.n1;
    pb.s_o0 = (long long)o.n2 * o.C;
    if (mul) {
      pb.mul = mul;
      pb.product = 1;
    }
    return launch_pencil(false, pb, (long long)nb * o.n1, st);
  }
  if (!o.has_z && mul) {   // in-plane: the product rides on the last (y) pass
    y.mul = mul;
    y.product = 1;
  }
  if (int rc = launch_pencil(false, y, (long long)nb * R[0], st)) return rc;
  if (!o.has_z) return SFM_OK;
  PencilArgs z = y;
  z.plan = o.pz;
  z.tw = o.twz;
  z.n_in = R[0];
  z.stride = o.plane;
  z.n_inner = static_cast<int>(o.plane);
  z.n_o0 = 1;
  z.s_o0 = 0;
  if (mul) {
    z.mul = mul;
    z.product = 1;
  }
  return launch_pencil(false, z, nb, st);
}

// Inverse of spec (mul != NULL: of spec * conj(mul), formed while loading) through
// `work` (may be spec itself when mul == NULL); the x pass either crops into the
// surface or writes the raw circular array.
int own_inverse(const OwnGeo& o, const int* F, int nb, const float2* spec, const float2* mul,
                float2* work, const XInvArgs& xi_in, hipStream_t st) {
  PencilArgs z = {};
  z.in = spec;
  z.out = work;
  z.mul = mul;
  z.plan = o.pz;
  z.tw = o.twz;
  z.n_in = F[0];
  z.stride = o.plane;
  z.n_inner = static_cast<int>(o.plane);
  z.n_o0 = 1;
  z.s_o0 = 0;
  z.n_o1 = nb;
  z.s_o1 = o.vol;
  z.product = mul ? 2 : 0;
  z.nin_step = 0;
  z.twl = nullptr;
  z.twl_conj = 0;
  if (o.has_z)
    if (int rc = launch_pencil(true, z, nb, st)) return rc;
  PencilArgs y = z;
  if (o.has_z) {
    y.in = work;
    y.mul = nullptr;
    y.product = 0;
  }   // (in-plane: the y pass is the first one and forms the product itself)
  y.plan = o.py;
  y.tw = o.twy;
  y.n_in = F[1];
  y.stride = o.C;
  y.n_inner = o.C;
  y.n_o0 = F[0];
  y.s_o0 = o.plane;
  if (o.n1) {
    // long y axis, backwards: over k2 for every k1 (conjugate twiddle on the
    // store), then over k1 for every n2 -> rows in natural order
    PencilArgs pb = y;     // carries the product-on-load of the first inverse pass
    pb.plan = o.py2;
    pb.tw = o.twy2;
    pb.n_in = o.n2;
    pb.stride = o.C;
    pb.n_o0 = o.n1;
    pb.s_o0 = (long long)o.n2 * o.C;
    pb.twl = o.twl;
    pb.twl_conj = 1;
    if (int rc = launch_pencil(true, pb, (long long)nb * o.n1, st)) return rc;
    PencilArgs pa = y;
    pa.in = work;
    pa.mul = nullptr;
    pa.product = 0;
    pa.plan = o.py;
    pa.tw = o.twy;
    pa.n_in = o.n1;
    pa.stride = (long long)o.n2 * o.C;
    pa.n_o0 = o.n2;
    pa.s_o0 = o.C;
    if (int rc = launch_pencil(true, pa, (long long)nb * o.n2, st)) return rc;
  } else if (int rc = launch_pencil(true, y, (long long)nb * F[0], st)) {
    return rc;
  }
  XInvArgs xi = xi_in;
  xi.in = work;
  xi.plan = o.px;
  xi.tw = o.twx;
  for (int i = 0; i < 3; ++i) xi.F[i] = F[i];
  xi.C = o.C;
  xi.nb = nb;
  return launch_xinv(xi, st);
}

}  // namespace

// Shapes this path takes: 2-D (F[0] == 1) and 3-D patches whose padded lengths
// are even and factor into 2, 3, 5; every axis fits one LDS tile (<= ~1000), the
// y axis of in-plane patches may be longer (four-step split).
bool own_fft_supported(int rank, const int* F) {
  if (rank != 2 && rank != 3) return false;
  for (int i = rank == 2 ? 1 : 0; i < 3; ++i)
    if (F[i] & 1) return false;
  if (rank == 3) return fits_tile(F[0]) && fits_tile(F[1]) && fits_tile(F[2]);
  int n1, n2;
  return F[0] == 1 && fits_tile(F[2]) && (fits_tile(F[1]) || split_long(F[1], &n1, &n2));
}

bool own_fft_fits_tile(int n) { return (n & 1) == 0 && fits_tile(n); }

// Un-masked correlation.  a0 / b0: [nb, Pn] / [nb, Qn] mean-subtracted patches; sa /
// sb: two half-spectrum buffers [nb, F0, F1, C]; surface: [nb, Sn]; smax: [nb] or
// NULL (zeroed).
int own_fft_correlate(const int* P, const int* Q, const int* S, const int* F, int nb,
                      const float* a0, const float* b0, float2* sa, float2* sb, float* surface,
                      unsigned int* smax, hipStream_t st) {
  OwnGeo o;
  if (int rc = own_setup(F, &o)) return rc;
  if (int rc = own_forward(o, P, F, nb, a0, 0, sa, nullptr, st)) return rc;
  if (int rc = own_forward(o, Q, F, nb, b0, 0, sb, sa, st)) return rc;  // sb = A conj(B)
  XInvArgs xi;
  xi.out = surface;
  xi.smax = smax;
  for (int i = 0; i < 3; ++i) {
    xi.S[i] = S[i];
    xi.Q[i] = Q[i];
  }
  xi.scale = 1.0f / (static_cast<float>(F[0]) * F[1] * F[2]);
  xi.raw = 0;
  return own_inverse(o, F, nb, sb, nullptr, sb, xi, st);
}

// Building blocks of the masked (six-term) correlation: spectrum of one plane
// (values or their squares), and the raw circular inverse of lhs * conj(rhs).
int own_fft_forward(const int* R, const int* F, int nb, const float* src, int square,
                    float2* spec, hipStream_t st) {
  OwnGeo o;
  if (int rc = own_setup(F, &o)) return rc;
  return own_forward(o, R, F, nb, src, square, spec, nullptr, st);
}

int own_fft_inverse_product(const int* F, int nb, const float2* lhs, const float2* rhs,
                            float2* work, float* real_out, hipStream_t st) {
  OwnGeo o;
  if (int rc = own_setup(F, &o)) return rc;
  XInvArgs xi;
  xi.out = real_out;
  xi.smax = nullptr;
  for (int i = 0; i < 3; ++i) {
    xi.S[i] = F[i];
    xi.Q[i] = 1;
  }
  xi.scale = 1.f;
  xi.raw = 1;
  return own_inverse(o, F, nb, lhs, rhs, work, xi, st);
}


// ---------------------------------------------------------------------------
// In-plane patches whose padded extent exceeds one LDS tile along BOTH axes
// (patches beyond ~860 px each way).  No half-spectrum tricks: the real patch is
// expanded to complex and TRANSPOSED, so that its x axis is the row index; the
// transform along the rows (one pass, or the four-step split) is the x
// transform; a complex transpose makes y the row index and the same routine
// transforms y.  The spectrum is [F1][F2] complex with both axes in the order
// the forward passes leave (element-wise products do not care); the inverse
// runs the same steps backwards and ends with a transposing real-part crop.
// ---------------------------------------------------------------------------
namespace {

struct AxisPlan {
  int N, n1, n2;          // n1 == 0: one pencil pass
  Plan p1, p2;
  const float2 *tw1, *tw2, *twl;
};

int make_axis(int n, AxisPlan* ax) {
  ax->N = n;
  ax->n1 = ax->n2 = 0;
  ax->tw2 = ax->twl = nullptr;
  if (fits_tile(n)) {
    if (!make_plan(n, &ax->p1)) return fail(SFM_ERR_INVALID, "own FFT: unsupported length %d", n);
    ax->tw1 = twiddles(n);
  } else {
    if (!split_long(n, &ax->n1, &ax->n2) || !make_plan(ax->n1, &ax->p1) ||
        !make_plan(ax->n2, &ax->p2))
      return fail(SFM_ERR_INVALID, "own FFT: unsupported length %d", n);
    ax->tw1 = twiddles(ax->n1);
    ax->tw2 = twiddles(ax->n2);
    ax->twl = twiddles(n);
    if (!ax->tw2 || !ax->twl) return fail(SFM_ERR_HIP, "own FFT: twiddle table allocation failed");
  }
  if (!ax->tw1) return fail(SFM_ERR_HIP, "own FFT: twiddle table allocation failed");
  return SFM_OK;
}

// FFT along the rows of nb arrays [N][C] (C contiguous pencils); forward: rows >=
// n_in are zero on input; mul / product as in PencilArgs (applied by the last
// forward pass / the first inverse pass).
int axis_rows(const AxisPlan& ax, bool inverse, const float2* in, float2* out, const float2* mul,
              int product, int n_in, int C, int nb, hipStream_t st) {
  PencilArgs y = {};
  y.in = in;
  y.out = out;
  y.plan = ax.p1;
  y.tw = ax.tw1;
  y.n_in = inverse ? ax.N : n_in;
  y.stride = C;
  y.n_inner = C;
  y.n_o0 = 1;
  y.s_o0 = 0;
  y.n_o1 = nb;
  y.s_o1 = (long long)ax.N * C;
  if (!ax.n1) {
    y.mul = mul;
    y.product = product;
    return launch_pencil(inverse, y, nb, st);
  }
  PencilArgs pa = y, pb = y;
  pa.stride = (long long)ax.n2 * C;    // over n1 for every n2
  pa.n_o0 = ax.n2;
  pa.s_o0 = C;
  pb.plan = ax.p2;                     // over n2 for every k1
  pb.tw = ax.tw2;
  pb.n_in = ax.n2;
  pb.n_o0 = ax.n1;
  pb.s_o0 = (long long)ax.n2 * C;
  if (!inverse) {
    pa.nin_step = ax.n2;
    pa.twl = ax.twl;
    if (int rc = launch_pencil(false, pa, (long long)nb * ax.n2, st)) return rc;
    pb.in = out;
    pb.mul = mul;
    pb.product = product;
    return launch_pencil(false, pb, (long long)nb * ax.n1, st);
  }
  pb.mul = mul;
  pb.product = product;
  pb.twl = ax.twl;
  pb.twl_conj = 1;
  if (int rc = launch_pencil(true, pb, (long long)nb * ax.n1, st)) return rc;
  pa.in = out;
  pa.n_in = ax.n1;
  return launch_pencil(true, pa, (long long)nb * ax.n2, st);
}

// dst[b][x][y] = (src[b][y][x] (squared), 0) for x < R2, y < R1: [R2][R1] complex
__global__ void __launch_bounds__(kThreads)
expand_t_kernel(const float* __restrict__ src, float2* __restrict__ dst, int R1, int R2,
                long long dst_batch, int square) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const float* s = src + b * (long long)R1 * R2;
  float2* d = dst + b * dst_batch;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const float v = s[(long long)min(y0 + j, R1 - 1) * R2 + min(x0 + tx, R2 - 1)];
    tile[j][tx] = square ? v * v : v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int x = x0 + j, y = y0 + tx;
    if (x < R2 && y < R1) d[(long long)x * R1 + y] = make_float2(tile[tx][j], 0.f);
  }
}

// dst[b][c][r] = src[b][r][c] for r < R, c < Cc (complex)
__global__ void __launch_bounds__(kThreads)
ctranspose_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int R, int Cc,
                  long long src_batch, long long dst_batch) {
  __shared__ float2 tile[32][33];
  const long long b = blockIdx.z;
  const float2* s = src + b * src_batch;
  float2* d = dst + b * dst_batch;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8)
    tile[j][tx] = s[(long long)min(r0 + j, R - 1) * Cc + min(c0 + tx, Cc - 1)];
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < Cc && r < R) d[(long long)c * R + r] = tile[tx][j];
  }
}

// Real part of w [b][F2][F1] (x-major), transposed: the cropped, scaled surface
// [S1][S2] (+ its maximum) or the raw circular array [F1][F2].
struct RealTArgs {
  const float2* w;
  float* out;
  unsigned int* smax;
  int F1, F2, S1, S2, Q1, Q2;
  float scale;
  int raw;
};

__global__ void __launch_bounds__(kThreads) real_t_kernel(RealTArgs g) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int O1 = g.raw ? g.F1 : g.S1, O2 = g.raw ? g.F2 : g.S2;
  const int kx0 = blockIdx.x * 32, ky0 = blockIdx.y * 32;
  // gather with y fastest (contiguous in w), store with x fastest
  for (int j = ty; j < 32; j += 8) {
    const int kx = min(kx0 + j, O2 - 1), ky = min(ky0 + tx, O1 - 1);
    int dx = kx, dy = ky;
    if (!g.raw) {
      dx = kx - (g.Q2 - 1);
      if (dx < 0) dx += g.F2;
      dy = ky - (g.Q1 - 1);
      if (dy < 0) dy += g.F1;
    }
    tile[j][tx] = g.w[b * (long long)g.F1 * g.F2 + (long long)dx * g.F1 + dy].x * g.scale;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = ty; j < 32; j += 8) {
    const int ky = ky0 + j, kx = kx0 + tx;
    if (ky < O1 && kx < O2) {
      const float v = tile[tx][j];
      g.out[b * (long long)O1 * O2 + (long long)ky * O2 + kx] = v;
      mx = fmaxf(mx, v);
    }
  }
  if (g.smax && !g.raw) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0 && mx > -INFINITY) {
      const unsigned u = __float_as_uint(mx);
      const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      if (o > __atomic_load_n(&g.smax[b], __ATOMIC_RELAXED)) atomicMax(&g.smax[b], o);
    }
  }
}

dim3 tgrid(int cols, int rows, int nb) { return dim3((cols + 31) / 32, (rows + 31) / 32, nb); }

struct BigGeo {
  AxisPlan ax, ay;   // transforms of length F2 (x) and F1 (y)
  long long fn;      // F1 * F2
};

int big_setup(const int* F, BigGeo* g) {
  if (int rc = make_axis(F[2], &g->ax)) return rc;
  if (int rc = make_axis(F[1], &g->ay)) return rc;
  g->fn = (long long)F[1] * F[2];
  return SFM_OK;
}

// spec [nb][F1][F2] = transform of the zero-padded src [nb][R1][R2] (mul: mul * conj of
// it); tmp: scratch of the same size.
int big_forward(const BigGeo& g, const int* R, const int* F, int nb, const float* src, int square,
                float2* spec, float2* tmp, const float2* mul, hipStream_t st) {
  const int R1 = R[1], R2 = R[2];
  // tmp as [F2][R1]: x is the row index
  hipLaunchKernelGGL(expand_t_kernel, tgrid(R2, R1, nb), dim3(kThreads), 0, st, src, tmp, R1, R2,
                     (long long)F[2] * R1, square);
  SFM_LAUNCH_CHECK();
  // (batch stride of the [F2][R1] arrays is F2 * R1: axis_rows assumes N * C)
  if (int rc = axis_rows(g.ax, false, tmp, tmp, nullptr, 0, R2, R1, nb, st)) return rc;
  // -> spec as [R1 (of F1)][F2]: y is the row index
  hipLaunchKernelGGL(ctranspose_kernel, tgrid(R1, F[2], nb), dim3(kThreads), 0, st, tmp, spec, F[2],
                     R1, (long long)F[2] * R1, g.fn);
  SFM_LAUNCH_CHECK();
  return axis_rows(g.ay, false, spec, spec, mul, mul ? 1 : 0, R1, F[2], nb, st);
}

// tmp [nb][F2][F1] = inverse transform of spec (* conj(mul)), x-major, un-scaled
int big_inverse(const BigGeo& g, const int* F, int nb, const float2* spec, const float2* mul,
                float2* work, float2* tmp, hipStream_t st) {
  if (int rc = axis_rows(g.ay, true, spec, work, mul, mul ? 2 : 0, F[1], F[2], nb, st)) return rc;
  hipLaunchKernelGGL(ctranspose_kernel, tgrid(F[2], F[1], nb), dim3(kThreads), 0, st, work, tmp, F[1],
                     F[2], g.fn, g.fn);
  SFM_LAUNCH_CHECK();
  return axis_rows(g.ax, true, tmp, tmp, nullptr, 0, F[2], F[1], nb, st);
}

}  // namespace

bool own_fft_big_supported(int rank, const int* F) {
  if (rank != 2 || F[0] != 1 || (F[1] & 1) || (F[2] & 1)) return false;
  AxisPlan a;
  int n1, n2;
  for (int i = 1; i < 3; ++i)
    if (!fits_tile(F[i]) && !split_long(F[i], &n1, &n2)) return false;
  (void)a;
  return true;
}

// The three entry points above for patches that are long along both axes; every
// spectrum / scratch array holds nb * F1 * F2 complex values.
int own_fft_big_correlate(const int* P, const int* Q, const int* S, const int* F, int nb,
                          const float* a0, const float* b0, float2* sa, float2* sb, float2* tmp,
                          float* surface, unsigned int* smax, hipStream_t st) {
  BigGeo g;
  if (int rc = big_setup(F, &g)) return rc;
  if (int rc = big_forward(g, P, F, nb, a0, 0, sa, tmp, nullptr, st)) return rc;
  if (int rc = big_forward(g, Q, F, nb, b0, 0, sb, tmp, sa, st)) return rc;   // sb = A conj(B)
  if (int rc = big_inverse(g, F, nb, sb, nullptr, sb, tmp, st)) return rc;
  RealTArgs r = {tmp, surface, smax, F[1], F[2], S[1], S[2], Q[1], Q[2],
                 1.0f / (static_cast<float>(F[1]) * F[2]), 0};
  hipLaunchKernelGGL(real_t_kernel, tgrid(S[2], S[1], nb), dim3(kThreads), 0, st, r);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

int own_fft_big_forward(const int* R, const int* F, int nb, const float* src, int square,
                        float2* spec, float2* tmp, hipStream_t st) {
  BigGeo g;
  if (int rc = big_setup(F, &g)) return rc;
  return big_forward(g, R, F, nb, src, square, spec, tmp, nullptr, st);
}

int own_fft_big_inverse_product(const int* F, int nb, const float2* lhs, const float2* rhs,
                                float2* work, float2* tmp, float* real_out, hipStream_t st) {
  BigGeo g;
  if (int rc = big_setup(F, &g)) return rc;
  if (int rc = big_inverse(g, F, nb, lhs, rhs, work, tmp, st)) return rc;
  RealTArgs r = {tmp, real_out, nullptr, F[1], F[2], F[1], F[2], 1, 1, 1.f, 1};
  hipLaunchKernelGGL(real_t_kernel, tgrid(F[2], F[1], nb), dim3(kThreads), 0, st, r);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

}  // namespace sfm

// Test hook (not part of the public header): batched 1-D complex FFT of
// contiguous pencils through the strided-pencil kernel, pencil p at in + p,
// samples n at stride n_pencils.
extern "C" int sfm_debug_fft1d(const void* in, void* out, int n, int n_in, int n_pencils,
                               int inverse, void* stream) {
  using namespace sfm;
  Plan pl;
  if (!make_plan(n, &pl)) return fail(SFM_ERR_INVALID, "fft1d: unsupported length %d", n);
  PencilArgs a = {};
  a.in = static_cast<const float2*>(in);
  a.out = static_cast<float2*>(out);
  a.mul = nullptr;
  a.plan = pl;
  a.tw = twiddles(n);
  a.n_in = n_in;
  a.stride = n_pencils;
  a.n_inner = n_pencils;
  a.n_o0 = 1;
  a.s_o0 = 0;
  a.n_o1 = 1;
  a.s_o1 = 0;
  a.product = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return launch_pencil(inverse != 0, a, 1, st);
}
