// Where does the power of the correlation kernel go?  The chip clocks the
// shader to its power budget, so the sustained clock under a synthetic loop is a
// power meter.  Variants of a loop of 110 back-to-back v_mfma_i32_16x16x64_i8 per
// "row group" (the shape of xcorr_mfma_kernel<10,11>'s inner loop), 2 waves per
// SIMD on every CU:
//   0  constant operand registers                          (bare pipe)
//   1  operands with pixel-like int8 data, 10 A x 11 B register fragments
//   2  (1) + the loop's LDS reads: 10 ds_read_b128 + 23 ds_read2_b32 per group
//   3  (2) + the 44 v_alignbyte_b32 funnel shifts
//   4  (3) with all-zero data in LDS (same instruction stream, no toggling)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/measure/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NCA = 10, NCE = 11, NQ = NCA + NCE - 1;

template <int MODE>
__global__ void __launch_bounds__(256, 2) loop(long long* out, const unsigned* seed, int groups) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 1024; i += 256)
    reinterpret_cast<unsigned*>(lds)[i] = MODE == 4 ? 0u : seed[i];
  __syncthreads();
  long long c0 = clock64(), w0 = wall_clock64();
  v4i acc[NQ];
  for (int q = 0; q < NQ; ++q) acc[q] = v4i{0, 0, 0, 0};
  v4i af[NCA], bf[NCE];
  for (int i = 0; i < NCA; ++i)
    af[i] = MODE == 0 ? v4i{0x01010101, 0x7f7f7f7f, 0x12345678, -1}
                      : *reinterpret_cast<const v4i*>(lds + ((lane * 16 + i * 1024) & 0xfff0));
  for (int i = 0; i < NCE; ++i)
    bf[i] = MODE == 0 ? v4i{0x01020304, 0x0f0e0d0c, 0x55aa55aa, 0x11223344}
                      : *reinterpret_cast<const v4i*>(lds + ((lane * 16 + 16384 + i * 1024) & 0xfff0));
  const unsigned char* ap = lds + (lane & 15) * 176 + (lane >> 4) * 176 * 16;
  const unsigned char* bp = lds + 32768 + (lane >> 4) * 208 + ((lane & 15) & ~3);
  const int sh = lane & 3;
  for (int g = 0; g < groups; ++g) {
    if (MODE >= 2) {
      const int row = (g & 31) * 4;
#pragma unroll
      for (int i = 0; i < NCA; ++i)
        af[i] = *reinterpret_cast<const v4i*>(ap + row * 176 + 16 * i);
      unsigned d[4 * NCE + 2];
#pragma unroll
      for (int j = 0; j < 4 * NCE + 2; j += 2) {
        const unsigned long long t = *reinterpret_cast<const unsigned long long*>(
            bp + ((row * 208 + 4 * j) & ~7));
        d[j] = (unsigned)t;
        d[j + 1] = (unsigned)(t >> 32);
      }
      if (MODE >= 3) {
#pragma unroll
        for (int c = 0; c < NCE; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            bf[c][k] = (int)__builtin_amdgcn_alignbyte(d[4 * c + k + 1], d[4 * c + k], sh);
      } else {
#pragma unroll
        for (int c = 0; c < NCE; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k) bf[c][k] = (int)d[4 * c + k];
      }
    }
#pragma unroll
    for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
      for (int c = 0; c < NCE; ++c)
        acc[ca - c + NCE - 1] =
            __builtin_amdgcn_mfma_i32_16x16x64_i8(af[ca], bf[c], acc[ca - c + NCE - 1], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  int s = 0;
  for (int q = 0; q < NQ; ++q) s ^= acc[q][0] ^ acc[q][1] ^ acc[q][2] ^ acc[q][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = s; }
}

int main(int argc, char** argv) {
  const int groups = argc > 1 ? atoi(argv[1]) : 60000;
  long long* d;
  hipMalloc(&d, 64);
  std::vector<unsigned> h(16 * 1024);
  srand(1);
  for (auto& w : h) {
    // four int8 "centred pixels": roughly gaussian, sigma ~ 30
    unsigned v = 0;
    for (int b = 0; b < 4; ++b) {
      int s = 0;
      for (int k = 0; k < 6; ++k) s += rand() % 51 - 25;
      v |= (unsigned)(s & 0xff) << (8 * b);
    }
    w = v;
  }
  unsigned* seed;
  hipMalloc(&seed, h.size() * 4);
  hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  long long r[3];
  auto run = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, d, seed, groups);
      hipMemcpy(r, d, 24, hipMemcpyDeviceToHost);
    }
    const double mfma = 110.0 * groups;
    printf("%-46s clock %7.1f MHz  %6.2f cyc/MFMA/wave  chip %7.1f TOPS (%.3f of 5000)\n", name,
           r[0] * 100.0 / r[1], r[0] / mfma, 512.0 * 4 * mfma * 32768 / (r[1] * 1e-8) / 1e12,
           512.0 * 4 * mfma * 32768 / (r[1] * 1e-8) / 1e12 / 5000);
  };
  run(loop<0>, "0 constant registers");
  run(loop<1>, "1 pixel-like data in registers");
  run(loop<2>, "2 + LDS fragment reads");
  run(loop<3>, "3 + alignbyte funnel shifts");
  run(loop<4>, "4 = 3 with all-zero data");
  run(loop<0>, "0 again");
  return 0;
}
