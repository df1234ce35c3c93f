// Where does the power of the correlation kernel go?  The chip clocks the
// shader to its power budget, so the sustained clock under a synthetic loop is a
// power meter.  A loop of 110 back-to-back v_mfma_i32_16x16x64_i8 per "row
// group" (the shape of xcorr_mfma_kernel<10,11>'s inner loop: 10 A x 11 B
// register fragments into 20 accumulator tiles), operands in registers, no LDS
// or VALU work in the loop, 2 waves per SIMD on every CU; only the operand DATA
// differs between the runs:
//   const      one constant fragment everywhere (no toggling in the multipliers)
//   centred    int8 pixels centred at the mean, sigma ~ 30 (what the kernel feeds)
//   offset+64  the same pixels + 64 (mostly positive values: fewer sign flips)
//   sigma10    low-contrast pixels
//   uniform    uniformly random bytes
//   zeros      all zero
// Rates are computed from the HIP-event time of the whole launch.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/measure/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NCA = 10, NCE = 11, NQ = NCA + NCE - 1;

__global__ void __launch_bounds__(256, 2) loop(long long* out, const unsigned* seed, int groups) {
  const int lane = threadIdx.x & 63;
  long long c0 = clock64(), w0 = wall_clock64();
  v4i acc[NQ];
  for (int q = 0; q < NQ; ++q) acc[q] = v4i{0, 0, 0, 0};
  v4i af[NCA], bf[NCE];
  for (int i = 0; i < NCA; ++i)
    af[i] = *reinterpret_cast<const v4i*>(seed + ((lane * 4 + i * 256) & 0x3ffc));
  for (int i = 0; i < NCE; ++i)
    bf[i] = *reinterpret_cast<const v4i*>(seed + ((lane * 4 + 4096 + i * 256) & 0x3ffc));
  for (int g = 0; g < groups; ++g) {
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
  unsigned* seed;
  hipMalloc(&seed, 16384 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto fill = [&](int kind) {
    std::vector<unsigned> h(16384);
    srand(1);
    for (auto& w : h) {
      unsigned v = 0;
      for (int b = 0; b < 4; ++b) {
        int s = 0;
        if (kind == 0) s = 0x37;
        else if (kind == 4) s = rand() & 0xff;
        else if (kind == 5) s = 0;
        else {
          const int half = kind == 3 ? 8 : 25;   // sum of six uniforms: sigma ~ 1.4 half
          for (int k = 0; k < 6; ++k) s += rand() % (2 * half + 1) - half;
          if (kind == 2) s += 64;
          s = s < -128 ? -128 : s > 127 ? 127 : s;
        }
        v |= (unsigned)(s & 0xff) << (8 * b);
      }
      w = v;
    }
    hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  };
  const char* names[] = {"const", "centred sigma 30", "offset +64", "sigma 10", "uniform bytes", "zeros"};
  long long r[3];
  for (int pass = 0; pass < 2; ++pass)
    for (int kind = 0; kind < 6; ++kind) {
      fill(kind);
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(loop, dim3(512), dim3(256), 0, 0, d, seed, groups);
        hipEventRecord(e1, 0);
        hipMemcpy(r, d, 24, hipMemcpyDeviceToHost);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double mfma = 110.0 * groups;
      const double tops = 512.0 * 4 * mfma * 32768 / (ms * 1e-3) / 1e12;
      printf("%-18s clock %7.1f MHz  %6.2f cyc/MFMA/wave  launch %7.2f ms  chip %7.1f TOPS (%.3f of 5000)\n",
             names[kind], r[0] * 100.0 / r[1], r[0] / mfma, ms, tops, tops / 5000);
    }
  return 0;
}
