// Patch cross-correlation on the int8 matrix cores of gfx950 (CDNA4).
//
// Device replacement for masked_xcorr(use_jax=True) of the reference
// (flow_field.py:36-89, unmasked branch) + the mean subtraction of
// _batched_xcorr (flow_field.py:340-353) for uint8 2-D patches: the reference
// evaluates  out[dy, dx] = sum_{y,x} (A[y+dy, x+dx] - mA) (B[y, x] - mB)
// with zero-padded FFTs; here the sum is evaluated exactly.
//
// 1. Integer core.  With per-patch integer centres cA, cB (chosen so that
//    a' = a - cA and b' = b - cB fit int8) the product sum S = sum a' b' is an
//    exact int32 and
//        out = S - mA' SB - mB' SA + mA' mB' N
//    where mA' = mA - cA (|mA'| <= 0.5 unless clamped), SA / SB are the sums
//    of a' / b' over the overlap rectangle of the shift (box sums from a
//    per-patch integral image) and N the overlap area.
//
// 2. The product sum as a GEMM on v_mfma_i32_16x16x64_i8.  For an output tile
//    of 16 dy values (M) x 16 dx values (N) the contraction index is
//    k = (yb, xa): A-operand[dy, k] = a'[yb + dy, xa] and
//    B-operand[k, dx] = b'[yb, xa - dx].  One MFMA consumes four k-slots
//    (lane group g = lane >> 4 <-> patch row yb0 + g) of 16 consecutive xa
//    each (one "chunk" ca):
//      - the A fragment of lane (m, g) is an aligned 16-byte row segment
//        a'[yb0 + g + dy0 + m, 16 ca ..] read with ds_read_b128 from an LDS
//        copy of the pre patch that is zero padded above and below, so
//        shifts that run off the patch contribute zeros;
//      - the B fragment of lane (n, g) is b'[yb0 + g, 16 (ca - q) + t - n + ..]:
//        a byte-shifted window of row yb0 + g that depends on ca and the
//        output tile q only through c = ca - q.  Each lane reads the 4 NCE + 1
//        aligned dwords covering its windows once per row group and funnel
//        shifts them (v_alignbyte_b32) into NCE fragments, which are then
//        used against every A chunk: NCA x NCE MFMAs per row group into
//        NCA + NCE - 1 accumulator tiles, all from 14 LDS loads + NCE*4 VALU ops.
//    A wave owns one 16-row dy tile at a time and all dx tiles of it; tiles
//    only visit the patch rows that overlap (zero skipping), so the MFMA work
//    is ~1.24x the algorithmic 2 P^4 flop for P = Q = 160.
//
// 3. Epilogue.  For P == Q (the production case) every overlap rectangle is
//    anchored at a patch corner, and with yv = dy mod Py, xv = dx mod Px,
//    ey = -sign(dy), ex = -sign(dx) the whole correction collapses to
//        ey ex G[yv][xv] + ey Rrow[dx>=0][yv] + ex Rcol[dy>=0][xv] + const
//                        + mA' mB' ny nx
//    with ONE combined float table per patch,
//        G[yv][xv] = -mB' IA[yv][xv] - mA' IB[Py - yv][Px - xv]
//    (IA / IB = integral images of a' / b'), and four 1-D arrays, all written
//    by the prep kernel: one coalesced gather per output instead of eight.
//    For P != Q (post_patch_size) the general 8-lookup box-sum form is used.
//
// 4. Schedule.  Workgroups are persistent (two per CU) and take patches from
//    a dynamic queue (the issue arbiter favours the older workgroup of a CU);
//    staging is one global round trip (unaligned 16-byte loads of both
//    patches), the table G is pulled into L2 by LDS-direct loads nothing waits
//    for; the first-peak search runs on what the kernel leaves behind (surface
//    maximum + hot list) in mfma_first_peak_kernel.  One launch carries several
//    reference batches (SfmXcorrDesc.group keeps their coupling apart).
//
// 5. Masked images (MODE raw): the same kernel forms exact integer products of
//    operand planes (value, validity, split square); eight passes + an f64
//    assembly kernel give Padfield's normalised correlation.
//
// LDS per workgroup (P = Q = 160): pre patch (Py + 34) x 176 B + post patch
// (Qy + 3) x 208 B = 68 KB -> two workgroups (8 waves) per CU.  Row pitches
// 176 / 208 keep the b128 / b32 fragment reads bank-conflict free.
//
// Build switches (experiments): SFM_MFMA_TIMING
// (in-kernel phase ticks), SFM_ABLATE_EPILOGUE, SFM_ABLATE_STORE, SFM_NO_TOUCH, SFM_AF_PREFETCH,
// SFM_EPI_QG / SFM_EPI_DEPTH.
#include "sfm_common.h"

#ifndef SFM_EPI_QG
#define SFM_EPI_QG 2
#endif
#ifndef SFM_EPI_DEPTH
#define SFM_EPI_DEPTH 1
#endif
// Ping-pong prefetch of the A fragments across row groups: measured 6 % SLOWER
// on the <10,11> variant (200 bytes of spills at the 256-VGPR limit), neutral on
// the smaller ones; kept as an experiment switch.
#ifndef SFM_WIDE_BPREFETCH_MAX
#define SFM_WIDE_BPREFETCH_MAX 20   // widest variant whose B dwords are double-buffered too
#endif
#ifndef SFM_WIDE_HALVES
#define SFM_WIDE_HALVES 1   // search-window variants: 8 waves per CU, a row tile as two column halves
#endif
#ifndef SFM_WIDE_WAVES
#define SFM_WIDE_WAVES 8    // waves of the search-window variants' one workgroup per CU (8 or 12)
#endif
#ifndef SFM_WIDE_TRIP
#define SFM_WIDE_TRIP 2
#endif
#ifndef SFM_AF_PREFETCH
#define SFM_AF_PREFETCH 0
#endif
// Row-group loop with the A chunk outermost and in-place prefetch (see the
// kernel): the production order.  0 selects the round-1 order (B fragment
// outermost, all LDS fragment loads at the loop head).
#ifndef SFM_LOOP_CA_OUTER
#define SFM_LOOP_CA_OUTER 1
#endif
// The waves of a workgroup take dy tiles from a shared longest-first list (LDS
// counter) instead of a static deal: the four SIMDs do not run at the same
// speed (each shares its matrix pipe with a wave of the other workgroup of the
// CU), and with the static deal ~17 % of a workgroup's time was spent waiting
// for its slowest wave at the end of every patch.
// One LDS instruction per MFMA gap instead of a burst of four behind every
// chunk (see the row-group loop).
#ifndef SFM_LOOP_INTERLEAVE
#define SFM_LOOP_INTERLEAVE 1
#endif
#ifndef SFM_DYNAMIC_TILES
#define SFM_DYNAMIC_TILES 1
#endif

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <cstring>
#include <vector>

namespace sfm {
void mfma_i8_padded_dims(const SfmXcorrDesc* d, int* rows, int* pitch);
}  // namespace sfm

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kPadTop = 16;     // zero rows above the pre patch in LDS
constexpr int kPadBottom = 18;  // zero rows below
constexpr int kMaxTilesPerWave = 24;
// Row-loop variants of the correlation kernel: outer column tiles (each side)
// left out when their bound allows it.  With 20 column tiles (P = 160) and an
// NCC peak of 0.9+ the outer 4 qualify for most patches, the outer 3 for all.
__host__ __device__ constexpr int col_skip_hi(int nq) { return nq / 5; }
__host__ __device__ constexpr int col_skip_lo(int nq) { return 3 * nq / 20; }
__host__ __device__ constexpr int col_skip_2(int nq) { return nq / 4; }       // per-row-tile
__host__ __device__ constexpr int col_skip_3(int nq) { return 3 * nq / 10; }  // variants
// the narrowest row loop: the four central column tiles (entered only in flight, see
// narrow_after in the kernel)
__host__ __device__ constexpr int col_skip_c(int nq) { return (nq - 4) / 2; }
// tbound layout per patch: [0, 30) row tiles; [30], [31] outer col_skip_lo / _hi column
// tiles (any dy); [32 + 3 p + j] row tile p with the outer col_skip_hi / _2 / _3 column
// tiles (2-D bound from 16 x 16 block energies)
constexpr int kBoundStride = 256;   // floats per patch: bounds, then the row-energy prefixes
constexpr int kBoundTiles = 30;    // dy tiles per patch with a pruning bound
constexpr int kBoundCorr = 122;    // tbound slot: bound of the mean-correction terms
constexpr int kBlkRows = 16, kBlkCols = 12;  // 16 x 16 pixel blocks of a patch (<= 256 x 192)
constexpr int kBoundRows = 256;    // patch rows the prep kernel keeps energies for
// tbound[kRowPre + 64 s + k] (uint bits): sum over rows < min(4 k, rows) of side s of
// sum_x (pixel - centre)^2 -- the energies of the int8 operands, exact integers
// (< 2^31 for every patch the matrix path takes); k <= 63, i.e. rows <= kEarlyRows
constexpr int kRowPre = 128;
constexpr int kEarlyRows = 252;

struct PatchParams {  // written by the prep kernel, one per patch
  int y0[2], x0[2];   // clamped patch origin in the image (pre, post)
  int c[2];           // integer centres
  float mu[2];        // mean - centre
  int my0[2], mx0[2]; // masked path: clamped patch origin in the mask arrays
};

// Operand planes of the masked (Padfield) path, all int8:
//   VAL    (pixel - centre) on valid pixels, 0 on masked ones
//   VALID  1 on valid pixels
//   SQHI / SQLO  with s = VAL^2:  s = 128 (SQHI + 64) + SQLO on valid pixels
enum Plane { kPlaneVal = 0, kPlaneValid = 1, kPlaneSqHi = 2, kPlaneSqLo = 3 };

struct MfmaArgs {
  const unsigned char* img[2];
  int ishape[2][2];   // [side][y, x]
  const int* starts[2];
  int P[2], Q[2];     // pre / post patch [y, x]
  int S[2];           // surface [y, x]
  int batch;
  int use_mean;
  float mean;
  PatchParams* pp;
  int* integ[2];      // general path: [B, (py+1) * (px+1)] per side
  long long integ_stride[2];
  float* gtab;        // same-size path: [B, Py * Px] combined table G
  float* aux;         // same-size path: [B, 4 * aux_n + 4] 1-D arrays + consts
  int aux_n;          // max(Py, Px) + 1
  float* surface;     // [B, 16 NP, 16 NQ] (padded to whole tiles)
  // LDS geometry
  int pa, pb;         // row pitches (bytes)
  int ml;             // left margin of the post patch rows (bytes)
  int a_bytes, b_bytes;
  int r_bytes;        // aux arrays / reduction scratch behind the patches
  // static tile schedule: tiles (dy tile indices) per wave
  int tiles[kWaves][kMaxTilesPerWave];  // ints: scalar loads from the kernarg segment
  int n_tiles[kWaves];
  // dynamic tile schedule: all dy tiles, longest first; the waves of a
  // workgroup draw from it through an LDS counter
  int order[kWaves * kMaxTilesPerWave];
  int n_order;
  // fused first-peak search (flow_field.py:238-262); see FusedPeaks
  int do_peaks;
  float threshold_rel;
  int min_distance;
  int cand_cap;
  int* idx1;
  float* v1;
  int* zero_is_peak;
  int* cand_count;
  float* cand_val;
  int* cand_idx;
  unsigned* bitmap;
  int group, bitmap_words;
  int hot_cap;        // per-surface capacity of the hot list
  int* hot_count;     // [B]
  float* hot_val;     // [B, hot_cap]
  int* hot_idx;       // [B, hot_cap] flat index ky * Sx + kx
  int* skipmask;      // [B] bit p: row tile p was pruned, its surface rows are not stored
  int sx_pitch;       // padded surface: row pitch (floats) = 16 * NQ
  // raw-product mode (masked path)
  const unsigned char* mask[2];
  int mshape[2][2];
  int plane[2];       // Plane of the pre / post operand
  int* raw_out;       // [batch, raw_stride] int32 products (rows x sx_pitch used)
  long long raw_stride;
  int* nvalid;        // [batch, 2] un-masked pixels per patch side (prep output)
  // extra masked passes: item i = (patch list[i] >> 3, kMaskedPasses[list[i] & 7]),
  // i < *n_list; its products go to raw_out[i]
  const int* list;
  const int* n_list;
  // a' * b' pass of the masked path: lower bound of the batch maximum of the overlap
  // (masked_classify_kernel); row tiles below 0.3 x it are zeroed by the overlap rule
  const int* ov_lb;   // [groups]
  int ov_group;       // patches per reference batch (maxima are per batch)
  long long s_stride; // padded surface: floats per patch = 16 * NP * sx_pitch
  // dynamic patch queue (NULL: static striding over the workgroups)
  int* work_counter;
  // XCD-sharded form of the queue: the patches are cut into 8 contiguous ranges
  // (partition k = the blocks b == k mod 8 of the prep kernel, which the
  // dispatcher places on XCD k), every XCD has its own head word (128 bytes
  // apart) and a workgroup draws from the range of the XCD it runs on, then
  // steals from the others.  Neighbouring patches overlap by 75 %: their pixels
  // are then fetched once per XCD L2 instead of once per L2.
  int* xcd_heads;
  // shader-clock probe: workgroup 0 leaves (core cycles, 10 ns wall ticks) of its
  // residency here (bench.py reports the sustained clock under this kernel);
  // clk[2]: dy tiles skipped by the pruning (low word) / drawn (high word), whole
  // launch; clk[3]: column tiles left out of the computed dy tiles (low word) / dy tiles
  // abandoned inside their row loop (high word); clk[4]: matrix
  // instructions issued by the row loops (count_tiles)
  long long* clk;
  int prio_mode;      // experiment knob: 0 natural, 1 alternate per tile, 2 static
  // exact pruning of dy tiles (fused-peaks mode): tbound[b][p] bounds |surface|
  // over tile p widened by `guard` rows (prep output; see the tile loop)
  float* tbound;      // [batch, kBoundStride]; the last two entries: outer column tiles
  int prune;
  int touch_all;      // lazy modes: pull the whole correction table into L2, not only the requested tiles' rows (SFM_MFMA_TOUCH_ALL=1)
  int narrow;         // lazy modes: row loops drop provably cold outer column tiles in flight (SFM_MFMA_NARROW=0: off)
  int widen;          // initial store requests: the previous need mask, widened by a tile (SFM_MFMA_WIDEN=1)
  int early;          // lazy modes: abandon provably cold tiles inside the row loop (least distance of two tests, row groups; 0 = off)
  int count_tiles;    // report the pruning counts through clk (timing hooks on)
  int probe;          // seed the running maximum from a probe block (see the kernel)
  int guard, guard_x;
  // Correction table built where it is read (kModeSameExactLazyG): the prep kernel
  // leaves, instead of G, the centred column sums above every 16th patch row --
  // c16[side][I][x] = sum_{y < 16 I} (pixel[y][x] - centre), I <= Py / 16, ints (and
  // the prefixes T of the post patch's centred row sums) -- and a tile that is about
  // to run its epilogue first forms its 16 table rows from them and the int8 patches
  // in LDS, writes them into G (now a per-patch scratch in L2: ten kilobytes that the
  // same lanes read back at once) and runs the ordinary epilogue.
  int lazy_g;
  int* c16;           // [B, c16_stride]: [2][Py / 16 + 1][Px] column sums, then T[Py + 1]
  long long c16_stride;
  int nq;             // column tiles of the kernel variant
  int slot_bytes;     // kModePipe: LDS bytes of one patch slot (multiple of 16)
  int pipe_admit;     // kModePipe: tiles of a patch that may be drawn before its first tile is done
  int prune_k[4];     // outer column tiles (each side) of the row-loop variants (ascending)
};

__device__ __forceinline__ unsigned load_u32_guarded(const unsigned* base,
                                                     long long idx,
                                                     long long n_words) {
  return (idx >= 0 && idx < n_words) ? base[idx] : 0u;
}

constexpr int kXcds = 8;
constexpr int kHeadPitch = 32;   // ints between the per-XCD queue heads

// Range of XCD partition k of n items: as many items as there are indices
// b == k (mod 8) below n, partitions laid out one after the other.
__host__ __device__ __forceinline__ int xcd_part_len(int n, int k) { return (n - k + kXcds - 1) / kXcds; }
__host__ __device__ __forceinline__ int xcd_part_base(int n, int k) {
  int base = 0;
  for (int i = 0; i < k; ++i) base += xcd_part_len(n, i);
  return base;
}

// Patch of prep block `blk`: block b is dispatched to XCD b mod 8 (observed
// placement; only speed depends on it), so it prepares an item of partition b mod 8.
__device__ __forceinline__ int xcd_block_item(const MfmaArgs& a, int blk, int n) {
  if (!a.xcd_heads) return blk;
  const int k = blk % kXcds;
  return xcd_part_base(n, k) + blk / kXcds;
}

// ---------------------------------------------------------------------------
// prep: patch statistics, integer centre, integral image
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) mfma_prep_kernel(MfmaArgs a) {
  if (a.work_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    *a.work_counter = 0;  // the correlation kernel's patch queue
  if (a.clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    a.clk[2] = a.clk[3] = a.clk[4] = 0;
  if (a.xcd_heads && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kXcds)
    a.xcd_heads[kHeadPitch * threadIdx.x] = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int red[3][kThreads];
  const int b = xcd_block_item(a, blockIdx.x, a.batch), s = blockIdx.y;
  const int py = s == 0 ? a.P[0] : a.Q[0];
  const int px = s == 0 ? a.P[1] : a.Q[1];
  const int H = a.ishape[s][0], W = a.ishape[s][1];
  // lax.dynamic_slice start clamping (flow_field.py:320-325).
  const int y0 = min(max(a.starts[s][b * 2 + 0], 0), H - py);
  const int x0 = min(max(a.starts[s][b * 2 + 1], 0), W - px);
  const unsigned char* img = a.img[s];
  int mn = 255, mx = 0, sum = 0;
  for (int i = threadIdx.x; i < py * px; i += kThreads) {
    const int y = i / px, x = i - y * px;
    const int v = img[(long long)(y0 + y) * W + x0 + x];
    smem[i] = static_cast<unsigned char>(v);
    mn = min(mn, v);
    mx = max(mx, v);
    sum += v;
  }
  red[0][threadIdx.x] = mn;
  red[1][threadIdx.x] = mx;
  red[2][threadIdx.x] = sum;
  __syncthreads();
  for (int k = kThreads / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + k]);
      red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + k]);
      red[2][threadIdx.x] += red[2][threadIdx.x + k];
    }
    __syncthreads();
  }
  mn = red[0][0];
  mx = red[1][0];
  sum = red[2][0];
  const float n_f = static_cast<float>(py * px);
  const float mean = a.use_mean ? a.mean : static_cast<float>(sum) / n_f;
  // Centre: nearest integer to the mean that keeps every pixel in int8.
  int c = static_cast<int>(rintf(fminf(fmaxf(mean, 0.f), 255.f)));
  c = min(max(c, mx - 127), mn + 128);
  if (threadIdx.x == 0) {
    PatchParams* p = &a.pp[b];
    p->y0[s] = y0;
    p->x0[s] = x0;
    p->c[s] = c;
    p->mu[s] = a.use_mean
                   ? a.mean - static_cast<float>(c)
                   : static_cast<float>(
                         (static_cast<double>(sum) - static_cast<double>(c) * py * px) /
                         (static_cast<double>(py) * px));
  }
  int* I = a.integ[s] + b * a.integ_stride[s];
  const int ip = px + 1;
  // Column running sums -> I[y + 1][x + 1]; zero first row / column.
  for (int x = threadIdx.x; x <= px; x += kThreads) I[x] = 0;
  for (int y = threadIdx.x; y <= py; y += kThreads) I[y * ip] = 0;
  for (int x = threadIdx.x; x < px; x += kThreads) {
    int run = 0;
    for (int y = 0; y < py; ++y) {
      run += static_cast<int>(smem[y * px + x]) - c;
      I[(y + 1) * ip + x + 1] = run;
    }
  }
  __syncthreads();
  // Row-wise inclusive scan (wave scan with carry).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int y = 1 + wave; y <= py; y += kWaves) {
    int carry = 0;
    for (int x0c = 0; x0c < px; x0c += 64) {
      const int x = x0c + lane;
      int v = x < px ? I[y * ip + x + 1] : 0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
      }
      v += carry;
      if (x < px) I[y * ip + x + 1] = v;
      carry = __shfl(v, 63, 64);
    }
  }
}

// 16 image bytes at an arbitrary byte offset (global memory takes unaligned
// dwordx4 loads); only the last bytes of the image need the guarded path.
__device__ __forceinline__ v4i load_16_bytes(const unsigned char* img, long long off,
                                             long long img_bytes) {
  // ONE unconditional load from an address clamped into the image (a load
  // under a per-lane condition gets its own basic block, and the loads of a
  // staging round would be waited for one by one); at the very end of the image
  // the bytes are shifted into place afterwards and the rest is zero.
  // img_bytes >= 16 (mfma_i8_eligible).
  const long long last = img_bytes - 16;
  const long long o = off > last ? last : off;
  v4i v;
  __builtin_memcpy(&v, img + o, 16);
  const long long sh = off - o;
  if (sh != 0) {
    unsigned __int128 x;
    __builtin_memcpy(&x, &v, 16);
    x = sh >= 16 ? 0 : x >> (8 * static_cast<int>(sh));
    __builtin_memcpy(&v, &x, 16);
  }
  return v;
}

// ---------------------------------------------------------------------------
// prep of the search-window variants (pre patches up to 320 x 320): the same outputs as
// mfma_prep_kernel -- clamped origin, integer centre, mean - centre, the integral image
// I[y + 1][x + 1] = sum_{y' <= y, x' <= x} (pixel - centre) of one patch side -- for patches
// where that kernel's thread-per-column sweep (a byte load per pixel, Py dependent LDS round
// trips per column, a second pass over the table in global memory) took a third of the
// correlation launch's time.  One workgroup of eight waves per (patch, side):
//   1. the patch goes to LDS with 16-byte loads (minimum, maximum and sum on the way);
//   2. wave w owns the rows [w R, (w + 1) R); lane l the columns 8 l .. 8 l + 7 (one aligned
//      8-byte LDS read per row).  A first sweep leaves the column sums of every stripe, so
//      that each wave knows the column sums ABOVE its stripe;
//   3. a second sweep carries the running column sums down the stripe; per row the lane's
//      eight prefix sums plus a wave scan of the lane totals are the integral-image row,
//      stored as two 16-byte pieces per lane (contiguous across the wave).
// ---------------------------------------------------------------------------
constexpr int kWidePrepWaves = 8;
__device__ __forceinline__ int wave_scan_incl(int v);   // (defined with the same-size prep pass)
typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));

__global__ void __launch_bounds__(64 * kWidePrepWaves) mfma_prep_wide_kernel(MfmaArgs a) {
  if (a.work_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    *a.work_counter = 0;  // the correlation kernel's patch queue
  if (a.clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    a.clk[2] = a.clk[3] = a.clk[4] = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int red[3][kWidePrepWaves];
  constexpr int kColsPerLane = 8, kMaxCols = 64 * kColsPerLane;
  __shared__ int stripe[kWidePrepWaves][kMaxCols];
  const int b = blockIdx.x, s = blockIdx.y;
  const int py = s == 0 ? a.P[0] : a.Q[0];
  const int px = s == 0 ? a.P[1] : a.Q[1];
  const int H = a.ishape[s][0], W = a.ishape[s][1];
  // lax.dynamic_slice start clamping (flow_field.py:320-325).
  const int y0 = min(max(a.starts[s][b * 2 + 0], 0), H - py);
  const int x0 = min(max(a.starts[s][b * 2 + 1], 0), W - px);
  const unsigned char* img = a.img[s];
  const long long img_bytes = (long long)H * W;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_chunks = (px + 15) >> 4, pitch = 16 * n_chunks + 16;   // (+16: rows on different banks)
  int mn = 255, mx = 0, sum = 0;
  for (int item = threadIdx.x; item < py * n_chunks; item += 64 * kWidePrepWaves) {
    const int y = item / n_chunks, ch = item - y * n_chunks;
    v4i w = load_16_bytes(img, (long long)(y0 + y) * W + x0 + ch * 16, img_bytes);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned v = static_cast<unsigned>(w[k]);
      const int keep = min(px - (ch * 16 + 4 * k), 4);   // bytes of this dword inside the patch
      if (keep < 4) v = keep <= 0 ? 0u : (v & (0xffffffffu >> (8 * (4 - keep))));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pv = static_cast<int>((v >> (8 * j)) & 255u);
        if (j < keep) {
          mn = min(mn, pv);
          mx = max(mx, pv);
        }
        sum += pv;   // (bytes outside the patch are zero)
      }
      w[k] = static_cast<int>(v);
    }
    *reinterpret_cast<v4i*>(smem + y * pitch + ch * 16) = w;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mn = min(mn, __shfl_xor(mn, d, 64));
    mx = max(mx, __shfl_xor(mx, d, 64));
    sum += __shfl_xor(sum, d, 64);
  }
  if (lane == 0) {
    red[0][wave] = mn;
    red[1][wave] = mx;
    red[2][wave] = sum;
  }
  __syncthreads();
  mn = red[0][0];
  mx = red[1][0];
  sum = red[2][0];
#pragma unroll
  for (int w = 1; w < kWidePrepWaves; ++w) {
    mn = min(mn, red[0][w]);
    mx = max(mx, red[1][w]);
    sum += red[2][w];
  }
  const float n_f = static_cast<float>(py * px);
  const float mean = a.use_mean ? a.mean : static_cast<float>(sum) / n_f;
  // Centre: nearest integer to the mean that keeps every pixel in int8.
  int c = static_cast<int>(rintf(fminf(fmaxf(mean, 0.f), 255.f)));
  c = min(max(c, mx - 127), mn + 128);
  if (threadIdx.x == 0) {
    PatchParams* p = &a.pp[b];
    p->y0[s] = y0;
    p->x0[s] = x0;
    p->c[s] = c;
    p->mu[s] = a.use_mean
                   ? a.mean - static_cast<float>(c)
                   : static_cast<float>(
                         (static_cast<double>(sum) - static_cast<double>(c) * py * px) /
                         (static_cast<double>(py) * px));
  }
  int* I = a.integ[s] + b * a.integ_stride[s];
  const int ip = px + 1;
  const int R = (py + kWidePrepWaves - 1) / kWidePrepWaves;
  const int r0 = wave * R, r1 = min(py, r0 + R);
  const int xl = kColsPerLane * lane;
  const bool act = xl < px;
  // (columns past the patch hold zero bytes: they must not pick up -c)
  int cm[kColsPerLane];
#pragma unroll
  for (int k = 0; k < kColsPerLane; ++k) cm[k] = xl + k < px ? c : 0;
  auto row_pixels = [&](int y, int* pv) {
    const uint2 d = *reinterpret_cast<const uint2*>(smem + y * pitch + (act ? xl : 0));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pv[k] = static_cast<int>((d.x >> (8 * k)) & 255u) - cm[k];
      pv[4 + k] = static_cast<int>((d.y >> (8 * k)) & 255u) - cm[4 + k];
    }
  };
  int col[kColsPerLane];
#pragma unroll
  for (int k = 0; k < kColsPerLane; ++k) col[k] = 0;
  for (int y = r0; y < r1; ++y) {
    int pv[kColsPerLane];
    row_pixels(y, pv);
#pragma unroll
    for (int k = 0; k < kColsPerLane; ++k) col[k] += pv[k];
  }
#pragma unroll
  for (int k = 0; k < kColsPerLane; ++k) stripe[wave][xl + k] = act ? col[k] : 0;
  // zero first row / column of the table
  for (int x = threadIdx.x; x <= px; x += 64 * kWidePrepWaves) I[x] = 0;
  for (int y = threadIdx.x; y <= py; y += 64 * kWidePrepWaves) I[y * ip] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kColsPerLane; ++k) col[k] = 0;
  for (int w = 0; w < wave; ++w)
#pragma unroll
    for (int k = 0; k < kColsPerLane; ++k) col[k] += stripe[w][xl + k];
  for (int y = r0; y < r1; ++y) {
    int pv[kColsPerLane];
    row_pixels(y, pv);
    int run = 0, pre[kColsPerLane];
#pragma unroll
    for (int k = 0; k < kColsPerLane; ++k) {
      col[k] += pv[k];
      run += act ? col[k] : 0;
      pre[k] = run;
    }
    const int before = wave_scan_incl(run) - run;   // columns left of this lane
    if (act) {
      int* dst = I + (y + 1) * ip + xl + 1;
      const v4i_a4 lo = {pre[0] + before, pre[1] + before, pre[2] + before, pre[3] + before};
      const v4i_a4 hi = {pre[4] + before, pre[5] + before, pre[6] + before, pre[7] + before};
      if (xl + 8 <= px) {
        *reinterpret_cast<v4i_a4*>(dst) = lo;
        *reinterpret_cast<v4i_a4*>(dst + 4) = hi;
      } else {
#pragma unroll
        for (int k = 0; k < kColsPerLane; ++k)
          if (xl + k < px) dst[k] = (k < 4 ? lo[k] : hi[k - 4]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// prep for P == Q: centres, means, combined correction table G and the 1-D
// row / column arrays.  One 256-thread block per patch; LDS holds the two raw
// uint8 patches (2 Py Px bytes), so several blocks share a CU.
//
// Wave w sweeps the rows yv = [w R, (w + 1) R) of the table (R = ceil(Py / 4))
// keeping, per lane, the running column sums of the pre patch above row yv and
// of the post patch above row Py - yv; a wave scan over x turns them into the
// integral-image rows IA[yv][.] and IB[Py - yv][.]; raw pixel sums are used and
// the centre is folded in afterwards (I'[y][x] = Iraw[y][x] - c y x).
// ---------------------------------------------------------------------------
constexpr int kPrepCols = 3;  // columns per lane: Px <= 192

// One entry of the correction table, G = -mB' IA'[yv][xv] - mA' IB'[Py - yv][Px - xv],
// from the two (exact, integer-valued) integral-image entries: ONE expression for
// the prep kernel's table and for the table rows built in the correlation kernel's
// epilogue (a product, then a fused multiply-add -- pinned, so that the compiler
// cannot contract the two sites differently: the surfaces must agree bit for bit).
__device__ __forceinline__ float g_entry(float mua, float mub, float ia, float ib) {
  return __fmaf_rn(-mua, ib, __fmul_rn(-mub, ia));
}

// v_pk_min_u16 / v_pk_max_u16 on two 16-bit fields of a dword
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
  us2 x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  const us2 z = __builtin_elementwise_min(x, y);
  unsigned r;
  __builtin_memcpy(&r, &z, 4);
  return r;
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  us2 x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  const us2 z = __builtin_elementwise_max(x, y);
  unsigned r;
  __builtin_memcpy(&r, &z, 4);
  return r;
}

// Wave-wide inclusive add scan on the DPP network (row shifts inside each
// 16-lane row, then row broadcasts), ~12 VALU ops instead of 6 LDS permutes.
__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}

// LDS tables of the prep pass (behind the two raw uint8 patches).
// (LAZY: the band sums live in the dynamic area, behind the packed words they come from)
template <int WAVES, int ROWS, bool LAZY>
struct PrepTables {
  int s_c[2];
  float s_mu[2];
  alignas(16) int band_tot[2][LAZY ? 1 : WAVES + 1][LAZY ? 4 : 64 * kPrepCols];
  // pruning bounds: per-row sum and sum of squares of the raw pixels, later the
  // prefix sums of the row energies (doubles)
  // (sum in the low, sum of squares in the high word: one 64-bit LDS atomic per item)
  unsigned long long row_acc[2][ROWS];
  double row_pre[2][ROWS + 1];
  int col_sq[2][64 * kPrepCols];  // column sums of squares (post: mirrored)
  int wred[2][3][WAVES];          // min / max / sum per wave (LAZY)
  // sums / sums of squares per 16 x 16 block, later their 2-D prefix sums (in place)
  unsigned long long blk_acc[2][kBlkRows][kBlkCols];  // packed like row_acc
};

// The prep pass of patch `b` by a workgroup of WAVES waves: `smem` holds the two
// raw patches (2 Py Px bytes, 16-byte aligned halves), `tp` the tables.  Used by
// the stand-alone kernel (8 waves, one block per patch).  Round 3 also ran it with
// 4 waves at the head of every patch INSIDE the correlation kernel (its LDS patch
// area as scratch, outputs written to global memory and read back): bit-identical,
// but the correlation kernel grew from 15.6 to 18.7 ms per pair while the 2.6 ms
// launch went away -- 19.25 vs 18.9 ms per pair.  The two workgroups of a CU do
// not stay in anti-phase, so the extra VALU / LDS phase of a patch is not hidden
// behind the other workgroup's matrix loop; it simply adds.  Removed again.
// LAZY (MfmaArgs::lazy_g, its own instantiation): the pass never needs a pixel
// twice, so nothing is staged in LDS -- see the fused loop below.
template <int WAVES, int ROWS, bool LAZY>
__device__ __forceinline__ void prep_same_body(const MfmaArgs& a, const int b,
                                               unsigned char* smem,
                                               PrepTables<WAVES, ROWS, LAZY>* tp) {
  constexpr int kPrepWaves = WAVES;
  constexpr int kPrepThreads = 64 * WAVES;
  int (&s_c)[2] = tp->s_c;
  float (&s_mu)[2] = tp->s_mu;
  auto& band_tot = tp->band_tot;
  // reduction scratch of phase 2, aliased onto band_tot (separated by a barrier)
  static_assert(LAZY || sizeof(tp->band_tot) >= sizeof(int) * 2 * 3 * kPrepThreads, "alias");
  // LAZY: band sums [2][Py / 16][Px] over the row words of the fused loop (same size),
  // written once those are folded into the row table
  int* const lazy_bsum = reinterpret_cast<int*>(smem);
  // (LAZY: band_tot holds the band sums by then; the scratch is its own small array)
  constexpr int kRedPitch = LAZY ? WAVES : kPrepThreads;
  int* const redp = LAZY ? &tp->wred[0][0][0] : &band_tot[0][0][0];
#define SFM_RED(s, i, w) redp[((s) * 3 + (i)) * kRedPitch + (w)]
  unsigned long long (&row_acc)[2][ROWS] = tp->row_acc;
  double (&row_pre)[2][ROWS + 1] = tp->row_pre;
  int (&col_sq)[2][64 * kPrepCols] = tp->col_sq;
  unsigned long long (&blk_acc)[2][kBlkRows][kBlkCols] = tp->blk_acc;
  const int py = a.P[0], px = a.P[1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* pix[2] = {smem, smem + ((py * px + 15) & ~15)};
  if (a.prune) {
    for (int i = threadIdx.x; i < 2 * ROWS; i += kPrepThreads) {
      (&row_acc[0][0])[i] = 0;
    }
    for (int i = threadIdx.x; i < 2 * 64 * kPrepCols; i += kPrepThreads) (&col_sq[0][0])[i] = 0;
    for (int i = threadIdx.x; i < 2 * kBlkRows * kBlkCols; i += kPrepThreads) {
      (&blk_acc[0][0][0])[i] = 0;
    }
    __syncthreads();
  }
#ifdef SFM_MFMA_TIMING
  long long pt[6]; pt[0] = clock64();
#define PTICK(i) pt[i] = clock64();
#else
#define PTICK(i)
#endif

  // Phase 1 + 2: copy both patches into LDS (unaligned 16-byte loads, all of a
  // round in flight together) and take min / max / sum of the pixels on the
  // way, from the registers.
  int y0[2], x0[2];
  int mn[2] = {255, 255}, mx[2] = {0, 0}, sum[2] = {0, 0};
  const int n_chunks = (px + 15) / 16;
  const int n_items = py * n_chunks;
  long long img_bytes[2];
  for (int s = 0; s < 2; ++s) {
    const int H = a.ishape[s][0], W = a.ishape[s][1];
    y0[s] = min(max(a.starts[s][b * 2 + 0], 0), H - py);
    x0[s] = min(max(a.starts[s][b * 2 + 1], 0), W - px);
    img_bytes[s] = (long long)H * W;
  }
#ifndef SFM_PREP_ITEMS
#define SFM_PREP_ITEMS 4
#endif
  constexpr int kItems = SFM_PREP_ITEMS;  // items per thread and plane whose loads are in flight
  if constexpr (LAZY) {
    // LAZY: everything this pass needs of a pixel is a sum -- per row, per column
    // and band of 16 rows, per 16 x 16 block, minimum / maximum -- so the patches
    // are never staged in LDS.  A thread takes one half block (8 rows x 16 columns
    // of one side: eight 16-byte loads in flight, one global round trip per patch)
    // and leaves its eight row sums and sixteen column sums, each packed with its
    // sum of squares into one word, in two LDS arrays (plain 16-byte stores: the
    // first version added them to the shared tables with 41 LDS atomics per thread,
    // 6- to 10-way address conflicts each, and spent 25 k cycles per patch here);
    // a second pass (a thread per row / per column) folds them into the tables
    // the staged form fills.  2 x Py / 8 x Px / 16 items (400 for 160 x 160).  The
    // same exact integers in the same tables: everything below is shared.  Py, Px
    // multiples of 16 (host: lazy_g).
    const int NB = py >> 4, NC = px >> 4;
    const int per_side = 2 * NB * NC;
    // rowpart[s][y][chunk] = row sum (12 bits) | sum of squares << 12;
    // halfcol[s][half band][x] = column sum over 8 rows (11 bits) | sum of squares << 11
    unsigned* rowpart = reinterpret_cast<unsigned*>(smem);
    unsigned* halfcol = rowpart + 2 * py * NC;
    unsigned mnp[2] = {0x00ff00ffu, 0x00ff00ffu}, mxp[2] = {0u, 0u};
    for (int item = threadIdx.x; item < 2 * per_side; item += kPrepThreads) {
      const int s = item >= per_side ? 1 : 0;
      const int rem = item - s * per_side;
      const int bh = rem / NC, ch = rem - bh * NC;   // half band (8 rows), chunk (16 columns)
      const int yb = 8 * bh;
      const int W = s ? a.ishape[1][1] : a.ishape[0][1];
      const unsigned char* img = s ? a.img[1] : a.img[0];
      const int yy0 = s ? y0[1] : y0[0], xx0 = s ? x0[1] : x0[0];
      const long long ib = s ? img_bytes[1] : img_bytes[0];
      v4i w[8];
#pragma unroll
      for (int r = 0; r < 8; ++r)
        w[r] = load_16_bytes(img, (long long)(yy0 + yb + r) * W + xx0 + ch * 16, ib);
      unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};   // bytes 0 | 2, 1 | 3 as 16-bit fields
      unsigned q[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) q[j] = 0;
      unsigned mn2 = 0x00ff00ffu, mx2 = 0u;
      int tsum = 0;
      unsigned tsq = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int isum = 0, isq = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned v = static_cast<unsigned>(w[r][k]);
          const unsigned e = v & 0x00ff00ffu, o = (v >> 8) & 0x00ff00ffu;
          lo[k] += e;
          hi[k] += o;
          mn2 = pk_min_u16(pk_min_u16(mn2, e), o);
          mx2 = pk_max_u16(pk_max_u16(mx2, e), o);
          isum = static_cast<int>(__builtin_amdgcn_sad_u8(v, 0u, isum));
          isq = static_cast<int>(__builtin_amdgcn_udot4(v, v, isq, false));
          const unsigned b0 = v & 0xffu, b1 = (v >> 8) & 0xffu, b2 = (v >> 16) & 0xffu, b3 = v >> 24;
          q[4 * k + 0] = __umul24(b0, b0) + q[4 * k + 0];
          q[4 * k + 1] = __umul24(b1, b1) + q[4 * k + 1];
          q[4 * k + 2] = __umul24(b2, b2) + q[4 * k + 2];
          q[4 * k + 3] = __umul24(b3, b3) + q[4 * k + 3];
        }
        rowpart[(s * py + yb + r) * NC + ch] =
            static_cast<unsigned>(isum) | (static_cast<unsigned>(isq) << 12);
        tsum += isum;
        tsq += static_cast<unsigned>(isq);
      }
      atomicAdd(&blk_acc[s][bh >> 1][ch],
                (static_cast<unsigned long long>(tsq) << 32) | static_cast<unsigned>(tsum));
      v4i* hd = reinterpret_cast<v4i*>(halfcol + (s * 2 * NB + bh) * px + 16 * ch);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        hd[k] = v4i{static_cast<int>((lo[k] & 0xffffu) | (q[4 * k + 0] << 11)),
                    static_cast<int>((hi[k] & 0xffffu) | (q[4 * k + 1] << 11)),
                    static_cast<int>((lo[k] >> 16) | (q[4 * k + 2] << 11)),
                    static_cast<int>((hi[k] >> 16) | (q[4 * k + 3] << 11))};
      if (s) {
        mnp[1] = pk_min_u16(mnp[1], mn2);
        mxp[1] = pk_max_u16(mxp[1], mx2);
        sum[1] += tsum;
      } else {
        mnp[0] = pk_min_u16(mnp[0], mn2);
        mxp[0] = pk_max_u16(mxp[0], mx2);
        sum[0] += tsum;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      mn[s] = static_cast<int>(min(mnp[s] & 0xffffu, mnp[s] >> 16));
      mx[s] = static_cast<int>(max(mxp[s] & 0xffffu, mxp[s] >> 16));
    }
  } else
  for (int item0 = threadIdx.x; item0 < n_items; item0 += kPrepThreads * kItems) {
    v4i w[2][kItems];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < kItems; ++u) {
        const int item = min(item0 + u * kPrepThreads, n_items - 1);  // extra items: ignored
        const int y = item / n_chunks, ch = item - y * n_chunks;
        const long long off = (long long)(y0[s] + y) * a.ishape[s][1] + x0[s] + ch * 16;
        w[s][u] = load_16_bytes(a.img[s], off, img_bytes[s]);
      }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < kItems; ++u) {
        const int item = item0 + u * kPrepThreads;
        if (item >= n_items) break;
        const int y = item / n_chunks, ch = item - y * n_chunks;
        int isum = 0, isq = 0;  // this item's share of its row's sums
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned v = static_cast<unsigned>(w[s][u][k]);
          const int xb = ch * 16 + k * 4;
          const int keep = min(4, px - xb);  // valid bytes of this dword
          if (keep <= 0) continue;
          unsigned char* dst = pix[s] + y * px + xb;
          if (keep == 4 && ((y * px + xb) & 3) == 0) {
            *reinterpret_cast<unsigned*>(dst) = v;
          } else {
            for (int t = 0; t < keep; ++t) dst[t] = static_cast<unsigned char>(v >> (8 * t));
          }
          const unsigned live = keep == 4 ? 0xffffffffu : (0xffffffffu >> (8 * (4 - keep)));
          isum = static_cast<int>(__builtin_amdgcn_sad_u8(v & live, 0u, isum));
          isq = static_cast<int>(__builtin_amdgcn_udot4(v & live, v & live, isq, false));
          const unsigned lo = v | ~live, hi = v & live;  // dead bytes: 255 for min, 0 for max
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            mn[s] = min(mn[s], static_cast<int>((lo >> (8 * t)) & 0xffu));
            mx[s] = max(mx[s], static_cast<int>((hi >> (8 * t)) & 0xffu));
          }
        }
        sum[s] += isum;
        if (a.prune) {
          const unsigned long long both =
              (static_cast<unsigned long long>(static_cast<unsigned>(isq)) << 32) |
              static_cast<unsigned>(isum);
          atomicAdd(&row_acc[s][y], both);
          atomicAdd(&blk_acc[s][y >> 4][ch], both);
        }
      }
  }
  // wave reduction on the shuffle network, then one LDS exchange across waves
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mn[s] = min(mn[s], __shfl_xor(mn[s], d, 64));
      mx[s] = max(mx[s], __shfl_xor(mx[s], d, 64));
      sum[s] += __shfl_xor(sum[s], d, 64);
    }
    if (lane == 0) {
      SFM_RED(s, 0, wave) = mn[s];
      SFM_RED(s, 1, wave) = mx[s];
      SFM_RED(s, 2, wave) = sum[s];
    }
  }
  __syncthreads();
  PTICK(1)
  if constexpr (LAZY) {
    // second pass of the fused loop: a thread per column folds the half-band words
    // into the band sums and the column's sum of squares, a thread per row the
    // chunk words into the row's sums (the barrier behind the centres orders them)
    const int NB = py >> 4, NC = px >> 4;
    const unsigned* rowpart = reinterpret_cast<const unsigned*>(smem);
    const unsigned* halfcol = rowpart + 2 * py * NC;
    int* bsum = lazy_bsum;
    for (int t = threadIdx.x; t < 2 * py; t += kPrepThreads) {
      const int u = 2 * py - 1 - t;   // (rows from the other end of the workgroup than the columns)
      const unsigned* rp = rowpart + u * NC;
      unsigned wv[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) wv[c] = c < NC ? rp[c] : 0u;
      unsigned sm = 0, sq = 0;
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        sm += wv[c] & 0xfffu;
        sq += wv[c] >> 12;
      }
      (&row_acc[0][0])[(u >= py ? ROWS : 0) + (u >= py ? u - py : u)] =
          (static_cast<unsigned long long>(sq) << 32) | sm;
    }
    __syncthreads();   // the row words are consumed: the band sums take their place
    for (int t = threadIdx.x; t < 2 * px; t += kPrepThreads) {
      const int s = t >= px ? 1 : 0, x = t - s * px;
      const unsigned* hc = halfcol + s * 2 * NB * px + x;
      unsigned wv[24];
#pragma unroll
      for (int h = 0; h < 24; ++h) wv[h] = h < 2 * NB ? hc[h * px] : 0u;   // (all in flight together)
      unsigned sq = 0;
#pragma unroll
      for (int I = 0; I < 12; ++I)
        if (I < NB) {
          bsum[(s * NB + I) * px + x] = static_cast<int>((wv[2 * I] & 0x7ffu) + (wv[2 * I + 1] & 0x7ffu));
          sq += (wv[2 * I] >> 11) + (wv[2 * I + 1] >> 11);
        }
      // (post patch: mirrored index, like the sweep's column energies)
      col_sq[s][s ? px - 1 - x : x] = static_cast<int>(sq);
    }
  }
  // (LAZY: by two lanes of the last wave, which has no share of the second pass)
  const int ctr_lane = static_cast<int>(threadIdx.x) - (LAZY ? kPrepThreads - 64 : 0);
  if (ctr_lane >= 0 && ctr_lane < 2) {
    const int s = ctr_lane;
    int r_mn = 255, r_mx = 0, r_sum = 0;
    for (int w2 = 0; w2 < kPrepWaves; ++w2) {
      r_mn = min(r_mn, SFM_RED(s, 0, w2));
      r_mx = max(r_mx, SFM_RED(s, 1, w2));
      r_sum += SFM_RED(s, 2, w2);
    }
    SFM_RED(s, 0, 0) = r_mn;
    SFM_RED(s, 1, 0) = r_mx;
    SFM_RED(s, 2, 0) = r_sum;
  }
  if (ctr_lane >= 0 && ctr_lane < 2) {
    const int s = ctr_lane;
    const int mn = SFM_RED(s, 0, 0), mx = SFM_RED(s, 1, 0), sum = SFM_RED(s, 2, 0);
    const float mean =
        a.use_mean ? a.mean : static_cast<float>(sum) / static_cast<float>(py * px);
    int c = static_cast<int>(rintf(fminf(fmaxf(mean, 0.f), 255.f)));
    c = min(max(c, mx - 127), mn + 128);
    s_c[s] = c;
    s_mu[s] = a.use_mean
                  ? a.mean - static_cast<float>(c)
                  : static_cast<float>((static_cast<double>(sum) -
                                        static_cast<double>(c) * py * px) /
                                       (static_cast<double>(py) * px));
    PatchParams* p = &a.pp[b];
    p->y0[s] = y0[s];
    p->x0[s] = x0[s];
    p->c[s] = c;
    p->mu[s] = s_mu[s];
  }
#undef SFM_RED
  __syncthreads();  // `red` (aliased with band_tot) fully consumed
  PTICK(2)
  if (a.prune && wave < 2) {
    // Row energies e[y] = sum_x (pixel - mean)^2 of side `wave` (exact integers
    // in, doubles out) and their prefix sums: lane l owns rows 4 l .. 4 l + 3.
    const int s = wave;
    const double mu = static_cast<double>(s_c[s]) + static_cast<double>(s_mu[s]);
    double e[4], tot = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 4 * lane + k;
      e[k] = 0.0;
      if (y < py)
        e[k] = fmax(static_cast<double>(row_acc[s][y] >> 32) -
                        2.0 * mu * static_cast<double>(row_acc[s][y] & 0xffffffffull) + mu * mu * px,
                    0.0);
      tot += e[k];
      e[k] = tot;  // inclusive within the lane
    }
    double inc = tot;  // inclusive scan of the lane totals
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    const double excl = inc - tot;
    if (lane == 0) row_pre[s][0] = 0.0;
    {
      // the same about the integer centre c (what the matrix operands hold), every
      // fourth row: the correlation kernel bounds the rows a tile has not visited yet
      const long long c = s_c[s];
      unsigned tot_c = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = 4 * lane + k;
        if (y < py) {
          const long long sq = static_cast<long long>(row_acc[s][y] >> 32);
          const long long sm = static_cast<long long>(row_acc[s][y] & 0xffffffffull);
          tot_c += static_cast<unsigned>(sq - 2 * c * sm + c * c * px);
        }
      }
      unsigned inc_c = tot_c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc_c, d, 64);
        if (lane >= d) inc_c += o;
      }
      a.tbound[(long long)b * kBoundStride + kRowPre + 64 * s + lane] =
          __uint_as_float(inc_c - tot_c);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 4 * lane + k;
      if (y < py) row_pre[s][y + 1] = excl + e[k];
    }
  }
  if (a.prune && wave == 2) {
    // inclusive 2-D prefix sums of the four block tables, in place (exact
    // integers; one wave: rows first, then columns)
    // (both fields at once: neither prefix overflows its 32 bits)
    if (lane < 2 * kBlkRows) {  // a lane reads its whole row (loads in flight together)
      const int s = lane >> 4, r = lane & 15;
      unsigned long long v[kBlkCols];
#pragma unroll
      for (int c = 0; c < kBlkCols; ++c) v[c] = blk_acc[s][r][c];
#pragma unroll
      for (int c = 1; c < kBlkCols; ++c) v[c] += v[c - 1];
#pragma unroll
      for (int c = 0; c < kBlkCols; ++c) blk_acc[s][r][c] = v[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 32 && (lane & 15) < kBlkCols) {
      const int s = lane >> 4, c = lane & 15;
      unsigned long long v[kBlkRows];
#pragma unroll
      for (int r = 0; r < kBlkRows; ++r) v[r] = blk_acc[s][r][c];
#pragma unroll
      for (int r = 1; r < kBlkRows; ++r) v[r] += v[r - 1];
#pragma unroll
      for (int r = 0; r < kBlkRows; ++r) blk_acc[s][r][c] = v[r];
    }
  }

  const int xl = kPrepCols * lane;
  if (LAZY) {
    // The correction table is built where it is read (kModeSameExactLazyG): of the
    // 20 row tiles of a patch 2.8 ever reach an epilogue, so instead of sweeping
    // Py rows of G (102 KB per patch, 60 % of this kernel's time) the pass leaves
    // the centred column sums above every 16th row, c16[side][I][x] (14 KB), and
    // the four 1-D arrays, which need one wave scan each.  Py, Px multiples of 16,
    // both <= 192 (host).
    const int NB = py >> 4;
    int* bsum = lazy_bsum;   // [2][NB][px]: raw column sums per band of 16 rows
    // (the band sums, the column energies and the block sums were accumulated by
    // the fused loop above; the barriers since then order them)
    int* c16 = a.c16 + b * a.c16_stride;   // [2][NB + 1][px], then T[py + 1]
    if (threadIdx.x < 2 * px) {
      const int s = threadIdx.x >= px ? 1 : 0, x = threadIdx.x - s * px;
      const int c = s_c[s];
      int band_sum[12];
#pragma unroll
      for (int I = 0; I < 12; ++I) band_sum[I] = I < NB ? bsum[(s * NB + I) * px + x] : 0;
      int run = 0;
      int* out = c16 + s * (NB + 1) * px + x;
#pragma unroll
      for (int I = 0; I < 12; ++I) {
        if (I < NB) out[I * px] = run - c * 16 * I;
        run += band_sum[I];
      }
      out[NB * px] = run - c * py;
      bsum[s * NB * px + x] = run;   // the column's raw total (its own entries are consumed)
    }
    __syncthreads();
    PTICK(3)
    const int ca = s_c[0], cb = s_c[1];
    const float mua = s_mu[0], mub = s_mu[1];
    float* aux = a.aux + (long long)b * (4 * a.aux_n + 4);
    float* rrowA = aux;
    float* rrowB = aux + a.aux_n;
    float* rcolA = aux + 2 * a.aux_n;
    float* rcolB = aux + 3 * a.aux_n;
    if (wave == 2 || wave == 3) {
      // wave 2: exclusive prefixes over the centred ROW sums: IA'[y][Px], IB'[y][Px];
      // wave 3: over the centred COLUMN sums: IA'[Py][x], IB'[Py][x]  (lane l owns the
      // entries 3 l .. 3 l + 2, one wave scan per side)
      const bool rows = wave == 2;
      const int len = rows ? py : px, other = rows ? px : py;
      int va[kPrepCols], vb[kPrepCols], sa = 0, sb = 0;
#pragma unroll
      for (int k = 0; k < kPrepCols; ++k) {
        const int i = xl + k;
        int raw_a = 0, raw_b = 0;
        if (i < len) {
          raw_a = rows ? static_cast<int>(row_acc[0][i] & 0xffffffffull) : bsum[i];
          raw_b = rows ? static_cast<int>(row_acc[1][i] & 0xffffffffull) : bsum[NB * px + i];
        }
        va[k] = sa;   // exclusive within the lane
        vb[k] = sb;
        sa += i < len ? raw_a - ca * other : 0;
        sb += i < len ? raw_b - cb * other : 0;
      }
      const int ea = wave_scan_incl(sa) - sa, eb = wave_scan_incl(sb) - sb;
      float* oa = rows ? rrowA : rcolA;
      float* ob = rows ? rrowB : rcolB;
#pragma unroll
      for (int k = 0; k < kPrepCols; ++k) {
        const int i = xl + k;
        const float fa = static_cast<float>(ea + va[k]), fb = static_cast<float>(eb + vb[k]);
        // T = IB'[y][Px], y <= Py, as integers behind the two column-sum tables
        if (rows && i <= len) c16[2 * (NB + 1) * px + i] = eb + vb[k];
        if (i < len) oa[i] = -mub * fa;             // rrowA[yv] / rcolA[xv]
        if (i >= 1 && i <= len) ob[len - i] = mua * fb;   // rrowB[Py - y] / rcolB[Px - x]
        if (rows && i == len) {
          aux[4 * a.aux_n + 0] = -mub * fa;         // -mB' IA'[Py][Px]
          aux[4 * a.aux_n + 1] = -mua * fb;         // -mA' IB'[Py][Px]
        }
      }
    }
  } else {
    // Phase 3a: per-band column totals (band w = rows [w R, (w + 1) R)).
    // Lane l owns the kPrepCols consecutive columns kPrepCols * l + k, so one
    // wave scan over the per-lane totals gives the prefix along x.
    const int R = (py + kPrepWaves - 1) / kPrepWaves;
    const int ra0 = min(wave * R, py), ra1 = min(ra0 + R, py);
  #pragma unroll
    for (int k = 0; k < kPrepCols; ++k) {
      const int x = xl + k;
      int ta = 0, tb = 0, qa = 0, qb = 0;
      if (x < px) {
        // several rows per trip: the byte loads of a trip are in flight together
  #pragma unroll 5
        for (int y = ra0; y < ra1; ++y) {
          const int va = pix[0][y * px + x];
          const int vb = pix[1][y * px + (px - 1 - x)];  // post patch: mirrored columns
          ta += va;
          tb += vb;
          qa += va * va;
          qb += vb * vb;
        }
      }
      band_tot[0][wave][xl + k] = ta;
      band_tot[1][wave][xl + k] = tb;
      if (a.prune && x < px) {
        atomicAdd(&col_sq[0][x], qa);
        atomicAdd(&col_sq[1][x], qb);
      }
    }
    __syncthreads();
    const int ca = s_c[0], cb = s_c[1];
    const float mua = s_mu[0], mub = s_mu[1];
    float* G = a.gtab + (long long)b * py * px;
    float* aux = a.aux + (long long)b * (4 * a.aux_n + 4);
    float* rrowA = aux;
    float* rrowB = aux + a.aux_n;
    float* rcolA = aux + 2 * a.aux_n;
    float* rcolB = aux + 3 * a.aux_n;

    // Phase 3b: running column sums at the first row of the band.
    //   colA = sum of pre rows  [0, yv)      at yv = ra0
    //   colB = sum of post rows [0, py - yv) at yv = ra0
    int colA[kPrepCols], colB[kPrepCols];
  #pragma unroll
    for (int k = 0; k < kPrepCols; ++k) {
      colA[k] = 0;
      colB[k] = 0;
      for (int w2 = 0; w2 < kPrepWaves; ++w2) {
        const int lo = min(w2 * R, py), hi = min(lo + R, py);
        if (hi <= ra0) colA[k] += band_tot[0][w2][xl + k];
        if (hi <= py - ra0) {
          colB[k] += band_tot[1][w2][xl + k];
        } else if (lo < py - ra0) {
          const int x = xl + k;  // partial band: rows [lo, py - ra0)
          if (x < px)
            for (int y = lo; y < py - ra0; ++y) colB[k] += pix[1][y * px + (px - 1 - x)];
        }
      }
    }
    PTICK(3)
    // Sweep.  The last wave also emits the yv == py row (pre-patch totals).
    // The post patch is scanned in MIRRORED column order (lane l owns columns
    // px - 1 - (3 l + k)): the table needs IB[py - yv][px - xv], and
    //     sum_{x < px - xv} b[.][x]  =  TB - (mirrored prefix up to xv),
    // so the lane that holds IA[yv][xv] also holds the matching IB value and a
    // row of G leaves the registers directly -- no transposition through LDS, no
    // intra-wave fences (the first version spent 1.8 k cycles per row on them).
    int y_end = wave == kPrepWaves - 1 ? py + 1 : ra1;
  #ifdef SFM_ABLATE_SWEEP   // timing experiment only (garbage tables): the prep pass without its sweep
    y_end = ra0;
  #endif
    for (int yv = ra0; yv < y_end; ++yv) {
      const int yw = py - yv;
      // the pixels that move the column sums to row yv + 1: requested now, added
      // at the end of the trip (their LDS latency hides behind the scans)
      int nxt_a[kPrepCols], nxt_b[kPrepCols];
  #pragma unroll
      for (int k = 0; k < kPrepCols; ++k) {
        const int x = xl + k;
        const bool live = yv < py && x < px;
        nxt_a[k] = live ? pix[0][yv * px + x] : 0;
        nxt_b[k] = live ? pix[1][(yw - 1) * px + (px - 1 - x)] : 0;
      }
      int pa[kPrepCols], pb[kPrepCols];
      int sa = 0, sb = 0;
  #pragma unroll
      for (int k = 0; k < kPrepCols; ++k) {
        sa += xl + k < px ? colA[k] : 0;
        sb += xl + k < px ? colB[k] : 0;
        pa[k] = sa;
        pb[k] = sb;
      }
      const int inc_a = wave_scan_incl(sa), inc_b = wave_scan_incl(sb);
      const int ea = inc_a - sa, eb = inc_b - sb;   // exclusive prefixes of the lane totals
      const int ta = __builtin_amdgcn_readlane(inc_a, 63);  // IrawA[yv][px]
      const int tb = __builtin_amdgcn_readlane(inc_b, 63);  // IrawB[py - yv][px]
      // centred integral images: I'[y][x] = Iraw[y][x] - c y x
      // (c y x < 255 * 256 * 192 < 2^24: 24-bit multiplies, full rate; the 32-bit
      // v_mul_lo_u32 is a quarter-rate instruction and a row had twelve of them)
      const int cay = ca * yv, cby = cb * yw;
      auto ia = [&](int raw, int x) { return static_cast<float>(raw - __mul24(cay, x)); };
      auto ib = [&](int raw, int x) { return static_cast<float>(raw - __mul24(cby, x)); };
      const float ia_px = ia(ta, px), ib_px = ib(tb, px);
      if (yv < py) {
  #pragma unroll
        for (int k = 0; k < kPrepCols; ++k) {
          const int xv = xl + k + 1;
          if (xv < px)
            G[yv * px + xv] = g_entry(mua, mub, ia(ea + pa[k], xv), ib(tb - (eb + pb[k]), px - xv));
        }
        if (lane == 0) {
          G[yv * px] = g_entry(mua, mub, ia(0, 0), ib_px);
          rrowA[yv] = -mub * ia_px;
          rrowB[yv] = mua * ib_px;
        }
      }
      if (yv == 0) {
  #pragma unroll
        for (int k = 0; k < kPrepCols; ++k) {
          const int xv = xl + k + 1;
          if (xv < px) rcolB[xv] = mua * ib(tb - (eb + pb[k]), px - xv);
        }
        if (lane == 0) {
          rcolB[0] = mua * ib_px;
          aux[4 * a.aux_n + 1] = -mua * ib_px;
        }
      }
      if (yv == py) {
  #pragma unroll
        for (int k = 0; k < kPrepCols; ++k) {
          const int xv = xl + k + 1;
          if (xv < px) rcolA[xv] = -mub * ia(ea + pa[k], xv);
        }
        if (lane == 0) {
          rcolA[0] = -mub * ia(0, 0);
          aux[4 * a.aux_n + 0] = -mub * ia_px;
        }
      }
      // advance to row yv + 1
  #pragma unroll
      for (int k = 0; k < kPrepCols; ++k) {
        colA[k] += nxt_a[k];
        colB[k] -= nxt_b[k];
      }
    }
  }
  if (a.prune && threadIdx.x < kBoundTiles) {
    // |surface[dy][dx]| = |sum over the overlap of (a - mean_a)(b - mean_b)|
    //   <= sqrt(E_A(rows of the overlap) E_B(rows of the overlap))     (Cauchy-Schwarz,
    // the column range only shrinks the sums), and the row sets are nested in
    // |dy|: the bound of a tile widened by `guard` rows is the one of its row
    // nearest the centre.  (row_pre was finished before the barrier of phase 3a.)
    const int p = threadIdx.x;
    const int lo = 16 * p - (py - 1), hi = min(lo + 15, py - 1);
    float bound = INFINITY;  // tiles that do not exist are never asked for
    if (lo <= py - 1) {
      int d = 0;
      if (lo > 0) d = max(0, lo - a.guard);
      if (hi < 0) d = min(0, hi + a.guard);
      // out[dy] = sum_yb a[yb + dy] b[yb]:  dy >= 0: a rows [dy, py), b rows [0, py - dy)
      const double ea = d >= 0 ? row_pre[0][py] - row_pre[0][d] : row_pre[0][py + d];
      const double eb = d >= 0 ? row_pre[1][py - d] : row_pre[1][py] - row_pre[1][-d];
      // margins: float rounding of the correction terms in the kernel (< 1 abs)
      bound = static_cast<float>(sqrt(fmax(ea, 0.0) * fmax(eb, 0.0)) * 1.0005 + 4.0);
    }
    a.tbound[(long long)b * kBoundStride + p] = bound;
  }
  if (a.prune && wave == 1) {
    // The same along x for the outermost prune_k[j] column tiles on either side
    // (the correlation kernel has row-loop variants without them): column
    // energies, lane l owns columns 3 l .. 3 l + 2 (post patch: mirrored).
    const double mua_d = static_cast<double>(s_c[0]) + static_cast<double>(s_mu[0]);
    const double mub_d = static_cast<double>(s_c[1]) + static_cast<double>(s_mu[1]);
    double ea[kPrepCols], eb[kPrepCols];
#pragma unroll
    for (int k = 0; k < kPrepCols; ++k) {
      const int x = xl + k;
      int sa = 0, sb = 0;
      if (LAZY) {   // raw column totals, natural order (post patch: mirrored here)
        const int* tot = lazy_bsum;
        sa = x < px ? tot[x] : 0;
        sb = x < px ? tot[(py >> 4) * px + (px - 1 - x)] : 0;
      } else {
        for (int w2 = 0; w2 < kPrepWaves; ++w2) {
          sa += band_tot[0][w2][x];
          sb += band_tot[1][w2][x];
        }
      }
      ea[k] = x < px ? fmax(col_sq[0][x] - 2.0 * mua_d * sa + mua_d * mua_d * py, 0.0) : 0.0;
      eb[k] = x < px ? fmax(col_sq[1][x] - 2.0 * mub_d * sb + mub_d * mub_d * py, 0.0) : 0.0;
    }
    auto wave_sum = [&](double v) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
      return v;
    };
    // bound of every shift with |dx| >= |d| (d of either sign), all dy:
    //   out[., dx] = sum a[., xa] b[., xa - dx]:  dx >= 0: a cols [dx, px), b cols [0, px - dx)
    auto bound_x = [&](int d) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int k = 0; k < kPrepCols; ++k) {
        const int xa = xl + k, xb = px - 1 - (xl + k);  // actual columns of the two entries
        const bool in_a = d >= 0 ? xa >= d : xa < px + d;
        const bool in_b = d >= 0 ? xb < px - d : xb >= -d;
        if (xa < px && in_a) sa += ea[k];
        if (xa < px && in_b) sb += eb[k];
      }
      return static_cast<float>(sqrt(wave_sum(sa) * wave_sum(sb)) * 1.0005 + 4.0);
    };
    float c1d[4];  // 1-D bounds of the outer prune_k[j] column tiles (wave-uniform)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ks = a.prune_k[j];
      c1d[j] = INFINITY;  // no such variant: never taken
      if (ks > 0 && j < 2) {  // (the two widest variants rely on the 2-D bound alone)
        // nearest-to-centre columns of the outer tiles: kx = 16 ks - 1 (left),
        // kx = 16 (nq - ks) (right); dx = kx - (px - 1); widened by the guard
        const int dl = min(0, 16 * ks - px + a.guard_x);
        const int dr = max(0, 16 * (a.nq - ks) - (px - 1) - a.guard_x);
        c1d[j] = fmaxf(bound_x(dl), bound_x(dr));
      }
    }
    if (lane == 0) {
      a.tbound[(long long)b * kBoundStride + kBoundTiles + 0] = c1d[0];
      a.tbound[(long long)b * kBoundStride + kBoundTiles + 1] = c1d[1];
      // |surface - S| = |mA' SB + mB' SA - mA' mB' N| over any overlap, with
      // |SA| <= sqrt(N sum a'^2) (a' = pixel - centre; exact integer totals from
      // the block table): what the seed probe of the correlation kernel subtracts
      const unsigned long long ta = blk_acc[0][(py - 1) >> 4][(px - 1) >> 4];
      const unsigned long long tb2 = blk_acc[1][(py - 1) >> 4][(px - 1) >> 4];
      const double nn = static_cast<double>(py) * px;
      const double ca_d = s_c[0], cb_d = s_c[1];
      const double e_a = fmax(static_cast<double>(ta >> 32) -
                                  2.0 * ca_d * static_cast<double>(ta & 0xffffffffull) + ca_d * ca_d * nn, 0.0);
      const double e_b = fmax(static_cast<double>(tb2 >> 32) -
                                  2.0 * cb_d * static_cast<double>(tb2 & 0xffffffffull) + cb_d * cb_d * nn, 0.0);
      const double ma = fabs(static_cast<double>(s_mu[0])), mb = fabs(static_cast<double>(s_mu[1]));
      a.tbound[(long long)b * kBoundStride + kBoundCorr] = static_cast<float>(
          (ma * sqrt(nn * e_b) + mb * sqrt(nn * e_a) + ma * mb * nn) * 1.001 + 8.0);
    }
    // 2-D: row tile p (rows of its guard-widened innermost shift) x the outer column
    // tiles of prune_k[1..3], from the block energies of the covering block
    // rectangles (rounded outwards: still an upper bound).  Lane = 2 p + side.
    // energy about the mean of the block rectangle [r0, r1) x [c0, c1) of side s
    auto rect_energy = [&](int s, int r0, int r1, int c0, int c1) {
      auto at = [&](int r, int c) {
        return (r < 0 || c < 0) ? 0ull : blk_acc[s][r][c];
      };
      // inclusion-exclusion, field by field
      const unsigned long long lo_mask = 0xffffffffull;
      const unsigned long long a11 = at(r1 - 1, c1 - 1), a01 = at(r0 - 1, c1 - 1);
      const unsigned long long a10 = at(r1 - 1, c0 - 1), a00 = at(r0 - 1, c0 - 1);
      const double sq = static_cast<double>(a11 >> 32) - static_cast<double>(a01 >> 32) -
                        static_cast<double>(a10 >> 32) + static_cast<double>(a00 >> 32);
      const double sm = static_cast<double>(a11 & lo_mask) - static_cast<double>(a01 & lo_mask) -
                        static_cast<double>(a10 & lo_mask) + static_cast<double>(a00 & lo_mask);
      const double cnt = static_cast<double>(min(16 * r1, py) - 16 * r0) *
                         static_cast<double>(min(16 * c1, px) - 16 * c0);
      const double mu = s == 0 ? mua_d : mub_d;
      return fmax(sq - 2.0 * mu * sm + mu * mu * cnt, 0.0);
    };
    const int p = lane >> 1, side = lane & 1;
    const int lo = 16 * p - (py - 1), hi = min(lo + 15, py - 1);
    int dyy = 0;
    if (lo > 0) dyy = max(0, lo - a.guard);
    if (hi < 0) dyy = min(0, hi + a.guard);
    // out[dy][dx] = sum a[yb + dy][xa] b[yb][xa - dx]
    const int ra0 = dyy >= 0 ? dyy : 0, ra1 = dyy >= 0 ? py : py + dyy;
    const int rb0 = dyy >= 0 ? 0 : -dyy, rb1 = dyy >= 0 ? py - dyy : py;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const int ks = a.prune_k[j];
      float bound = INFINITY;
      if (ks > 0 && lo <= py - 1) {
        const int dxx = side == 0 ? min(0, 16 * ks - px + a.guard_x)
                                  : max(0, 16 * (a.nq - ks) - (px - 1) - a.guard_x);
        const int ca0 = dxx >= 0 ? dxx : 0, ca1 = dxx >= 0 ? px : px + dxx;
        const int cb0 = dxx >= 0 ? 0 : -dxx, cb1 = dxx >= 0 ? px - dxx : px;
        double ea2 = 0.0, eb2 = 0.0;
        if (ca1 > ca0 && ra1 > ra0)
          ea2 = rect_energy(0, ra0 >> 4, (ra1 + 15) >> 4, ca0 >> 4, (ca1 + 15) >> 4);
        if (cb1 > cb0 && rb1 > rb0)
          eb2 = rect_energy(1, rb0 >> 4, (rb1 + 15) >> 4, cb0 >> 4, (cb1 + 15) >> 4);
        bound = static_cast<float>(sqrt(ea2 * eb2) * 1.0005 + 4.0);
      }
      bound = fmaxf(bound, __shfl_xor(bound, 1, 64));  // left and right outer tiles
      bound = fminf(bound, c1d[j]);
      if (side == 0 && p < kBoundTiles)
        a.tbound[(long long)b * kBoundStride + 32 + 3 * p + (j - 1)] = bound;
    }
  }
#ifdef SFM_MFMA_TIMING
  PTICK(4)
  if (b == 100 && lane == 0 && (wave == 0 || wave == WAVES - 1))
    printf("PREP wave %d: load %lld reduce %lld bands %lld sweep %lld\n", wave, pt[1] - pt[0],
           pt[2] - pt[1], pt[3] - pt[2], pt[4] - pt[3]);
#endif
}


constexpr int kPrepWavesAlone = 8;   // bands of rows swept concurrently (stand-alone kernel)
// LAZY: three workgroups per CU (51 KB of LDS each; <= 80 VGPRs at six waves per SIMD)
#ifndef SFM_PREP_LB
#define SFM_PREP_LB 6
#endif
template <bool LAZY>
__global__ void __launch_bounds__(64 * kPrepWavesAlone, LAZY ? SFM_PREP_LB : 1)
mfma_prep_same_kernel(MfmaArgs a) {
  if (a.work_counter && blockIdx.x == 0 && threadIdx.x == 0)
    *a.work_counter = 0;  // the correlation kernel's patch queue
  if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) a.clk[2] = a.clk[3] = a.clk[4] = 0;  // tile counts
  if (a.xcd_heads && blockIdx.x == 0 && threadIdx.x < kXcds)
    a.xcd_heads[kHeadPitch * threadIdx.x] = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ PrepTables<kPrepWavesAlone, kBoundRows, LAZY> tables;
  prep_same_body<kPrepWavesAlone, kBoundRows, LAZY>(a, xcd_block_item(a, blockIdx.x, a.batch),
                                                    smem, &tables);
}

// ---------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------
// Loads the pre and the post patch into LDS as int8 (pixel - centre), 16 bytes
// per work item, from arbitrarily aligned global rows.  The aligned dword
// loads of BOTH patches are issued before anything is consumed (kStageBatch
// items per thread and plane cover a 160 x 160 patch in one round), so a
// patch costs one global-memory round trip.
struct StagePlane {
  const unsigned char* img;
  long long img_bytes;
  int W, y0, x0, py, px, centre;
  unsigned char* dst;
  int pitch, row_off, col_off, n_chunks;
};

constexpr int kStageBatch = 7;

// (first, limit: the range of work items this call stages -- everything by default)
template <int NT = kThreads>
__device__ __forceinline__ void stage_patches(const StagePlane& p0, const StagePlane& p1,
                                              const int tid, const int first = 0,
                                              const int limit = 0x7fffffff) {
  const StagePlane* pl[2] = {&p0, &p1};
  const int n_items0 = p0.py * p0.n_chunks, n_items1 = p1.py * p1.n_chunks;
  const int n_max = min(max(n_items0, n_items1), limit);
  for (int item0 = first + tid; item0 < n_max; item0 += NT * kStageBatch) {
    v4i w[2][kStageBatch];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const StagePlane& p = *pl[s];
      const int n_items = p.py * p.n_chunks;
#pragma unroll
      for (int u = 0; u < kStageBatch; ++u) {
        const int item = min(item0 + u * NT, n_items - 1);  // extra items: ignored
        const int y = item / p.n_chunks, ch = item - y * p.n_chunks;
        const long long off = (long long)(p.y0 + y) * p.W + p.x0 + ch * 16;
        w[s][u] = load_16_bytes(p.img, off, p.img_bytes);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const StagePlane& p = *pl[s];
      const unsigned cc = static_cast<unsigned>(p.centre) * 0x01010101u;
      const int n_items = p.py * p.n_chunks;
#pragma unroll
      for (int u = 0; u < kStageBatch; ++u) {
        const int item = item0 + u * NT;
        if (item >= n_items) break;
        const int y = item / p.n_chunks, ch = item - y * p.n_chunks;
        v4i out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned v = static_cast<unsigned>(w[s][u][k]);
          // bytewise v - centre (mod 256) == int8 value of pixel - centre
          const unsigned H = 0x80808080u;
          v = ((v | H) - (cc & ~H)) ^ ((v ^ ~cc) & H);
          // zero bytes beyond the patch width
          const int xb = ch * 16 + k * 4;
          if (xb + 4 > p.px) {
            const int keep = p.px - xb;  // < 4
            v = keep <= 0 ? 0u : (v & (0xffffffffu >> (8 * (4 - keep))));
          }
          out[k] = static_cast<int>(v);
        }
        *reinterpret_cast<v4i*>(p.dst + (p.row_off + y) * p.pitch + p.col_off + ch * 16) =
            out;
      }
    }
  }
}

// Masked path: stages one operand plane of a patch, 16 pixels per work item,
// from arbitrarily aligned image / mask rows (aligned dword loads + funnel
// shift, several items in flight).
__device__ void stage_plane(const unsigned char* __restrict__ img, long long img_bytes,
                            int W, int y0, int x0,
                            const unsigned char* __restrict__ mask,
                            long long mask_bytes, int MW, int my0, int mx0, int py,
                            int px, int centre, int plane,
                            unsigned char* __restrict__ dst, int pitch, int row_off,
                            int col_off, int n_chunks) {
  const unsigned* iw = reinterpret_cast<const unsigned*>(img);
  const long long n_iw = (img_bytes + 3) >> 2;
  // the mask base pointer may be byte aligned: split into aligned base + shift
  const unsigned long long mbase = reinterpret_cast<unsigned long long>(mask);
  const unsigned* mw = reinterpret_cast<const unsigned*>(mbase & ~3ull);
  const long long m_lead = static_cast<long long>(mbase & 3ull);
  const long long n_mw = (mask_bytes + m_lead + 3) >> 2;
  constexpr int kBatch = 2;
  const int n_items = py * n_chunks;
  for (int item0 = threadIdx.x; item0 < n_items; item0 += kThreads * kBatch) {
    unsigned wi[kBatch][5], wm[kBatch][5], shi[kBatch], shm[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int item = item0 + u * kThreads;
      const int y = item / n_chunks, ch = item - y * n_chunks;
      const long long off = (long long)(y0 + y) * W + x0 + ch * 16;
      const long long moff = (long long)(my0 + y) * MW + mx0 + ch * 16 + m_lead;
      shi[u] = static_cast<unsigned>(off & 3);
      shm[u] = static_cast<unsigned>(moff & 3);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        wi[u][k] = item < n_items ? load_u32_guarded(iw, (off >> 2) + k, n_iw) : 0u;
        wm[u][k] = (mask && item < n_items) ? load_u32_guarded(mw, (moff >> 2) + k, n_mw)
                                            : 0u;
      }
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int item = item0 + u * kThreads;
      if (item >= n_items) break;
      const int y = item / n_chunks, ch = item - y * n_chunks;
      v4i out;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned pv = __builtin_amdgcn_alignbyte(wi[u][k + 1], wi[u][k], shi[u]);
        const unsigned mv = __builtin_amdgcn_alignbyte(wm[u][k + 1], wm[u][k], shm[u]);
        unsigned o = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int x = ch * 16 + k * 4 + t;
          const int ap = static_cast<int>((pv >> (8 * t)) & 0xffu) - centre;
          const bool valid = ((mv >> (8 * t)) & 0xffu) == 0 && x < px;
          const int sq = ap * ap;
          int val = plane == kPlaneVal ? ap
                    : plane == kPlaneValid ? 1
                    : plane == kPlaneSqHi ? (sq >> 7) - 64
                                          : (sq & 127);
          if (!valid) val = 0;
          o |= static_cast<unsigned>(val & 0xff) << (8 * t);
        }
        out[k] = static_cast<int>(o);
      }
      *reinterpret_cast<v4i*>(dst + (row_off + y) * pitch + col_off + ch * 16) = out;
    }
  }
}

// Masked path prep: clamped origins in image and mask, masked mean, centre.
__global__ void __launch_bounds__(kThreads) mfma_prep_masked_kernel(MfmaArgs a) {
  __shared__ int red[4][kThreads];
  const int b = xcd_block_item(a, blockIdx.x, a.batch), s = blockIdx.y;
  const int py = s == 0 ? a.P[0] : a.Q[0];
  const int px = s == 0 ? a.P[1] : a.Q[1];
  const int H = a.ishape[s][0], W = a.ishape[s][1];
  const int sy = a.starts[s][b * 2 + 0], sx = a.starts[s][b * 2 + 1];
  const int y0 = min(max(sy, 0), H - py), x0 = min(max(sx, 0), W - px);
  const unsigned char* mask = a.mask[s];
  const int MH = a.mshape[s][0], MW = a.mshape[s][1];
  const int my0 = mask ? min(max(sy, 0), MH - py) : 0;
  const int mx0 = mask ? min(max(sx, 0), MW - px) : 0;
  int mn = 255, mx = 0, sum = 0, cnt = 0;
  // 16 pixels per work item: one 16-byte load of pixels and one of the mask
  // (byte loads only in the last, partial item of a row)
  const int n_chunks = (px + 15) >> 4;
  const int n_items = py * n_chunks;
  auto tally = [&](const unsigned* pw, const unsigned* mw, int nx) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int v = static_cast<int>((pw[j >> 2] >> (8 * (j & 3))) & 0xffu);
      const bool ok = j < nx && ((mw[j >> 2] >> (8 * (j & 3))) & 0xffu) == 0;
      mn = ok ? min(mn, v) : mn;
      mx = ok ? max(mx, v) : mx;
      sum += ok ? v : 0;
      cnt += ok ? 1 : 0;
    }
  };
  if ((px & 15) == 0) {
    // whole 16-byte items only: four items per round, their eight loads issued
    // as one straight-line group (items past the end repeat the last one and
    // are ignored; without a mask the image stands in and its bytes are dropped)
    const unsigned char* mbase = mask ? mask + (long long)my0 * MW + mx0
                                      : a.img[s] + (long long)y0 * W + x0;
    const long long mpitch = mask ? MW : W;
    const unsigned mkeep = mask ? 0xffffffffu : 0u;
    constexpr int kRound = 4;
    for (int item0 = threadIdx.x; item0 < n_items; item0 += kThreads * kRound) {
      unsigned pw[kRound][4], mw[kRound][4];
#pragma unroll
      for (int u = 0; u < kRound; ++u) {
        const int item = min(item0 + u * kThreads, n_items - 1);
        const int y = item / n_chunks, x = 16 * (item - y * n_chunks);
        __builtin_memcpy(pw[u], a.img[s] + (long long)(y0 + y) * W + x0 + x, 16);
        __builtin_memcpy(mw[u], mbase + y * mpitch + x, 16);
      }
#pragma unroll
      for (int u = 0; u < kRound; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mw[u][k] &= mkeep;
        tally(pw[u], mw[u], item0 + u * kThreads < n_items ? 16 : 0);
      }
    }
  } else {
    for (int item = threadIdx.x; item < n_items; item += kThreads) {
      const int y = item / n_chunks, ch = item - y * n_chunks;
      const int x = 16 * ch, nx = min(16, px - x);
      const unsigned char* ip = a.img[s] + (long long)(y0 + y) * W + x0 + x;
      const unsigned char* mp = mask ? mask + (long long)(my0 + y) * MW + mx0 + x : nullptr;
      unsigned pw[4] = {0, 0, 0, 0}, mw[4] = {0, 0, 0, 0};
      if (nx == 16) {
        __builtin_memcpy(pw, ip, 16);
        if (mp) __builtin_memcpy(mw, mp, 16);
      } else {
        for (int j = 0; j < nx; ++j) {
          pw[j >> 2] |= static_cast<unsigned>(ip[j]) << (8 * (j & 3));
          if (mp) mw[j >> 2] |= static_cast<unsigned>(mp[j]) << (8 * (j & 3));
        }
      }
      tally(pw, mw, nx);
    }
  }
  red[0][threadIdx.x] = mn;
  red[1][threadIdx.x] = mx;
  red[2][threadIdx.x] = sum;
  red[3][threadIdx.x] = cnt;
  __syncthreads();
  for (int k = kThreads / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + k]);
      red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + k]);
      red[2][threadIdx.x] += red[2][threadIdx.x + k];
      red[3][threadIdx.x] += red[3][threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    mn = red[0][0];
    mx = red[1][0];
    sum = red[2][0];
    cnt = red[3][0];
    PatchParams* p = &a.pp[b];
    p->y0[s] = y0;
    p->x0[s] = x0;
    p->my0[s] = my0;
    p->mx0[s] = mx0;
    if (a.nvalid) a.nvalid[b * 2 + s] = cnt;
    int c = 128;
    float mu = 0.f;
    if (cnt > 0) {
      // nanmean over the unmasked pixels (flow_field.py:340-347)
      const float mean =
          a.use_mean ? a.mean : static_cast<float>(sum) / static_cast<float>(cnt);
      c = static_cast<int>(rintf(fminf(fmaxf(mean, 0.f), 255.f)));
      c = min(max(c, mx - 127), mn + 128);
      mu = a.use_mean ? a.mean - static_cast<float>(c)
                      : static_cast<float>((static_cast<double>(sum) -
                                            static_cast<double>(c) * cnt) /
                                           static_cast<double>(cnt));
    }
    p->c[s] = c;
    p->mu[s] = mu;
  }
}

// One element of the Padfield assembly (flow_field.py:113-131).  Inputs are
// exact integers: xc = sum a'b', s_a / s_b = sums of a' / b', sq_a / sq_b =
// sums of a'^2 / b'^2, n = number of pixel pairs, all over the pairs valid on
// both sides at this shift (a', b' = pixel - integer centre).
//
// The reference subtracts the masked patch means first and then removes the
// overlap means again; in exact arithmetic the patch means (and the centres)
// cancel, and multiplying numerator and denominator by n leaves
//     NCC = (n xc - s_a s_b) / sqrt((n sq_a - s_a^2)(n sq_b - s_b^2)),
// three differences of integers below 2^53: exact in double, no division and
// no cancellation error.  `den` (the reference's denominator, = den_n / n) is
// only compared with the batch tolerance, so float precision is enough for it.
struct PadfieldTerms {
  double num_n;  // n * numerator
  float den_n;   // n * denominator
  float den, ov;
};

__device__ __forceinline__ PadfieldTerms padfield_terms(int xc, int s_a, int s_b, int n,
                                                        double sq_a, double sq_b) {
  PadfieldTerms t;
  const double nd = n, sa = s_a, sb = s_b;
  const double pd = fmax(fma(nd, sq_a, -(sa * sa)), 0.0);
  const double cd = fmax(fma(nd, sq_b, -(sb * sb)), 0.0);
  t.num_n = fma(nd, static_cast<double>(xc), -(sa * sb));
  t.den_n = __builtin_amdgcn_sqrtf(static_cast<float>(pd * cd));  // 1 ulp
  t.ov = fmaxf(static_cast<float>(n), 1.1920928955078125e-07f);
  t.den = t.den_n * __builtin_amdgcn_rcpf(t.ov);
  return t;
}

// SQHI / SQLO products back to the sum of squares (exact in double).
__device__ __forceinline__ double square_sum(int hi, int lo, int n) {
  return 128.0 * static_cast<double>(hi) + static_cast<double>(lo) +
         8192.0 * static_cast<double>(n);
}

// ---------------------------------------------------------------------------
// Masked path, fast form.  Of the eight Padfield products only a' * b' always
// needs the matrix cores: a product with the VALID plane of a side WITHOUT
// masked pixels is a box sum of the other operand over the overlap rectangle,
// read from per-patch integral images.  Per patch:
//   class 0  no masked pixel on either side   1 matrix pass, 4 tables
//   class 1  pre side masked, post side clean 4 passes, 3 tables (pre planes)
//   class 2  pre side clean, post side masked 4 passes, 3 tables (post planes)
//   class 3  both sides masked                8 passes
// The assembly runs twice over the integer data (batch maxima first, then the
// normalised surface) instead of spilling numerator, denominator and overlap
// surfaces in between.
// ---------------------------------------------------------------------------
struct MaskedFastArgs {
  const PatchParams* pp;
  const int* nvalid;    // [batch, 2]
  int* cls;             // [batch] class
  int* first;           // [batch] index of the patch's first extra product surface
  int* items;           // extra passes: patch * 8 + index into kMaskedPasses
  int* n_items;
  int* tab;             // [batch, 4, tab_elems] inclusive integral images
  long long tab_elems;  // Py * Px
  const int* raw0;      // [batch, elems]   a' * b'
  const int* rawd;      // [n_items, elems] extra products
  long long elems;      // padded surface elements per patch
  long long raw_stride; // ints between consecutive product surfaces
  int pitch;
  const unsigned char* img[2];
  const unsigned char* mask[2];
  int ishape[2][2], mshape[2][2];
  int P[2], Q[2], S[2];
  int batch;
  int xcd_map;
  int all_passes;       // test switch: every patch takes the eight passes (class 3)
  int dead_rows;        // final phase: skip rows below the overlap threshold (SFM_MASKED_DEADROWS=0: off)
  int row_blocks;       // workgroups per surface in the assembly kernels
  int axis_max;         // class 0 maxima come from masked_axis_max_kernel (P == Q)
  int* sweep0;          // [batch] 1: a class 0 patch whose maxima need the sweep after all
  int group;            // patches per reference batch: maxima / ov_lb are per batch
  int* ov_lb;           // [groups] lower bound of the batch maximum of the overlap
  unsigned int* maxima; // [groups, 2]
  float* out;           // [batch, elems] final normalised surface
  unsigned int* smax;   // [batch] or NULL: per-surface maximum (ordered bits)
  float* blkmax;        // [batch, row_blocks] or NULL: maximum of every block of 32 surface rows
};

// One workgroup: classes and the list of extra passes in patch order.
__global__ void __launch_bounds__(1024) masked_classify_kernel(MaskedFastArgs g) {
  __shared__ int wsum[16];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < g.batch; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    int cls = 0;
    if (b < g.batch)
      cls = g.all_passes ? 3
                         : (g.nvalid[2 * b] < g.P[0] * g.P[1] ? 1 : 0) +
                               (g.nvalid[2 * b + 1] < g.Q[0] * g.Q[1] ? 2 : 0);
    const int k = b < g.batch ? (cls == 0 ? 0 : cls == 3 ? 7 : 3) : 0;
    const int incl = wave_scan_incl(k);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int first = base + incl - k;
    if (b < g.batch) {
      g.cls[b] = cls;
      g.first[b] = first;
      // class 1: VALID_A * {VAL, SQHI, SQLO}_B; class 2: {VAL, SQHI, SQLO}_A * VALID_B
      for (int j = 0; j < k; ++j) {
        const int pass =
            cls == 3 ? 1 + j : cls == 1 ? (j == 0 ? 2 : 5 + j) : (j == 0 ? 1 : 3 + j);
        g.items[first + j] = b * 8 + pass;
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) s_base = base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *g.n_items = s_base;
  // A lower bound of the batch maximum of the overlap, known before any product:
  // at the zero shift a same-size patch pair overlaps in at least
  // nvalid_A + nvalid_B - Py Px pixels (= Py Px for a clean pair).  The raw pass of
  // a' * b' uses it to leave out row tiles that the overlap rule zeroes anyway.
  // One bound per reference batch (`group` patches; zeroed by the host).
  if (g.P[0] == g.Q[0] && g.P[1] == g.Q[1])
    for (int b0 = 0; b0 < g.batch; b0 += 1024) {
      const int b = b0 + threadIdx.x;
      int lb = 0, gid = -1;
      if (b < g.batch) {
        lb = g.nvalid[2 * b] + g.nvalid[2 * b + 1] - g.P[0] * g.P[1];
        gid = b / g.group;
      }
      // one atomic per wave where its 64 patches share a batch (same-address atomics
      // serialise in the L2)
      const int g0 = __builtin_amdgcn_readfirstlane(gid);
      if (__ballot(gid != g0) == 0) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) lb = max(lb, __shfl_xor(lb, d, 64));
        if (lane == 0 && g0 >= 0 && lb > 0) atomicMax(&g.ov_lb[g0], lb);
      } else if (gid >= 0 && lb > 0) {
        atomicMax(&g.ov_lb[gid], lb);
      }
    }
}

// Inclusive integral images T[y][x] = sum over rows <= y, columns <= x of the
// planes a patch's box sums need (masked pixels count as 0):
//   class 0: a', a'^2, b', b'^2;  class 1: a', a'^2, valid_a;  class 2: b', b'^2, valid_b
// Wave w of the workgroup builds table w: a lane owns four adjacent columns
// (one dword of pixels per row), so a row costs one wave scan; the running
// column sums of the row prefixes stay in registers while it walks down.
__global__ void __launch_bounds__(kThreads) masked_tables_kernel(MaskedFastArgs g) {
  const int b = blockIdx.x;
  const int cls = g.cls[b];
  const int lane = threadIdx.x & 63;
  const int t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (cls == 3 || (cls != 0 && t == 3)) return;
  const int s = cls == 0 ? t >> 1 : cls - 1;
  const int kind = cls == 0 ? (t & 1) : t;  // 0 value, 1 square, 2 valid
  const int py = s ? g.Q[0] : g.P[0], px = s ? g.Q[1] : g.P[1];
  const PatchParams pp = g.pp[b];
  const int c = pp.c[s];
  const int W = g.ishape[s][1], MW = g.mshape[s][1];
  const unsigned char* src = g.img[s] + (long long)pp.y0[s] * W + pp.x0[s];
  const unsigned char* msk =
      cls != 0 && g.mask[s] ? g.mask[s] + (long long)pp.my0[s] * MW + pp.mx0[s] : nullptr;
  int* T = g.tab + ((long long)b * 4 + t) * g.tab_elems;
  constexpr int kAhead = 8;
  // Columns 4 lane .. 4 lane + 3; a lane whose dword would cross the end of the
  // row fetches the last whole dword of the row instead (unconditional loads: a
  // load under a per-lane condition ends up in its own block and the loads of
  // a batch would be waited for one by one).
  const int x0 = 4 * lane;
  const int xl = min(x0, max(px - 4, 0));
  const bool vec = (px & 3) == 0 && px >= 4;
  int acc[4] = {0, 0, 0, 0};
  // The loads of a batch of rows must be ONE straight-line group (any branch
  // between them, even a wave-uniform one, gets a wait of its own: measured,
  // eight serial round trips per batch): the mask pointer falls back to the
  // image (its bytes are then ignored) and narrow patches take another loop.
  // (Cutting the rows into bands swept by more waves was measured slower:
  // 192 vs 127 us per 1024 patches.)
  const unsigned char* mbase = msk ? msk : src;
  const long long mpitch = msk ? MW : W;
  const unsigned mkeep = msk ? 0xffffffffu : 0u;
  const bool wide = px >= 4;
  for (int y0 = 0; y0 < py; y0 += kAhead) {
    unsigned pw[kAhead], mw[kAhead];
    if (wide) {
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int yc = min(y0 + u, py - 1);
        __builtin_memcpy(&pw[u], src + (long long)yc * W + xl, 4);
        __builtin_memcpy(&mw[u], mbase + yc * mpitch + xl, 4);
      }
#pragma unroll
      for (int u = 0; u < kAhead; ++u) mw[u] &= mkeep;
    } else {  // patches narrower than a dword: byte by byte
      for (int u = 0; u < kAhead; ++u) {
        const int yc = min(y0 + u, py - 1);
        pw[u] = mw[u] = 0;
        for (int j = 0; j < px; ++j) {
          pw[u] |= static_cast<unsigned>(src[(long long)yc * W + j]) << (8 * j);
          if (msk) mw[u] |= static_cast<unsigned>(msk[(long long)yc * MW + j]) << (8 * j);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int y = y0 + u;
      const bool row_in = y < py;  // (rows beyond the patch repeat the last one: not stored)
      int v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // a re-based dword (row tail) holds column x0 + j in byte j + x0 - xl
        const bool in = x0 + j < px;
        const int sh = in ? 8 * (j + x0 - xl) : 0;
        const int pv = static_cast<int>((pw[u] >> sh) & 0xffu) - c;
        const bool valid = in && ((mw[u] >> sh) & 0xffu) == 0;
        const int val = kind == 2 ? 1 : kind == 1 ? pv * pv : pv;
        v[j] = valid ? val : 0;
      }
      const int p1 = v[0] + v[1], p2 = p1 + v[2], p3 = p2 + v[3];
      const int incl = wave_scan_incl(p3);
      const int excl = incl - p3;
      acc[0] += excl + v[0];
      acc[1] += excl + p1;
      acc[2] += excl + p2;
      acc[3] += incl;
      if (vec) {
        if (row_in && x0 < px)
          *reinterpret_cast<v4i*>(T + y * px + x0) = v4i{acc[0], acc[1], acc[2], acc[3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (row_in && x0 + j < px) T[y * px + x0 + j] = acc[j];
      }
    }
  }
}

// Batch maxima of a CLEAN same-size pair (class 0, P == Q) without visiting its
// 4 P^2 shifts.  den(dy, dx) = sqrt(SSD_A(rect) SSD_B(rect)) in exact arithmetic,
// SSD = sum of squared deviations from the rectangle's own mean, which only grows
// with the rectangle; the overlap rectangles of a shift are (rows of dy) x (columns
// of dx) on both sides, so  den(dy, dx) <= min(den(dy, 0), den(0, dx)).  The float
// value the assembly computes (padfield_terms: < 2^-20 relative error) can therefore
// exceed the largest value found on the two axes only where BOTH axis values are
// within 1e-5 of it -- for textured patches the zero shift alone.  The kernel
// evaluates the axes (row / column totals of the integral images), then every
// shift of the candidate cross product with the assembly's own expression, and
// merges the maximum: the same bits as the sweep over all shifts
// (masked_phase_rows<0, false>), which these patches then skip.  The overlap
// maximum of a clean pair is Py Px (zero shift).  One wave per patch.
constexpr int kAxisMaxLen = 192;   // patch side (matrix path: <= 160 wide, <= 192 tall)
constexpr int kAxisMaxPairs = 2048;

__device__ __forceinline__ float axis_den(const int (*I)[kAxisMaxLen + 1], int P, int other,
                                          int d) {
  // shift d along one axis, zero along the other: side A rows [a0, a1), side B rows
  // [a0 - d, a1 - d); I[t][i] = total of rows (columns) < i of table t
  const int a0 = max(0, d), a1 = min(P, P + d);
  const int s_a = I[0][a1] - I[0][a0];
  const double sq_a = static_cast<double>(I[1][a1] - I[1][a0]);
  const int s_b = I[2][a1 - d] - I[2][a0 - d];
  const double sq_b = static_cast<double>(I[3][a1 - d] - I[3][a0 - d]);
  return padfield_terms(0, s_a, s_b, (a1 - a0) * other, sq_a, sq_b).den;
}

__global__ void __launch_bounds__(64) masked_axis_max_kernel(MaskedFastArgs g) {
  const int b = blockIdx.x;
  if (g.cls[b] != 0) return;
  __shared__ int R[4][kAxisMaxLen + 1], Cc[4][kAxisMaxLen + 1];
  __shared__ short cand[2][2 * kAxisMaxLen];
  __shared__ int n_cand[2];
  const int lane = threadIdx.x;
  const int Py = g.P[0], Px = g.P[1];
  const int* tab = g.tab + (long long)b * 4 * g.tab_elems;
  // totals of the rows / columns before index i (inclusive tables: last column / row)
  for (int i = lane; i < Py; i += 64)
#pragma unroll
    for (int t = 0; t < 4; ++t) R[t][i + 1] = tab[t * g.tab_elems + (long long)i * Px + Px - 1];
  for (int i = lane; i < Px; i += 64)
#pragma unroll
    for (int t = 0; t < 4; ++t) Cc[t][i + 1] = tab[t * g.tab_elems + (long long)(Py - 1) * Px + i];
  if (lane < 4) R[lane][0] = Cc[lane][0] = 0;
  if (lane < 2) n_cand[lane] = 0;
  __syncthreads();
  constexpr int kPer = (2 * kAxisMaxLen + 63) / 64;   // shifts per lane and axis
  float dy_den[kPer], dx_den[kPer];
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int ky = lane + 64 * u, kx = lane + 64 * u;
    dy_den[u] = ky < 2 * Py - 1 ? axis_den(R, Py, Px, ky - (Py - 1)) : -1.f;
    dx_den[u] = kx < 2 * Px - 1 ? axis_den(Cc, Px, Py, kx - (Px - 1)) : -1.f;
    amax = fmaxf(amax, fmaxf(dy_den[u], dx_den[u]));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d, 64));
  // candidate rows / columns: axis value within 1e-5 of the axis maximum
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    if (dy_den[u] >= 0.f && dy_den[u] * 1.00001f >= amax)
      cand[0][atomicAdd(&n_cand[0], 1)] = static_cast<short>(lane + 64 * u - (Py - 1));
    if (dx_den[u] >= 0.f && dx_den[u] * 1.00001f >= amax)
      cand[1][atomicAdd(&n_cand[1], 1)] = static_cast<short>(lane + 64 * u - (Px - 1));
  }
  __syncthreads();
  float mden = amax;
  const int ny_c = n_cand[0], nx_c = n_cand[1];
  // (amax == 0: every denominator is exactly 0.)  A patch with wide flat margins
  // can have thousands of candidate shifts: it takes the ordinary sweep instead.
  const bool sweep = amax > 0.f && ny_c * nx_c > kAxisMaxPairs;
  if (lane == 0) g.sweep0[b] = sweep ? 1 : 0;
  if (sweep) return;
  if (amax > 0.f && ny_c * nx_c > 1) {
    // the shifts off the axes that could reach the maximum: the assembly's expression
    // on the four box sums (inclusive tables: corner (y1 - 1, x1 - 1) etc.)
    for (int i = lane; i < ny_c * nx_c; i += 64) {
      const int dy = cand[0][i / nx_c], dx = cand[1][i % nx_c];
      const int ya0 = max(0, dy), ya1 = min(Py, Py + dy);
      const int xa0 = max(0, dx), xa1 = min(Px, Px + dx);
      int box[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int* T = tab + t * g.tab_elems;
        const int y0 = t < 2 ? ya0 : ya0 - dy, y1 = t < 2 ? ya1 : ya1 - dy;
        const int x0 = t < 2 ? xa0 : xa0 - dx, x1 = t < 2 ? xa1 : xa1 - dx;
        int v = T[(long long)(y1 - 1) * Px + x1 - 1];
        if (y0 > 0) v -= T[(long long)(y0 - 1) * Px + x1 - 1];
        if (x0 > 0) v -= T[(long long)(y1 - 1) * Px + x0 - 1];
        if (y0 > 0 && x0 > 0) v += T[(long long)(y0 - 1) * Px + x0 - 1];
        box[t] = v;
      }
      mden = fmaxf(mden, padfield_terms(0, box[0], box[2], (ya1 - ya0) * (xa1 - xa0),
                                        static_cast<double>(box[1]), static_cast<double>(box[3])).den);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mden = fmaxf(mden, __shfl_xor(mden, d, 64));
  }
  if (lane == 0) {
    unsigned int* mx = g.maxima + 2 * (b / g.group);
    const unsigned db = __float_as_uint(mden);
    const unsigned ob = __float_as_uint(static_cast<float>(Py * Px));
    if (db > __atomic_load_n(&mx[0], __ATOMIC_RELAXED)) atomicMax(&mx[0], db);
    if (ob > __atomic_load_n(&mx[1], __ATOMIC_RELAXED)) atomicMax(&mx[1], ob);
  }
}

// Padfield assembly from the integer data.  FINAL = false: batch maxima of the
// denominator and the overlap only; FINAL = true: the normalised surface
// (flow_field.py:132-156) with the tolerance / overlap threshold of those maxima.
// Grid (row blocks, batch); a wave takes whole surface rows; where box sums
// are used it first differences the integral images over the row range of the
// shift into 1-D arrays in LDS (D[x] = sum over those rows, columns < x), so
// that a box sum is two LDS reads.
constexpr int kAsmRowsPerWave = 8;
constexpr int kAsmMaxCols = 208;

struct PhaseOut {
  float mden, mov, rmax;
};

// The rows of one wave for a patch of class CLS (compile-time: every register
// array below is indexed with constants).  D0 / D1: this wave's LDS arrays,
//   class 0: D0[x] = (a', a'^2) sums, D1[x] = (b', b'^2) sums,
//   class 1: D0[x] = (a', a'^2),      D1[x].x = valid_a count,
//   class 2: D0[x] = (b', b'^2),      D1[x].x = valid_b count
// over the row range of the shift and columns < x.
template <int CLS, bool FINAL>
__device__ __forceinline__ void masked_phase_rows(const MaskedFastArgs& g, int b,
                                                  int row_block, int2* D0, int2* D1,
                                                  float tol, float px_thr,
                                                  PhaseOut* po) {
  constexpr int kTabs = CLS == 0 ? 4 : CLS == 3 ? 0 : 3;
  constexpr int kRaw = CLS == 0 ? 1 : CLS == 3 ? 8 : 4;  // product surfaces read
  constexpr int kTabCols = 3;  // table columns per lane: Px <= 192
  constexpr int kOutCols = 5;  // surface columns per lane: Sx <= 319 (patches <= 160 wide)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Py = g.P[0], Px = g.P[1], Qy = g.Q[0], Qx = g.Q[1];
  const int Sy = g.S[0], Sx = g.S[1];
  const int* raw0 = g.raw0 + (long long)b * g.raw_stride;
  const int* rawd = g.rawd + (long long)g.first[b] * g.raw_stride;
  const int* tab = g.tab + (long long)b * 4 * g.tab_elems;
  // Every global load of a row is issued before the first one is used: the
  // integral-image rows of the NEXT surface row travel while this one is
  // assembled, and the product surfaces of a row are requested together.
  int thi[kTabs ? kTabs : 1][kTabCols], tlo[kTabs ? kTabs : 1][kTabCols];
  const int row_base = (row_block * kWaves + wave) * kAsmRowsPerWave;
  // FINAL: a surface row whose largest possible overlap, ny * Qx, is below the
  // overlap threshold (flow_field.py:151-155) is zero whatever its terms are
  // (overlap <= ny * Qx for every class): nothing is loaded or assembled for it.
  // With a clean patch in the batch the threshold is 0.3 Py Px: 30 % of the rows.
  auto dead_row = [&](int ky) {
    if (!FINAL || !g.dead_rows) return false;
    const int dy = ky - (Qy - 1);
    const int ny = min(Py, Qy + dy) - max(0, dy);
    return static_cast<float>(ny * Qx) < px_thr;
  };
  auto fetch_tables = [&](int ky) {
    if (ky >= Sy || dead_row(ky)) return;
    const int dy = ky - (Qy - 1);
    const int ya0 = max(0, dy), ya1 = min(Py, Qy + dy);
#pragma unroll
    for (int j = 0; j < kTabs; ++j) {
      // side of table j: class 0: A A B B, class 1: A A A, class 2: B B B
      const bool side_b = CLS == 2 || (CLS == 0 && j >= 2);
      const int w = side_b ? Qx : Px;
      const int y0 = side_b ? ya0 - dy : ya0, y1 = side_b ? ya1 - dy : ya1;
      const int* T = tab + j * g.tab_elems;
#pragma unroll
      for (int k = 0; k < kTabCols; ++k) {
        // (clamped addresses, unconditional loads: see masked_tables_kernel)
        const int x = min(lane + 64 * k, w - 1);
        thi[j][k] = T[(y1 - 1) * w + x];  // y1 >= 1 on every valid row
        const int lo = T[max(y0 - 1, 0) * w + x];
        tlo[j][k] = y0 > 0 ? lo : 0;
      }
    }
  };
  // product surfaces: row r of the current surface row, also one row ahead
  int rv[kRaw][kOutCols];
  auto fetch_raw = [&](int ky) {
    if (ky >= Sy || dead_row(ky)) return;
    const long long row = (long long)ky * g.pitch;
#pragma unroll
    for (int c = 0; c < kOutCols; ++c) {
      const int kx = lane + 64 * c;
#pragma unroll
      for (int q = 0; q < kRaw; ++q) {
        if (!FINAL && q == 0) continue;  // the maxima do not involve a' * b'
        const int* src = q == 0 ? raw0 : rawd + (long long)(q - 1) * g.raw_stride;
        rv[q][c] = src[row + min(kx, Sx - 1)];
      }
    }
  };
  if (kTabs) fetch_tables(row_base);
  fetch_raw(row_base);
  for (int i = 0; i < kAsmRowsPerWave; ++i) {
    const int ky = row_base + i;
    const bool row_ok = ky < Sy;
    const int dy = ky - (Qy - 1);
    const int ya0 = max(0, dy), ya1 = min(Py, Qy + dy);
    const int ny = ya1 - ya0;
    const bool dead = row_ok && dead_row(ky);
    if (kTabs) {
      __syncthreads();  // previous row's arrays consumed
      if (row_ok && !dead) {
#pragma unroll
        for (int k = 0; k < kTabCols; ++k) {
          const int x = lane + 64 * k + 1;
          D0[x] = make_int2(thi[0][k] - tlo[0][k], thi[1 % (kTabs ? kTabs : 1)][k] -
                                                       tlo[1 % (kTabs ? kTabs : 1)][k]);
          D1[x] = make_int2(thi[2 % (kTabs ? kTabs : 1)][k] - tlo[2 % (kTabs ? kTabs : 1)][k],
                            kTabs == 4 ? thi[3 % (kTabs ? kTabs : 1)][k] -
                                             tlo[3 % (kTabs ? kTabs : 1)][k]
                                       : 0);
        }
        if (lane == 0) {
          D0[0] = make_int2(0, 0);
          D1[0] = make_int2(0, 0);
        }
      }
      __syncthreads();
      if (i + 1 < kAsmRowsPerWave) fetch_tables(ky + 1);
    }
    if (!row_ok) continue;
    const long long row = (long long)ky * g.pitch;
    if (dead) {  // (FINAL only) the whole row is zero
      if (i + 1 < kAsmRowsPerWave) fetch_raw(ky + 1);
#pragma unroll
      for (int c = 0; c < kOutCols; ++c) {
        const int kx = lane + 64 * c;
        if (kx < Sx) g.out[(long long)b * g.elems + row + kx] = 0.f;
      }
      po->rmax = fmaxf(po->rmax, 0.f);
      continue;
    }
    // this row's products move to `cur`; the next row's loads are issued now
    int cur[kRaw][kOutCols];
#pragma unroll
    for (int c = 0; c < kOutCols; ++c)
#pragma unroll
      for (int q = 0; q < kRaw; ++q) cur[q][c] = (!FINAL && q == 0) ? 0 : rv[q][c];
    if (i + 1 < kAsmRowsPerWave) fetch_raw(ky + 1);
#pragma unroll
    for (int c = 0; c < kOutCols; ++c) {
      const int kx = lane + 64 * c;
      if (kx < Sx) {
        const int dx = kx - (Qx - 1);
        const int xa0 = max(0, dx), xa1 = min(Px, Qx + dx);
        const int xb0 = xa0 - dx, xb1 = xa1 - dx;
        int s_a, s_b, n;
        double sq_a, sq_b;
        if (CLS == 0) {
          const int2 a1 = D0[xa1], a0 = D0[xa0], b1 = D1[xb1], b0 = D1[xb0];
          n = ny * (xa1 - xa0);
          s_a = a1.x - a0.x;
          sq_a = a1.y - a0.y;
          s_b = b1.x - b0.x;
          sq_b = b1.y - b0.y;
        } else if (CLS == 1) {
          const int2 a1 = D0[xa1], a0 = D0[xa0];
          n = D1[xa1].x - D1[xa0].x;
          s_a = a1.x - a0.x;
          sq_a = a1.y - a0.y;
          s_b = cur[1 % kRaw][c];
          sq_b = square_sum(cur[2 % kRaw][c], cur[3 % kRaw][c], n);
        } else if (CLS == 2) {
          const int2 b1 = D0[xb1], b0 = D0[xb0];
          n = D1[xb1].x - D1[xb0].x;
          s_b = b1.x - b0.x;
          sq_b = b1.y - b0.y;
          s_a = cur[1 % kRaw][c];
          sq_a = square_sum(cur[2 % kRaw][c], cur[3 % kRaw][c], n);
        } else {
          s_a = cur[1 % kRaw][c];
          s_b = cur[2 % kRaw][c];
          n = cur[3 % kRaw][c];
          sq_a = square_sum(cur[4 % kRaw][c], cur[5 % kRaw][c], n);
          sq_b = square_sum(cur[6 % kRaw][c], cur[7 % kRaw][c], n);
        }
        const PadfieldTerms t = padfield_terms(cur[0][c], s_a, s_b, n, sq_a, sq_b);
        if (FINAL) {
          float v = t.den > tol ? static_cast<float>(t.num_n) / t.den_n : 0.f;
          v = fminf(fmaxf(v, -1.f), 1.f);
          if (t.ov < px_thr) v = 0.f;
          g.out[(long long)b * g.elems + row + kx] = v;
          po->rmax = fmaxf(po->rmax, v);
        } else {
          po->mden = fmaxf(po->mden, t.den);
          po->mov = fmaxf(po->mov, t.ov);
        }
      }
    }
  }
}

// End of an assembly workgroup: surface maximum (FINAL) or batch maxima.
template <bool FINAL>
__device__ __forceinline__ void phase_finish(const MaskedFastArgs& g, int b, int row_block,
                                             const PhaseOut& po) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float mden = po.mden, mov = po.mov, rmax = po.rmax;
  if (FINAL) {
    // surface maximum for the peak search (monotonic uint image of the float)
    if (g.smax) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, d, 64));
      if (g.blkmax) {
        // and the maximum of this workgroup's 32 rows: the peak sweep skips blocks
        // that hold nothing above its threshold
        __shared__ float bred[kWaves];
        if (lane == 0) bred[wave] = rmax;
        __syncthreads();
        if (threadIdx.x == 0) {
          float m = bred[0];
          for (int w = 1; w < kWaves; ++w) m = fmaxf(m, bred[w]);
          g.blkmax[(long long)b * g.row_blocks + row_block] = m;
        }
      }
      if (lane == 0 && rmax > -INFINITY) {
        const unsigned u = __float_as_uint(rmax);
        const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        if (o > __atomic_load_n(&g.smax[b], __ATOMIC_RELAXED)) atomicMax(&g.smax[b], o);
      }
    }
  } else {
    // one pair of atomics per workgroup, and only when it can raise a maximum
    // (tens of thousands of same-address atomics serialise in the L2)
    __shared__ float red[2][kWaves];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mden = fmaxf(mden, __shfl_xor(mden, d, 64));
      mov = fmaxf(mov, __shfl_xor(mov, d, 64));
    }
    if (lane == 0) {
      red[0][wave] = mden;
      red[1][wave] = mov;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      float m = red[threadIdx.x][0];
      for (int w = 1; w < kWaves; ++w) m = fmaxf(m, red[threadIdx.x][w]);
      const unsigned bits = __float_as_uint(m);
      unsigned int* mx = g.maxima + 2 * (b / g.group) + threadIdx.x;
      if (bits > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, bits);
    }
  }
}

template <bool FINAL>
__global__ void __launch_bounds__(kThreads) masked_phase_kernel(MaskedFastArgs g) {
  __shared__ int2 D[kWaves][2][kAsmMaxCols];
  // Workgroups are dealt to the 8 XCDs round robin; the row blocks of a patch
  // share its integral images, so they are mapped to ONE XCD (one L2): XCD x
  // takes the x-th eighth of the (patch, row block) list.  Grid = 8 * chunk.
  const int chunk = gridDim.x >> 3;
  const int item = g.xcd_map ? (blockIdx.x & 7) * chunk + (blockIdx.x >> 3) : blockIdx.x;
  const int b = item / g.row_blocks;
  if (b >= g.batch) return;
  const int row_block = item - b * g.row_blocks;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cls = g.cls[b];
  float tol = 0.f, px_thr = 0.f;
  if (FINAL) {
    const unsigned int* mx = g.maxima + 2 * (b / g.group);
    tol = 1e3f * 1.1920928955078125e-07f * __uint_as_float(mx[0]);
    px_thr = 0.3f * __uint_as_float(mx[1]);
  }
  PhaseOut po = {0.f, 0.f, -INFINITY};
  if (cls == 3) return;  // masked_phase3_kernel
  if (!FINAL && cls == 0 && g.axis_max && !g.sweep0[b]) return;  // masked_axis_max_kernel
  switch (cls) {
    case 0: masked_phase_rows<0, FINAL>(g, b, row_block, D[wave][0], D[wave][1], tol, px_thr, &po); break;
    case 1: masked_phase_rows<1, FINAL>(g, b, row_block, D[wave][0], D[wave][1], tol, px_thr, &po); break;
    default: masked_phase_rows<2, FINAL>(g, b, row_block, D[wave][0], D[wave][1], tol, px_thr, &po); break;
  }
  phase_finish<FINAL>(g, b, row_block, po);
}

// Class 3 (both sides masked): all eight products come from memory and no
// row structure is needed, so the surface is walked flat, four elements per
// thread with 16-byte loads, the next four in flight while these are assembled.
template <bool FINAL>
__global__ void __launch_bounds__(kThreads) masked_phase3_kernel(MaskedFastArgs g) {
  const int chunk = gridDim.x >> 3;
  const int item = g.xcd_map ? (blockIdx.x & 7) * chunk + (blockIdx.x >> 3) : blockIdx.x;
  const int b = item / g.row_blocks;
  if (b >= g.batch || g.cls[b] != 3) return;
  const int row_block = item - b * g.row_blocks;
  const int Sy = g.S[0], Sx = g.S[1];
  float tol = 0.f, px_thr = 0.f;
  if (FINAL) {
    const unsigned int* mx = g.maxima + 2 * (b / g.group);
    tol = 1e3f * 1.1920928955078125e-07f * __uint_as_float(mx[0]);
    px_thr = 0.3f * __uint_as_float(mx[1]);
  }
  const long long e0 = (long long)row_block * kWaves * kAsmRowsPerWave * g.pitch;
  const int n4 = static_cast<int>(
      min((long long)kWaves * kAsmRowsPerWave * g.pitch, g.elems - e0) >> 2);
  const v4i* src[8];
  src[0] = reinterpret_cast<const v4i*>(g.raw0 + (long long)b * g.raw_stride + e0);
#pragma unroll
  for (int q = 1; q < 8; ++q)
    src[q] = reinterpret_cast<const v4i*>(g.rawd + (long long)(g.first[b] + q - 1) * g.raw_stride + e0);
  float* out = g.out + (long long)b * g.elems + e0;
  PhaseOut po = {0.f, 0.f, -INFINITY};
  v4i nxt[8];
  // FINAL: rows whose largest possible overlap is below the overlap threshold are
  // zero (see masked_phase_rows): their eight product rows are not read
  auto dead_row = [&](int ky) {
    if (!FINAL || !g.dead_rows) return false;
    const int dy = ky - (g.Q[0] - 1);
    const int ny = min(g.P[0], g.Q[0] + dy) - max(0, dy);
    return static_cast<float>(ny * g.Q[1]) < px_thr;
  };
  auto fetch = [&](int f) {
    const int fc = min(f, n4 - 1);
    if (dead_row(static_cast<int>((e0 + 4LL * fc) / g.pitch))) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (!FINAL && q == 0) continue;  // the maxima do not involve a' * b'
      nxt[q] = src[q][fc];
    }
  };
  fetch(threadIdx.x);
  for (int f = threadIdx.x; f < n4; f += kThreads) {
    v4i cur[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cur[q] = (!FINAL && q == 0) ? v4i{0, 0, 0, 0} : nxt[q];
    fetch(f + kThreads);
    const long long e = e0 + 4LL * f;
    const int ky = static_cast<int>(e / g.pitch);
    const int kx0 = static_cast<int>(e - (long long)ky * g.pitch);
    if (dead_row(ky)) {  // (rows past Sy: ny <= 0, dead as well; padding, never read)
      *reinterpret_cast<float4*>(out + 4 * f) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ky < Sy && kx0 < Sx) po.rmax = fmaxf(po.rmax, 0.f);
      continue;
    }
    float v4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = cur[3][j];
      const PadfieldTerms t =
          padfield_terms(cur[0][j], cur[1][j], cur[2][j], n, square_sum(cur[4][j], cur[5][j], n),
                         square_sum(cur[6][j], cur[7][j], n));
      const bool ok = ky < Sy && kx0 + j < Sx;
      if (FINAL) {
        float v = t.den > tol ? static_cast<float>(t.num_n) / t.den_n : 0.f;
        v = fminf(fmaxf(v, -1.f), 1.f);
        if (t.ov < px_thr) v = 0.f;
        v4[j] = v;
        if (ok) po.rmax = fmaxf(po.rmax, v);
      } else if (ok) {
        po.mden = fmaxf(po.mden, t.den);
        po.mov = fmaxf(po.mov, t.ov);
      }
    }
    if (FINAL) *reinterpret_cast<float4*>(out + 4 * f) = make_float4(v4[0], v4[1], v4[2], v4[3]);
  }
  phase_finish<FINAL>(g, b, row_block, po);
}

__device__ __forceinline__ int box_sum(const int* __restrict__ I, int ip, int y0,
                                       int y1, int x0, int x1) {
  // I has a zero first row and column.
  int s = I[y1 * ip + x1];
  if (y0 > 0) s -= I[y0 * ip + x1];
  if (x0 > 0) s -= I[y1 * ip + x0];
  if (y0 > 0 && x0 > 0) s += I[y0 * ip + x0];
  return s;
}

// First-peak search of one finished surface from what the correlation kernel
// left behind: the surface maximum and the hot list (every element that
// exceeded threshold_rel * running maximum when its tile was produced).
// Same result as peaks_first_kernel in sfm_xcorr.hip: peak = element that
// equals its (2 m + 1)^2 zero-padded window maximum and exceeds
// threshold_rel * max(surface) (flow_field.py:238-262).
__device__ __forceinline__ bool peak_better(float v, int i, float bv, int bi) {
  return v > bv || (v == bv && i < bi);
}

// First peak of one surface by ONE WAVE (the kernel below runs four surfaces per
// workgroup, a wave each: 40 401 workgroups of a few hundred hot elements were
// priced by their dispatch and by a chain of dependent round trips per
// workgroup).  The first hot-list entry of every lane is requested together with
// the surface's maximum and fill count, the candidate slots are counted in LDS
// (a returning global atomic per candidate was a round trip of its own), and the
// arg-max runs on the shuffle network: no barrier anywhere.
__device__ void wave_first_peak(const MfmaArgs& a, int b, const float* surf,
                                int Sy, int Sx, int* cand_lds) {
  const int lane = threadIdx.x & 63;
  // (unconditional: entries beyond the fill count are stale and ignored)
  const float hv0 = a.hot_val[(long long)b * a.hot_cap + min(lane, a.hot_cap - 1)];
  const int hi0 = a.hot_idx[(long long)b * a.hot_cap + min(lane, a.hot_cap - 1)];
  const float mx = a.v1[b];          // the surface maximum on entry
  const int n_hot = a.hot_count[b];
  if (lane == 0) *cand_lds = 0;
  // (the reset, the other lanes' atomics and the read-back below are LDS operations of ONE
  // wave, which the hardware runs in program order; the fences keep the compiler from
  // reordering them around the divergent loops in between)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int pitch = a.sx_pitch;
  const int m = a.min_distance;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  // Tests one element that exceeds the threshold and records it if it is a
  // window maximum.
  auto consider = [&](int y, int x, float v) {
    // window maximum from clamped addresses: unconditional loads (all in flight
    // together); a clamped position repeats an element of the window, which
    // cannot change the maximum
    float wm = -INFINITY;
    const bool outside = y - m < 0 || y + m >= Sy || x - m < 0 || x + m >= Sx;
    if (m == 2) {  // the default min_distance: 25 loads in flight instead of a loop of waits
      float w[25];
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx)
          w[(dy + 2) * 5 + dx + 2] = surf[(long long)min(max(y + dy, 0), Sy - 1) * pitch +
                                          min(max(x + dx, 0), Sx - 1)];
#pragma unroll
      for (int k = 0; k < 25; ++k) wm = fmaxf(wm, w[k]);
    } else {
      for (int dy = -m; dy <= m; ++dy) {
        const float* wrow = surf + (long long)min(max(y + dy, 0), Sy - 1) * pitch;
        for (int dx = -m; dx <= m; ++dx) wm = fmaxf(wm, wrow[min(max(x + dx, 0), Sx - 1)]);
      }
    }
    if (outside) wm = fmaxf(wm, 0.f);
    if (v != wm) return;
    const int i = y * Sx + x;
    if (peak_better(v, i, bv, bi)) {
      bv = v;
      bi = i;
    }
    const int slot = atomicAdd(cand_lds, 1);
    if (slot < a.cand_cap) {
      a.cand_val[(long long)b * a.cand_cap + slot] = v;
      a.cand_idx[(long long)b * a.cand_cap + slot] = i;
    }
    if (i == 0) a.zero_is_peak[b] = 1;
  };
  if (mx > 0.f) {
    const float thr = a.threshold_rel * mx;
    if (n_hot <= a.hot_cap) {
      const float* hv = a.hot_val + (long long)b * a.hot_cap;
      const int* hi = a.hot_idx + (long long)b * a.hot_cap;
      for (int e = lane; e < n_hot; e += 64) {
        const float v = e == lane ? hv0 : hv[e];
        if (!(v > thr)) continue;
        const int i = e == lane ? hi0 : hi[e];
        const int y = i / Sx;
        consider(y, i - y * Sx, v);
      }
    } else {
      // Hot list overflowed (flat surfaces): sweep the stored surface.
      const int n4 = (Sx + 3) >> 2;
      const int skipped = a.skipmask ? a.skipmask[b] : 0;  // pruned row tiles: not stored
      for (int y = 0; y < Sy; ++y) {
        if ((skipped >> (y >> 4)) & 1) continue;
        const float* row = surf + (long long)y * pitch;
        for (int c4 = lane; c4 < n4; c4 += 64) {
          const float4 q4 = *reinterpret_cast<const float4*>(row + 4 * c4);
          const float vals[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (vals[t] > thr && 4 * c4 + t < Sx) consider(y, 4 * c4 + t, vals[t]);
        }
      }
    }
  }
  // (value, index) arg-max across the wave, first index wins ties (a total
  // order: any reduction tree finds the same winner).
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float ov = __shfl_xor(bv, d, 64);
    const int oi = __shfl_xor(bi, d, 64);
    if (peak_better(ov, oi, bv, bi)) {
      bv = ov;
      bi = oi;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) {
    a.cand_count[b] = *cand_lds;   // (behind every lane's atomics in this wave's LDS queue)
    const float v1 = bv;
    const int i1 = v1 == -INFINITY ? 0 : bi;  // argmax of an all -inf row is 0
    a.idx1[b] = i1;
    a.v1[b] = v1;
    sfm::set_bit_once(&a.bitmap[(long long)(b / a.group) * a.bitmap_words + (i1 >> 5)],
                       1u << (i1 & 31));
  }
}

// kModeSameExact: P == Q and Px == 16 NCA (production 160, 128, 96, 64, 48): the
// per-column sign / index arithmetic of the epilogue becomes compile-time.
constexpr int kModeGeneral = 0, kModeSame = 1, kModeRaw = 2, kModeSameExact = 3;
// Lazy surface stores (r4), the flow path's form of kModeSame / kModeSameExact.
// The flow path never returns the surface, and the peak kernels read a few
// hundred of its 102 400 values: the 5 x 5 window of every hot element, the
// 11 x 11 sharpness window of the first peak, the hot elements themselves.  A
// row tile is therefore stored only if it may hold a hot element (its maximum
// exceeds threshold_rel x the running maximum, the test that also feeds the hot
// list) or lies within the guard band (max(min_distance, 2 peak_radius) rows) of
// a tile that may.  A tile learns that it is needed from a request mask in LDS
// if the hot neighbour finished first; if it had already finished without
// storing, it is recomputed at the end of the patch (redo mask; rare, because
// tiles are drawn innermost first and the peak sits there).  What is left
// un-stored has the property of a pruned tile -- every element of the tile and
// of its guard band is below threshold_rel x the final maximum -- and goes into
// the same skip mask for the overflow sweeps of the peak kernels.  Values and
// decisions that reach the output are unchanged: bit-identical results.
constexpr int kModeSameLazy = 4, kModeSameExactLazy = 5;
// kModeSameExactLazy with the correction table built in the epilogue of the tiles
// that are stored (MfmaArgs::lazy_g; Py a multiple of 16): no table G in memory.
constexpr int kModeSameExactLazyG = 6;
// kModeSameExactLazyG as a cross-patch pipeline: ONE workgroup of eight waves per CU, TWO
// patch slots in LDS, the tile queue running across the patch boundary (see the kernel).
constexpr int kModePipe = 7;
constexpr int kPipeCtl = 32;  // ints of a slot's control header (behind its lazy-store state;
                              // 256 ints of seed-probe sums follow)

// Inclusive add scan over the 16 lanes of a DPP row (the 16 columns of a tile).
__device__ __forceinline__ int row16_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  return v;
}
// Lane 15 of every 16-lane row to the whole row (ds_swizzle, bit mode: lane' =
// (lane & 0x10) | 0x0f inside each half wave; no LDS memory is touched).
__device__ __forceinline__ int row16_last(int v) {
  return __builtin_amdgcn_ds_swizzle(v, (0x0f << 5) | 0x10);
}
// Lane n <- lane 15 - n of the same row.
__device__ __forceinline__ int row16_mirror(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror
}

// Wave-wide maximum of non-negative values on the DPP network (no LDS round
// trips): row shifts inside each 16-lane row, then row broadcasts; the result
// is read from lane 63 into a scalar.
#define SFM_DPP_MAXF(x, ctrl, rmask, bc)                                            \
  fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl,   \
                                                      rmask, 0xf, bc)))
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = SFM_DPP_MAXF(v, 0x111, 0xf, true);   // row_shr:1
  v = SFM_DPP_MAXF(v, 0x112, 0xf, true);   // row_shr:2
  v = SFM_DPP_MAXF(v, 0x114, 0xf, true);   // row_shr:4
  v = SFM_DPP_MAXF(v, 0x118, 0xf, true);   // row_shr:8
  v = SFM_DPP_MAXF(v, 0x142, 0xa, false);  // row_bcast:15
  v = SFM_DPP_MAXF(v, 0x143, 0xc, false);  // row_bcast:31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The same for non-negative integers (six DPP steps on the VALU; the shuffle
// form of the reduction is six dependent LDS round trips, next to the matrix
// loops' fragment reads).
#define SFM_DPP_MAXI(x, ctrl, rmask, bc) \
  max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, bc))
__device__ __forceinline__ int wave_max_nonneg_i(int v) {
  v = SFM_DPP_MAXI(v, 0x111, 0xf, true);   // row_shr:1
  v = SFM_DPP_MAXI(v, 0x112, 0xf, true);   // row_shr:2
  v = SFM_DPP_MAXI(v, 0x114, 0xf, true);   // row_shr:4
  v = SFM_DPP_MAXI(v, 0x118, 0xf, true);   // row_shr:8
  v = SFM_DPP_MAXI(v, 0x142, 0xa, false);  // row_bcast:15
  v = SFM_DPP_MAXI(v, 0x143, 0xc, false);  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// Bitwise OR over the wave, the same way.
#define SFM_DPP_OR(x, ctrl, rmask, bc) \
  ((x) | __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, bc))
__device__ __forceinline__ int wave_or(int v) {
  v = SFM_DPP_OR(v, 0x111, 0xf, true);   // row_shr:1
  v = SFM_DPP_OR(v, 0x112, 0xf, true);   // row_shr:2
  v = SFM_DPP_OR(v, 0x114, 0xf, true);   // row_shr:4
  v = SFM_DPP_OR(v, 0x118, 0xf, true);   // row_shr:8
  v = SFM_DPP_OR(v, 0x142, 0xa, false);  // row_bcast:15
  v = SFM_DPP_OR(v, 0x143, 0xc, false);  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// base[byte_off]: a 32-bit byte offset on a wave-uniform base selects the
// scalar-base + VGPR-offset addressing mode (no 64-bit VALU address math).
__device__ __forceinline__ float at_byte(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// One WAVE per surface, four surfaces per workgroup; v1[b] holds the surface
// maximum on entry.
__global__ void __launch_bounds__(kThreads) mfma_first_peak_kernel(MfmaArgs a) {
  __shared__ int s_cand[kWaves];
  const int wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kWaves + wave;
  if (b >= a.batch) return;
  wave_first_peak(a, b, a.surface + b * a.s_stride, a.S[0], a.S[1], &s_cand[wave]);
}

// XCD-sharded queue: next item of this workgroup (n_items: none left).  Called
// by all threads; `state` (thread 0 only): bits 0..2 the partition drawn from,
// bits 8..15 the partitions found empty.
__device__ __forceinline__ int pull_item(const MfmaArgs& a, int n_items, int* state,
                                         int* next_lds) {
  __syncthreads();
  if (threadIdx.x == 0) {
    int item = n_items, st = *state;
    for (int tries = 0; tries < kXcds; ++tries) {
      const int x = st & 7;
      if (!((st >> (8 + x)) & 1)) {
        const int p = atomicAdd(&a.xcd_heads[kHeadPitch * x], 1);
        if (p < xcd_part_len(n_items, x)) {
          item = xcd_part_base(n_items, x) + p;
          break;
        }
        st |= 1 << (8 + x);
      }
      st = (st & ~7) | ((x + 1) & 7);   // steal from the next XCD's range
    }
    *state = st;
    *next_lds = item;
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*next_lds);
}

// Next patch of this workgroup; called by all threads at the end of a patch.
__device__ __forceinline__ int next_patch(const MfmaArgs& a, int b, int* next_lds) {
  if (!a.work_counter) return b + gridDim.x;
  __syncthreads();
  if (threadIdx.x == 0) *next_lds = gridDim.x + atomicAdd(a.work_counter, 1);
  __syncthreads();
  // wave-uniform: keeps every per-patch base address in scalar registers
  return __builtin_amdgcn_readfirstlane(*next_lds);
}

template <int NCA, int NCE, int MODE>
__global__ void __launch_bounds__(MODE == kModePipe ? 2 * kThreads
                                  : (NCA > 10 && SFM_WIDE_HALVES) ? 64 * SFM_WIDE_WAVES : kThreads,
                                  (MODE == kModePipe || NCA > 10) ? 1 : 2) xcorr_mfma_kernel(MfmaArgs a) {
  // Search-window geometry (NCA > 10: pre patches 161 .. 320 wide against a post patch of up
  // to 160, processor/flow.py:577,792-803).  The pre patch alone is up to 119 KB of LDS, so a
  // CU holds ONE workgroup = one wave per SIMD -- which then owns the SIMD's whole register
  // file: 512 registers per lane, the 25 .. 30 accumulator tiles of a row tile in the upper
  // half (the compiler places them in AGPRs), the 15 .. 20 A fragments double-buffered in the
  // lower one (the fragments of the next row group are in flight while the 165 .. 220 matrix
  // instructions of this one issue).  kModeGeneral semantics; no pruning.  (That is the FIRST
  // form, SFM_WIDE_HALVES=0; the default is the second one, WIDE8 below.)
  // Cross-patch pipeline (kModePipe; everything else is kModeSameExactLazyG).  The other
  // modes run two workgroups of four waves per CU, each on its own patch, and every patch
  // has phases in which its four waves wait for each other: the staging round trip, the
  // seed probe's barrier, the waves that find the tile queue empty while the long tiles of
  // the patch drain, the publication barrier.  Here ONE workgroup of EIGHT waves owns the
  // CU and LDS holds TWO patch slots (pixels, 1-D arrays, bounds, lazy-store state, a
  // control header each: 2 x 72 KB at 160^2).  A wave is not bound to a slot: it draws a
  // row tile from whichever slot has one (control word = generation << 8 | next tile, so
  // the draw is atomic with the patch's identity), and there is no workgroup barrier after
  // start-up.  The wave whose tile completes a patch (per-slot completion counter) is its
  // CLOSER: alone it runs the end-of-patch pass (recomputation of band tiles that finished
  // un-stored, against the final maximum), publishes the patch, claims the next patch from
  // the global queue, stages it into the slot it just freed (pixels, tables, seed probe,
  // table touches -- one wave's loads, while the other seven work on the other slot) and
  // re-arms the slot's control word.  The state a slot carries from patch to patch (seed
  // block, previous need mask, hot columns) stays with the slot, i.e. every slot behaves
  // like a workgroup of the other modes; results do not depend on any of it.
  constexpr bool PIPE = MODE == kModePipe;
  constexpr bool LAZYG = MODE == kModeSameExactLazyG || PIPE;
  constexpr bool SAME = MODE == kModeSame || MODE == kModeSameExact || MODE == kModeSameLazy ||
                        MODE == kModeSameExactLazy || LAZYG;
  constexpr bool EXACT = MODE == kModeSameExact || MODE == kModeSameExactLazy || LAZYG;
  constexpr bool LAZY = MODE == kModeSameLazy || MODE == kModeSameExactLazy || LAZYG;
  constexpr bool RAW = MODE == kModeRaw;
  constexpr int NQ = NCA + NCE - 1;
  // Search-window variants, second form (SFM_WIDE_HALVES, the default): the CU's one workgroup
  // has EIGHT waves (two per SIMD: a partner covers a wave's fragment round trips and runs
  // its matrix loop under the other's epilogue) and a tile job is one column HALF of a row
  // tile -- NQH accumulator tiles in plain VGPRs, the 12 .. 15 A chunks that reach them.
  constexpr bool WIDE8 = NCA > 10 && MODE == kModeGeneral && SFM_WIDE_HALVES;
  constexpr int NQH = (NQ + 1) / 2;
  constexpr int kCq0 = NCE - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float touch_junk[kThreads];  // sink of the LDS-direct G touches
  __shared__ int probe_lds[kThreads];     // K-split sums of the seed probe (a.prune)
  // (kModePipe: everything below exists once per slot, `bind_slot` re-points the names)
  unsigned char* A_lds = smem;
  unsigned char* B_lds = smem + a.a_bytes;
  float* R_lds = reinterpret_cast<float*>(smem + a.a_bytes + a.b_bytes);
  // 16 bytes behind the aux arrays: running maximum of the current surface.
  int* pmax_lds = reinterpret_cast<int*>(smem + a.a_bytes + a.b_bytes + a.r_bytes);
  int* hot_lds = pmax_lds + 1;  // hot-list fill count of the current patch
  float* tb_lds = reinterpret_cast<float*>(pmax_lds + 4);  // pruning bounds (a.prune)
  // Pruning state of the workgroup, behind the bounds:
  //   [0] (row tile << 8 | column tile) that held the maximum of the previous patch:
  //       where the next patch is probed first
  //   [1] bit mask of the row tiles pruned in the current patch
  //   [2] anything pruned in the current patch?  [3] patches done  [4] probe this patch?
  int* best_lds = reinterpret_cast<int*>(tb_lds + kBoundStride);
  // Lazy stores, per patch: [0] row tiles that should be stored (requests of
  // tiles that may be hot), [1] tiles finished, [2] tiles stored, [3] tiles
  // claimed for recomputation; lz_tmax[p]: maximum of finished tile p
  int* lz = best_lds + 8;
  float* lz_tmax = reinterpret_cast<float*>(lz + 4);
  int* lz_prev = reinterpret_cast<int*>(lz_tmax + 31);   // (tiles 0 .. 30) what the previous patch needed
  // Column side of the lazy stores: lz_cq[0], [1] lowest / highest column tile that
  // held a possibly hot element in this patch, [2], [3] the same of the previous
  // patch of the workgroup; lz_ks[p]: outer column tiles row tile p dropped inside its
  // row loop (0: none beyond the a-priori ones)
  int* lz_cq = lz_prev + 1;
  int* lz_ks = lz_cq + 4;
  // lz_rows[p]: first | last << 8 row (0 .. 15) of tile p with a possibly hot element
  int* lz_rows = lz_ks + 32;
  // kModePipe, control header of a slot:
  //   [0] generation << 8 | next tile of the tile order (>= n_order: nothing to draw; written
  //       by the opener LAST, so a successful draw implies a completely staged patch)
  //   [1] tiles of the patch that are done (pruned, abandoned, cold or finished)
  //   [2] patch index   [3] slot retired (no patch left for it)
  //   [4..7] float bits of mean_A - c_A, mean_B - c_B, const_a, const_b of the patch
  int* ctl = lz_rows + 32;
  int* pr_lds = probe_lds;   // seed-probe sums (kModePipe: the slot's own 256 words)
  auto bind_slot = [&](int s) {
    unsigned char* base = smem + s * a.slot_bytes;
    A_lds = base;
    B_lds = base + a.a_bytes;
    R_lds = reinterpret_cast<float*>(base + a.a_bytes + a.b_bytes);
    pmax_lds = reinterpret_cast<int*>(base + a.a_bytes + a.b_bytes + a.r_bytes);
    hot_lds = pmax_lds + 1;
    tb_lds = reinterpret_cast<float*>(pmax_lds + 4);
    best_lds = reinterpret_cast<int*>(tb_lds + kBoundStride);
    lz = best_lds + 8;
    lz_tmax = reinterpret_cast<float*>(lz + 4);
    lz_prev = reinterpret_cast<int*>(lz_tmax + 31);
    lz_cq = lz_prev + 1;
    lz_ks = lz_cq + 4;
    lz_rows = lz_ks + 32;
    ctl = lz_rows + 32;
    pr_lds = ctl + kPipeCtl;
  };
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int Py = a.P[0], Px = a.P[1], Qy = a.Q[0], Qx = a.Q[1];
  const int Sy = a.S[0], Sx = a.S[1];

  const long long probe_c0 = clock64(), probe_w0 = wall_clock64();
  // Zero the whole LDS image once: pad rows / margins stay zero afterwards.
  for (int i = threadIdx.x * 16; i < (PIPE ? 2 * a.slot_bytes : a.a_bytes + a.b_bytes);
       i += (PIPE ? 2 * kThreads : WIDE8 ? 64 * SFM_WIDE_WAVES : kThreads) * 16)
    *reinterpret_cast<v4i*>(smem + i) = v4i{0, 0, 0, 0};
  if constexpr (PIPE) {
    __syncthreads();
    if (lane == 0 && wave < 2) {
      bind_slot(wave);
      *best_lds = (((a.Q[0] - 1) / 16) << 8) | ((a.Q[1] - 1) / 16);  // the zero shift
      best_lds[2] = 1;
      lz_cq[0] = NQ;
      lz_cq[1] = -1;
      lz_cq[2] = lz_cq[3] = (a.Q[1] - 1) / 16;
      ctl[0] = 255;    // generation 0, nothing to draw
      ctl[2] = -1;
    }
    bind_slot(0);
    __syncthreads();
  }

  if (!PIPE && SAME && (a.prune || LAZY) && threadIdx.x == 0) {
    *best_lds = (((a.Q[0] - 1) / 16) << 8) | ((a.Q[1] - 1) / 16);  // the zero shift
    best_lds[2] = 1;  // pruning events of the previous patch (optimistic start)
    best_lds[3] = 0;  // patches of this workgroup so far
    if (LAZY) {
      *lz_prev = 0;
      lz_cq[0] = NQ;
      lz_cq[1] = -1;
      lz_cq[2] = lz_cq[3] = (a.Q[1] - 1) / 16;   // the zero shift
    }
  }
  const long long bytes0 = (long long)a.ishape[0][0] * a.ishape[0][1];
  const long long bytes1 = (long long)a.ishape[1][0] * a.ishape[1][1];
  const int cq0 = NCE - 1;
  // Per-lane start of the B windows inside a padded post-patch row.
  const int pos0 = a.ml + Qx - 1 - n - 16 * cq0;
  const int sh = pos0 & 3;

#ifdef SFM_MFMA_TIMING
  const long long wstart = wall_clock64();
  const long long cstart = clock64();
  long long tph[18] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; long long tc = clock64(); int npat = 0;
#define TICK(i) { long long tn = clock64(); tph[i] += tn - tc; tc = tn; }
#else
#define TICK(i)
#endif
  // Patches are handed out dynamically: the two workgroups of a CU do not run
  // at the same speed (see the priority note below) and finish whole patches
  // at different times.
  int* next_lds = hot_lds + 1;
  int tiles_drawn = 0, tiles_skipped = 0, cols_skipped = 0, tiles_early = 0;  // (wave-uniform; reported through a.clk)
  long long mfma_issued = 0;  // matrix instructions this wave issued in the row loops
  if (a.prio_mode == 3) {
    // HW_REG_LDS_ALLOC[7:0] = LDS_BASE: 0 for the first workgroup of the CU.
    const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | 6) & 0xff;
    if (lds_base != 0)
      __builtin_amdgcn_s_setprio(3);
    else
      __builtin_amdgcn_s_setprio(0);
  }
  // Work items: patches, or the extra (patch, operand-plane pair) passes of
  // the masked path (see masked_classify_kernel).
  const int n_items = (RAW && a.list) ? *a.n_list : a.batch;
  // HW_REG_XCC_ID (20), bits [3:0]: the XCD this workgroup runs on
  int q_state = a.xcd_heads ? static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) : 0;
  // (g0, gs: first row group and stride of the calling wave; alone: no other wave takes part)
  // (kind 0: the four waves of a workgroup, sums through LDS atomics, barrier, seed;
  // kind 2, kModePipe: this wave's share of the row groups into the slot's sums only --
  // seed_finish() follows when every share is in)
  auto seed_finish = [&]() {
    int sm = max(max(pr_lds[lane], pr_lds[64 + lane]), max(pr_lds[128 + lane], pr_lds[192 + lane]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sm = max(sm, __shfl_xor(sm, d, 64));
    const float m_lo = __int2float_rd(sm) - tb_lds[kBoundCorr];
    if (lane == 0 && m_lo > 0.f) atomicMax(pmax_lds, __float_as_int(m_lo));
  };
  auto seed_probe = [&](const int g0, const int gs, const int kind) {
      // Seed of the running maximum.  The first tiles are drawn before any tile
      // has finished, i.e. with nothing to prune against.  So the 16 x 16 block of
      // shifts that held the previous patch's maximum is evaluated first, its
      // patch rows split over the four waves (exact integer sums S, the same
      // fragments the row loop would use), summed through LDS atomics, and
      //   max(surface) >= max(S over the block) - |correction|max =: m_lo
      // (tbound[kBoundCorr], prep kernel) goes into the running maximum.  m_lo is
      // strictly below a real element, so the final maximum is unaffected.
      const int pq = __builtin_amdgcn_readfirstlane(*best_lds);
      const int ps = pq >> 8, qs = pq & 255;
      const int pdy0 = 16 * ps - (Qy - 1);
      const int pylo = max(0, -pdy0 - 15), pyhi = min(Qy, Py - pdy0);
      const unsigned char* pap = A_lds + (kPadTop + pylo + g + pdy0 + n) * a.pa;
      const unsigned char* pbp = B_lds + (pylo + g) * a.pb + (pos0 & ~3);
      // two row groups per trip: their loads are in flight together and they
      // accumulate into separate registers (no dependent MFMA chain of 2 NCA)
      v4i pacc = v4i{0, 0, 0, 0}, pacc2 = v4i{0, 0, 0, 0};
      const int n_grp = (pyhi - pylo + 3) >> 2;
      // The pairs (ca, c) on the diagonal of column tile qs are ca = ca_lo + i,
      // c = c_lo + i, i < n_on: contiguous A chunks against a contiguous run of B
      // dwords (4 n_on + 1, shared between neighbouring fragments like in the row
      // loop).  Pairs past n_on, and groups past the tile, read A from a zero
      // padding row instead (branch-free).
      const int ca_lo = max(0, qs - cq0), c_lo = ca_lo - qs + cq0;
      const int n_on = min(NCA - ca_lo, NCE - c_lo);
      const unsigned char* zero_row = A_lds + n * a.pa;  // inside the top padding
      constexpr int kPD = 4 * NCA + 1;
      auto load_group = [&](int grp, v4i* paf, unsigned* pd) {
        const bool live = grp < n_grp;
        const int gc = min(grp, n_grp - 1);
        const unsigned char* ag = pap + 4 * gc * a.pa + 16 * ca_lo;
        const unsigned char* bg = pbp + 4 * gc * a.pb + 16 * c_lo;
#pragma unroll
        for (int i = 0; i < NCA; ++i)
          paf[i] = *reinterpret_cast<const v4i*>((live && i < n_on) ? ag + 16 * i : zero_row);
#pragma unroll
        for (int j = 0; j < kPD; ++j) pd[j] = *reinterpret_cast<const unsigned*>(bg + 4 * j);
      };
      auto mma_group = [&](const v4i* paf, const unsigned* pd, v4i& acc_out) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          v4i bf;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            bf[k] = static_cast<int>(
                __builtin_amdgcn_alignbyte(pd[4 * i + k + 1], pd[4 * i + k], sh));
          acc_out = __builtin_amdgcn_mfma_i32_16x16x64_i8(paf[i], bf, acc_out, 0, 0, 0);
        }
      };
      for (int grp = g0; grp < n_grp; grp += 2 * gs) {
        v4i paf[NCA], paf2[NCA];
        unsigned pd[kPD], pd2[kPD];
        load_group(grp, paf, pd);
        load_group(grp + gs, paf2, pd2);
        mma_group(paf, pd, pacc);
        mma_group(paf2, pd2, pacc2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pacc[r] += pacc2[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&pr_lds[r * 64 + lane], pacc[r]);
      if (kind == 2) return;
      __syncthreads();
      seed_finish();
  };
  // kModePipe: opening a slot, i.e. everything the other modes do per patch with 256
  // threads and three barriers (staging, tables, state reset, seed probe, table touches),
  // cut into UNITS that any wave without a tile can take:
  //   unit 0            (the closer itself) the 1-D arrays, the pruning bounds, the zeroed
  //                     probe sums, the table touches
  //   units 1 .. NA     staging: 7 x 64 sixteen-byte items of either patch each
  //   units NA+1 .. +NB the seed probe: every NB-th row group of the probe block (only
  //                     after units 0 .. NA: they read the staged pixels)
  // Claims are one LDS atomic (ctl[8]); the wave whose unit is the last one done (ctl[9])
  // seeds the running maximum from the probe sums and arms the slot.  The patch that goes
  // into a slot was claimed from the global queue -- and its PatchParams fetched -- while
  // the previous patch of the slot was still running (ctl[18 ..]), so nothing of an
  // opening waits for a global round trip except unit 0's own table loads.
  constexpr int kProbeUnits = 4;
  auto n_stage_units = [&]() {
    const int n_max = max(Py * NCA, Qy * ((Qx + 15) / 16));
    return (n_max + 64 * kStageBatch - 1) / (64 * kStageBatch);
  };
  // (lane 0) claims the patch after next for the bound slot and parks its parameters
  auto prefetch_next = [&](const int first) {
    int nb = first;
    if (nb < 0) {
      if (lane == 0) nb = 2 * static_cast<int>(gridDim.x) + atomicAdd(a.work_counter, 1);
      nb = __builtin_amdgcn_readfirstlane(nb);
    }
    if (nb < n_items) {
      const PatchParams pp = a.pp[nb];
      if (lane == 0) {
        ctl[19] = pp.y0[0]; ctl[20] = pp.x0[0]; ctl[21] = pp.c[0];
        ctl[22] = pp.y0[1]; ctl[23] = pp.x0[1]; ctl[24] = pp.c[1];
        ctl[25] = __float_as_int(pp.mu[0]); ctl[26] = __float_as_int(pp.mu[1]);
      }
    }
    if (lane == 0) ctl[18] = nb;
  };
  auto open_unit_done = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    int dn = 0;
    if (lane == 0) dn = atomicAdd(&ctl[9], 1) + 1;
    dn = __builtin_amdgcn_readfirstlane(dn);
    if (dn == __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[10]))) {
      // the last unit: seed, then arm (generation + 1, tile 0; every LDS write of the
      // opening is older than this one)
      if (__builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[11]))) seed_finish();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) {
        const int gen = (*const_cast<volatile int*>(&ctl[0]) >> 8) + 1;
        *const_cast<volatile int*>(&ctl[0]) = gen << 8;
      }
    }
  };
  // unit c >= 1 of the bound slot's opening (claimed by the caller)
  auto open_unit = [&](const int c) {
    const int na = n_stage_units();
    if (c <= na) {
      const int y0a = __builtin_amdgcn_readfirstlane(ctl[12]), x0a = __builtin_amdgcn_readfirstlane(ctl[13]);
      const int cca = __builtin_amdgcn_readfirstlane(ctl[14]);
      const int y0b = __builtin_amdgcn_readfirstlane(ctl[15]), x0b = __builtin_amdgcn_readfirstlane(ctl[16]);
      const int ccb = __builtin_amdgcn_readfirstlane(ctl[17]);
      const StagePlane sa = {a.img[0], bytes0, a.ishape[0][1], y0a, x0a, Py, Px,
                             cca, A_lds, a.pa, kPadTop, 0, NCA};
      const StagePlane sb = {a.img[1], bytes1, a.ishape[1][1], y0b, x0b, Qy, Qx,
                             ccb, B_lds, a.pb, 0, a.ml, (Qx + 15) / 16};
      stage_patches<64>(sa, sb, lane, (c - 1) * 64 * kStageBatch, c * 64 * kStageBatch);
      TICK(15)
    } else {
      // (the pixels and the zeroed sums first)
      while (__builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[9])) < na + 1)
        __builtin_amdgcn_s_sleep(2);
      seed_probe(c - na - 1, kProbeUnits, 2);
      TICK(16)
    }
    open_unit_done();
  };
  // the closer (or, at start-up, the wave that owns the slot): next patch into the bound slot
  auto open_begin = [&]() {
    const int nb = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[18]));
    if (nb >= n_items) {
      if (lane == 0) ctl[3] = 1;   // retired: the control word stays at "nothing to draw"
      return false;
    }
#ifdef SFM_MFMA_TIMING
    ++npat;
#endif
    const int na = n_stage_units();
    if (lane == 0) {
      ctl[2] = nb;
      ctl[12] = ctl[19]; ctl[13] = ctl[20]; ctl[14] = ctl[21];
      ctl[15] = ctl[22]; ctl[16] = ctl[23]; ctl[17] = ctl[24];
      ctl[4] = ctl[25]; ctl[5] = ctl[26];
      *pmax_lds = 0;
      *hot_lds = 0;
      // initial store requests: as in the other lazy modes (see there)
      const int pt = *best_lds >> 8, gt = (a.guard + 15) >> 4;
      const int lo_t = max(pt - gt, 0), hi_t = min(pt + gt, a.n_order - 1);
      const int pv = *lz_prev;
      lz[0] = (pv && !a.widen ? 0 : static_cast<int>(((2u << hi_t) - 1u) & ~((1u << lo_t) - 1u))) |
              pv | (a.widen ? (pv << 1) | (pv >> 1) : 0);
      lz[1] = lz[2] = lz[3] = 0;
      if (lz_cq[1] >= lz_cq[0]) {
        lz_cq[2] = lz_cq[0];
        lz_cq[3] = lz_cq[1];
      }
      lz_cq[0] = NQ;
      lz_cq[1] = -1;
      int probe_on = 0;
      if (a.prune) {
        best_lds[1] = 0;
        const int n_done = best_lds[3];
        best_lds[3] = n_done + 1;
        probe_on = (a.probe && (best_lds[2] > 0 || (n_done & 7) == 0)) ? 1 : 0;
        best_lds[4] = probe_on;
        best_lds[2] = 0;
      }
      ctl[1] = 0;
      ctl[9] = 0;
      ctl[10] = 1 + na + (probe_on ? kProbeUnits : 0);
      ctl[11] = probe_on;
      // (in order behind everything above: the staging units may start)
      *const_cast<volatile int*>(&ctl[8]) = 1;
    }
    // unit 0
    const float* aux = a.aux + (long long)nb * (4 * a.aux_n + 4);
    constexpr int kAuxP = (4 * (16 * NCA + 1) + 63) / 64;   // (Py <= Px = 16 NCA)
    float auxv[kAuxP], tbv[kBoundStride / 64];
#pragma unroll
    for (int k = 0; k < kAuxP; ++k) {
      const int i = lane + 64 * k;
      auxv[k] = i < 4 * a.aux_n ? aux[i] : 0.f;
    }
    const float ca_new = aux[4 * a.aux_n + 0], cb_new = aux[4 * a.aux_n + 1];
#pragma unroll
    for (int k = 0; k < kBoundStride / 64; ++k)
      tbv[k] = a.tbound[(long long)nb * kBoundStride + lane + 64 * k];
#ifndef SFM_NO_TOUCH
    {
      // the column-sum tables this patch's finishing tiles build their table rows from
      const char* gt = reinterpret_cast<const char*>(a.c16 + nb * a.c16_stride);
      const unsigned junk_off = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
          (__attribute__((address_space(3))) float*)touch_junk));
      const int n_lines = (2 * ((Py >> 4) + 1) * Px + Py + 1 + 15) >> 4;
      for (int k = lane; k < n_lines; k += 64) {
        const char* src = gt + (size_t)k * 64;
        unsigned saved_m0;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
            "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(saved_m0)
            : "v"(src), "s"(junk_off)
            : "memory");
      }
    }
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) pr_lds[lane + 64 * k] = 0;
    if (lane < 32) lz_ks[lane] = 0;
#pragma unroll
    for (int k = 0; k < kAuxP; ++k) {
      const int i = lane + 64 * k;
      if (i < 4 * a.aux_n) R_lds[i] = auxv[k];
    }
#pragma unroll
    for (int k = 0; k < kBoundStride / 64; ++k) tb_lds[lane + 64 * k] = tbv[k];
    if (lane == 0) {
      ctl[6] = __float_as_int(ca_new);
      ctl[7] = __float_as_int(cb_new);
    }
    TICK(17)
    open_unit_done();
    return true;
  };
  int cur = 0;   // kModePipe: the slot this wave is bound to
  if constexpr (PIPE) {
    // start-up: waves 0 and 1 own the first opening of slot 0 / 1, everybody helps
    cur = wave & 1;
    bind_slot(cur);
    if (wave < 2) {
      prefetch_next(2 * static_cast<int>(blockIdx.x) + wave);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (open_begin()) prefetch_next(-1);
    }
  }
  // (kModePipe: ONE trip -- its tile loop runs across all patches of the workgroup)
  for (int item = PIPE ? 0 : a.xcd_heads ? pull_item(a, n_items, &q_state, next_lds) : blockIdx.x;
       PIPE ? item == 0 : item < n_items;
       item = PIPE ? 1 : a.xcd_heads ? pull_item(a, n_items, &q_state, next_lds)
                                      : next_patch(a, item, next_lds)) {
    int b = item;
    int plane0 = a.plane[0], plane1 = a.plane[1];
    if (RAW && a.list) {
      const int code = a.list[item], pass = code & 7;  // kMaskedPasses[pass]
      b = code >> 3;
      plane0 = pass == 1 ? kPlaneVal : pass == 4 ? kPlaneSqHi : pass == 5 ? kPlaneSqLo
                                                                          : kPlaneValid;
      plane1 = pass == 2 ? kPlaneVal : pass == 6 ? kPlaneSqHi : pass == 7 ? kPlaneSqLo
                                                                          : kPlaneValid;
    }
#ifdef SFM_MFMA_TIMING
    ++npat;
#endif
    TICK(7)
    float const_a = 0.f, const_b = 0.f;
    float mua = 0.f, mub = 0.f, muab = 0.f;
    if constexpr (!PIPE) {
    __syncthreads();  // previous patch fully consumed / zero fill done
    TICK(0)
    const PatchParams pp = a.pp[b];
    // covers aux_n <= 192 and Py Px <= 32768 in registers; taller patches
    // take the remainder loops below
    constexpr int kAuxRegs = 3;
    float auxv[kAuxRegs];
    float tbv = 0.f;
    if (RAW) {
      stage_plane(a.img[0], bytes0, a.ishape[0][1], pp.y0[0], pp.x0[0], a.mask[0],
                  (long long)a.mshape[0][0] * a.mshape[0][1], a.mshape[0][1],
                  pp.my0[0], pp.mx0[0], Py, Px, pp.c[0], plane0, A_lds, a.pa,
                  kPadTop, 0, NCA);
      stage_plane(a.img[1], bytes1, a.ishape[1][1], pp.y0[1], pp.x0[1], a.mask[1],
                  (long long)a.mshape[1][0] * a.mshape[1][1], a.mshape[1][1],
                  pp.my0[1], pp.mx0[1], Qy, Qx, pp.c[1], plane1, B_lds, a.pb, 0,
                  a.ml, (Qx + 15) / 16);
    } else {
      // Everything staging needs from global memory is requested before the
      // first wait (the four 1-D correction arrays and the pixels of both
      // patches): loads return in order, so staging costs one round trip.
      if (SAME) {
        const float* aux = a.aux + (long long)b * (4 * a.aux_n + 4);
#pragma unroll
        for (int k = 0; k < kAuxRegs; ++k) {
          const int i = threadIdx.x + k * kThreads;
          auxv[k] = i < 4 * a.aux_n ? aux[i] : 0.f;
        }
        const_a = aux[4 * a.aux_n + 0];
        const_b = aux[4 * a.aux_n + 1];
        // (unconditional: a load under a condition would be waited for on its own)
        tbv = a.tbound[(long long)b * kBoundStride + (threadIdx.x & (kBoundStride - 1))];
      }
      const StagePlane sa = {a.img[0], bytes0, a.ishape[0][1], pp.y0[0], pp.x0[0], Py, Px,
                             pp.c[0], A_lds, a.pa, kPadTop, 0, NCA};
      const StagePlane sb = {a.img[1], bytes1, a.ishape[1][1], pp.y0[1], pp.x0[1], Qy, Qx,
                             pp.c[1], B_lds, a.pb, 0, a.ml, (Qx + 15) / 16};
      stage_patches<WIDE8 ? 64 * SFM_WIDE_WAVES : kThreads>(sa, sb, threadIdx.x);
      TICK(8)
    }
    if (threadIdx.x == 0) {
      *pmax_lds = 0;  // float bits of max(surface, 0)
      *hot_lds = 0;
      pmax_lds[3] = 0;  // tile counter of this patch
      if (LAZY) {
        // requested from the start: what the previous patch of this workgroup
        // turned out to need, and the tiles around the row tile that held its
        // maximum (flow fields are coherent; the tiles next to the peak tile are
        // drawn together with it and would otherwise finish un-stored and be
        // recomputed: measured 2.4 redone tiles per patch without any request,
        // 0.75 with the peak tile's band alone -- the peak's blob often straddles
        // two tiles, whose bands are four tiles)
        const int pt = *best_lds >> 8, gt = (a.guard + 15) >> 4;
        const int lo_t = max(pt - gt, 0), hi_t = min(pt + gt, a.n_order - 1);
        const int pv = *lz_prev;   // (widened by a tile: a store costs less than a recomputation)
        // (the band of the peak tile only while there is no previous need mask: the
        // mask counts the guard band in rows and is usually a tile narrower)
        lz[0] = (pv && !a.widen ? 0 : static_cast<int>(((2u << hi_t) - 1u) & ~((1u << lo_t) - 1u))) |
                pv | (a.widen ? (pv << 1) | (pv >> 1) : 0);
        lz[1] = lz[2] = lz[3] = 0;
        if (lz_cq[1] >= lz_cq[0]) {   // (the previous patch had hot elements)
          lz_cq[2] = lz_cq[0];
          lz_cq[3] = lz_cq[1];
        }
        lz_cq[0] = NQ;
        lz_cq[1] = -1;
      }
      if (SAME && a.prune) {
        best_lds[1] = 0;  // row tiles pruned in this patch
        // The seed probe pays only where something gets pruned: it runs if the
        // previous patch of this workgroup pruned anything, and every eighth
        // patch regardless (to notice when the data improve).
        const int n_done = best_lds[3];
        best_lds[3] = n_done + 1;
        best_lds[4] = (best_lds[2] > 0 || (n_done & 7) == 0) ? 1 : 0;
        best_lds[2] = 0;
      }
    }
    if (SAME) {
#pragma unroll
      for (int k = 0; k < kAuxRegs; ++k) {
        const int i = threadIdx.x + k * kThreads;
        if (i < 4 * a.aux_n) R_lds[i] = auxv[k];
      }
      if (a.prune && threadIdx.x < kBoundStride) tb_lds[threadIdx.x] = tbv;
      if (a.prune) probe_lds[threadIdx.x] = 0;
      if (LAZY && threadIdx.x < 32) lz_ks[threadIdx.x] = 0;
      const float* aux = a.aux + (long long)b * (4 * a.aux_n + 4);
      for (int i = threadIdx.x + kAuxRegs * kThreads; i < 4 * a.aux_n; i += kThreads)
        R_lds[i] = aux[i];
    }
    TICK(9)
    __syncthreads();

    if (SAME && a.prune && a.probe && __builtin_amdgcn_readfirstlane(best_lds[4]))
      seed_probe(wave, kWaves, 0);

    TICK(1)
    mua = pp.mu[0];
    mub = pp.mu[1];
    muab = mua * mub;
    asm volatile("" : "+v"(muab));  // every global load so far has been consumed
#ifndef SFM_NO_TOUCH
    if (SAME) {
      // Pull this patch's correction table G (written by the prep kernel, by now
      // in HBM / Infinity Cache) into this XCD's L2, one touch per 64-byte line.
      // It is first needed by the epilogue of the first tile, a whole matrix
      // loop away, so the touches use the LDS-direct load path into a junk
      // area: no VGPR result, and nothing waits for them before that epilogue.
      // (The lane offset is made opaque so that the per-touch addresses are
      // scalar base + one VGPR instead of loop-invariant VGPRs that spill.)
      // (Written as inline assembly: the compiler's global_load_lds builtin does
      // not initialise M0, the LDS destination base, on this toolchain.  The
      // asm loads are invisible to the compiler's vmcnt bookkeeping, which is
      // safe because every counted wait that follows is for YOUNGER loads and
      // the counter retires in order.)
      const char* gt = LAZYG ? reinterpret_cast<const char*>(a.c16 + b * a.c16_stride)
                             : reinterpret_cast<const char*>(a.gtab + (long long)b * Py * Px);
      const unsigned junk_off = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
          (__attribute__((address_space(3))) float*)touch_junk));
      constexpr int kTouches = (32768 / 16 + kThreads - 1) / kThreads;
      // opaque per patch: otherwise the eight lane addresses are hoisted out of
      // the patch loop as 16 VGPRs and spilled
      unsigned lane_off;
      asm volatile("v_lshlrev_b32 %0, 6, %1" : "=v"(lane_off) : "v"(threadIdx.x));
      // Lazy modes: the epilogue runs on the few tiles that may be hot or are asked
      // for, so only the table rows of the tiles requested at this point are pulled
      // (row yv serves the shifts dy = yv and dy = yv - Py: two row tiles); a tile
      // that turns out hot without having been predicted takes its L2 misses.
      const int req_now = LAZY ? *const_cast<volatile int*>(&lz[0]) : -1;
#pragma unroll
      for (int k = 0; k < kTouches; ++k) {
        const int i = (threadIdx.x + k * kThreads) * 16;
        bool wanted = i < Py * Px;
        // (the column-sum tables the table rows are built from: all of them)
        if (LAZYG) wanted = i < 2 * ((Py >> 4) + 1) * Px + Py + 1;
        if (!LAZYG && LAZY && a.prune && !a.touch_all) {
          const int yv = min(i / Px, Py - 1), yv2 = min((i + 15) / Px, Py - 1);   // (a line may straddle two rows)
          const unsigned t_mask = (1u << ((yv + Py - 1) >> 4)) | (1u << (max(yv - 1, 0) >> 4)) |
                                  (1u << ((yv2 + Py - 1) >> 4)) | (1u << (max(yv2 - 1, 0) >> 4));
          wanted = wanted && (t_mask & static_cast<unsigned>(req_now)) != 0;
        }
        if (wanted) {
          const char* src = gt + (size_t)k * kThreads * 64 + lane_off;
          unsigned saved_m0;  // M0 is the LDS base of the load; put it back
          asm volatile(
              "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
              "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
              : "=&s"(saved_m0)
              : "v"(src), "s"(junk_off)
              : "memory");
        }
      }
    }
#endif
    }  // !PIPE: the patch's staging
    const int* IA = (SAME || RAW) ? nullptr : a.integ[0] + b * a.integ_stride[0];
    const int* IB = (SAME || RAW) ? nullptr : a.integ[1] + b * a.integ_stride[1];
    const float* G = SAME ? a.gtab + (long long)b * Py * Px : nullptr;   // (kModePipe: per draw)
    float* surf = a.surface + b * a.s_stride;

#if SFM_DYNAMIC_TILES
    bool redo_phase = false;   // LAZY: the patch's tiles are finished, recompute what was missed
    for (;;) {
      int ti = 0, p = 0;
      bool forced = false;
      if (PIPE && !redo_phase) {
        // A tile of either slot.  The control word is tested with a plain read and drawn
        // from with ONE atomic; the patch header is read in the same round (it cannot
        // change while this wave holds a tile of the patch).
        bool got = false;
        for (int spin = 0;; ++spin) {
          const int v = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[0]));
          const int dead_here = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[3]));
          // (the first tiles of a patch run against the seed of the running maximum alone:
          // only pipe_admit of them are handed out before one is done)
          const int done_here = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[1]));
          if ((v & 255) < a.n_order && ((v & 255) < a.pipe_admit || done_here > 0)) {
            int v2 = 0, hb = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
            if (lane == 0) v2 = atomicAdd(&ctl[0], 1);
            hb = *const_cast<volatile int*>(&ctl[2]);
            h4 = *const_cast<volatile int*>(&ctl[4]);
            h5 = *const_cast<volatile int*>(&ctl[5]);
            h6 = *const_cast<volatile int*>(&ctl[6]);
            h7 = *const_cast<volatile int*>(&ctl[7]);
            v2 = __builtin_amdgcn_readfirstlane(v2);
            if ((v2 & 255) < a.n_order) {
              ti = v2 & 255;
              b = __builtin_amdgcn_readfirstlane(hb);
              mua = __int_as_float(__builtin_amdgcn_readfirstlane(h4));
              mub = __int_as_float(__builtin_amdgcn_readfirstlane(h5));
              const_a = __int_as_float(__builtin_amdgcn_readfirstlane(h6));
              const_b = __int_as_float(__builtin_amdgcn_readfirstlane(h7));
              muab = mua * mub;
              G = a.gtab + (long long)b * Py * Px;
              surf = a.surface + b * a.s_stride;
              got = true;
              break;
            }
          }
          // no tile here: a unit of this slot's opening, if it is being opened
          {
            const int u = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[8]));
            const int nu = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[10]));
            if (u >= 1 && u < nu) {
              int c = 0;
              if (lane == 0) c = atomicAdd(&ctl[8], 1);
              c = __builtin_amdgcn_readfirstlane(c);
              if (c < nu) {
                TICK(13)
                open_unit(c);
                spin = 0;
                continue;   // (same slot: more units, or its first tiles)
              }
            }
          }
          cur ^= 1;
          bind_slot(cur);
          if (dead_here &&
              __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&ctl[3])))
            break;   // both slots retired: this wave is done
          if (spin & 1) __builtin_amdgcn_s_sleep(4);
        }
        TICK(13)   // (timing build) looking for a tile
        if (!got) break;
        p = __builtin_amdgcn_readfirstlane(a.order[ti]);
      }
      if (!PIPE && !redo_phase) {
        if (lane == 0) ti = atomicAdd(pmax_lds + 3, 1);
        ti = __builtin_amdgcn_readfirstlane(ti);
        if (ti >= a.n_order) {
          if (!LAZY) break;
          __syncthreads();   // (once per wave and patch) every tile is finished: the redo set is final
          redo_phase = true;
        } else {
          p = __builtin_amdgcn_readfirstlane(a.order[ti]);
        }
      }
      int half = 0;   // WIDE8: which column half of row tile p this job is
      if constexpr (WIDE8) {
        half = p >> 8;
        p &= 255;
      }
      if (LAZY && redo_phase) {
        // What has to be in memory, now that the maximum is final: the tiles that
        // hold an element above threshold_rel x the maximum and their guard bands.
        // The former stored themselves (the running maximum they were judged by
        // was not larger); of the latter, those that finished before anybody asked
        // are recomputed -- every wave forms the same set and claims its share.
        const float thr_f = a.threshold_rel * __int_as_float(*const_cast<volatile int*>(pmax_lds));
        const int done = *const_cast<volatile int*>(&lz[1]);
        // (one tile per lane: as loops over the tiles these were forty dependent LDS
        // round trips per wave and patch, under the fragment traffic of the other
        // workgroup's matrix loops)
        const int t_l = min(lane, 30);
        const bool mine = lane < a.n_order && ((done >> lane) & 1);
        const float tm_l = lz_tmax[t_l];
        const int rr_l = lz_rows[t_l], ks_l = lz_ks[t_l];
        const int hq_lo = *const_cast<volatile int*>(&lz_cq[0]);
        const int hq_hi = *const_cast<volatile int*>(&lz_cq[1]);
        int mask_l = 0;
        if (mine && tm_l > thr_f) {
          // (the guard band is counted in rows from the tile's hot rows: 10 rows
          // reach both neighbouring tiles from a quarter of the positions only)
          const int lo_t = max(16 * t_l + (rr_l & 255) - a.guard, 0) >> 4;
          const int hi_t = min((16 * t_l + (rr_l >> 8) + a.guard) >> 4, a.n_order - 1);
          mask_l = static_cast<int>(((2u << hi_t) - 1u) & ~((1u << lo_t) - 1u));
        }
        const int need = wave_or(mask_l);
        if (lane == 0) *lz_prev = need;   // (every wave writes the same value)
        // stored tiles that dropped column tiles in flight and turn out to have a
        // possibly hot column (of any row tile) within a tile of what they dropped:
        // recomputed in full like the band tiles that finished un-stored
        const int bad = static_cast<int>(__ballot(mine && hq_hi >= hq_lo && ks_l > 0 &&
                                                  (hq_lo < ks_l + 1 || hq_hi > NQ - 2 - ks_l)));
        int todo = need & done & (~*const_cast<volatile int*>(&lz[2]) | bad) &
                   ~*const_cast<volatile int*>(&lz[3]);
        todo = __builtin_amdgcn_readfirstlane(todo);
        if (PIPE && todo == 0) {
          // the patch is complete: publish it (what the peak kernels read), then put the
          // next patch into this slot
          if (lane == 0) {
            a.v1[b] = __int_as_float(*const_cast<volatile int*>(pmax_lds));
            a.hot_count[b] = *const_cast<volatile int*>(hot_lds);
            a.skipmask[b] = ~*const_cast<volatile int*>(&lz[2]) &
                            static_cast<int>((2u << (a.n_order - 1)) - 1u);
          }
          TICK(14)   // (timing build) end-of-patch pass of the closer
          if (open_begin()) prefetch_next(-1);
          TICK(17)
          redo_phase = false;
          continue;
        }
        if (todo == 0) break;
        p = __builtin_ctz(todo);
        int old = 0;
        if (lane == 0) old = atomicOr(&lz[3], 1 << p);
        old = __builtin_amdgcn_readfirstlane(old);
        if ((old >> p) & 1) continue;   // another wave took it
        forced = true;
      }
#else
    const int n_my_tiles = __builtin_amdgcn_readfirstlane(a.n_tiles[wave]);
    for (int ti = 0; ti < n_my_tiles; ++ti) {
      const int p = __builtin_amdgcn_readfirstlane(a.tiles[wave][ti]);
      constexpr bool forced = false;
      static_assert(!LAZY, "lazy stores need the dynamic tile queue");
#endif
      // (the tile's body is a block of its own: a `continue` inside it ends the tile and
      // falls through to the completion count of kModePipe behind it)
      do {
      // The two workgroups of a CU share each SIMD's MFMA pipe, and the issue
      // arbiter always favours the older wave: the younger workgroup would
      // run ~25 % slower and finish long after its partner.  Alternating the
      // priority per tile (opposite phase for the second dispatch wave of
      // workgroups) evens the two out.
      if (a.prio_mode == 1) {
        if ((ti & 1) ^ (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0))
          __builtin_amdgcn_s_setprio(1);
        else
          __builtin_amdgcn_s_setprio(0);
      } else if (a.prio_mode == 2) {
        if (blockIdx.x >= (gridDim.x >> 1))
          __builtin_amdgcn_s_setprio(1);
        else
          __builtin_amdgcn_s_setprio(0);
      }
      int raw_col_skip = 0;
      if (RAW && a.ov_lb) {
        // largest ny = overlap rows of a shift in this tile (ny rises to min(Py, Qy)
        // at dy in [0, Py - Qy] and falls): at the tile's row nearest that range
        const int ky0 = 16 * p, ky1 = min(16 * p + 15, Sy - 1);
        const int dyc = min(max(min(max(0, ky0 - (Qy - 1)), Py - Qy), ky0 - (Qy - 1)), ky1 - (Qy - 1));
        const int ny_max = min(Py, Qy + dyc) - max(0, dyc);
        const int lb = __builtin_amdgcn_readfirstlane(a.ov_lb[b / a.ov_group]);
        const float ov_thr = 0.3f * static_cast<float>(lb);
        if (static_cast<float>(ny_max * Qx) < ov_thr) continue;
        // the same along x: outer column tiles whose widest overlap nx, times
        // ny_max, stays below the threshold are zeroed as well (row-loop variants)
        auto cols_dead = [&](int ks) {
          if (ks <= 0) return false;
          auto nx_at = [&](int kx) {
            const int dx = kx - (Qx - 1);
            return kx < Sx ? min(Px, Qx + dx) - max(0, dx) : 0;
          };
          const int nx_max = max(nx_at(16 * ks - 1), nx_at(16 * (NQ - ks)));
          return static_cast<float>(ny_max * nx_max) < ov_thr;
        };
        raw_col_skip = cols_dead(col_skip_3(NQ))   ? col_skip_3(NQ)
                       : cols_dead(col_skip_2(NQ)) ? col_skip_2(NQ)
                       : cols_dead(col_skip_hi(NQ)) ? col_skip_hi(NQ)
                       : cols_dead(col_skip_lo(NQ)) ? col_skip_lo(NQ)
                                                    : 0;
      }
      int col_skip = raw_col_skip;  // outer column tiles (each side) this tile leaves out
      // (lazy modes) read with the pruning bounds, used by the test in front of the row loop
      unsigned hd_ea = 0, hd_eb = 0;
      float hd_corr = 0.f, hd_thr = 0.f;
      int hd_req = 0, hd_cq = 0;
      if (SAME && a.prune && !forced) {
        // Exact pruning.  Every element of this tile and of the `guard` rows
        // around it is bounded by tb (prep kernel).  If tb < threshold_rel x (the
        // running maximum, which never exceeds the final one), none of them is
        // the maximum, a peak candidate, inside a candidate's max-filter window
        // or inside the first peak's sharpness window: the tile is never read
        // for its values.  It is not stored at all; the overflow fall-backs of the
        // peak kernels, which sweep whole surfaces, get the set of pruned tiles
        // (a.skipmask) and skip / zero them.
        // (everything the decisions of this tile read from LDS is requested here, in
        // one round: under the fragment traffic of eight waves a dependent LDS read
        // costs several hundred cycles, and the tests below were six of them in a row)
        const int mrun_bits = *const_cast<volatile int*>(pmax_lds);
        const float tb_p = tb_lds[p], tb_c0 = tb_lds[kBoundTiles + 0], tb_c1 = tb_lds[kBoundTiles + 1];
        const float tb_2a = tb_lds[32 + 3 * p + 0], tb_2b = tb_lds[32 + 3 * p + 1];
        const float tb_2c = tb_lds[32 + 3 * p + 2];
        if (LAZY) {
          const int dy0h = 16 * p - (Qy - 1);
          const int yloh = max(0, -dy0h - 15), yhih = min(Qy, Py - dy0h);
          const unsigned* rp = reinterpret_cast<const unsigned*>(tb_lds + kRowPre);
          const int a_lo = max(0, yloh + dy0h), a_hi = min(Py, yhih + dy0h + 15);
          hd_ea = rp[(a_hi + 3) >> 2] - rp[a_lo >> 2];
          hd_eb = rp[64 + ((yhih + 3) >> 2)] - rp[64 + (yloh >> 2)];
          hd_corr = tb_lds[kBoundCorr];
          hd_req = *const_cast<volatile int*>(&lz[0]);
          hd_cq = min(lz_cq[2], NQ - 1 - lz_cq[3]);
        }
        const float mrun = __int_as_float(__builtin_amdgcn_readfirstlane(mrun_bits));
        hd_thr = a.threshold_rel * mrun;
        ++tiles_drawn;
        if (tb_p < a.threshold_rel * mrun) {
          ++tiles_skipped;
          if (lane == 0) {
            atomicOr(&best_lds[1], 1 << p);
            best_lds[2] = 1;  // (benign race: any non-zero value)
          }
          continue;
        }
        // Along x the same argument holds for the outer column tiles (bounds of
        // the outermost kKs1 / kKs2 tiles on either side, widened by the
        // guard): the row loop below has variants that leave them out.
        const float t = a.threshold_rel * mrun;
        if (tb_c0 < t) col_skip = col_skip_lo(NQ);
        if (tb_c1 < t) col_skip = col_skip_hi(NQ);
        // this row tile with fewer columns still (2-D block bounds; each already
        // capped by the 1-D column bound of its variant)
        if (tb_2a < t) col_skip = max(col_skip, col_skip_hi(NQ));
        if (tb_2b < t) col_skip = max(col_skip, col_skip_2(NQ));
        if (tb_2c < t) col_skip = max(col_skip, col_skip_3(NQ));
        cols_skipped += 2 * col_skip;
        if (col_skip > 0 && lane == 0) best_lds[2] = 1;
      }
      const int dy0 = 16 * p - (Qy - 1);
      const int ylo = max(0, -dy0 - 15);
      const int yhi = min(Qy, Py - dy0);
      if constexpr (NCA > 10 && MODE == kModeGeneral) {
        // Search-window variants: the rows of the two integral images this tile's epilogue
        // will gather from (written by the prep launch: in HBM by now) are pulled towards
        // this XCD's L2 while the matrix loop runs -- LDS-direct loads into a junk area,
        // no register result, nothing waits for them.  For the 16 shifts dy of the tile
        // the epilogue reads the rows ya0 = max(0, dy) and ya1 = min(Py, Qy + dy) of IA and
        // the rows ya0 - dy, ya1 - dy of IB: four runs of at most 16 consecutive rows.
        const int dlo = dy0, dhi = min(dy0 + 15, Sy - 1 - (Qy - 1));
        const unsigned junk_off = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
            (__attribute__((address_space(3))) float*)touch_junk));
        auto touch_rows = [&](const int* tab, int ipitch, int row_lo, int row_hi) {
          const char* base = reinterpret_cast<const char*>(tab + (long long)row_lo * ipitch);
          const int n_lines = ((row_hi - row_lo + 1) * ipitch * 4 + 63) >> 6;
          for (int k = lane; k < n_lines; k += 64) {
            const char* src = base + (size_t)k * 64;
            unsigned saved_m0;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(saved_m0)
                : "v"(src), "s"(junk_off)
                : "memory");
          }
        };
        const int* IAt = a.integ[0] + b * a.integ_stride[0];
        const int* IBt = a.integ[1] + b * a.integ_stride[1];
        const int a0lo = max(0, dlo), a0hi = max(0, dhi);
        const int a1lo = min(Py, Qy + dlo), a1hi = min(Py, Qy + dhi);
        if (a0hi > 0) touch_rows(IAt, Px + 1, a0lo, a0hi);        // (row 0 is all zero: one line)
        touch_rows(IAt, Px + 1, a1lo, a1hi);
        const int b0lo = min(a0lo - dlo, a0hi - dhi), b0hi = max(a0lo - dlo, a0hi - dhi);
        const int b1lo = min(a1lo - dlo, a1hi - dhi), b1hi = max(a1lo - dlo, a1hi - dhi);
        if (b0hi > 0) touch_rows(IBt, Qx + 1, b0lo, b0hi);
        touch_rows(IBt, Qx + 1, b1lo, b1hi);
      }
      v4i acc[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = v4i{0, 0, 0, 0};
      const bool chk_early = LAZY && a.early > 0 && a.prune && !forced;
      bool abandoned = false;   // (lazy modes) given up inside the row loop: see check_after
      bool narrowed = false;    // (lazy modes) column tiles dropped inside the row loop
      int yb0 = ylo, y_checked = -1;   // row-loop position (shared by the loop variants)
      int y_next = ylo;                // row of the next in-loop test (see check_after)
      // widest narrowing the previous patch's hot column tiles allow (one tile of margin)
      int kmax_pred = 0;
      if (LAZY && chk_early && a.narrow && a.guard_x <= 16)
        kmax_pred = min(__builtin_amdgcn_readfirstlane(hd_cq) - 1, a.narrow);

      const unsigned char* ap =
          A_lds + (kPadTop + ylo + g + dy0 + n) * a.pa;
      const unsigned char* bp = B_lds + (ylo + g) * a.pb + (pos0 & ~3);
      // One row group: the B dwords of THIS group and the A fragments of the
      // NEXT group are requested together, then the NCA x NCE MFMAs run on the A
      // fragments that were prefetched a group ago -- only the first B dwords
      // are on the critical path at the loop head.  (The last prefetch reads up
      // to 8 rows past the tile's last row: still inside the workgroup's LDS,
      // never used.)
      auto row_group = [&](const v4i* af, v4i* af_next, const unsigned char* ap_next,
                           const unsigned char* bp_cur) {
        unsigned d[4 * NCE + 1];
#pragma unroll
        for (int j = 0; j < 4 * NCE + 1; ++j)
          d[j] = *reinterpret_cast<const unsigned*>(bp_cur + 4 * j);
        if (af_next) {
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca)
            af_next[ca] = *reinterpret_cast<const v4i*>(ap_next + 16 * ca);
        }
        // (one read of the caller's array per chunk: with NCA x NCE uses per call site the
        // double-buffered fragment arrays of the wide variants were left in scratch memory)
        v4i afl[NCA];
#pragma unroll
        for (int ca = 0; ca < NCA; ++ca) afl[ca] = af[ca];
#pragma unroll
        for (int c = 0; c < NCE; ++c) {
          v4i bf;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            bf[k] = static_cast<int>(
                __builtin_amdgcn_alignbyte(d[4 * c + k + 1], d[4 * c + k], sh));
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca) {
            const int q = ca - c + cq0;
            acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afl[ca], bf, acc[q], 0,
                                                           0, 0);
          }
        }
      };
      // Search-window variants: the B dwords of the NEXT row group are requested with its A
      // fragments, behind this group's own loads -- a lone wave per SIMD has no partner to
      // cover the round trip at the head of a row group (measured 20.6 cycles per matrix
      // instruction with the dwords loaded where they are used).
      constexpr int kNDW = 4 * NCE + 1;
      auto row_group_w = [&](const v4i* af, v4i* af_next, const unsigned char* ap_next,
                             const unsigned* dc, unsigned* dn, const unsigned char* bp_next) {
        if (af_next) {
#pragma unroll
          for (int j = 0; j < kNDW; ++j)
            dn[j] = *reinterpret_cast<const unsigned*>(bp_next + 4 * j);
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca)
            af_next[ca] = *reinterpret_cast<const v4i*>(ap_next + 16 * ca);
        }
        v4i afl[NCA];
        unsigned dl[kNDW];
#pragma unroll
        for (int ca = 0; ca < NCA; ++ca) afl[ca] = af[ca];
#pragma unroll
        for (int j = 0; j < kNDW; ++j) dl[j] = dc[j];
#pragma unroll
        for (int c = 0; c < NCE; ++c) {
          v4i bf;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            bf[k] = static_cast<int>(
                __builtin_amdgcn_alignbyte(dl[4 * c + k + 1], dl[4 * c + k], sh));
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca) {
            const int q = ca - c + cq0;
            acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afl[ca], bf, acc[q], 0,
                                                           0, 0);
          }
        }
      };
      // (the general P != Q epilogue needs the registers: old order there)
      if constexpr (SFM_LOOP_CA_OUTER && MODE != kModeGeneral) {
        // Software-pipelined order: A chunk outer, B fragment inner.  All NCE B
        // fragments of the row group stay in registers (4 NCE VGPRs) and every A
        // fragment is reloaded IN PLACE for the next row group right after its
        // last MFMA has been issued; the B dwords of the next group arrive in
        // small batches between the A chunks and are funnel-shifted into the
        // fragment registers behind the last chunk's MFMAs.  Nothing is waited
        // for at the loop head: the matrix pipe never drains between row
        // groups, even when the wave is alone on its SIMD.  (The prefetches of
        // the last group read up to 4 rows past the tile: inside the
        // workgroup's LDS, never used.)
        constexpr int kND = 4 * NCE + 1;
        constexpr int kPerCa = (kND + NCA - 2) / (NCA - 1);  // dwords fetched per chunk
        // A fragments live in a rotating window of kW register sets: chunk ca
        // uses set ca % kW and, once its MFMAs are issued, the set is reloaded
        // with the chunk kW positions ahead (of this row group or the next).
        constexpr int kW = (NCA > 6 && NCA % 2 == 0) ? NCA / 2 : NCA;
        static_assert(NCA % kW == 0, "window must divide the chunk count");
        // The row loop exists in up to three variants: KS outer column tiles on
        // either side are left out (exact pruning along x, see col_skip below).
        // Cold before it is finished (lazy modes).  After the patch rows yb < y the
        // accumulators hold exact partial sums; what the rows y .. yhi - 1 can still
        // add to any element of the tile is at most
        //   sqrt(E_A(rows y + dy0 .. yhi + dy0 + 14) E_B(rows y .. yhi - 1))
        // (Cauchy-Schwarz over the remaining operand bytes, whole rows: shifts and
        // the column window only shrink the sums; rows outside a patch are zero
        // padding) -- integer row energies about the operand centres from the prep
        // kernel (tb_lds[kRowPre ...], every fourth row, rounded outwards).  With the
        // bound of the mean correction added, a tile whose every element stays below
        // threshold_rel x the running maximum is exactly what the test behind the
        // loop calls cold: not hot, not a maximum, and -- unless requested -- never
        // read.  It is abandoned here, with the rest of its row loop unissued (before
        // the first row group this is the tile's own bound, without the guard band
        // the a-priori test has to include); requested later, it is recomputed like
        // any tile that finished un-stored.
        // (returns -1: abandoned; otherwise the number of outer column tiles the rest
        // of the row loop may leave out, >= ks_now)
        // Narrower in flight.  The same bound column by column: if every shift of the
        // outer K + 1 column tiles on either side (one tile of margin: guard_x <= 16)
        // already satisfies partial sum + rest + |correction|max < threshold_rel x the
        // running maximum, the K outer tiles are cold whatever the remaining rows hold,
        // and the loop continues in the variant without them; the epilogue zeroes them
        // like tiles left out a priori.  Zeros stand in for cold values wherever a cold
        // value may be read (a max-filter window); the windows that need true values
        // belong to hot elements, which this tile does not have within a tile of the
        // columns it dropped -- but another row tile of the band might.  So every tile
        // that may be hot records its hot column tiles (lz_cq), a stored tile records
        // how far it was narrowed (lz_ks), and the end-of-patch pass recomputes in full
        // what turns out to lie within a tile of a hot column (rare: the loop only
        // narrows to what the previous patch's hot columns leave, kmax_pred).
        // The tests are not free (80 integer maxima, wave reductions, a square root:
        // measured 2 300 cycles each next to the other workgroup's matrix loop, 23 %
        // of a wave's time in the tile loop when run every four row groups), so each
        // one schedules the next: the rest bound falls about linearly with the rows
        // that are left, which puts the earliest row at which the tile (or its outer
        // column tiles) can pass at  yhi - (rows left) x gap / rest  for the current
        // gap = threshold - partial maximum - |correction|max.
        auto check_after = [&](int y, int ks_now) {
          int run = 0, m_k[5] = {0, 0, 0, 0, 0};
          constexpr int kCand[5] = {col_skip_lo(NQ), col_skip_hi(NQ), col_skip_2(NQ),
                                    col_skip_3(NQ), col_skip_c(NQ)};
          if (y > ylo) {   // (before the first row group every sum is zero)
#pragma unroll
            for (int j = 0; j < (NQ + 1) / 2; ++j) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                run = max(run, acc[j][r]);
                run = max(run, acc[NQ - 1 - j][r]);
              }
#pragma unroll
              for (int i = 0; i < 5; ++i)
                if (kCand[i] == j) m_k[i] = run;   // tiles q <= K and q >= NQ - 1 - K
            }
          }
          const int ms = wave_max_nonneg_i(run);
          const unsigned* rp = reinterpret_cast<const unsigned*>(tb_lds + kRowPre);
          const int a_lo = max(0, y + dy0), a_hi = min(Py, yhi + dy0 + 15);
          const unsigned ea = rp[(a_hi + 3) >> 2] - rp[a_lo >> 2];
          const unsigned eb = rp[64 + ((yhi + 3) >> 2)] - rp[64 + (y >> 2)];
          // (margins: a few ulp of the product, the root and the two sums)
          const float rest = sqrtf(__uint2float_ru(ea) * __uint2float_ru(eb)) * 1.000002f + 2.f;
          const float corr = tb_lds[kBoundCorr];
          const float ub = (__int2float_ru(ms) + rest + corr) * 1.000002f + 2.f;
          const float thr_now = a.threshold_rel * __int_as_float(__builtin_amdgcn_readfirstlane(
                                                      *const_cast<volatile int*>(pmax_lds)));
          const int req = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&lz[0]));
          const bool requested = (req >> p) & 1;
          if (ub < thr_now && !requested) {
            if (lane == 0) {
              lz_tmax[p] = ub;
              atomicOr(&lz[1], 1 << p);
            }
            return -1;
          }
          // earliest row at which a set of columns with partial maximum m can pass
          const float left = static_cast<float>(yhi - y);
          auto passes_at = [&](int m) {
            const float gap = thr_now - (__int2float_ru(m) + corr);
            return gap > 0.f ? yhi - static_cast<int>(left * (gap / rest)) : yhi;
          };
          int y_at = requested ? yhi : passes_at(ms);
          // (before the first row group nothing is known about the columns: the first
          // row at which anything can pass -- requested tiles are tested for narrowing only)
          if (y == ylo && kmax_pred > ks_now) y_at = min(y_at, passes_at(0));
          int ks_new = ks_now;
          if (y > ylo && kmax_pred > ks_now) {
            bool open = true;   // widest first: the sets are nested
#pragma unroll
            for (int i = 4; i >= 0; --i) {
              if (i < 4 && kCand[i] == kCand[i + 1]) continue;
              if (kCand[i] <= 0 || 2 * kCand[i] + 2 > NQ) continue;
              if (!open || kCand[i] <= ks_now || kCand[i] > kmax_pred) continue;   // (wave-uniform)
              const int mo = wave_max_nonneg_i(m_k[i]);
              const float ubk = (__int2float_ru(mo) + rest + corr) * 1.000002f + 2.f;
              if (ubk < thr_now) {
                ks_new = kCand[i];
                open = false;
              } else {
                y_at = min(y_at, passes_at(mo));
              }
            }
          }
          y_next = max(y + 4 * a.early, ylo + ((y_at - ylo + 3) & ~3));
          return ks_new;
        };
        auto rows = [&](auto ks_const) {
        constexpr int KS = decltype(ks_const)::value;
        v4i af[kW], bf[NCE];
        unsigned dn[kND];
#pragma unroll
        for (int j = 0; j < kND; ++j) dn[j] = *reinterpret_cast<const unsigned*>(bp + 4 * j);
#pragma unroll
        for (int w = 0; w < kW; ++w)
          af[w] = *reinterpret_cast<const v4i*>(ap + 16 * w);
#pragma unroll
        for (int c = 0; c < NCE; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            bf[c][k] = static_cast<int>(
                __builtin_amdgcn_alignbyte(dn[4 * c + k + 1], dn[4 * c + k], sh));
        // Lazy modes: every a.early row groups (and before the first) the tile is
        // tested for being provably cold already, as a whole or in its outer column
        // tiles -- see check_after above.
        const int y_enter = yb0;
        for (;;) {
        int yseg = yhi;
        if (LAZY && chk_early) {
          if (yhi - yb0 > 8 && yb0 >= y_next && yb0 != y_checked) {
            y_checked = yb0;
            TICK(12)   // (timing build) row groups (and the loop prologue)
            const int ks_new = check_after(yb0, KS);
            TICK(11)   // (timing build) the in-loop tests
            if (ks_new < 0) {
              abandoned = true;
              break;
            }
            if (ks_new > KS) {   // the rest of the tile runs in a narrower variant
              col_skip = ks_new;
              narrowed = true;
              break;
            }
          }
          yseg = min(yhi, max(y_next, yb0 + 4));
        }
        for (; yb0 < yseg; yb0 += 4) {
          bp += 4 * a.pb;
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca) {
#if SFM_LOOP_INTERLEAVE
            // the register set of the PREVIOUS chunk is free now: its reload and
            // this chunk's share of the next B dwords can sit in the MFMA gaps
            // (one LDS instruction per gap) instead of in a burst behind them
            if (ca > 0) {
              const int cp = ca - 1;
              if (cp + kW < NCA)
                af[cp % kW] = *reinterpret_cast<const v4i*>(ap + 16 * (cp + kW));
              else
                af[cp % kW] =
                    *reinterpret_cast<const v4i*>(ap + 4 * a.pa + 16 * (cp + kW - NCA));
            }
            if (ca < NCA - 1) {
#pragma unroll
              for (int j = ca * kPerCa; j < (ca + 1) * kPerCa && j < kND; ++j)
                dn[j] = *reinterpret_cast<const unsigned*>(bp + 4 * j);
            }
#endif
#pragma unroll
            for (int c = 0; c < NCE; ++c) {
              const int q = ca - c + cq0;
              if (q >= KS && q < NQ - KS)
                acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[ca % kW], bf[c], acc[q], 0,
                                                               0, 0);
              if (ca == NCA - 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  bf[c][k] = static_cast<int>(
                      __builtin_amdgcn_alignbyte(dn[4 * c + k + 1], dn[4 * c + k], sh));
              }
            }
#if SFM_LOOP_INTERLEAVE
            if (ca == NCA - 1)  // the last chunk's set: nothing follows in this group
              af[ca % kW] =
                  *reinterpret_cast<const v4i*>(ap + 4 * a.pa + 16 * (ca + kW - NCA));
            if (ca < NCA - 1) {
              // MFMAs of this chunk: columns q = ca - c + cq0 inside [KS, NQ - KS)
              const int c_lo = ca + cq0 - (NQ - KS - 1) > 0 ? ca + cq0 - (NQ - KS - 1) : 0;
              const int c_hi = ca + cq0 - KS < NCE - 1 ? ca + cq0 - KS : NCE - 1;
              const int n_mfma = c_hi - c_lo + 1;
              // (the builtin wants literal counts; ca is a constant after unrolling)
#define SFM_PAIR(i)                                                             \
  if (n_mfma > i) {                                                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA */           \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* one LDS read */       \
  }
              SFM_PAIR(0) SFM_PAIR(1) SFM_PAIR(2) SFM_PAIR(3)
#undef SFM_PAIR
              switch (n_mfma - 4) {
                case 1: __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); break;
                case 2: __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); break;
                case 3: __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); break;
                case 4: __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); break;
                case 5: __builtin_amdgcn_sched_group_barrier(0x008, 5, 0); break;
                case 6: __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); break;
                case 7: __builtin_amdgcn_sched_group_barrier(0x008, 7, 0); break;
                case 8: __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); break;
                default: break;
              }
              static_assert(NCE - 4 <= 8, "extend the switch");
            }
#else
            if (ca + kW < NCA)
              af[ca % kW] = *reinterpret_cast<const v4i*>(ap + 16 * (ca + kW));
            else
              af[ca % kW] =
                  *reinterpret_cast<const v4i*>(ap + 4 * a.pa + 16 * (ca + kW - NCA));
            if (ca < NCA - 1) {
#pragma unroll
              for (int j = ca * kPerCa; j < (ca + 1) * kPerCa && j < kND; ++j)
                dn[j] = *reinterpret_cast<const unsigned*>(bp + 4 * j);
            }
#endif
            // keep the chunk order: the machine scheduler would otherwise sink
            // every prefetch to the end of the body (= no prefetch distance)
            __builtin_amdgcn_sched_barrier(0);
          }
          ap += 4 * a.pa;
        }
        if (yb0 >= yhi) break;
        }
        {
          // instructions of one row group of this variant: (ca, c) pairs whose
          // column tile q = ca - c + cq0 is kept
          int per_group = 0;
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
            for (int c = 0; c < NCE; ++c)
              per_group += (ca - c + NCE - 1 >= KS && ca - c + NCE - 1 < NQ - KS) ? 1 : 0;
          mfma_issued += (long long)((yb0 - y_enter) / 4) * per_group;  // (groups issued)
        }
        };
        constexpr int kKs1 = col_skip_lo(NQ), kKs2 = col_skip_hi(NQ);
        constexpr int kKs3 = col_skip_2(NQ), kKs4 = col_skip_3(NQ);
        constexpr int kKs5 = (LAZY && 2 * col_skip_c(NQ) + 2 <= NQ) ? col_skip_c(NQ) : 0;
        static_assert(kKs1 <= kKs2 && kKs2 <= kKs3 && kKs3 <= kKs4, "ascending");
        // (a variant returns when the tile is finished, abandoned or to be continued
        // in a narrower variant)
        TICK(10)   // (timing build) tile draw, pruning tests, setup
        // ascending and straight-line: a tile only ever moves to a narrower variant
        // (a loop around one dispatch made the register allocator spill 255 VGPRs)
        // (the test before the first row group runs here: a tile it abandons never
        // loads the loop's first fragments)
        if (LAZY && chk_early && yhi - ylo > 8) {
          // check_after(ylo) on the values read with the pruning bounds: every partial
          // sum is zero, the bound is the tile's own a-priori bound
          y_checked = ylo;
          const float rest = sqrtf(__uint2float_ru(hd_ea) * __uint2float_ru(hd_eb)) * 1.000002f + 2.f;
          const float ub = (rest + hd_corr) * 1.000002f + 2.f;
          const bool requested = (__builtin_amdgcn_readfirstlane(hd_req) >> p) & 1;
          if (ub < hd_thr && !requested) {
            if (lane == 0) {
              lz_tmax[p] = ub;
              atomicOr(&lz[1], 1 << p);
            }
            abandoned = true;
          } else {
            const float gap = hd_thr - hd_corr;
            const int y0_at = gap > 0.f ? yhi - static_cast<int>(static_cast<float>(yhi - ylo) * (gap / rest)) : yhi;
            const int y_at = (requested && kmax_pred <= col_skip) ? yhi : y0_at;
            y_next = max(ylo + 4 * a.early, ylo + ((y_at - ylo + 3) & ~3));
          }
        }
        if (!abandoned && (col_skip == 0 || (kKs1 == 0 && col_skip == kKs1)))
          rows(std::integral_constant<int, 0>{});
        if (kKs1 > 0 && col_skip == kKs1 && !abandoned && yb0 < yhi)
          rows(std::integral_constant<int, kKs1>{});
        if (kKs2 > kKs1 && col_skip == kKs2 && !abandoned && yb0 < yhi)
          rows(std::integral_constant<int, kKs2>{});
        if (kKs3 > kKs2 && col_skip == kKs3 && !abandoned && yb0 < yhi)
          rows(std::integral_constant<int, kKs3>{});
        if (kKs4 > kKs3 && col_skip == kKs4 && !abandoned && yb0 < yhi)
          rows(std::integral_constant<int, kKs4>{});
        if (kKs5 > kKs4 && col_skip == kKs5 && !abandoned && yb0 < yhi)
          rows(std::integral_constant<int, kKs5>{});
      } else if constexpr (WIDE8) {
        auto half_rows = [&](auto h_const) {
          constexpr int H = decltype(h_const)::value;
          constexpr int Q0 = H ? NQH : 0, Q1 = H ? NQ : NQH;
          // A chunks with a pair (ca, c) whose column tile ca - c + kCq0 lies in [Q0, Q1)
          constexpr int CA0 = Q0 - kCq0 > 0 ? Q0 - kCq0 : 0;
          constexpr int CA1 = Q1 - 1 < NCA - 1 ? Q1 - 1 : NCA - 1;
          constexpr int NC = CA1 - CA0 + 1;
          int n_pairs = 0;
          // (the A chunks of the next row group are requested behind this group's B dwords:
          // two register sets, swapped by unrolling two groups per trip)
          auto group = [&](const v4i* af, v4i* af_next) {
            unsigned d[kNDW];
#pragma unroll
            for (int j = 0; j < kNDW; ++j) d[j] = *reinterpret_cast<const unsigned*>(bp + 4 * j);
            if (af_next) {
#pragma unroll
              for (int i = 0; i < NC; ++i)
                af_next[i] = *reinterpret_cast<const v4i*>(ap + 4 * a.pa + 16 * (CA0 + i));
            }
            v4i afl[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) afl[i] = af[i];
#pragma unroll
            for (int c = 0; c < NCE; ++c) {
              v4i bf;
#pragma unroll
              for (int k = 0; k < 4; ++k)
                bf[k] = static_cast<int>(
                    __builtin_amdgcn_alignbyte(d[4 * c + k + 1], d[4 * c + k], sh));
#pragma unroll
              for (int i = 0; i < NC; ++i) {
                const int q = CA0 + i - c + kCq0;
                if (q >= Q0 && q < Q1)
                  acc[q - Q0] =
                      __builtin_amdgcn_mfma_i32_16x16x64_i8(afl[i], bf, acc[q - Q0], 0, 0, 0);
              }
            }
            ap += 4 * a.pa;
            bp += 4 * a.pb;
          };
          if constexpr (SFM_WIDE_WAVES <= 8) {
            v4i afA[NC], afB[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i)
              afA[i] = *reinterpret_cast<const v4i*>(ap + 16 * (CA0 + i));
            int yy = ylo;
            for (; yy + 4 < yhi; yy += 8) {
              group(afA, afB);
              group(afB, afA);
            }
            if (yy < yhi) group(afA, nullptr);
          } else {   // three waves per SIMD (170 registers each): no second fragment set
            for (int yy = ylo; yy < yhi; yy += 4) {
              v4i af1[NC];
#pragma unroll
              for (int i = 0; i < NC; ++i)
                af1[i] = *reinterpret_cast<const v4i*>(ap + 16 * (CA0 + i));
              group(af1, nullptr);
            }
          }
#pragma unroll
          for (int c = 0; c < NCE; ++c)
#pragma unroll
            for (int i = 0; i < NC; ++i)
              n_pairs += (CA0 + i - c + kCq0 >= Q0 && CA0 + i - c + kCq0 < Q1) ? 1 : 0;
          mfma_issued += (long long)((yhi - ylo + 3) / 4) * n_pairs;
        };
        if (half)
          half_rows(std::integral_constant<int, 1>{});
        else
          half_rows(std::integral_constant<int, 0>{});
      } else if constexpr (NCA > 10 && NCA <= SFM_WIDE_BPREFETCH_MAX) {
        v4i afA[NCA], afB[NCA];
        unsigned dA[kNDW], dB[kNDW];
#pragma unroll
        for (int ca = 0; ca < NCA; ++ca)
          afA[ca] = *reinterpret_cast<const v4i*>(ap + 16 * ca);
#pragma unroll
        for (int j = 0; j < kNDW; ++j) dA[j] = *reinterpret_cast<const unsigned*>(bp + 4 * j);
        int yb0 = ylo;
        for (; yb0 + 4 < yhi; yb0 += 8) {
          row_group_w(afA, afB, ap + 4 * a.pa, dA, dB, bp + 4 * a.pb);
          row_group_w(afB, afA, ap + 8 * a.pa, dB, dA, bp + 8 * a.pb);
          ap += 8 * a.pa;
          bp += 8 * a.pb;
        }
        if (yb0 < yhi) row_group_w(afA, nullptr, nullptr, dA, nullptr, nullptr);
        mfma_issued += (long long)((yhi - ylo + 3) / 4) * (NCA * NCE);
      } else if constexpr (SFM_AF_PREFETCH || NCA > 10) {
        v4i afA[NCA], afB[NCA];
#pragma unroll
        for (int ca = 0; ca < NCA; ++ca)
          afA[ca] = *reinterpret_cast<const v4i*>(ap + 16 * ca);
        int yb0 = ylo;
        for (; yb0 + 4 < yhi; yb0 += 8) {
          row_group(afA, afB, ap + 4 * a.pa, bp);
          row_group(afB, afA, ap + 8 * a.pa, bp + 4 * a.pb);
          ap += 8 * a.pa;
          bp += 8 * a.pb;
        }
        if (yb0 < yhi) row_group(afA, nullptr, nullptr, bp);
        mfma_issued += (long long)((yhi - ylo + 3) / 4) * (NCA * NCE);
      } else {
        for (int yb0 = ylo; yb0 < yhi; yb0 += 4) {
          v4i af[NCA];
#pragma unroll
          for (int ca = 0; ca < NCA; ++ca)
            af[ca] = *reinterpret_cast<const v4i*>(ap + 16 * ca);
          row_group(af, nullptr, nullptr, bp);
          ap += 4 * a.pa;
          bp += 4 * a.pb;
        }
        mfma_issued += (long long)((yhi - ylo + 3) / 4) * (NCA * NCE);
      }

      TICK(2)
      if (LAZY && abandoned) {
        ++tiles_early;
        continue;
      }
      if constexpr (LAZY) {
        // Cold tile, known before its epilogue: every output is S + correction, the
        // integer sums S are in the accumulators and |correction| has the patch-wide
        // bound of the seed probe (tbound[kBoundCorr]).  If max S + that bound stays
        // below threshold_rel x the running maximum, no element of the tile is hot,
        // none raises the maximum, and -- unless a hot tile asked for it -- nothing
        // of it is stored: the epilogue (40 table gathers and ~400 VALU operations
        // per lane) has nothing to deliver.  Asked for later after all, it is
        // recomputed like any band tile that finished un-stored.
        if (a.prune && !forced) {
          int ms = 0;   // (max(S, 0): a bound of max S all the same)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) ms = max(ms, acc[q][r]);
          ms = wave_max_nonneg_i(ms);
          const float ub = __int2float_ru(ms) + tb_lds[kBoundCorr];
          const float mrun = __int_as_float(__builtin_amdgcn_readfirstlane(
              *const_cast<volatile int*>(pmax_lds)));
          const int req = __builtin_amdgcn_readfirstlane(*const_cast<volatile int*>(&lz[0]));
          if (ub < a.threshold_rel * mrun && !((req >> p) & 1)) {
            if (lane == 0) {
              lz_tmax[p] = ub;
              atomicOr(&lz[1], 1 << p);
            }
            continue;
          }
        }
      }
      if constexpr (LAZYG) {
        // The 16 rows of the correction table this tile's epilogue reads, built here
        // (the prep kernel of this mode leaves no table): rows yv = 16 Ib + j + 1, j < 16,
        // Ib = p mod (Py / 16), of
        //   G[yv][xv] = g_entry( IA'[yv][xv], IB'[Py - yv][Px - xv] ),
        //   IA'[yv][xv]           = sum_{x < xv} colA[x],  colA = c16[0][Ib] + rows 16 Ib .. yv - 1 of a'
        //   IB'[Py - yv][Px - xv] = T - sum_{x >= Px - xv} colB[x],
        //                           colB = c16[1][NB - Ib] - rows Py - yv .. 16 (NB - Ib) - 1 of b'
        // (a', b': the int8 operand copies in LDS; c16: centred column sums above every
        // 16th row, prep kernel; T = sum of colB over all columns).  This is the prep
        // kernel's sweep over 16 rows, run by the wave that needs them: lane l owns the
        // four columns 4 l .. 4 l + 3 of the pre patch and the MIRRORED columns of the post
        // patch (one aligned LDS dword per row and side), keeps their running column sums,
        // and a wave scan over the lane totals gives the exclusive prefixes, so the lane
        // leaves G[yv][4 l .. 4 l + 3] as one 16-byte store.  The same integers as the
        // table of the other modes, the same expression (g_entry): the same bits.  The rows
        // go to this patch's G area, now a scratch that stays in L2: every entry the
        // epilogue below gathers is stored here by this wave first.
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz) : : "memory");
        const int NB = Py >> 4;
        const int Ib = p < NB ? p : p - NB;
        const int* c16 = a.c16 + b * a.c16_stride;
        const int l4 = 4 * lane + oz;
        const bool act = l4 < Px;
        const int xa = act ? l4 : 0;              // (idle lanes read column 0 and store nothing)
        const int xb = Px - 4 - xa;               // first of the four mirrored columns
        v4i ca = *reinterpret_cast<const v4i*>(c16 + Ib * Px + xa);
        const v4i cbn = *reinterpret_cast<const v4i*>(c16 + (NB + 1 + NB - Ib) * Px + xb);
        // tile NB - 1: its last row is dy = 0, i.e. yv = 0: IA' = 0 and IB' over ALL rows
        v4i call = v4i{0, 0, 0, 0};
        if (p == NB - 1) call = *reinterpret_cast<const v4i*>(c16 + (NB + 1 + NB) * Px + xb);
        int cb[4] = {cbn[3], cbn[2], cbn[1], cbn[0]};     // mirrored order
        if (!act) {
          ca = v4i{0, 0, 0, 0};
          call = v4i{0, 0, 0, 0};
#pragma unroll
          for (int k = 0; k < 4; ++k) cb[k] = 0;
        }
        const unsigned char* ap = A_lds + (kPadTop + 16 * Ib) * a.pa + xa;
        const unsigned char* bp = B_lds + (16 * (NB - Ib) - 1) * a.pb + a.ml + xb;
        float* Gw = a.gtab + (long long)b * Py * Px + xa;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
          const unsigned da = *reinterpret_cast<const unsigned*>(ap + j * a.pa);
          const unsigned db = *reinterpret_cast<const unsigned*>(bp - j * a.pb);
          if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              ca[k] += static_cast<int>(static_cast<signed char>(da >> (8 * k)));
              cb[k] -= static_cast<int>(static_cast<signed char>(db >> (8 * (3 - k))));
            }
          }
          const bool zero_row = p == NB - 1 && j == 15;      // yv = 0 (wave-uniform)
          int va[4], vb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            va[k] = zero_row ? 0 : ca[k];
            vb[k] = zero_row ? call[3 - k] : cb[k];
          }
          const int sa = va[0] + va[1] + va[2] + va[3], sb = vb[0] + vb[1] + vb[2] + vb[3];
          const int inc_a = wave_scan_incl(sa), inc_b = wave_scan_incl(sb);
          const int tb = __builtin_amdgcn_readlane(inc_b, 63);   // T = IB'[Py - yv][Px]
          int ia = inc_a - sa, ib = tb - (inc_b - sb);
          v4i out;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            out[k] = __float_as_int(g_entry(mua, mub, static_cast<float>(ia), static_cast<float>(ib)));
            ia += va[k];
            ib -= vb[k];
          }
          const int yv = zero_row ? 0 : 16 * Ib + j + 1;
          // (row 15 of the last tile is padding: there is no row yv = Py)
          if (act && yv < Py) *reinterpret_cast<v4i*>(Gw + yv * Px) = out;
        }
        // The epilogue below gathers these rows back (other lanes' stores included) through
        // loads the compiler cannot relate to the stores above: every store of this wave has
        // to have reached the L2 (vector memory is write-through) before the first gather
        // is issued.  (Tiles p and p + NB of a patch write the SAME values into the same
        // rows, possibly from two waves at once: duplicate writers of identical bits.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // The epilogue's table addresses do not depend on the MFMA loop; without
      // this opaque zero the compiler hoists its ~100 gathers above the loop
      // and spills the accumulators.  After the loop there are >140 free VGPRs,
      // so all gathers of a tile can be in flight together.
      int opaque_zero;
      asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero) : : "memory");
      // Epilogue: lane holds rows ky = 16 p + 4 g + r, columns kx = 16 q + n.
      // Branch-free: indices are clamped so every table load is in bounds;
      // only the store is predicated.
      // The surface buffer is padded to whole tiles (pitch 16 NQ, 16 NP rows),
      // so every lane stores unconditionally.
      int srow[4];
      bool rowok[4];
      float tmax = 0.f;  // max(tile, 0): only positive maxima matter
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        srow[r] = (16 * p + 4 * g + r) * a.sx_pitch + n;
        rowok[r] = 16 * p + 4 * g + r < Sy;
      }
#ifdef SFM_ABLATE_EPILOGUE
      // timing experiment only (results are garbage): how fast is the kernel
      // when a tile costs nothing after its matrix loop?
      {
        int sink = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) sink ^= acc[q][0] ^ acc[q][1] ^ acc[q][2] ^ acc[q][3];
        if (sink == 0x7fffffff) a.surface[0] = 1.f;
        continue;
      }
#endif
      // One column tile's share of the hot list (see the tile's peak pass below).
      int hq_lo = NQ, hq_hi = -1;   // column tiles with elements that may be hot
      auto hot_insert = [&](const int q, const float v0, const float v1, const float v2,
                            const float v3, const float thr_t) {
        float* hv = a.hot_val + (long long)b * a.hot_cap;
        int* hi = a.hot_idx + (long long)b * a.hot_cap;
        const float vv[4] = {v0, v1, v2, v3};
        const float vm = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
        if (__any(vm > thr_t)) {  // wave-uniform, rarely taken
          hq_lo = min(hq_lo, q);
          hq_hi = q;
          // one list reservation per column tile (ballots + one LDS atomic): an
          // atomic per lane and row was four dependent LDS round trips here
          unsigned long long bal[4];
          int total = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bal[r] = __ballot(vv[r] > thr_t);
            total += __builtin_popcountll(bal[r]);
          }
          int base = 0;
          if (lane == 0) base = atomicAdd(hot_lds, total);
          base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = vv[r];
            if (v > thr_t) {
              const int slot = base + static_cast<int>(__builtin_amdgcn_mbcnt_hi(
                                          static_cast<unsigned>(bal[r] >> 32),
                                          __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(bal[r]), 0u)));
              if (slot < a.hot_cap) {
                hv[slot] = v;
                hi[slot] = (16 * p + 4 * g + r) * Sx + 16 * q + n;
              }
            }
            base += __builtin_popcountll(bal[r]);
          }
        }
      };
      // kModeGeneral, search-window variants (NCA > 10): the epilogue (pass 0) and the
      // hot-list pass (pass 1) over the column tiles of this row tile.
      // Search-window variants: a lone wave per SIMD pays every round trip of its table
      // look-ups in full, and eight of them per output (two box sums from two integral
      // images) made the epilogue as long as the matrix loop.  With Px >= Qx a shift dx
      // is in one of three regimes, and in each of them half of the eight corner terms
      // are a row constant or zero:
      //   L  dx <= 0:            xa0 = 0 (IA[.][0] = 0),  xb1 = Qx (row total of B)
      //   M  0 < dx < Px - Qx:   xb0 = 0,  xb1 = Qx: the post patch lies inside -- SB is
      //                          the row total, SA the only box with four corners
      //   R  dx >= Px - Qx:      xa1 = Px (row total of A),  xb0 = 0
      // so FOUR gathers per output are enough, selected per lane without a branch:
      //   v0, v1 = IA[oa1 | oa0][L, M: xa1;  R: xa0]
      //   v2, v3 = L: IB[ob1 | ob0][xb0]   M: IA[oa1 | oa0][xa0]   R: IB[ob1 | ob0][xb1]
      //   SA = L: v0 - v1   M: (v0 - v1) - (v2 - v3)   R: total_A - (v0 - v1)
      //   SB = L: total_B - (v2 - v3)   M: total_B   R: v2 - v3
      // -- the same integers as the eight-corner form, so the same bits.  The gathers of
      // a group of column tiles are all in flight before the previous group is consumed.
      auto wide_general = [&](const int pass, const float thr) {
        if constexpr (NCA > 10 && MODE == kModeGeneral) {
          const int ipa = Px + 1, ipb = Qx + 1;
          int oa0[4], oa1[4], ob0[4], ob1[4], ny[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ky = min(16 * p + 4 * g + r, Sy - 1);
            const int dy = ky - (Qy - 1);
            const int ya0 = max(0, dy), ya1 = min(Py, Qy + dy);
            oa0[r] = ya0 * ipa + opaque_zero;
            oa1[r] = ya1 * ipa;
            ob0[r] = (ya0 - dy) * ipb + opaque_zero;
            ob1[r] = (ya1 - dy) * ipb;
            ny[r] = ya1 - ya0;
          }
          int ta[4], tb[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ta[r] = IA[oa1[r] + Px] - IA[oa0[r] + Px];
            tb[r] = IB[ob1[r] + Qx] - IB[ob0[r] + Qx];
          }
          // (regimes as all-ones / zero lane masks: nested selects came out as divergent
          // branches, one basic block per output)
          const long long ib_delta = IB - IA;   // IB's entries addressed from IA
          // The column tiles are walked by a RUN-TIME loop, two tiles per trip: unrolled over
          // 25 .. 30 tiles the scheduler hoisted every gather of a tile's 100 .. 120 outputs
          // per lane to the front and spilled a thousand registers.  The accumulator tile of
          // a run-time q is reached through a switch over its compile-time names (a compare
          // tree on a scalar: the accumulators must never be indexed dynamically) and is
          // only ever READ here -- a loop that also wrote the finished values back through
          // such a switch left every register array of the kernel in scratch memory.  The
          // hot-list pass (pass 1, only for a tile that may hold a hot element: a few per
          // patch) therefore recomputes its values from the integer sums instead.
          auto acc_get = [&](int q, int* dst) {
            switch (q) {
#define SFM_ACC_CASE(k)                                                   \
  case k:                                                                 \
    if constexpr (k < NQ) {                                               \
      dst[0] = acc[k][0]; dst[1] = acc[k][1]; dst[2] = acc[k][2]; dst[3] = acc[k][3]; \
    }                                                                     \
    break;
              SFM_ACC_CASE(0) SFM_ACC_CASE(1) SFM_ACC_CASE(2) SFM_ACC_CASE(3) SFM_ACC_CASE(4)
              SFM_ACC_CASE(5) SFM_ACC_CASE(6) SFM_ACC_CASE(7) SFM_ACC_CASE(8) SFM_ACC_CASE(9)
              SFM_ACC_CASE(10) SFM_ACC_CASE(11) SFM_ACC_CASE(12) SFM_ACC_CASE(13) SFM_ACC_CASE(14)
              SFM_ACC_CASE(15) SFM_ACC_CASE(16) SFM_ACC_CASE(17) SFM_ACC_CASE(18) SFM_ACC_CASE(19)
              SFM_ACC_CASE(20) SFM_ACC_CASE(21) SFM_ACC_CASE(22) SFM_ACC_CASE(23) SFM_ACC_CASE(24)
              SFM_ACC_CASE(25) SFM_ACC_CASE(26) SFM_ACC_CASE(27) SFM_ACC_CASE(28) SFM_ACC_CASE(29)
#undef SFM_ACC_CASE
              default: break;
            }
          };
          static_assert(NQ <= 30, "extend the accumulator switch");
          const int qb_half = WIDE8 ? half * NQH : 0;
          const int qe_half = WIDE8 ? min(NQ, qb_half + NQH) : NQ;
          // One trip: the column tiles q0 .. q0 + kT - 1 of the run that ends at hi.  REG:
          // the regime every lane of both tiles is in (0 L, 1 M, 2 R: scalar table bases,
          // 32-bit byte offsets, no selects) or 3 (the one or two tiles a regime boundary
          // runs through: per-lane masks).
          constexpr int kT = SFM_WIDE_TRIP;   // column tiles per trip (16 kT gathers in flight)
          // (REG 0 .. 2: consecutive tiles are 16 columns = 64 bytes apart in every table row
          // and in the surface, so a trip forms its addresses once, for its first tile, and
          // reaches the others through the instructions' immediate offsets)
          auto trip = [&](auto reg_const, auto t_const, const int q0) {
            constexpr int REG = decltype(reg_const)::value;
            constexpr int T = decltype(t_const)::value;
            int sv[T][4], gv[T][4][4], mLs[T], mRs[T], xw[T];
#pragma unroll
            for (int u = 0; u < T; ++u) acc_get(q0 + u - qb_half, sv[u]);
            if constexpr (REG == 3) {
#pragma unroll
              for (int u = 0; u < T; ++u) {
                const int q = q0 + u;
                const int kx = 16 * q + n;
                const int dx = min(kx, Sx - 1) - (Qx - 1);
                const int xa0 = max(0, dx), xa1 = min(Px, Qx + dx);
                xw[u] = xa1 - xa0;
                const int mL = -static_cast<int>(dx <= 0);
                const int mR = ~mL & -static_cast<int>(dx >= Px - Qx);
                const int mM = ~(mL | mR);
                const int xb0 = xa0 - dx, xb1 = xa1 - dx;
                const int c01 = (xa0 & mR) | (xa1 & ~mR);
                const int c23 = (xb0 & mL) | (xa0 & mM) | (xb1 & mR);
                const long long d23 = ib_delta & ~static_cast<long long>(mM);
                mLs[u] = mL;
                mRs[u] = mR;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  gv[u][r][0] = IA[oa1[r] + c01];
                  gv[u][r][1] = IA[oa0[r] + c01];
                  gv[u][r][2] = IA[d23 + (((oa1[r] & mM) | (ob1[r] & ~mM)) + c23)];
                  gv[u][r][3] = IA[d23 + (((oa0[r] & mM) | (ob0[r] & ~mM)) + c23)];
                }
              }
            } else {
              const int dx0 = 16 * q0 + n - (Qx - 1);     // (no clamped column in these runs)
              // L: IA[.][xa1 = Qx + dx], IB[.][xb0 = -dx]   M: IA[.][Qx + dx], IA[.][xa0 = dx]
              // R: IA[.][xa0 = dx], IB[.][xb1 = Px - dx]   -- step per tile: +16, except the
              // two IB columns, which step -16
              const int c01 = REG == 2 ? dx0 : Qx + dx0;
              const int c23 = REG == 0 ? -dx0 : (REG == 1 ? dx0 : Px - dx0);
              constexpr int kStep23 = REG == 1 ? 16 : -16;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int* p0 = IA + (oa1[r] + c01);
                const int* p1 = IA + (oa0[r] + c01);
                const int* p2 = REG == 1 ? IA + (oa1[r] + c23) : IB + (ob1[r] + c23);
                const int* p3 = REG == 1 ? IA + (oa0[r] + c23) : IB + (ob0[r] + c23);
#pragma unroll
                for (int u = 0; u < T; ++u) {
                  gv[u][r][0] = p0[16 * u];
                  gv[u][r][1] = p1[16 * u];
                  gv[u][r][2] = p2[kStep23 * u];
                  gv[u][r][3] = p3[kStep23 * u];
                }
              }
#pragma unroll
              for (int u = 0; u < T; ++u) {
                const int dx = dx0 + 16 * u;
                xw[u] = REG == 0 ? Qx + dx : (REG == 1 ? Qx : Px - dx);
              }
            }
            float* srow_p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) srow_p[r] = surf + (srow[r] + 16 * q0);
#pragma unroll
            for (int u = 0; u < T; ++u) {
              const int q = q0 + u;
              const int kx = 16 * q + n;
              const float fnx = static_cast<float>(xw[u]);
              float outv[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int d01 = gv[u][r][0] - gv[u][r][1], d23 = gv[u][r][2] - gv[u][r][3];
                int sa, sb;
                if constexpr (REG == 3) {
                  const int mL = mLs[u], mR = mRs[u], mM = ~(mL | mR);
                  sa = (d01 & ~mR) + ((ta[r] - d01) & mR) - (d23 & mM);
                  sb = ((tb[r] - d23) & mL) + (tb[r] & mM) + (d23 & mR);
                } else {
                  sa = REG == 0 ? d01 : (REG == 1 ? d01 - d23 : ta[r] - d01);
                  sb = REG == 0 ? tb[r] - d23 : (REG == 1 ? tb[r] : d23);
                }
                // (explicit fused operations: the passes and regimes are copies of this code,
                // and the hot list must hold the very bits pass 0 stored -- the first-peak
                // kernel compares a hot value with the stored surface around it)
                float v = static_cast<float>(sv[u][r]);
                v = __builtin_fmaf(-mua, static_cast<float>(sb), v);
                v = __builtin_fmaf(-mub, static_cast<float>(sa), v);
                v = __builtin_fmaf(muab, __fmul_rn(static_cast<float>(ny[r]), fnx), v);
                // (only the last column tile has columns past the surface)
                const bool ok = REG == 3 ? rowok[r] && (q < NQ - 1 || kx < Sx) : rowok[r];
                if (pass == 0) {
                  __builtin_nontemporal_store(v, srow_p[r] + 16 * u);
                  tmax = fmaxf(tmax, ok ? v : 0.f);
                }
                outv[r] = ok ? v : -INFINITY;
              }
              if (pass == 1) hot_insert(q, outv[0], outv[1], outv[2], outv[3], thr);
            }
          };
          auto run = [&](auto reg_const, int lo, int hi) {   // tiles [lo, hi)
            int q0 = lo;
#pragma unroll 1
            for (; q0 + kT <= hi; q0 += kT) trip(reg_const, std::integral_constant<int, kT>{}, q0);
#pragma unroll 1
            for (; q0 < hi; ++q0) trip(reg_const, std::integral_constant<int, 1>{}, q0);
          };
          // Column tiles by regime (dx = 16 q + n - (Qx - 1), n = 0 .. 15; the clamped
          // columns past the surface count as R):
          //   L  every lane dx <= 0:            q < qL
          //   M  every lane 0 < dx < Px - Qx:   qM0 <= q < qM1
          //   R  every lane dx >= Px - Qx:      q >= qR
          const int qL = Qx >= 16 ? min((Qx - 16) / 16 + 1, NQ) : 0;
          const int qM0 = min((Qx + 15) / 16, NQ);
          const int qM1 = min(max(Px >= 17 ? (Px - 17) / 16 + 1 : 0, qM0), NQ);
          // (tiles with columns past the surface -- from column Sx on; several tiles when the
          // variant is wider than the patch -- take the per-lane code: their shifts are clamped)
          const int qc = min(Sx / 16, NQ);
          const int qR = min(max((Px + 14) / 16, qM1), qc);
          // (WIDE8: this job's column half only)
          auto run_c = [&](auto reg_const, int lo, int hi) {
            run(reg_const, max(lo, qb_half), min(hi, qe_half));
          };
          run_c(std::integral_constant<int, 0>{}, 0, min(qL, qc));
          run_c(std::integral_constant<int, 3>{}, min(qL, qc), min(qM0, qc));
          run_c(std::integral_constant<int, 1>{}, min(qM0, qc), min(qM1, qc));
          run_c(std::integral_constant<int, 3>{}, min(qM1, qc), qR);
          run_c(std::integral_constant<int, 2>{}, qR, qc);
          run_c(std::integral_constant<int, 3>{}, qc, NQ);
        }
      };
      if (RAW) {
        // exact integer products; the Padfield assembly happens afterwards
        int* raw = a.raw_out + (a.list ? item : b) * a.raw_stride;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) raw[srow[r] + 16 * q] = acc[q][r];
        continue;
      }
      if (SAME) {
        // corr = ey ex G[yv][xv] + ey Rrow[sx][yv] + ex Rcol[sy][xv] + const
        //        + mua mub ny nx        (header comment, item 3)
        // arranged as  (ey ex) g + A[r][sx] + B[q][sy] + fny[r] fnx[q]  with the
        // constants folded into A.  Rows / columns of the tile padding get
        // NaN through fny / fnx: NaN never wins fmaxf and never passes "> thr".
        const float* rrowA = R_lds;
        const float* rrowB = R_lds + a.aux_n;
        const float* rcolA = R_lds + 2 * a.aux_n;
        const float* rcolB = R_lds + 3 * a.aux_n;
        const int nn = n + opaque_zero;
        int grow[4];
        float ey[4], a_neg[4], a_pos[4], fny[4];
        bool sy[4];
        unsigned rowp[4];  // byte offsets into `surf` (scalar base + 32-bit offset)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ky_raw = 16 * p + 4 * g + r;
          const int ky = min(ky_raw, Sy - 1);
          const int dy = ky - (Py - 1);
          sy[r] = dy >= 0;
          const int yv = sy[r] ? dy : dy + Py;
          ey[r] = sy[r] ? -1.f : 1.f;
          grow[r] = yv * Px + opaque_zero;
          // A[r][sx]: sx = 0 (dx < 0) and sx = 1 (dx >= 0)
          a_neg[r] = ey[r] * rrowB[yv] + (sy[r] ? 0.f : const_b);
          a_pos[r] = ey[r] * rrowA[yv] + (sy[r] ? const_a : 0.f);
          fny[r] = ky_raw < Sy ? muab * static_cast<float>(Py - abs(dy)) : NAN;
          rowp[r] = 4u * static_cast<unsigned>(ky_raw * a.sx_pitch + nn);
        }
        if constexpr (EXACT) {
          // Px == 16 NCA: column tile q < NCA holds kx = 16 q + n < Px - 1 (dx < 0,
          // xv = kx + 1, ex = +1, fnx = kx + 1); tile q + NCA holds the shifts
          // dx = kx + 1 > 0 with the SAME xv (ex = -1, fnx = Px - xv).  The only
          // lanes that deviate are n = 15 of tile NCA - 1 (dx = 0: xv = 0) and of
          // tile 2 NCA - 1 (kx = Sx: padding).  Table addresses are one VGPR per
          // row plus an immediate, signs are folded into the instruction.
          const int xvb = nn + 1;                         // xv of tile 0
          const bool last = nn == 15;                     // the deviating lane
          const float fxb = static_cast<float>(xvb);
          const float* rcolA = R_lds + 2 * a.aux_n;
          const float* rcolB = R_lds + 3 * a.aux_n;
          unsigned gofs[4];                               // byte offsets of G[yv][xvb]
#pragma unroll
          for (int r = 0; r < 4; ++r) gofs[r] = 4u * static_cast<unsigned>(grow[r] + xvb);
          constexpr int kQG = SFM_EPI_QG;
          constexpr int kGroupsE = (NCA + kQG - 1) / kQG;
          float gb[2][kQG][4];
          float rca_e[NCA], rcb_e[NCA];
#pragma unroll
          for (int qq = 0; qq < NCA; ++qq) {
            // the deviating lane of the last tile reads xv = 0 (first half); its
            // second-half value is padding
            const int xv = (qq == NCA - 1 && last) ? 0 : xvb + 16 * qq;
            rca_e[qq] = rcolA[xv];
            rcb_e[qq] = rcolB[xv];
          }
          auto fetch_e = [&](int grp, int slot) {
#pragma unroll
            for (int u = 0; u < kQG; ++u) {
              const int qq = grp * kQG + u;
              if (qq >= NCA) break;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                unsigned off = gofs[r] + 64u * qq;
                if (qq == NCA - 1)  // xv = 0 for the deviating lane (also keeps the
                  off = last ? 4u * static_cast<unsigned>(grow[r]) : off;  // read in bounds)
                gb[slot][u][r] = at_byte(G, off);
              }
            }
          };
          auto store4 = [&](int q, int r, float v) {
            if (q < col_skip || q >= NQ - col_skip) v = 0.f;  // column tile left out
#ifndef SFM_ABLATE_STORE  // timing experiment: the kernel without its surface writes
            if constexpr (!LAZY)
              __builtin_nontemporal_store(
                  v, reinterpret_cast<float*>(reinterpret_cast<char*>(surf) +
                                              (static_cast<size_t>(rowp[r]) + 64u * q)));
#endif
            tmax = fmaxf(tmax, v);
            acc[q][r] = __float_as_int(v);
          };
          auto emit_e = [&](int qq, const float* gv) {
            // first half: dx < 0 (except the deviating lane of tile NCA - 1: dx = 0)
            {
              const int q = qq;
              const bool dev = qq == NCA - 1 && last;
              const float fnx = dev ? static_cast<float>(Px) : fxb + static_cast<float>(16 * qq);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float bsel = sy[r] ? rca_e[qq] : rcb_e[qq];
                float corr = dev ? fmaf(-ey[r], gv[r], a_pos[r]) - bsel
                                 : fmaf(ey[r], gv[r], a_neg[r]) + bsel;
                corr = fmaf(fny[r], fnx, corr);
                store4(q, r, static_cast<float>(acc[q][r]) + corr);
              }
            }
            // second half: dx = 16 qq + n + 1 > 0, same xv; the deviating lane of
            // the last tile is the padding column kx = Sx
            if (qq + NCA < NQ) {
              const int q = qq + NCA;
              const bool dev = qq == NCA - 1 && last;
              const float fnx =
                  dev ? NAN : static_cast<float>(Px) - (fxb + static_cast<float>(16 * qq));
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float bsel = sy[r] ? rca_e[qq] : rcb_e[qq];
                float corr = fmaf(-ey[r], gv[r], a_pos[r]) - bsel;
                corr = fmaf(fny[r], fnx, corr);
                store4(q, r, static_cast<float>(acc[q][r]) + corr);
              }
            }
          };
          fetch_e(0, 0);
#pragma unroll
          for (int grp = 0; grp < kGroupsE; ++grp) {
            __builtin_amdgcn_sched_barrier(0);
            if (grp + 1 < kGroupsE) fetch_e(grp + 1, (grp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kQG; ++u) {
              const int qq = grp * kQG + u;
              if (qq >= NCA) break;
              emit_e(qq, gb[grp & 1][u]);
            }
          }
        } else {
        const bool paired = Px == 16 * NCA;
        constexpr int kQG = SFM_EPI_QG;        // output columns per gather group
        constexpr int kDepth = SFM_EPI_DEPTH;  // groups in flight ahead of the stores
        constexpr int kSlots = kDepth + 1;
        constexpr int kGroups = (NCA + kQG - 1) / kQG;
        auto xv_of = [&](int q) {
          const int dx = min(16 * q + nn, Sx - 1) - (Px - 1);
          return dx >= 0 ? dx : dx + Px;
        };
        // Column terms of all NQ output columns, read from LDS in one batch
        // (one round trip instead of one per column while the other waves
        // keep the LDS pipeline busy with matrix fragments).
        float rca[NQ], rcb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int xv = xv_of(q);
          rca[q] = rcolA[xv];
          rcb[q] = rcolB[xv];
        }
        auto emit = [&](int q, const float* gv) {
          const int kx = 16 * q + nn;
          const int dx = min(kx, Sx - 1) - (Px - 1);
          const bool sx = dx >= 0;
          const int xv = sx ? dx : dx + Px;
          const float ex = sx ? -1.f : 1.f;
          const float b_pos = ex * rca[q];  // sy = 1
          const float b_neg = ex * rcb[q];  // sy = 0
          const float fnx = (q < NQ - 1 || kx < Sx)
                                ? static_cast<float>(Px - abs(dx)) : NAN;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float corr = fmaf(ey[r] * ex, gv[r], sx ? a_pos[r] : a_neg[r]);
            corr += sy[r] ? b_pos : b_neg;
            corr = fmaf(fny[r], fnx, corr);
            float v = static_cast<float>(acc[q][r]) + corr;
            if (q < col_skip || q >= NQ - col_skip) v = 0.f;  // column tile left out
#ifndef SFM_ABLATE_STORE  // timing experiment: the kernel without its surface writes
            if constexpr (!LAZY)
              __builtin_nontemporal_store(
                  v, reinterpret_cast<float*>(reinterpret_cast<char*>(surf) +
                                              (static_cast<size_t>(rowp[r]) + 64u * q)));
#endif
            tmax = fmaxf(tmax, v);
            acc[q][r] = __float_as_int(v);
          }
        };
        // gbuf[slot][u][0] = G for column q, [1] = G for column q + NCA
        float gbuf[kSlots][kQG][2][4];
        auto fetch = [&](int grp, int slot) {
#pragma unroll
          for (int u = 0; u < kQG; ++u) {
            const int q = grp * kQG + u;
            if (q >= NCA) break;
            const int x0 = xv_of(q);
#pragma unroll
            for (int r = 0; r < 4; ++r) gbuf[slot][u][0][r] = at_byte(G, 4u * static_cast<unsigned>(grow[r] + x0));
            // Px == 16 NCA: column q + NCA reads the same table entries (the
            // copy is taken at use time; copying here would wait for the loads)
            if (q + NCA < NQ && !paired) {
              const int x1 = xv_of(q + NCA);
#pragma unroll
              for (int r = 0; r < 4; ++r) gbuf[slot][u][1][r] = at_byte(G, 4u * static_cast<unsigned>(grow[r] + x1));
            }
          }
        };
#pragma unroll
        for (int grp = 0; grp < kDepth; ++grp)
          if (grp < kGroups) fetch(grp, grp % kSlots);
#pragma unroll
        for (int grp = 0; grp < kGroups; ++grp) {
          __builtin_amdgcn_sched_barrier(0);
          if (grp + kDepth < kGroups) fetch(grp + kDepth, (grp + kDepth) % kSlots);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < kQG; ++u) {
            const int q = grp * kQG + u;
            if (q >= NCA) break;
            emit(q, gbuf[grp % kSlots][u][0]);
            if (q + NCA < NQ) {
              float g1[4];
#pragma unroll
              for (int r = 0; r < 4; ++r)
                g1[r] = paired ? gbuf[grp % kSlots][u][0][r] : gbuf[grp % kSlots][u][1][r];
              emit(q + NCA, g1);
            }
          }
        }
        }  // generic (not EXACT) column handling
      } else {
        const int ipa = Px + 1, ipb = Qx + 1;
        int oa0[4], oa1[4], ob0[4], ob1[4], ny[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ky = min(16 * p + 4 * g + r, Sy - 1);
          const int dy = ky - (Qy - 1);
          const int ya0 = max(0, dy), ya1 = min(Py, Qy + dy);
          oa0[r] = ya0 * ipa + opaque_zero;
          oa1[r] = ya1 * ipa;
          ob0[r] = (ya0 - dy) * ipb + opaque_zero;
          ob1[r] = (ya1 - dy) * ipb;
          ny[r] = ya1 - ya0;
        }
        if constexpr (NCA > 10) {
          wide_general(0, 0.f);
        } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int kx = 16 * q + n;
          const int dx = min(kx, Sx - 1) - (Qx - 1);
          const int xa0 = max(0, dx), xa1 = min(Px, Qx + dx);
          const int xb0 = xa0 - dx, xb1 = xa1 - dx;
          const float fnx = static_cast<float>(xa1 - xa0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int sa = IA[oa1[r] + xa1] - IA[oa0[r] + xa1] -
                           IA[oa1[r] + xa0] + IA[oa0[r] + xa0];
            const int sb = IB[ob1[r] + xb1] - IB[ob0[r] + xb1] -
                           IB[ob1[r] + xb0] + IB[ob0[r] + xb0];
            float v = static_cast<float>(acc[q][r]);
            v = v - mua * static_cast<float>(sb);
            v = v - mub * static_cast<float>(sa);
            v = v + muab * (static_cast<float>(ny[r]) * fnx);
            surf[srow[r] + 16 * q] = v;
            const bool ok = rowok[r] && (q < NQ - 1 || kx < Sx);
            tmax = fmaxf(tmax, ok ? v : 0.f);
            acc[q][r] = __float_as_int(ok ? v : -INFINITY);
          }
        }
        }
      }
      TICK(3)
      if (a.do_peaks && forced) {
        // LAZY, recomputed tile: it is cold (nothing for the running maximum or
        // the hot list), it only has to reach memory
        if constexpr (LAZY) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              __builtin_nontemporal_store(
                  __int_as_float(acc[q][r]),
                  reinterpret_cast<float*>(reinterpret_cast<char*>(surf) +
                                           (4 * static_cast<size_t>(srow[r]) + 64u * q)));
          if (lane == 0) atomicOr(&lz[2], 1 << p);
        }
      } else if (a.do_peaks) {
        tmax = wave_max_nonneg(tmax);
        // non-negative floats order like their bit patterns; the atomic hands
        // back the running maximum of the tiles finished so far
        int prev_bits = 0;
        if (lane == 0) prev_bits = atomicMax(pmax_lds, __float_as_int(tmax));
        prev_bits = __builtin_amdgcn_readfirstlane(prev_bits);
        // Hot list: every element above threshold_rel * (running maximum) can
        // still turn out to be a peak; the final filter runs when the surface
        // is complete.  The running maximum never exceeds the final one, so
        // nothing that matters is dropped.
        const float mrun = fmaxf(tmax, __int_as_float(prev_bits));
        const float thr_t = a.threshold_rel * mrun;
        if (SAME && (a.prune || LAZY) && tmax > __int_as_float(prev_bits)) {
          // new running maximum (rare): remember its column tile for the seed
          // probe of the next patch
          int qbest = 0;
#pragma unroll
          for (int q = NQ - 1; q >= 0; --q) {
            const float vm = fmaxf(fmaxf(__int_as_float(acc[q][0]), __int_as_float(acc[q][1])),
                                   fmaxf(__int_as_float(acc[q][2]), __int_as_float(acc[q][3])));
            if (__any(vm == tmax)) qbest = q;
          }
          if (lane == 0) *best_lds = (p << 8) | qbest;
        }
        if constexpr (LAZY) {
          // Store this tile?  `hot`: it may hold an element above threshold_rel x
          // the final maximum (the running maximum only grows, so this errs on the
          // side of storing).  A hot tile asks for the tiles of its guard band; a
          // tile of the band that had finished before is dealt with at the end of
          // the patch, against the FINAL maximum (redo phase above).
          const bool hot = tmax > thr_t;
          int nbmask = 0;
          if (hot) {
            // rows of the tile that hold a possibly hot element -> the tiles within
            // `guard` rows of them
            unsigned rowmask = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float vm = __int_as_float(acc[0][r]);
#pragma unroll
              for (int q = 1; q < NQ; ++q) vm = fmaxf(vm, __int_as_float(acc[q][r]));
              const unsigned long long bal = __ballot(vm > thr_t);
#pragma unroll
              for (int gg = 0; gg < 4; ++gg)
                if ((bal >> (16 * gg)) & 0xffffull) rowmask |= 1u << (4 * gg + r);
            }
            const int r_lo = __builtin_ctz(rowmask | 0x10000u);
            const int r_hi = 31 - __builtin_clz(rowmask | 1u);
            const int lo_t = max(16 * p + r_lo - a.guard, 0) >> 4;
            const int hi_t = min((16 * p + r_hi + a.guard) >> 4, a.n_order - 1);
            nbmask = static_cast<int>(((2u << hi_t) - 1u) & ~((1u << lo_t) - 1u)) | (1 << p);
            if (lane == 0) lz_rows[p] = r_lo | (r_hi << 8);
          }
          int req = 0;
          if (lane == 0) {
            if (hot) atomicOr(&lz[0], nbmask);   // later tiles of the band store right away
            req = *const_cast<volatile int*>(&lz[0]);
          }
          req = __builtin_amdgcn_readfirstlane(req);
          const bool need = hot || ((req >> p) & 1);
          if (need) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                __builtin_nontemporal_store(
                    __int_as_float(acc[q][r]),
                    reinterpret_cast<float*>(reinterpret_cast<char*>(surf) +
                                             (4 * static_cast<size_t>(srow[r]) + 64u * q)));
          }
          if (lane == 0) {
            lz_tmax[p] = tmax;
            lz_ks[p] = narrowed ? col_skip : 0;
            if (need) atomicOr(&lz[2], 1 << p);
            atomicOr(&lz[1], 1 << p);
          }
        }
        // tmax is the wave-wide tile maximum: nothing to do unless it clears
        // the running threshold (the common case away from the peak).
        if (tmax > thr_t) {
        if constexpr (NCA > 10) wide_general(1, thr_t);
#pragma unroll
        for (int q = 0; q < (NCA > 10 ? 0 : NQ); ++q) {
          const float v0 = __int_as_float(acc[q][0]), v1 = __int_as_float(acc[q][1]);
          const float v2 = __int_as_float(acc[q][2]), v3 = __int_as_float(acc[q][3]);
          hot_insert(q, v0, v1, v2, v3, thr_t);
        }
        if (LAZY && lane == 0 && hq_hi >= 0) {
          atomicMin(&lz_cq[0], hq_lo);
          atomicMax(&lz_cq[1], hq_hi);
        }
        }
      }
      TICK(4)
      } while (0);
      if constexpr (PIPE) {
        TICK(9)   // (timing build) tiles that end without an epilogue
        if (!forced) {
          // done with this tile, whatever became of it; the wave that completes the
          // patch closes it (every LDS write of this wave is older than the count)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          int dn = 0;
          if (lane == 0) dn = atomicAdd(&ctl[1], 1) + 1;
          if (__builtin_amdgcn_readfirstlane(dn) == a.n_order) redo_phase = true;
        }
      }
    }
    TICK(5)
    if (!PIPE && a.do_peaks) {
      // Publish the running maximum and the hot-list fill; the first-peak
      // search over them is mfma_first_peak_kernel (every patch in parallel,
      // off the matrix pipeline's critical path).
      __syncthreads();
      if (threadIdx.x == 0) {
        a.v1[b] = __int_as_float(*pmax_lds);
        a.hot_count[b] = *hot_lds;
        if (SAME && a.prune) a.skipmask[b] = best_lds[1];
        // un-stored tiles (pruned ones included) are what the sweeps must leave out
        if (LAZY) a.skipmask[b] = ~lz[2] & static_cast<int>((2u << (a.n_order - 1)) - 1u);
      }
    }
    TICK(6)
  }
  if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) {
    a.clk[0] = clock64() - probe_c0;
    a.clk[1] = wall_clock64() - probe_w0;
  }
  // (only while the timing hooks are on: same-address atomics are serialised in the
  // L2 -- three per wave cost a launch of one patch per workgroup 100 us at its end)
  if (SAME && a.prune && a.count_tiles && a.clk && lane == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(a.clk + 2),
              (static_cast<unsigned long long>(tiles_drawn) << 32) |
                  static_cast<unsigned long long>(tiles_skipped));
    atomicAdd(reinterpret_cast<unsigned long long*>(a.clk + 3),
              (static_cast<unsigned long long>(tiles_early) << 32) |
                  static_cast<unsigned long long>(cols_skipped));
  }
  if (a.count_tiles && a.clk && lane == 0)
    atomicAdd(reinterpret_cast<unsigned long long*>(a.clk + 4),
              static_cast<unsigned long long>(mfma_issued));
#ifdef SFM_MFMA_TIMING
  if (lane == 0 && wave == 0) {
    // HW_REG_HW_ID (4): [11:8] CU, [12] SH, [15:13] SE; HW_REG_XCC_ID (20): [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    const long long wall = wall_clock64() - wstart, cyc = clock64() - cstart;
    printf("WG %d xcc %u se %u cu %u patches %d wall %lld cycles %lld MHz %lld mfma %lld epi %lld\n",
           blockIdx.x, xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, npat, wall, cyc,
           wall ? cyc * 100 / wall : 0, tph[2] / (npat ? npat : 1),
           tph[3] / (npat ? npat : 1));
  }
  if (PIPE && blockIdx.x == 7 && lane == 0) {
    // totals of this wave over the launch (cycles): divide by the workgroup's patches
    printf("pipewave %d opened %d tiles %d: total %lld seek %lld setup %lld tests %lld groups %lld loop-tail %lld "
           "epi %lld hot %lld exit %lld close %lld stage %lld probe %lld arm %lld\n",
           wave, npat, tiles_drawn, clock64() - cstart, tph[13], tph[10], tph[11], tph[12], tph[2], tph[3],
           tph[4], tph[9], tph[14], tph[15], tph[16], tph[17]);
  }
  if (!PIPE && blockIdx.x == 7 && lane == 0)
    printf("wave %d patches %d: next %lld sync %lld pix %lld aux+touch %lld stagesync %lld mfma %lld (+ setup %lld tests %lld groups %lld) epi %lld hot %lld tail %lld peaks %lld\n",
           wave, npat, tph[7] / npat, tph[0] / npat, tph[8] / npat, tph[9] / npat, tph[1] / npat, tph[2] / npat,
           tph[10] / npat, tph[11] / npat, tph[12] / npat, tph[3] / npat,
           tph[4] / npat, tph[5] / npat, tph[6] / npat);
#endif
}

struct Variant {
  int nca, nce;
};
// Instantiated chunk geometries: NCA >= ceil(Px / 16), NCE >= floor((Qx + 14) / 16) + 1.
// ({15, 11}, {20, 11}: search-window geometry, pre patches up to 240 / 320 wide)
constexpr Variant kVariants[] = {{3, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {8, 9}, {10, 11},
                                 {15, 11}, {20, 11}};

bool exact_enabled() {
  const char* e = sfm::measure_option("SFM_MFMA_EXACT");
  return !(e && e[0] == '0');
}

// SFM_MFMA_PRUNE=0: every dy tile is computed (tests, measurements).
bool prune_enabled() {
  const char* e = sfm::option("SFM_MFMA_PRUNE");
  return !(e && e[0] == '0');
}

int pick_variant(int px, int qx) {
  const int nca = (px + 15) / 16, nce = (qx + 14) / 16 + 1;
  for (size_t i = 0; i < sizeof(kVariants) / sizeof(kVariants[0]); ++i)
    if (kVariants[i].nca >= nca && kVariants[i].nce >= nce) return static_cast<int>(i);
  return -1;
}

struct Layout {
  int nca, nce, pa, pb, ml, a_bytes, b_bytes;
};

Layout make_layout(const SfmXcorrDesc* d, const Variant& v) {
  Layout l;
  l.nca = v.nca;
  l.nce = v.nce;
  const int py = d->patch[1], qy = d->post_patch[1], qx = d->post_patch[2];
  // Pre patch rows: 16 * NCA data bytes, pitch = odd multiple of 16 bytes so
  // that 16 consecutive rows hit 16 different 16-byte LDS slots.
  int pa16 = v.nca;
  if ((pa16 & 1) == 0) pa16 += 1;
  l.pa = pa16 * 16;
  l.a_bytes = (py + kPadTop + kPadBottom) * l.pa;
  // Post patch rows: left margin so that every lane window starts >= 0.
  const int cq0 = v.nce - 1;
  int ml = 16 * cq0 + 16 - qx;
  if (ml < 0) ml = 0;
  ml = (ml + 15) / 16 * 16;
  l.ml = ml;
  // Furthest byte touched: window start (n = 0) + 16 * NCE + 4 bytes read-ahead.
  int need = ml + qx - 1 - 16 * cq0 + 16 * v.nce + 4;
  need = std::max(need, ml + (qx + 15) / 16 * 16);
  int pb16 = (need + 15) / 16;
  // pitch / 4 mod 32 in {20, 12, ...}: keep 4 consecutive rows on distinct
  // banks for the dword reads: use pitch = 16 * odd with (pitch / 4) % 8 == 4.
  while (((pb16 * 4) % 8) != 4) ++pb16;
  l.pb = pb16 * 16;
  l.b_bytes = (qy + 3) * l.pb;
  l.a_bytes = (l.a_bytes + 15) / 16 * 16;
  l.b_bytes = (l.b_bytes + 15) / 16 * 16;
  return l;
}

bool same_size(const SfmXcorrDesc* d) {
  return d->patch[1] == d->post_patch[1] && d->patch[2] == d->post_patch[2] &&
         d->patch[2] <= 64 * kPrepCols;  // columns per lane in the prep kernel
}

struct Ws {
  int* c16;
  long long c16_stride;
  PatchParams* pp;
  int* integ[2];
  long long stride[2];
  float* gtab;
  float* aux;
  int aux_n;
  float* tbound;
  int* counter;
  size_t bytes;
};

Ws carve_ws(const SfmXcorrDesc* d, void* base) {
  sfm::Carver c(base);
  Ws w;
  std::memset(&w, 0, sizeof(w));
  const size_t B = d->batch;
  w.pp = c.take<PatchParams>(B);
  w.counter = c.take<int>(64 + kXcds * kHeadPitch);  // queue head, clk probe, per-XCD heads
  if (same_size(d)) {
    w.aux_n = std::max(d->patch[1], d->patch[2]) + 1;
    w.gtab = c.take<float>(B * d->patch[1] * d->patch[2]);
    w.aux = c.take<float>(B * (4 * w.aux_n + 4));
    w.tbound = c.take<float>(B * kBoundStride);
    // column sums above every 16th row + row-sum prefixes (kModeSameExactLazyG)
    w.c16_stride = 2LL * (d->patch[1] / 16 + 1) * d->patch[2] + d->patch[1] + 1;
    w.c16_stride = (w.c16_stride + 63) / 64 * 64;
    w.c16 = c.take<int>(B * (size_t)w.c16_stride);
  } else {
    w.stride[0] = (long long)(d->patch[1] + 1) * (d->patch[2] + 1);
    w.stride[1] = (long long)(d->post_patch[1] + 1) * (d->post_patch[2] + 1);
    w.integ[0] = c.take<int>(B * w.stride[0]);
    w.integ[1] = c.take<int>(B * w.stride[1]);
  }
  w.bytes = c.total();
  return w;
}

int device_cus();

// kModePipe: one workgroup of eight waves per CU, two patch slots in LDS.
template <int NCA, int NCE>
int launch_pipe(const MfmaArgs& a, int grid, hipStream_t st) {
  const size_t lds = 2 * static_cast<size_t>(a.slot_bytes);
  static size_t attr_set = 0;
  if (lds > attr_set) {
    SFM_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&xcorr_mfma_kernel<NCA, NCE, kModePipe>),
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    attr_set = lds;
  }
  grid = std::min((grid + 1) / 2, device_cus());
  {
    const char* g = sfm::option("SFM_MFMA_GRID");   // (tests: many patches per workgroup)
    if (g && std::atoi(g) > 0) grid = std::min(grid, std::max(1, std::atoi(g) / 2));
  }
  sfm::prof_begin(sfm::kProfXcorr, st);
  hipLaunchKernelGGL((xcorr_mfma_kernel<NCA, NCE, kModePipe>), dim3(grid),
                     dim3(2 * kThreads), lds, st, a);
  sfm::prof_end(sfm::kProfXcorr, st);
  sfm::prof_clock(sfm::kProfXcorr, a.clk, st, 5);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

template <int NCA, int NCE, int MODE>
int launch_one(const MfmaArgs& a, int grid, size_t lds, hipStream_t st) {
  // (search-window variants: eight waves per workgroup, see WIDE8 in the kernel)
  constexpr int kThreadsV = (NCA > 10 && SFM_WIDE_HALVES) ? 64 * SFM_WIDE_WAVES : kThreads;
  static size_t attr_set = 0;
  if (lds > attr_set) {
    SFM_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&xcorr_mfma_kernel<NCA, NCE, MODE>),
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    attr_set = lds;
  }
  // Persistent grid: as many workgroups as the CUs hold at once (registers and
  // LDS decide: 2 per CU for 160-wide patches, 3-4 for the small variants).
  static int per_cu = 0;
  static size_t per_cu_lds = 0;
  if (per_cu == 0 || per_cu_lds != lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &n, reinterpret_cast<const void*>(&xcorr_mfma_kernel<NCA, NCE, MODE>), kThreadsV,
            lds) != hipSuccess || n < 1)
      n = lds * 2 <= 160 * 1024 ? 2 : 1;
    per_cu = n;
    per_cu_lds = lds;
  }
  int wg_per_cu = per_cu;
  {
    // (a measurement switch; read per call so that a test can flip it)
    const char* cap = sfm::measure_option("SFM_MFMA_MAX_WG_PER_CU");
    if (cap && std::atoi(cap) > 0) wg_per_cu = std::min(wg_per_cu, std::atoi(cap));
  }
  grid = std::min(grid, device_cus() * wg_per_cu);
  {
    // SFM_MFMA_GRID=n (tests): at most n workgroups, so that a small batch runs
    // MANY patches through one workgroup -- the state a workgroup carries from
    // patch to patch (previous need mask / hot columns / seed block, the probe's
    // self-switch-off) is otherwise only exercised by full-size fields.
    const char* g = sfm::option("SFM_MFMA_GRID");
    if (g && std::atoi(g) > 0) grid = std::min(grid, std::atoi(g));
  }
  sfm::prof_begin(sfm::kProfXcorr, st);
  hipLaunchKernelGGL((xcorr_mfma_kernel<NCA, NCE, MODE>), dim3(grid),
                     dim3(kThreadsV), lds, st, a);
  sfm::prof_end(sfm::kProfXcorr, st);
  sfm::prof_clock(sfm::kProfXcorr, a.clk, st, 5);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

template <int NCA, int NCE>
int launch_variant(const MfmaArgs& a, int mode, int grid, size_t lds,
                   hipStream_t st) {
  if constexpr (NCA > 10) {   // search-window variants: un-masked general mode only
    if (mode != kModeGeneral) return sfm::fail(SFM_ERR_INVALID, "wide MFMA variant: general mode only");
    return launch_one<NCA, NCE, kModeGeneral>(a, grid, lds, st);
  } else
  switch (mode) {
    case kModeSame: return launch_one<NCA, NCE, kModeSame>(a, grid, lds, st);
    case kModeSameExact: return launch_one<NCA, NCE, kModeSameExact>(a, grid, lds, st);
    case kModeSameLazy: return launch_one<NCA, NCE, kModeSameLazy>(a, grid, lds, st);
    case kModeSameExactLazy: return launch_one<NCA, NCE, kModeSameExactLazy>(a, grid, lds, st);
    case kModeSameExactLazyG: return launch_one<NCA, NCE, kModeSameExactLazyG>(a, grid, lds, st);
    case kModePipe:
      // (instantiated for the 160-wide variant; the narrower ones keep three or four
      // workgroups per CU and are not priced by the per-patch phases)
      if constexpr (NCA == 10)
        return launch_pipe<NCA, NCE>(a, grid, st);
      else
        return launch_one<NCA, NCE, kModeSameExactLazyG>(a, grid, lds, st);
    case kModeRaw: return launch_one<NCA, NCE, kModeRaw>(a, grid, lds, st);
    default: return launch_one<NCA, NCE, kModeGeneral>(a, grid, lds, st);
  }
}

int launch_mode(int vi, const MfmaArgs& a, int mode, int grid, size_t lds,
                hipStream_t st) {
  switch (vi) {
#ifndef SFM_DEV_ONLY_WIDE   // (development: compile the search-window variants alone)
    case 0: return launch_variant<3, 4>(a, mode, grid, lds, st);
    case 1: return launch_variant<4, 5>(a, mode, grid, lds, st);
    case 2: return launch_variant<5, 6>(a, mode, grid, lds, st);
    case 3: return launch_variant<6, 7>(a, mode, grid, lds, st);
    case 4: return launch_variant<7, 8>(a, mode, grid, lds, st);
    case 5: return launch_variant<8, 9>(a, mode, grid, lds, st);
    case 6: return launch_variant<10, 11>(a, mode, grid, lds, st);
#endif
    case 7: return launch_variant<15, 11>(a, mode, grid, lds, st);
    case 8: return launch_variant<20, 11>(a, mode, grid, lds, st);
  }
  return sfm::fail(SFM_ERR_INVALID, "no MFMA variant");
}

int device_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  }
  return cus;
}

// Geometry, image pointers, LDS layout and the static tile schedule shared by
// every mode.
int fill_common(const SfmXcorrDesc* d, const Layout& l, MfmaArgs* ap) {
  MfmaArgs& a = *ap;
  std::memset(&a, 0, sizeof(a));
  a.img[0] = static_cast<const unsigned char*>(d->pre_image);
  a.img[1] = static_cast<const unsigned char*>(d->post_image);
  a.mask[0] = d->pre_mask;
  a.mask[1] = d->post_mask;
  for (int k = 0; k < 2; ++k) {
    a.ishape[0][k] = d->pre_shape[1 + k];
    a.ishape[1][k] = d->post_shape[1 + k];
    a.mshape[0][k] = d->pre_mask_shape[1 + k];
    a.mshape[1][k] = d->post_mask_shape[1 + k];
    a.P[k] = d->patch[1 + k];
    a.Q[k] = d->post_patch[1 + k];
    a.S[k] = a.P[k] + a.Q[k] - 1;
  }
  a.starts[0] = d->pre_starts;
  a.starts[1] = d->post_starts;
  a.batch = d->batch;
  a.use_mean = d->use_mean;
  a.mean = d->mean;
  {
    int rows = 0, pitch = 0;
    sfm::mfma_i8_padded_dims(d, &rows, &pitch);
    a.sx_pitch = pitch;
    a.s_stride = (long long)rows * pitch;
  }
  a.pa = l.pa;
  a.pb = l.pb;
  a.ml = l.ml;
  a.a_bytes = l.a_bytes;
  a.b_bytes = l.b_bytes;
  // Static schedule: dy tiles sorted by the number of patch rows they visit,
  // dealt to the 4 waves longest-first.
  const int np = (a.S[0] + 15) / 16;
  std::vector<std::pair<int, int>> work;
  for (int p = 0; p < np; ++p) {
    const int dy0 = 16 * p - (a.Q[0] - 1);
    const int ylo = std::max(0, -dy0 - 15), yhi = std::min(a.Q[0], a.P[0] - dy0);
    work.push_back({std::max(0, (yhi - ylo + 3) / 4), p});
  }
  std::sort(work.begin(), work.end(),
            [](const std::pair<int, int>& x, const std::pair<int, int>& y) {
              return x.first > y.first || (x.first == y.first && x.second < y.second);
            });
  if (work.size() > sizeof(a.order) / sizeof(a.order[0]))
    return sfm::fail(SFM_ERR_INVALID, "too many dy tiles");
  a.n_order = static_cast<int>(work.size());
  for (size_t i = 0; i < work.size(); ++i) a.order[i] = work[i].second;
  if (l.nca > 10 && SFM_WIDE_HALVES) {
    // search-window variants: a tile job is one column half of a row tile (p | half << 8)
    if (2 * work.size() > sizeof(a.order) / sizeof(a.order[0]))
      return sfm::fail(SFM_ERR_INVALID, "too many dy tiles");
    a.n_order = static_cast<int>(2 * work.size());
    for (size_t i = 0; i < work.size(); ++i) {
      a.order[2 * i] = work[i].second;
      a.order[2 * i + 1] = work[i].second | (1 << 8);
    }
  }
  int load[kWaves] = {0, 0, 0, 0};
  for (auto& t : work) {
    int best = 0;
    for (int k = 1; k < kWaves; ++k)
      if (load[k] < load[best]) best = k;
    if (a.n_tiles[best] >= kMaxTilesPerWave)
      return sfm::fail(SFM_ERR_INVALID, "too many dy tiles");
    a.tiles[best][a.n_tiles[best]++] = t.second;
    load[best] += t.first + 1;  // + epilogue
  }
  return SFM_OK;
}

// Masked path.  Pass index -> operand planes (pre, post); pass 0 runs for every
// patch, the others only where a side has masked pixels (masked_classify_kernel).
const int kMaskedPasses[8][2] = {
    {kPlaneVal, kPlaneVal},     {kPlaneVal, kPlaneValid},  {kPlaneValid, kPlaneVal},
    {kPlaneValid, kPlaneValid}, {kPlaneSqHi, kPlaneValid}, {kPlaneSqLo, kPlaneValid},
    {kPlaneValid, kPlaneSqHi},  {kPlaneValid, kPlaneSqLo}};

// SFM_MASKED_FAST=0 (tests): every patch takes all eight passes.
bool masked_all_passes() {
  const char* e = sfm::option("SFM_MASKED_FAST");
  return e && e[0] == '0';
}

struct MaskedWs {
  PatchParams* pp;
  int *nvalid, *cls, *first, *items, *n_items, *ov_lb, *sweep0, *tab, *raw0, *rawd;
  int group, n_groups;
  long long tab_elems, raw_stride;
  size_t bytes;
};

MaskedWs carve_masked(const SfmXcorrDesc* d, void* base) {
  sfm::Carver c(base);
  MaskedWs w;
  std::memset(&w, 0, sizeof(w));
  int rows = 0, pitch = 0;
  sfm::mfma_i8_padded_dims(d, &rows, &pitch);
  const size_t B = d->batch;
  w.pp = c.take<PatchParams>(B);
  w.nvalid = c.take<int>(2 * B);
  w.cls = c.take<int>(B);
  w.first = c.take<int>(B);
  w.items = c.take<int>(7 * B);
  w.n_items = c.take<int>(16);
  // reference batches in this call: the tolerances are maxima over a batch
  w.group = d->group > 0 && d->group < d->batch ? d->group : d->batch;
  w.n_groups = (d->batch + w.group - 1) / w.group;
  w.ov_lb = c.take<int>(w.n_groups);
  w.sweep0 = c.take<int>(B);
  // + 64 ints: consecutive tables (and product surfaces below) of a patch are read
  // together; strides that are multiples of 4 KB would put them on one memory channel
  w.tab_elems = (long long)d->patch[1] * d->patch[2] + 64;
  w.tab = c.take<int>(B * 4 * (size_t)w.tab_elems);
  w.raw_stride = (long long)rows * pitch + 64;
  w.raw0 = c.take<int>(B * (size_t)w.raw_stride);
  w.rawd = c.take<int>(B * 7 * (size_t)w.raw_stride);
  w.bytes = c.total();
  return w;
}

}  // namespace

namespace sfm {

bool mfma_i8_eligible(const SfmXcorrDesc* d) {
  if (!d || d->ndim != 2 || d->dtype != SFM_DTYPE_U8) return false;
  if (d->patch[0] != 1 || d->post_patch[0] != 1) return false;
  const int py = d->patch[1], px = d->patch[2];
  const int qy = d->post_patch[1], qx = d->post_patch[2];
  const int vi = pick_variant(px, qx);
  if (vi < 0) return false;
  if (py < 1 || qy < 1 || qy > py || qx > px) return false;
  if (kVariants[vi].nca <= 10) {
    if ((long long)py * px > 32768) return false;      // prep kernel LDS copy
  } else {
    // search-window variants: un-masked, the prep kernel's LDS copy of a patch and the
    // correlation kernel's two patches + scratch within one CU's LDS
    if (d->pre_mask || d->post_mask) return false;
    if ((long long)py * px > 140 * 1024) return false;
    const Layout l = make_layout(d, kVariants[vi]);
    if ((size_t)l.a_bytes + l.b_bytes + 8 * 1024 > 160 * 1024) return false;
  }
  if ((long long)qy * qx * 16384 > 0x7fffffffLL) return false;  // int32 sums
  if ((py + qy - 1 + 15) / 16 > kWaves * kMaxTilesPerWave) return false;
  // (the staging loads are 16 bytes wide and clamped into the image)
  if ((long long)d->pre_shape[1] * d->pre_shape[2] < 16 ||
      (long long)d->post_shape[1] * d->post_shape[2] < 16)
    return false;
  if ((reinterpret_cast<uintptr_t>(d->pre_image) & 3) ||
      (reinterpret_cast<uintptr_t>(d->post_image) & 3))
    return false;
  return true;
}

size_t mfma_i8_workspace_bytes(const SfmXcorrDesc* d) {
  if (d->pre_mask || d->post_mask) return carve_masked(d, nullptr).bytes;
  return carve_ws(d, nullptr).bytes;
}

void mfma_i8_padded_dims(const SfmXcorrDesc* d, int* rows, int* pitch) {
  const int vi = pick_variant(d->patch[2], d->post_patch[2]);
  const int sy = d->patch[1] + d->post_patch[1] - 1;
  *rows = (sy + 15) / 16 * 16;
  *pitch = vi < 0 ? 0 : 16 * (kVariants[vi].nca + kVariants[vi].nce - 1);
}

int mfma_i8_surface(const SfmXcorrDesc* d, void* ws_base, float* surface,
                    const FusedPeaks* fp) {
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int vi = pick_variant(d->patch[2], d->post_patch[2]);
  if (vi < 0) return fail(SFM_ERR_INVALID, "patch too wide for the MFMA path");
  const Layout l = make_layout(d, kVariants[vi]);
  Ws w = carve_ws(d, ws_base);
  MfmaArgs a;
  if (int rc = fill_common(d, l, &a)) return rc;
  a.pp = w.pp;
  const bool same = same_size(d);
  a.integ[0] = w.integ[0];
  a.integ[1] = w.integ[1];
  a.integ_stride[0] = w.stride[0];
  a.integ_stride[1] = w.stride[1];
  a.gtab = w.gtab;
  a.c16 = w.c16;
  a.c16_stride = w.c16_stride;
  a.aux = w.aux;
  a.aux_n = w.aux_n;
  a.surface = surface;
  {
    const char* e = sfm::measure_option("SFM_MFMA_QUEUE");
    a.work_counter = (e && e[0] == '0') ? nullptr : w.counter;
    // Measured on MI355X (8192^2 warped pair, A/B on one box, pruned and not):
    // 15.48-15.53 / 20.0-20.06 ms either way -- the staging round trip is hidden
    // behind the other workgroup's matrix phase, the kernel is matrix-pipe / power
    // bound, so L2 locality of the pixel loads buys nothing.  Opt-in ("1").
    const char* x = sfm::option("SFM_MFMA_XCD");
    a.xcd_heads = (a.work_counter && x && x[0] == '1') ? w.counter + 64 : nullptr;
    a.clk = reinterpret_cast<long long*>(w.counter + 16);
    const char* p = sfm::measure_option("SFM_MFMA_PRIO");
    a.prio_mode = p ? std::atoi(p) : 0;
  }
  if (fp) {
    a.do_peaks = 1;
    a.threshold_rel = d->threshold_rel;
    a.min_distance = d->min_distance;
    a.cand_cap = fp->cand_cap;
    a.idx1 = fp->idx1;
    a.v1 = fp->v1;
    a.zero_is_peak = fp->zero_is_peak;
    a.cand_count = fp->cand_count;
    a.cand_val = fp->cand_val;
    a.cand_idx = fp->cand_idx;
    a.bitmap = fp->bitmap;
    a.group = fp->group;
    a.bitmap_words = fp->bitmap_words;
    a.hot_cap = fp->hot_cap;
    a.hot_count = fp->hot_count;
    a.hot_val = fp->hot_val;
    a.hot_idx = fp->hot_idx;
    a.skipmask = fp->skipmask;
    // exact pruning of dy tiles that cannot matter to the peak statistics
    a.tbound = w.tbound;
    a.guard = std::max(d->min_distance, 2 * d->peak_radius[1]);
    a.guard_x = std::max(d->min_distance, 2 * d->peak_radius[2]);
    a.nq = kVariants[vi].nca + kVariants[vi].nce - 1;
    a.prune_k[0] = col_skip_lo(a.nq);
    a.prune_k[1] = col_skip_hi(a.nq);
    a.prune_k[2] = col_skip_2(a.nq);
    a.prune_k[3] = col_skip_3(a.nq);
    {
      const char* e = sfm::measure_option("SFM_MFMA_PROBE");  // "0": no seed probe
      a.probe = !(e && e[0] == '0');
    }
    a.count_tiles = sfm::profiling() ? 1 : 0;
    {
      // least number of row groups between two tests inside the row loop of the
      // lazy modes ("0": no tests); each test schedules the next, see check_after
      const char* e = sfm::option("SFM_MFMA_EARLY");
      a.early = e ? std::atoi(e) : 1;
      // initial store requests of a patch: what the previous patch of the workgroup
      // needed, "1": widened by a row tile on either side.  (Before the in-loop test
      // a spare request cost a store and saved a recomputation when the peak moved
      // to the next tile; now a requested tile also runs its whole row loop where
      // an unrequested one is abandoned after ~3/4 of it.  Measured on the 8192^2
      // warped pair: 12.29 ms per launch without, 12.54 ms with the widening.)
      const char* wd = sfm::option("SFM_MFMA_WIDEN");
      a.widen = wd ? std::atoi(wd) : 0;
      const char* ta = sfm::measure_option("SFM_MFMA_TOUCH_ALL");
      a.touch_all = ta && ta[0] == '1';
      const char* nw = sfm::option("SFM_MFMA_NARROW");
      a.narrow = nw ? std::atoi(nw) : 64;   // widest narrowing allowed (0: off)
      if (a.early < 0 || a.P[0] > kEarlyRows) a.early = 0;
    }
    a.prune = same && prune_enabled() && a.n_order <= kBoundTiles &&
              a.P[0] <= 16 * kBlkRows && a.P[1] <= 16 * kBlkCols &&
              a.P[0] <= kBoundRows && d->threshold_rel > 0.f && d->threshold_rel <= 1.f &&
              a.guard >= 0;
  }
  if (!a.tbound) a.tbound = w.tbound;  // read (not used) by every same-size launch
  const bool exact = same && d->patch[2] == 16 * kVariants[vi].nca && exact_enabled();
  // lazy surface stores: the flow path only (fused peak search: nobody else
  // reads the surface), tile masks of 31 bits, SFM_MFMA_LAZY=0 switches it off
  bool lazy = same && fp != nullptr && a.n_order <= 31 && a.skipmask != nullptr;
  {
    const char* e = sfm::option("SFM_MFMA_LAZY");
    if (e && e[0] == '0') lazy = false;
  }
  // The correction table built in the epilogue of the tiles that are stored instead
  // of by the prep kernel (kModeSameExactLazyG): pays where few tiles reach an
  // epilogue, i.e. with the pruning on (the un-pruned launch, every tile finished,
  // keeps the table).  SFM_MFMA_LAZYG=0: off.
  {
    const char* e = sfm::option("SFM_MFMA_LAZYG");
    a.lazy_g = exact && lazy && a.prune && !(e && e[0] == '0') && a.P[0] % 16 == 0 &&
               a.P[0] <= 160 && a.P[1] <= 160 && a.P[0] >= 32;
  }
  // Region behind the patches: the four 1-D correction arrays, reused as the
  // arg-max scratch of the fused peak search, then the running-max word.
  size_t r_bytes = same ? (size_t)4 * w.aux_n * 4 : 0;
  r_bytes = std::max(r_bytes, (size_t)kThreads * 8);
  r_bytes = (r_bytes + 15) / 16 * 16;
  a.r_bytes = static_cast<int>(r_bytes);
  if (same) {
    const size_t prep_lds = 2 * (((size_t)a.P[0] * a.P[1] + 15) & ~(size_t)15);
    static size_t prep_attr = 0;
    if (!a.lazy_g && prep_lds > prep_attr) {
      SFM_HIP_CHECK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&mfma_prep_same_kernel<false>),
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(prep_lds)));
      prep_attr = prep_lds;
    }
    if (a.lazy_g) {   // no pixels in LDS: the packed row / half-band column words
      const size_t lazy_lds =
          sizeof(unsigned) * (2 * (size_t)a.P[0] * (a.P[1] / 16) + 2 * 2 * (size_t)(a.P[0] / 16) * a.P[1]);
      hipLaunchKernelGGL(mfma_prep_same_kernel<true>, dim3(d->batch),
                         dim3(64 * kPrepWavesAlone), lazy_lds, st, a);
    }
    else
      hipLaunchKernelGGL(mfma_prep_same_kernel<false>, dim3(d->batch),
                         dim3(64 * kPrepWavesAlone), prep_lds, st, a);
  } else {
    const size_t prep_lds = (size_t)a.P[0] * a.P[1];
    static size_t prep_gen_attr = 0;
    if (prep_lds > 48 * 1024 && prep_lds > prep_gen_attr) {
      SFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_prep_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(prep_lds)));
      prep_gen_attr = prep_lds;
    }
    if (kVariants[vi].nca > 10 && a.P[1] <= 512 && !a.xcd_heads) {
      // search-window variants: the wide prep pass (see mfma_prep_wide_kernel)
      const size_t wide_lds = (size_t)a.P[0] * (16 * ((a.P[1] + 15) / 16) + 16);
      static size_t wide_attr = 0;
      if (wide_lds > wide_attr) {
        SFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_prep_wide_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(wide_lds)));
        wide_attr = wide_lds;
      }
      hipLaunchKernelGGL(mfma_prep_wide_kernel, dim3(d->batch, 2), dim3(64 * kWidePrepWaves),
                         wide_lds, st, a);
    } else
    hipLaunchKernelGGL(mfma_prep_kernel, dim3(d->batch, 2), dim3(kThreads),
                       prep_lds, st, a);
  }
  SFM_LAUNCH_CHECK();
  const size_t lds = (size_t)l.a_bytes + l.b_bytes + r_bytes + 16 + 4 * kBoundStride + 48 + 128 + 160 + 128;
  a.slot_bytes = static_cast<int>((lds + 4 * kPipeCtl + 4 * 256 + 15) / 16 * 16);
  const int grid = d->batch;  // capped to the resident workgroups in launch_one
  // cross-patch pipeline (kModePipe): SFM_MFMA_PIPE=1
  bool pipe = false;
  {
    const char* e = sfm::option("SFM_MFMA_PIPE");
    pipe = e && e[0] == '1' && a.lazy_g && kVariants[vi].nca == 10 && a.work_counter &&
           !a.xcd_heads && a.P[0] <= a.P[1] && a.n_order < 128 &&
           2 * (size_t)a.slot_bytes + 2048 <= 160 * 1024;
  }
  {
    const char* e = sfm::option("SFM_MFMA_PIPE_ADMIT");
    a.pipe_admit = e && std::atoi(e) > 0 ? std::atoi(e) : 2;
  }
  const int mode = !same ? kModeGeneral
                   : exact ? (pipe ? kModePipe
                              : a.lazy_g ? kModeSameExactLazyG
                                       : lazy ? kModeSameExactLazy : kModeSameExact)
                           : (lazy ? kModeSameLazy : kModeSame);
  if (int rc = launch_mode(vi, a, mode, grid, lds, st)) return rc;
  if (fp) {
    hipLaunchKernelGGL(mfma_first_peak_kernel, dim3((d->batch + kWaves - 1) / kWaves), dim3(kThreads), 0, st, a);
    SFM_LAUNCH_CHECK();
  }
  return SFM_OK;
}

// Masked (Padfield) correlation on the matrix cores.  Outputs, all padded to
// whole tiles [batch, rows, pitch]: num (numerator), den, ov, and the
// batch-global maxima the finalize step needs (flow_field.py:137, 151).
// Masked (Padfield) correlation on the matrix cores: writes the normalised
// surface, padded to whole tiles [batch, rows, pitch]; `maxima` = scratch for
// the batch maxima (flow_field.py:137, 151), two words per reference batch
// (`d->group` patches); `smax` (optional, zeroed by the caller) receives the
// ordered bits of every surface maximum.
int mfma_i8_masked(const SfmXcorrDesc* d, void* ws_base, float* surface,
                   unsigned int* maxima, unsigned int* smax, float* blkmax) {
  hipStream_t st = static_cast<hipStream_t>(d->stream);
  const int vi = pick_variant(d->patch[2], d->post_patch[2]);
  if (vi < 0) return fail(SFM_ERR_INVALID, "patch too wide for the MFMA path");
  const Layout l = make_layout(d, kVariants[vi]);
  MaskedWs w = carve_masked(d, ws_base);
  MfmaArgs a;
  if (int rc = fill_common(d, l, &a)) return rc;
  for (int k = 0; k < 2; ++k)
    if (a.mask[k] && (a.mshape[k][0] < (k ? a.Q[0] : a.P[0]) ||
                      a.mshape[k][1] < (k ? a.Q[1] : a.P[1])))
      return fail(SFM_ERR_INVALID, "mask smaller than patch");
  a.pp = w.pp;
  a.prio_mode = 1;
  a.nvalid = w.nvalid;
  hipLaunchKernelGGL(mfma_prep_masked_kernel, dim3(d->batch, 2), dim3(kThreads), 0,
                     st, a);
  SFM_LAUNCH_CHECK();
  SFM_HIP_CHECK(hipMemsetAsync(maxima, 0, 2 * w.n_groups * sizeof(unsigned int), st));
  SFM_HIP_CHECK(hipMemsetAsync(w.ov_lb, 0, w.n_groups * sizeof(int), st));
  size_t r_bytes = (size_t)kThreads * 8;
  a.r_bytes = static_cast<int>(r_bytes);
  const size_t lds = (size_t)l.a_bytes + l.b_bytes + r_bytes + 16;
  MaskedFastArgs g;
  std::memset(&g, 0, sizeof(g));
  g.pp = w.pp;
  g.nvalid = w.nvalid;
  g.cls = w.cls;
  g.first = w.first;
  g.items = w.items;
  g.n_items = w.n_items;
  g.tab = w.tab;
  g.tab_elems = w.tab_elems;
  g.raw0 = w.raw0;
  g.rawd = w.rawd;
  g.elems = a.s_stride;
  g.raw_stride = w.raw_stride;
  a.raw_stride = w.raw_stride;
  g.pitch = a.sx_pitch;
  for (int k = 0; k < 2; ++k) {
    g.img[k] = a.img[k];
    g.mask[k] = a.mask[k];
    g.ishape[k][0] = a.ishape[k][0];
    g.ishape[k][1] = a.ishape[k][1];
    g.mshape[k][0] = a.mshape[k][0];
    g.mshape[k][1] = a.mshape[k][1];
    g.P[k] = a.P[k];
    g.Q[k] = a.Q[k];
    g.S[k] = a.S[k];
  }
  g.batch = d->batch;
  g.all_passes = masked_all_passes() ? 1 : 0;
  {
    const char* e = sfm::option("SFM_MASKED_DEADROWS");
    g.dead_rows = !(e && e[0] == '0');
  }
  {
    const char* e = sfm::option("SFM_PHASE_XCD");
    g.xcd_map = !(e && e[0] == '0');
  }
  g.maxima = maxima;
  {
    // class 0 maxima from the axes (SFM_MASKED_AXISMAX=0: the sweep over all shifts)
    const char* e = sfm::option("SFM_MASKED_AXISMAX");
    g.axis_max = !(e && e[0] == '0') && a.P[0] == a.Q[0] && a.P[1] == a.Q[1] &&
                 a.P[0] <= kAxisMaxLen && a.P[1] <= kAxisMaxLen;
  }
  g.group = w.group;
  g.ov_lb = w.ov_lb;
  g.sweep0 = w.sweep0;
  g.out = surface;
  g.smax = smax;
  g.blkmax = smax ? blkmax : nullptr;
  const int rows_per_wg = kWaves * kAsmRowsPerWave;
  static_assert(kWaves * kAsmRowsPerWave == 32, "blkmax blocks: kMaskedBlkRows in sfm_xcorr.hip");
  g.row_blocks = (a.S[0] + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL(masked_classify_kernel, dim3(1), dim3(1024), 0, st, g);
  hipLaunchKernelGGL(masked_tables_kernel, dim3(d->batch), dim3(kThreads), 0, st, g);
  SFM_LAUNCH_CHECK();
  MfmaArgs c = a;
  c.plane[0] = kMaskedPasses[0][0];
  c.plane[1] = kMaskedPasses[0][1];
  c.raw_out = w.raw0;
  c.ov_lb = g.dead_rows ? w.ov_lb : nullptr;
  c.ov_group = w.group;
  if (int rc = launch_mode(vi, c, kModeRaw, d->batch, lds, st)) return rc;
  c.ov_lb = nullptr;  // the other products feed the maxima of every element
  c.raw_out = w.rawd;
  c.list = w.items;
  c.n_list = w.n_items;
  const long long extra = 7LL * d->batch;
  if (int rc = launch_mode(vi, c, kModeRaw, static_cast<int>(std::min<long long>(extra, 1 << 20)),
                           lds, st))
    return rc;
  const dim3 grid(static_cast<unsigned>(((long long)g.row_blocks * d->batch + 7) / 8 * 8));
  if (g.axis_max)
    hipLaunchKernelGGL(masked_axis_max_kernel, dim3(d->batch), dim3(64), 0, st, g);
  hipLaunchKernelGGL(masked_phase_kernel<false>, grid, dim3(kThreads), 0, st, g);
  hipLaunchKernelGGL(masked_phase3_kernel<false>, grid, dim3(kThreads), 0, st, g);
  hipLaunchKernelGGL(masked_phase_kernel<true>, grid, dim3(kThreads), 0, st, g);
  hipLaunchKernelGGL(masked_phase3_kernel<true>, grid, dim3(kThreads), 0, st, g);
  SFM_LAUNCH_CHECK();
  return SFM_OK;
}

}  // namespace sfm
