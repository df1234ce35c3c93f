// Shared host-side helpers for libsofima_amd.so (gfx950 only).
#ifndef SFM_COMMON_H_
#define SFM_COMMON_H_

#include <hip/hip_runtime.h>

#include <string>

#include <cstdarg>
#include <cstdio>

#include "../../include/sofima_amd.h"

namespace sfm {

// Thread-local error text behind sfm_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

#define SFM_HIP_CHECK(expr)                                                  \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess)                                                    \
      return ::sfm::fail(SFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr,        \
                         hipGetErrorString(e_), __FILE__, __LINE__);         \
  } while (0)

#define SFM_LAUNCH_CHECK() SFM_HIP_CHECK(hipGetLastError())

// Behaviour switches (sfm_set_option; the environment variable of the same name
// is the default).  Returns the value or nullptr when unset.  The pointer stays
// valid for the life of the process (values are interned, never freed).  Entry
// points resolve their switches once, outside their launch loops.
const char* option(const char* name);
// Measurement-only switches (SFM_MFMA_PROBE / TOUCH_ALL / EXACT / QUEUE / PRIO /
// MAX_WG_PER_CU: they select schedules nobody ships, for A/B runs under tools/measure/):
// read only by a library built with -DSFM_MEASUREMENT_SWITCHES (SFM_BUILD_FLAGS of
// sofima_amd/_build.py; tools/measure/build_timing_lib.sh builds one); the production
// library ignores them.  sfm_get_option("SFM_BUILD_MEASUREMENT_SWITCHES") says which it is.
inline const char* measure_option(const char* name) {
#ifdef SFM_MEASUREMENT_SWITCHES
  return option(name);
#else
  (void)name;
  return nullptr;
#endif
}
std::string option_str(const char* name);   // copy; empty when unset

// FIRE scalars as the mesh kernels keep them on the device (sfm_mesh.hip) and
// the pending per-step corrections derived from the previous step's sums.
struct MeshScalars {
  float dt, alpha;
  int n_pos;
  float cap;
  float gate;
  float mx[3];
  float mv[3];
};

// How a kernel outside sfm_mesh.hip reads the position a node has AFTER the
// position update of the running step, x' = x + dt v + dt^2/2 a (with the pending
// velocity gate / drift removal of the previous step, mesh.py:439, 492-497),
// from the state before it -- the same expression, operation for operation, as
// advance_kernel / the fused integrators evaluate.  v == nullptr: x is final.
struct AdvanceView {
  const float* v;
  const float* a;
  const MeshScalars* scal;   // FIRE: dt, gate, drift means of the step
  int fire;
  int pending;               // the previous step's gate / drift still to apply
  int remove_drift;
  float vv_dt;               // damped Verlet: fixed dt
  const float* colmean;      // pending per-x-column drift means [6][X] (5-D states) or nullptr
};

// sfm_maps.hip: prev = target_mesh(x') written to `out`.  strips_only: nodes
// outside every neighbour's paste region are left untouched (they are NaN after
// one full evaluation and never change).
int launch_target_mesh(const SfmTargetMeshDesc* d, const float* x, float* out,
                       hipStream_t st, const AdvanceView* adv = nullptr,
                       bool strips_only = false, const int* block_list = nullptr);
// The node blocks that touch a paste region (in-plane montages), built once per
// chunk for the strips-only launches: `list` holds target_list_ints(d) ints.
size_t target_list_ints(const SfmTargetMeshDesc* d);
int build_target_list(const SfmTargetMeshDesc* d, int* list, hipStream_t st);

// sfm_comm.hip: grouped point-to-point building blocks (RCCL).
int comm_group_begin(SfmComm* c);
int comm_send(SfmComm* c, const float* buf, size_t count, int peer, hipStream_t st);
int comm_recv(SfmComm* c, float* buf, size_t count, int peer, hipStream_t st);
int comm_group_end(SfmComm* c, int rc);
int comm_rank(const SfmComm* c);
int comm_size(const SfmComm* c);

// bench.py timing hooks (sfm_profile_*): no-ops unless enabled.
constexpr int kProfXcorr = 0;
constexpr int kProfMesh = 1;
bool profiling();
void prof_begin(int kind, hipStream_t st);
void prof_end(int kind, hipStream_t st);
// Samples a kernel's shader-clock probe: dev_pair = {core cycles, 10 ns ticks}.
void prof_clock(int kind, const long long* dev_pair, hipStream_t st, int n = 2);

// Buffers of the peak search that the MFMA kernel fills itself when the first
// pass is fused into it (sfm_xcorr_mfma.hip: fused_first_peak).
struct FusedPeaks {
  int cand_cap;
  int* idx1;
  float* v1;
  int* zero_is_peak;
  int* cand_count;
  float* cand_val;
  int* cand_idx;
  unsigned* bitmap;   // [n_groups, bitmap_words]
  int group;          // rows per coupling group
  int bitmap_words;
  int hot_cap;     // per-surface capacity of the hot list
  int* hot_count;  // zeroed with the rest of the per-batch state
  float* hot_val;
  int* hot_idx;
  int* skipmask;   // [B] bit t: 16-row surface tile t was pruned (never stored)
};

#ifdef __HIPCC__
// Sets bits of a shared word; most callers of a batch set the SAME bit (equal
// first-peak indices), and same-address atomics are serialised in the L2: look first.
__device__ __forceinline__ void set_bit_once(unsigned int* word, unsigned int bits) {
  if ((__atomic_load_n(word, __ATOMIC_RELAXED) & bits) != bits) atomicOr(word, bits);
}
#endif

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves aligned sub-buffers out of the caller's workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
  size_t total() const { return align_up(off, 256); }
};

}  // namespace sfm

#endif  // SFM_COMMON_H_
