// RCCL exchange steps of the multi-GPU mesh path: one process
// per GPU; neighbouring bands of one mesh exchange their boundary rows over
// xGMI point to point, and the per-band partial sums (FIRE power, drift means)
// are all-gathered so that every rank reduces them in the same (rank) order.
//
// The reference has no multi-device code; these entry points exist for the
// band-sharded relaxation of sofima_amd/dist.py (mesh.py:448-499 is the step
// whose reductions they carry across GPUs).
//
// RCCL is resolved at run time (dlsym / dlopen), never at link time: the
// library must load on hosts without RCCL, and a process that already carries
// a RCCL (PyTorch bundles one) keeps using that copy.
#include "sfm_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

struct SfmComm {
  ncclComm_t comm;
  int rank;
  int n_ranks;
  int device;
};

namespace {

struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  void* handles[4] = {RTLD_DEFAULT, nullptr, nullptr, nullptr};
  const char* names[3] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  int n = 1;
  if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
    for (const char* nm : names) {
      void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (h) {
        handles[n++] = h;
        break;
      }
    }
  }
  auto find = [&](const char* sym) -> void* {
    for (int i = 0; i < n; ++i)
      if (void* p = dlsym(handles[i], sym)) return p;
    return nullptr;
  };
#define SFM_SYM(field, name) \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(find(name))
  SFM_SYM(get_unique_id, "ncclGetUniqueId");
  SFM_SYM(comm_init_rank, "ncclCommInitRank");
  SFM_SYM(comm_destroy, "ncclCommDestroy");
  SFM_SYM(group_start, "ncclGroupStart");
  SFM_SYM(group_end, "ncclGroupEnd");
  SFM_SYM(send, "ncclSend");
  SFM_SYM(recv, "ncclRecv");
  SFM_SYM(all_gather, "ncclAllGather");
  SFM_SYM(all_reduce, "ncclAllReduce");
  SFM_SYM(error_string, "ncclGetErrorString");
#undef SFM_SYM
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy &&
              g_rccl.group_start && g_rccl.group_end && g_rccl.send && g_rccl.recv &&
              g_rccl.all_gather && g_rccl.all_reduce && g_rccl.error_string;
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok)
    return sfm::fail(SFM_ERR_NO_DEVICE, "RCCL (librccl.so) could not be loaded");
  return SFM_OK;
}

#define SFM_NCCL_CHECK(expr)                                                   \
  do {                                                                         \
    ncclResult_t r_ = (expr);                                                  \
    if (r_ != ncclSuccess)                                                     \
      return ::sfm::fail(SFM_ERR_HIP, "%s failed: %s", #expr, g_rccl.error_string(r_)); \
  } while (0)

int check_comm(const SfmComm* c) {
  if (!c || !c->comm) return sfm::fail(SFM_ERR_INVALID, "comm is NULL");
  return need_rccl();
}

}  // namespace

// Building blocks for callers inside the library that need more than one
// neighbour pair per group (sfm_mesh_relax_banded): every send / recv between
// group_begin and group_end progresses concurrently.
namespace sfm {

int comm_group_begin(SfmComm* c) {
  if (int rc = check_comm(c)) return rc;
  SFM_NCCL_CHECK(g_rccl.group_start());
  return SFM_OK;
}

int comm_send(SfmComm* c, const float* buf, size_t count, int peer, hipStream_t st) {
  ncclResult_t r = g_rccl.send(buf, count, ncclFloat32, peer, c->comm, st);
  return r == ncclSuccess ? SFM_OK
                          : fail(SFM_ERR_HIP, "ncclSend failed: %s", g_rccl.error_string(r));
}

int comm_recv(SfmComm* c, float* buf, size_t count, int peer, hipStream_t st) {
  ncclResult_t r = g_rccl.recv(buf, count, ncclFloat32, peer, c->comm, st);
  return r == ncclSuccess ? SFM_OK
                          : fail(SFM_ERR_HIP, "ncclRecv failed: %s", g_rccl.error_string(r));
}

// Always call after comm_group_begin succeeded, also when a send / recv failed
// (`rc` = the first error so far, returned unless closing the group fails too).
int comm_group_end(SfmComm* c, int rc) {
  (void)c;
  ncclResult_t r = g_rccl.group_end();
  if (rc) return rc;
  return r == ncclSuccess ? SFM_OK
                          : fail(SFM_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.error_string(r));
}

int comm_rank(const SfmComm* c) { return c ? c->rank : 0; }
int comm_size(const SfmComm* c) { return c ? c->n_ranks : 1; }

}  // namespace sfm

extern "C" {

int sfm_comm_unique_id(void* id128) {
  if (!id128) return sfm::fail(SFM_ERR_INVALID, "id is NULL");
  if (int rc = need_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == SFM_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  SFM_NCCL_CHECK(g_rccl.get_unique_id(&id));
  std::memcpy(id128, &id, sizeof(id));
  return SFM_OK;
}

int sfm_comm_init(SfmComm** out, const void* id128, int rank, int n_ranks) {
  if (!out || !id128) return sfm::fail(SFM_ERR_INVALID, "comm/id is NULL");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return sfm::fail(SFM_ERR_INVALID, "rank %d of %d", rank, n_ranks);
  if (int rc = need_rccl()) return rc;
  int dev = 0;
  SFM_HIP_CHECK(hipGetDevice(&dev));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  SFM_NCCL_CHECK(g_rccl.comm_init_rank(&comm, n_ranks, id, rank));
  SfmComm* c = new (std::nothrow) SfmComm;
  if (!c) {
    g_rccl.comm_destroy(comm);
    return sfm::fail(SFM_ERR_INVALID, "out of host memory");
  }
  c->comm = comm;
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->device = dev;
  *out = c;
  return SFM_OK;
}

int sfm_comm_destroy(SfmComm* c) {
  if (!c) return SFM_OK;
  if (int rc = need_rccl()) return rc;
  if (c->comm) SFM_NCCL_CHECK(g_rccl.comm_destroy(c->comm));
  delete c;
  return SFM_OK;
}

int sfm_comm_halo_exchange(SfmComm* c, int peer_lo, const float* send_lo, float* recv_lo,
                           int peer_hi, const float* send_hi, float* recv_hi, size_t count,
                           void* stream) {
  if (int rc = check_comm(c)) return rc;
  for (int peer : {peer_lo, peer_hi})
    if (peer < -1 || peer >= c->n_ranks)
      return sfm::fail(SFM_ERR_INVALID, "peer %d of %d ranks", peer, c->n_ranks);
  if ((peer_lo >= 0 && (!send_lo || !recv_lo)) || (peer_hi >= 0 && (!send_hi || !recv_hi)))
    return sfm::fail(SFM_ERR_INVALID, "halo buffers missing for an existing neighbour");
  if (count == 0 || (peer_lo < 0 && peer_hi < 0)) return SFM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one grouped call: both directions progress concurrently on the two xGMI
  // links to the neighbours
  SFM_NCCL_CHECK(g_rccl.group_start());
  // the group is always closed, also after a failed send / recv: a dangling
  // group would swallow every later collective of this thread
  ncclResult_t first = ncclSuccess;
  auto note = [&](ncclResult_t r) {
    if (first == ncclSuccess && r != ncclSuccess) first = r;
  };
  if (peer_lo >= 0) {
    note(g_rccl.send(send_lo, count, ncclFloat32, peer_lo, c->comm, st));
    note(g_rccl.recv(recv_lo, count, ncclFloat32, peer_lo, c->comm, st));
  }
  if (peer_hi >= 0) {
    note(g_rccl.send(send_hi, count, ncclFloat32, peer_hi, c->comm, st));
    note(g_rccl.recv(recv_hi, count, ncclFloat32, peer_hi, c->comm, st));
  }
  note(g_rccl.group_end());
  if (first != ncclSuccess)
    return sfm::fail(SFM_ERR_HIP, "halo exchange failed: %s", g_rccl.error_string(first));
  return SFM_OK;
}

int sfm_comm_allgather(SfmComm* c, const float* send, float* recv, size_t count,
                       void* stream) {
  if (int rc = check_comm(c)) return rc;
  if (!send || !recv) return sfm::fail(SFM_ERR_INVALID, "allgather: NULL buffer");
  SFM_NCCL_CHECK(g_rccl.all_gather(send, recv, count, ncclFloat32, c->comm,
                                   static_cast<hipStream_t>(stream)));
  return SFM_OK;
}

int sfm_comm_allreduce_scalars(SfmComm* c, float* inout, size_t count, int op,
                               void* stream) {
  if (int rc = check_comm(c)) return rc;
  if (!inout) return sfm::fail(SFM_ERR_INVALID, "allreduce: NULL buffer");
  if (op != SFM_REDUCE_SUM && op != SFM_REDUCE_MAX)
    return sfm::fail(SFM_ERR_INVALID, "allreduce: op %d", op);
  SFM_NCCL_CHECK(g_rccl.all_reduce(inout, inout, count, ncclFloat32,
                                   op == SFM_REDUCE_SUM ? ncclSum : ncclMax, c->comm,
                                   static_cast<hipStream_t>(stream)));
  return SFM_OK;
}

}  // extern "C"
