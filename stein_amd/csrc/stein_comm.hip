// stein_comm.hip -- the row-sharded step with its collectives issued by the library itself: RCCL over xGMI, one
// communicator per process-rank.  One C call per step: both all-gathers as one RCCL group, the rank
// segments of steinhip.hip, the median's all-reduce(s) and the |phi|^2 all-reduce, all queued on the caller's stream.
//
// RCCL is not a link-time dependency: the library is looked up when the first communicator is made
// (dlopen("librccl.so.1")), which returns the copy the process has loaded already when there is one (PyTorch-ROCm brings
// its own under the same soname) -- two RCCL runtimes in one process would each keep their own IPC / bootstrap state.
// Only the NCCL 2.x core entry points are used; <rccl/rccl.h> supplies types and enums, never a symbol.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>

#include "stein_common.h"
#include "stein_x3.h"
#include "steinhip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.handle) return STEIN_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return stein_fail(STEIN_E_RCCL, "RCCL not found: %s", dlerror());
  Rccl r;
  r.handle = h;
#define SYM(field, name)                                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                      \
  if (!r.field) return stein_fail(STEIN_E_RCCL, "RCCL lacks %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl = r;
  return STEIN_OK;
}

#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return stein_fail(STEIN_E_RCCL, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)

constexpr unsigned COMM_MAGIC = 0x57E1C0DEu;
struct SteinComm {
  unsigned magic;
  ncclComm_t comm;
  int nranks, rank, device;
  unsigned* flags_host;   // page-locked landing place of the window's hit flag (28 bytes used)
  hipEvent_t flags_ready;
};

SteinComm* as_comm(void* p) {
  SteinComm* c = static_cast<SteinComm*>(p);
  return (c && c->magic == COMM_MAGIC) ? c : nullptr;
}

}   // namespace

extern "C" int stein_comm_unique_id(void* id_out, size_t id_bytes) {
  if (!id_out) return stein_fail(STEIN_E_BADARG, "NULL pointer");
  if (id_bytes != STEIN_COMM_ID_BYTES) return stein_fail(STEIN_E_BADARG, "id buffer must be %d bytes", STEIN_COMM_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == STEIN_COMM_ID_BYTES, "ncclUniqueId size");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  RCCL_TRY(g_rccl.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return STEIN_OK;
}

extern "C" int stein_comm_init(const void* id, size_t id_bytes, int nranks, int rank, void** comm_out) {
  if (!id || !comm_out) return stein_fail(STEIN_E_BADARG, "NULL pointer");
  if (id_bytes != STEIN_COMM_ID_BYTES) return stein_fail(STEIN_E_BADARG, "id buffer must be %d bytes", STEIN_COMM_ID_BYTES);
  if (nranks < 1 || rank < 0 || rank >= nranks) return stein_fail(STEIN_E_BADARG, "rank %d of %d", rank, nranks);
  int rc = rccl_load();
  if (rc) return rc;
  SteinComm* c = new SteinComm();
  c->nranks = nranks;
  c->rank = rank;
  c->flags_host = nullptr;
  c->flags_ready = nullptr;
  c->comm = nullptr;
  auto cleanup = [&](int code) {
    if (c->flags_ready) (void)hipEventDestroy(c->flags_ready);
    if (c->flags_host) (void)hipHostFree(c->flags_host);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return code;
  };
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->flags_host), 64, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->flags_ready, hipEventDisableTiming);
  if (e != hipSuccess) return cleanup(stein_fail(STEIN_E_HIP, "communicator set-up: %s", hipGetErrorString(e)));
  std::memset(c->flags_host, 0, 64);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);   // collective over the ranks; blocks until all joined
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return cleanup(stein_fail(STEIN_E_RCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)));
  }
  c->magic = COMM_MAGIC;
  *comm_out = c;
  return STEIN_OK;
}

extern "C" int stein_comm_destroy(void* comm) {
  SteinComm* c = as_comm(comm);
  if (!c) return stein_fail(STEIN_E_BADARG, "not a communicator");
  c->magic = 0;
  (void)hipEventDestroy(c->flags_ready);
  (void)hipHostFree(c->flags_host);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  delete c;
  if (r != ncclSuccess) return stein_fail(STEIN_E_RCCL, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
  return STEIN_OK;
}

extern "C" int stein_comm_info(void* comm, int* nranks_out, int* rank_out) {
  SteinComm* c = as_comm(comm);
  if (!c || !nranks_out || !rank_out) return stein_fail(STEIN_E_BADARG, "not a communicator / NULL pointer");
  *nranks_out = c->nranks;
  *rank_out = c->rank;
  return STEIN_OK;
}

extern "C" int stein_rank_step(void* comm, const void* theta_local, const void* score_local, void* theta_all,
                               void* score_all, int64_t n, int64_t d, int dtype, float* phi_local, float* h2_out,
                               float* median_out, double* sqnorm_out, float* dK_out, void* workspace, size_t ws_bytes,
                               int flags, int* window_hit_out, void* stream_v) {
  SteinComm* c = as_comm(comm);
  if (!c) return stein_fail(STEIN_E_BADARG, "not a communicator");
  if (!theta_local || !score_local || !theta_all || !score_all || !phi_local || !h2_out || !median_out || !sqnorm_out ||
      !workspace)
    return stein_fail(STEIN_E_BADARG, "NULL pointer");
  if (dtype != STEIN_F32 && dtype != STEIN_BF16) return stein_fail(STEIN_E_BADARG, "dtype %d", dtype);
  if (n < 2 || d < 1) return stein_fail(STEIN_E_BADARG, "need n >= 2 and d >= 1");
  if (n % c->nranks) return stein_fail(STEIN_E_SHAPE, "n = %lld is not divisible by %d ranks", (long long)n, c->nranks);
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev != c->device) return stein_fail(STEIN_E_BADARG, "communicator belongs to device %d, current device is %d", c->device, dev);
  const int64_t n_local = n / c->nranks, row0 = (int64_t)c->rank * n_local;
  const hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const bool window = (flags & STEIN_FLAG_RANK_WINDOW) != 0;
  const int seg_flags = flags & (STEIN_FLAG_X3 | STEIN_FLAG_RANK_WINDOW);
  if (flags & ~(STEIN_FLAG_X3 | STEIN_FLAG_RANK_WINDOW | STEIN_FLAG_TIMING | STEIN_FLAG_TILED))
    return stein_fail(STEIN_E_BADARG, "unknown flags 0x%x", flags);

  // the sections the collectives touch (the same layout the segments derive)
  SteinLayout L;
  int rc = stein_make_layout(n_local, n, d, dtype, (flags & STEIN_FLAG_X3) | STEIN_FLAG_TILED, &L);
  if (rc) return rc;
  if (ws_bytes < L.total) return stein_fail(STEIN_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  u64* hist = reinterpret_cast<u64*>(ws + L.off[STEIN_WS_HIST]);
  u64* table = reinterpret_cast<u64*>(ws + L.off[STEIN_WS_SPEC]) + (SPEC_SLOTS * 8 + SPEC_CAP);

  // (1) every rank's rows of theta and of the score: ONE grouped launch
  const ncclDataType_t dt = dtype == STEIN_BF16 ? ncclBfloat16 : ncclFloat32;
  const size_t count = (size_t)n_local * (size_t)d;
  RCCL_TRY(g_rccl.GroupStart());
  ncclResult_t r1 = g_rccl.AllGather(theta_local, theta_all, count, dt, c->comm, stream);
  ncclResult_t r2 = g_rccl.AllGather(score_local, score_all, count, dt, c->comm, stream);
  RCCL_TRY(g_rccl.GroupEnd());
  RCCL_TRY(r1);
  RCCL_TRY(r2);

  // (2) row norms, theta's planes, the [n_local, n] distance block with the median's first counts
  if ((rc = stein_rank_begin(theta_all, n, d, row0, n_local, dtype, workspace, ws_bytes, seg_flags, stream))) return rc;

  auto score_planes = [&]() -> int {
    if (!(flags & STEIN_FLAG_X3)) return STEIN_OK;
    return stein_x3_split(nullptr, score_all, dtype, n, d, L, ws + L.off[STEIN_WS_PLANES], stream);
  };
  auto level_sum = [&](int level) -> int {   // hist[level] summed over the ranks, in place
    u64* h = hist + (size_t)level * 2 * STEIN_HIST_BINS;
    RCCL_TRY(g_rccl.AllReduce(h, h, 2 * STEIN_HIST_BINS, ncclUint64, ncclSum, c->comm, stream));
    return STEIN_OK;
  };
  auto radix = [&](bool need_level0_pass) -> int {
    int rr;
    if (need_level0_pass &&
        (rr = stein_rank_radix(0, 1, n, d, row0, n_local, dtype, workspace, ws_bytes, seg_flags, h2_out, median_out, stream)))
      return rr;
    for (int level = 0; level < STEIN_HIST_LEVELS; ++level) {
      if ((rr = level_sum(level))) return rr;
      if ((rr = stein_rank_radix(level, 0, n, d, row0, n_local, dtype, workspace, ws_bytes, seg_flags, h2_out, median_out, stream)))
        return rr;
    }
    return STEIN_OK;
  };

  // (3) the median of the n^2 distances, identical on every rank
  int hit = -1;
  if (window) {
    RCCL_TRY(g_rccl.AllReduce(table, table, SPEC_TABLE, ncclUint64, ncclSum, c->comm, stream));
    if ((rc = stein_rank_pick(n, d, row0, n_local, dtype, workspace, ws_bytes, seg_flags, h2_out, median_out, c->flags_host,
                              stream)))
      return rc;
    HIP_TRY(hipEventRecord(c->flags_ready, stream));
    if ((rc = score_planes())) return rc;   // work that does not depend on the flag: covers the host's wake-up
    HIP_TRY(hipEventSynchronize(c->flags_ready));
    hit = c->flags_host[0] != 0;
    const bool skip_l0 = c->flags_host[6] != 0;
    if (!hit && (rc = radix(!skip_l0))) return rc;   // every rank read the same table: every rank takes the same branch
  } else {
    if ((rc = radix(false))) return rc;
    if ((rc = score_planes())) return rc;
  }
  if (window_hit_out) *window_hit_out = hit;

  // (4) the contraction on the local rows, phi, and the global |phi|^2
  if ((rc = stein_rank_finish(theta_all, score_all, n, d, row0, n_local, dtype, h2_out, phi_local, sqnorm_out, dK_out,
                              workspace, ws_bytes, seg_flags | (flags & STEIN_FLAG_TIMING), stream)))
    return rc;
  RCCL_TRY(g_rccl.AllReduce(sqnorm_out, sqnorm_out, 1, ncclFloat64, ncclSum, c->comm, stream));
  return STEIN_OK;
}
