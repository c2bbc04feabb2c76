// stein_comm.hip -- the row-sharded step with its collectives issued by the library itself: RCCL over xGMI, one
// communicator per process-rank.  One C call per step: the all-gather of theta, the rank segments of
// steinhip.hip, the median's all-reduce(s) and the |phi|^2 all-reduce queued on the caller's stream; the all-gather of the
// score rows on a side stream of the communicator, beside the distance pass, joined in front of the score's planes.
// A failure behind a step's first collective aborts the communicator (ncclCommAbort), so the peers fail instead of hanging.
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
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional: absent from very old builds
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
  r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
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
  hipStream_t side;       // the score rows are gathered here, beside the distance pass on the caller's stream
  hipEvent_t fork, join;
  bool dead;              // aborted after a failure behind the first collective of a step: every later call fails at once
};

// A rank that fails AFTER its peers may already be inside a collective must not simply return: the peers would wait in RCCL
// for ever (the communicator has no watchdog).  Abort it, so that their pending and later collectives fail instead.
int fail_in_step(SteinComm* c, int rc) {
  if (!c->dead) {
    c->dead = true;
    if (g_rccl.CommAbort && c->comm) (void)g_rccl.CommAbort(c->comm);
    c->comm = nullptr;
  }
  return rc;
}

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
  c->side = nullptr;
  c->fork = c->join = nullptr;
  c->dead = false;
  auto cleanup = [&](int code) {
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->flags_ready) (void)hipEventDestroy(c->flags_ready);
    if (c->flags_host) (void)hipHostFree(c->flags_host);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return code;
  };
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->flags_host), 64, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->flags_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming);
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
  (void)hipStreamSynchronize(c->side);
  (void)hipStreamDestroy(c->side);
  (void)hipEventDestroy(c->fork);
  (void)hipEventDestroy(c->join);
  (void)hipEventDestroy(c->flags_ready);
  (void)hipHostFree(c->flags_host);
  ncclResult_t r = c->comm ? g_rccl.CommDestroy(c->comm) : ncclSuccess;   // (an aborted communicator is gone already)
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
  if (c->dead) return stein_fail(STEIN_E_RCCL, "the communicator was aborted after a failed step; make a new one");
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

  // Everything a rank can get wrong by itself has been checked above, before its first collective.  From here on a failure
  // goes through fail_in_step: the communicator is aborted, so the peers' collectives fail instead of waiting for ever.
#define STEP_TRY(expr) do { const int rc_ = (expr); if (rc_) return fail_in_step(c, rc_); } while (0)
#define STEP_RCCL(expr)                                                                                   \
  do {                                                                                                    \
    ncclResult_t r_ = (expr);                                                                             \
    if (r_ != ncclSuccess) return fail_in_step(c, stein_fail(STEIN_E_RCCL, "%s: %s", #expr, g_rccl.GetErrorString(r_))); \
  } while (0)
#define STEP_HIP(expr)                                                                                    \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail_in_step(c, stein_fail(STEIN_E_HIP, "%s: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  // (1) every rank's rows of theta (the caller's stream: the distance pass needs them) and of the score (the side stream:
  // they are not needed before the contraction, so their gather runs beside the distance pass).  RCCL executes the
  // operations of one communicator in the order they were issued, whatever their streams.
  const ncclDataType_t dt = dtype == STEIN_BF16 ? ncclBfloat16 : ncclFloat32;
  const size_t count = (size_t)n_local * (size_t)d;
  STEP_HIP(hipEventRecord(c->fork, stream));                 // the score rows are the caller's stream's product
  STEP_HIP(hipStreamWaitEvent(c->side, c->fork, 0));
  STEP_RCCL(g_rccl.AllGather(theta_local, theta_all, count, dt, c->comm, stream));
  STEP_RCCL(g_rccl.AllGather(score_local, score_all, count, dt, c->comm, c->side));
  STEP_HIP(hipEventRecord(c->join, c->side));

  // (2) row norms, theta's planes, the [n_local, n] distance block with the median's first counts
  STEP_TRY(stein_rank_begin(theta_all, n, d, row0, n_local, dtype, workspace, ws_bytes, seg_flags, stream));

  auto score_planes = [&]() -> int {   // first reader of the gathered score rows: the caller's stream joins the side stream
    if (hipStreamWaitEvent(stream, c->join, 0) != hipSuccess) return stein_fail(STEIN_E_HIP, "hipStreamWaitEvent failed");
    if (!(flags & STEIN_FLAG_X3)) return STEIN_OK;
    return stein_x3_split(nullptr, score_all, dtype, n, d, L, ws + L.off[STEIN_WS_PLANES], stream);
  };
  auto level_sum = [&](int level) -> int {   // hist[level] summed over the ranks, in place
    u64* h = hist + (size_t)level * 2 * STEIN_HIST_BINS;
    ncclResult_t r_ = g_rccl.AllReduce(h, h, 2 * STEIN_HIST_BINS, ncclUint64, ncclSum, c->comm, stream);
    if (r_ != ncclSuccess) return stein_fail(STEIN_E_RCCL, "ncclAllReduce(histogram): %s", g_rccl.GetErrorString(r_));
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
    STEP_RCCL(g_rccl.AllReduce(table, table, SPEC_TABLE, ncclUint64, ncclSum, c->comm, stream));
    STEP_TRY(stein_rank_pick(n, d, row0, n_local, dtype, workspace, ws_bytes, seg_flags, h2_out, median_out, c->flags_host,
                             stream));
    STEP_HIP(hipEventRecord(c->flags_ready, stream));
    STEP_TRY(score_planes());   // work that does not depend on the flag: covers the host's wake-up
    STEP_HIP(hipEventSynchronize(c->flags_ready));
    hit = c->flags_host[0] != 0;
    const bool skip_l0 = c->flags_host[6] != 0;
    if (!hit) STEP_TRY(radix(!skip_l0));   // every rank read the same table: every rank takes the same branch
  } else {
    STEP_TRY(radix(false));
    STEP_TRY(score_planes());
  }
  if (window_hit_out) *window_hit_out = hit;

  // (4) the contraction on the local rows, phi, and the global |phi|^2
  STEP_TRY(stein_rank_finish(theta_all, score_all, n, d, row0, n_local, dtype, h2_out, phi_local, sqnorm_out, dK_out,
                             workspace, ws_bytes, seg_flags | (flags & STEIN_FLAG_TIMING), stream));
  STEP_RCCL(g_rccl.AllReduce(sqnorm_out, sqnorm_out, 1, ncclFloat64, ncclSum, c->comm, stream));
#undef STEP_TRY
#undef STEP_RCCL
#undef STEP_HIP
  return STEIN_OK;
}
