// The data-parallel reducer below Python (last row): what a host in any language needs to reduce
// the gradient arena's buckets over RCCL while backward is still running -- the transport half next to the notification half
// (wavlm_dp_set_listener, rowops.hip).  Reference semantics: one all-reduce of the flat gradient buffer per bucket, averaged over
// the ranks (src/fairseq/distributed/legacy_distributed_data_parallel.py:82-120, src/fairseq/distributed/utils.py:273-288);
// here the buckets go out in the order the caller reports them, on a communication stream of their own, each behind an event
// on the stream that wrote it, and wavlm_dp_finish makes that stream wait for all of them (the reference's flush at the end of
// backward, legacy_distributed_data_parallel.py:122-165).
//
// RCCL is NOT a link-time dependency of libwavlm_hip.so: a process that never calls wavlm_dp_init never touches it, and a process
// that already carries an RCCL (PyTorch loads its own copy) must not get a second one -- the entry points are resolved at the
// first wavlm_dp_unique_id / wavlm_dp_init: the copy that is already loaded (dlopen RTLD_NOLOAD), else the system's.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"
#include "../../include/wavlm_hip.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::mutex g_mu;

bool rccl_load() {  // g_mu held
  if (g_rccl.ok) return true;
  static const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", nullptr};
  void* h = nullptr;
  for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);  // a copy this process already carries
  for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return false;
  g_rccl.lib = h;
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
  return g_rccl.ok;
}

constexpr int DP_EVENTS = 64;  // ring of "bucket written" events: a bucket's event is consumed by the communication stream's
                               // wait long before 64 more buckets have been reported
struct DpState {
  bool up = false;
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  ncclRedOp_t op = ncclSum;
  hipStream_t cstream = nullptr;
  hipEvent_t ready[DP_EVENTS] = {};
  hipEvent_t done = nullptr;
  int next = 0;
  uint64_t pending = 0;  // buckets enqueued since the last finish
};
DpState g_dp;

void dp_teardown() {  // g_mu held
  if (g_dp.comm && g_rccl.ok) g_rccl.CommDestroy(g_dp.comm);
  for (int i = 0; i < DP_EVENTS; ++i)
    if (g_dp.ready[i]) (void)hipEventDestroy(g_dp.ready[i]);
  if (g_dp.done) (void)hipEventDestroy(g_dp.done);
  if (g_dp.cstream) (void)hipStreamDestroy(g_dp.cstream);
  g_dp = DpState();
}

}  // namespace

extern "C" {

int wavlm_dp_unique_id(void* id128) {
  if (!id128) return WL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!rccl_load()) return WL_ELAUNCH;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return WL_ELAUNCH;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  __builtin_memcpy(id128, &id, sizeof(id));
  return WL_OK;
}

int wavlm_dp_init(int32_t rank, int32_t world, const void* id128, int32_t average) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) return WL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_dp.up) return WL_EINVAL;  // one communicator per process (one process per GPU); wavlm_dp_destroy first
  if (!rccl_load()) return WL_ELAUNCH;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WL_ELAUNCH;
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  g_dp.rank = rank; g_dp.world = world; g_dp.device = dev;
  g_dp.op = average ? ncclAvg : ncclSum;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = the numerically lowest value = the highest priority
  if (hipStreamCreateWithPriority(&g_dp.cstream, hipStreamNonBlocking, hi) != hipSuccess) { dp_teardown(); return WL_ELAUNCH; }
  for (int i = 0; i < DP_EVENTS; ++i)
    if (hipEventCreateWithFlags(&g_dp.ready[i], hipEventDisableTiming) != hipSuccess) { dp_teardown(); return WL_ELAUNCH; }
  if (hipEventCreateWithFlags(&g_dp.done, hipEventDisableTiming) != hipSuccess) { dp_teardown(); return WL_ELAUNCH; }
  if (g_rccl.CommInitRank(&g_dp.comm, world, id, rank) != ncclSuccess) { g_dp.comm = nullptr; dp_teardown(); return WL_ELAUNCH; }
  g_dp.up = true;
  return WL_OK;
}

int wavlm_dp_bucket_ready(void* base, uint64_t count, int32_t dtype, void* compute_stream) {
  if (!base || (dtype != WL_F32 && dtype != WL_BF16)) return WL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_dp.up) return WL_EINVAL;
  if (count == 0) return WL_OK;
  hipStream_t st = (hipStream_t)compute_stream;
  hipEvent_t ev = g_dp.ready[g_dp.next];
  g_dp.next = (g_dp.next + 1) % DP_EVENTS;
  if (hipEventRecord(ev, st) != hipSuccess) return WL_ELAUNCH;                 // the bucket is complete in `st` order here
  if (hipStreamWaitEvent(g_dp.cstream, ev, 0) != hipSuccess) return WL_ELAUNCH;
  if (g_rccl.AllReduce(base, base, (size_t)count, dtype == WL_BF16 ? ncclBfloat16 : ncclFloat32, g_dp.op, g_dp.comm,
                       g_dp.cstream) != ncclSuccess)
    return WL_ELAUNCH;
  ++g_dp.pending;
  return WL_OK;
}

int wavlm_dp_finish(void* compute_stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_dp.up) return WL_EINVAL;
  if (g_dp.pending == 0) return WL_OK;
  // streams are in order: behind the last bucket's all-reduce = behind all of them
  if (hipEventRecord(g_dp.done, g_dp.cstream) != hipSuccess) return WL_ELAUNCH;
  if (hipStreamWaitEvent((hipStream_t)compute_stream, g_dp.done, 0) != hipSuccess) return WL_ELAUNCH;
  g_dp.pending = 0;
  return WL_OK;
}

int wavlm_dp_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_dp.up) return WL_OK;
  (void)hipStreamSynchronize(g_dp.cstream);
  dp_teardown();
  return WL_OK;
}

}  // extern "C"
