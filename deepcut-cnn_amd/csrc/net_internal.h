// net_internal.h — what the translation units of the graph runtime share and nobody else sees (net.h is the interface).
// Round 5: csrc/net.cpp (3 700 lines) was split by concern, no behaviour change:
//   runtime.cpp   Caffe context, the process-wide runtime lock, device allocation / fills / uploads, Storage (SyncedMemory + Blob)
//   net_init.cpp  Net::Init semantics: InsertSplits, layer set-up and shape inference, weight files
//   net_lower.cpp the lowering to a launch plan and the per-shape plan cache
//   net_tune.cpp  tile selection by measurement, the tune-cache file, set_tile / reports
//   net_run.cpp   execution: uploads, SyncedMemory moves, launches, hipGraph capture, the batch / request entries, map output
//   net_image.cpp image entry (pre-processing on the device), pose decode, multi-person consumers, plan / profile / debug text
//   net_group.cpp NetGroup: pyramid-grouped execution
//   streams.cpp   the process-wide pool of executor streams chosen by measurement (round 5)
//   multi_gpu.cpp in-process multi-GPU forward of the C ABI (round 5)
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <sstream>
#include <string>

#include "../../include/deepcut_hip.h"
#include "net.h"

namespace dc {

#define HIPCHECK(expr)                                                                               \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      throw DcError(DC_EDEVICE, std::string(#expr) + " failed: " + hipGetErrorString(e_));           \
  } while (0)
#define KCHECK(expr)                                                                                 \
  do {                                                                                               \
    int e_ = (expr);                                                                                 \
    if (e_ != 0)                                                                                     \
      throw DcError(DC_EDEVICE, std::string(#expr) + " failed: " + hipGetErrorString((hipError_t)e_)); \
  } while (0)

// ---- runtime.cpp ----------------------------------------------------------------------------------
std::recursive_mutex& runtime_mu();
struct RuntimeLock {
  std::lock_guard<std::recursive_mutex> lk;
  RuntimeLock() : lk(runtime_mu()) {}
};
hipStream_t util_stream();  // utility stream of the current device (non-blocking, never destroyed); caller holds the runtime lock
void dev_zero(void* p, size_t bytes, void* s);
void dev_upload(void* dst, const void* src, size_t bytes, void* s);
void dev_free(void* p);
void dev_alloc(void** p, size_t bytes);
void storage_to_device_impl(Storage& s, void* stream, bool wait);
// library's own and only kernel launches are recorded — nothing another thread does can belong to the capture.
template <class F>
hipGraphExec_t capture_graph(void* cs, F&& enqueue) {
  RuntimeLock rl;
  hipGraph_t graph;
  HIPCHECK(hipStreamBeginCapture((hipStream_t)cs, hipStreamCaptureModeRelaxed));
  try {
    enqueue(cs);
  } catch (...) {
    hipGraph_t g2;
    (void)hipStreamEndCapture((hipStream_t)cs, &g2);
    throw;
  }
  HIPCHECK(hipStreamEndCapture((hipStream_t)cs, &graph));
  hipGraphExec_t ge;
  const hipError_t e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) throw DcError(DC_EDEVICE, std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e));
  return ge;
}

// ---- streams.cpp ------------------------------------------------------------------------------------
std::vector<void*> executor_stream_pool(int device, size_t want);  // a COPY of the first `want` process-wide candidates
void pool_stream_acquire(void* s);
void pool_stream_release(void* s);
int pool_stream_users(void* s);

// ---- small helpers -----------------------------------------------------------------------------------
inline int env_int(const char* k, int def) {
  const char* v = std::getenv(k);
  return v ? std::atoi(v) : def;
}
uint64_t content_hash(const float* p, size_t n);               // net_lower.cpp
void load_tune_cache_locked(ModelShared& shared);        // net_tune.cpp: DC_TUNE_CACHE file -> table (once per model)
void write_tune_cache_locked(ModelShared& shared);       // net_tune.cpp: table (united with the file) -> file, atomically

}  // namespace dc
