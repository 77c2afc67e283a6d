// runtime.cpp — see net_internal.h: context, runtime lock, device memory, Storage, SyncedMemory moves.
#include "net_internal.h"

namespace dc {

// ---- context ------------------------------------------------------------------------------------
Context& Context::get() {
  static thread_local Context c;
  return c;
}
int device_count() {
  static int cached = -1;
  if (cached >= 0) return cached;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  (void)hipGetLastError();
  cached = n;
  return n;
}

// ---- stream capture against everything else ------------------------------------------------------
// While ANY stream of the process is being captured, this HIP runtime fails the synchronous legacy-stream calls of EVERY thread
// (hipMemset, hipMemcpy: hipErrorStreamCaptureImplicit, "would make the legacy stream depend on a capturing blocking stream" —
// whatever the capturing stream's flags and the capture mode) and invalidates the capture on top; allocation and release
// synchronise the device, which a capturing stream cannot take either.  Seen with three groups driven by three host threads
// (tools/stress_groups.py): one thread's buffer growth killed another's graph capture.  So (1) the library makes no
// legacy-stream call: fills and uploads go asynchronously to a utility stream of its own and are waited for there; and (2)
// its captures and its allocations / releases / device-wide waits exclude each other through one process-wide lock (captures
// take a millisecond and happen once per shape; allocations likewise).  Lock order: ModelShared::mu before this one.
std::recursive_mutex& runtime_mu() {
  static std::recursive_mutex* m = new std::recursive_mutex();  // never destroyed: blobs may be released at exit
  return *m;
}
// utility stream of the current device (non-blocking, never destroyed); caller holds the runtime lock
hipStream_t util_stream() {
  static std::map<int, hipStream_t> streams;
  int d = 0;
  HIPCHECK(hipGetDevice(&d));
  auto it = streams.find(d);
  if (it != streams.end()) return it->second;
  hipStream_t st;
  HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  streams[d] = st;
  return st;
}
// zero-fill / upload, complete on return, no legacy stream involved: on the executor's own stream `s` — the utility stream is
// for blobs that belong to no net only (a stream more in the process moves every later stream to another hardware queue:
// with a utility stream created beside the first executor, four forwards in flight fell from 486 to 429 images/s)
void dev_zero(void* p, size_t bytes, void* s) {
  RuntimeLock rl;
  hipStream_t us = s ? (hipStream_t)s : util_stream();
  HIPCHECK(hipMemsetAsync(p, 0, bytes, us));
  HIPCHECK(hipStreamSynchronize(us));
}
void dev_upload(void* dst, const void* src, size_t bytes, void* s) {
  RuntimeLock rl;
  hipStream_t us = s ? (hipStream_t)s : util_stream();
  HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, us));
  HIPCHECK(hipStreamSynchronize(us));
}
void dev_free(void* p) {
  if (!p) return;
  RuntimeLock rl;
  (void)hipFree(p);
}
// pinned (page-locked) host memory: allocation and release are device-wide events like hipMalloc / hipFree and exclude graph captures
void* host_alloc_pinned(size_t bytes) {
  RuntimeLock rl;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    throw DcError(DC_EDEVICE, "hipHostMalloc of " + std::to_string(bytes) + " bytes failed");
  }
  return p;
}
void host_free_pinned(void* p) {
  if (!p) return;
  RuntimeLock rl;
  if (hipHostFree(p) != hipSuccess) {
    (void)hipGetLastError();
    throw DcError(DC_EINVAL, "not a pinned allocation of this library");
  }
}
void dev_alloc(void** p, size_t bytes) {
  RuntimeLock rl;
  HIPCHECK(hipMalloc(p, bytes));
}
// The launches `enqueue(cs)` puts on stream cs, as an executable graph.  Relaxed mode: cs is a non-blocking stream of the
// ---- Storage ------------------------------------------------------------------------------------
Storage::~Storage() {
  if (host) {
    if (host_pinned) {
      RuntimeLock rl;
      (void)hipHostFree(host);
    } else std::free(host);
  }
  dev_free(dev);
  dev_free(stage);
}
size_t Storage::count() const {
  size_t c = 1;
  for (int d : shape) c *= (size_t)d;
  return c;
}
int Storage::cp() const {
  int c = dim(1);
  const int v = 16 / esize;  // elements per 16-byte vector
  return pad4 ? (c + v - 1) / v * v : c;
}
size_t Storage::dev_count() const { return (size_t)dim(0) * dim(2) * dim(3) * cp(); }
void Storage::reshape(const std::vector<int>& s) {
  long long total = 1;
  for (int d : s) {
    if (d < 0) throw DcError(DC_ESHAPE, "negative blob dimension");
    // Blob::Reshape: CHECK_LE(shape[i], INT_MAX / count_) << "blob size exceeds INT_MAX" (blob.cpp:31-34) — four dimensions of
    // 65536 would otherwise wrap the element count to 0
    if (d != 0 && total > 0x7fffffffLL / d) throw DcError(DC_ESHAPE, "blob size exceeds INT_MAX");
    total *= d;
  }
  shape = s;
  if (count() > host_cap && host) {
    // Blob::Reshape replaces the SyncedMemory when capacity grows (blob.cpp:37-41)
    if (host_pinned) {
      RuntimeLock rl;
      (void)hipHostFree(host);
    } else std::free(host);
    host = nullptr;
    host_cap = 0;
    head = UNINITIALIZED;
  }
  // the same on the device side: a blob that lives only there (HEAD_AT_GPU without a host copy: mutable_gpu_data of a
  // stand-alone blob, a Layer top, a net input after a device-side forward) and is reshaped beyond its device allocation
  // must not keep handing out the old, too small pointer — gpu_data()/mutable_gpu_data()/to_host all return early on
  // HEAD_AT_GPU.  Back to UNINITIALIZED: the next device access allocates (zero-filled), as a fresh SyncedMemory would.
  if (dev && std::max<size_t>(dev_count(), 8) * (size_t)esize > dev_cap && (head == HEAD_AT_GPU || head == SYNCED)) {
    if (head == SYNCED && host && count() <= host_cap) head = HEAD_AT_CPU;  // the (large enough) host copy stays authoritative
    else {
      head = UNINITIALIZED;
      // a fresh SyncedMemory hands out zero-filled memory on first touch (syncedmem.cpp:25-31): a host buffer kept from
      // before (older than the device image that is now gone) must not show through host_ptr()
      if (host) std::memset(host, 0, host_cap * sizeof(float));
    }
  }
}
float* Storage::host_ptr() {
  size_t n = std::max<size_t>(count(), 1);
  if (!host) {
    host_pinned = false;
    if (device_count() > 0 && !is_param) {
      void* p = nullptr;
      RuntimeLock rl;
      if (hipHostMalloc(&p, n * sizeof(float), hipHostMallocDefault) == hipSuccess) {
        host = (float*)p;
        host_pinned = true;
      } else {
        (void)hipGetLastError();
      }
    }
    if (!host) host = (float*)std::malloc(n * sizeof(float));
    if (!host) throw DcError(DC_EDEVICE, "out of host memory");
    std::memset(host, 0, n * sizeof(float));
    host_cap = n;
  }
  return host;
}
void Storage::ensure_dev(size_t n) {
  const size_t bytes = std::max<size_t>(n, 8) * (size_t)esize;
  if (bytes <= dev_cap && dev) return;
  dev_free(dev);
  dev = nullptr;
  dev_alloc((void**)&dev, bytes);
  // pitch-padding channels stay 0.  The fill is complete when dev_zero returns: the executors' streams are non-blocking and
  // order themselves after nothing, a fill still in flight could land AFTER the first kernels of a forward had written the
  // buffer (seen once as garbage in a clone's first request, when the fill went to the NULL stream unwaited).
  dev_zero(dev, bytes, owner ? owner->stream : nullptr);
  dev_cap = bytes;
  if (owner) {  // captured graphs carry the old address: they are re-captured lazily (PlanState::graph_buf_gen)
    ++owner->buf_gen_;
    ++owner->stats.buffer_growths;
  }
}
void Storage::ensure_stage(size_t n) {
  if (n <= stage_cap && stage) return;
  dev_free(stage);
  stage = nullptr;
  dev_alloc((void**)&stage, std::max<size_t>(n, 4) * sizeof(float));
  stage_cap = n;
}

void storage_to_device(Storage& s, void* stream) { storage_to_device_impl(s, stream, true); }
// wait = false: the upload is enqueued and the caller synchronises the stream before the host copy can change again
// (Net::forward: its own final synchronisation covers the inputs it sent up — one round trip less per forward)
void storage_to_device_impl(Storage& s, void* stream, bool wait) {
  if (s.head == HEAD_AT_GPU || s.head == SYNCED) return;
  if (device_count() <= 0) throw DcError(DC_EDEVICE, "no HIP device visible");
  size_t n = s.count();
  s.ensure_dev(s.dev_count());
  if (s.head == UNINITIALIZED) {
    HIPCHECK(hipMemsetAsync(s.dev, 0, std::max<size_t>(s.dev_count(), 8) * (size_t)s.esize, (hipStream_t)stream));
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    s.head = HEAD_AT_GPU;
    return;
  }
  if (s.shape.size() == 4) {
    s.ensure_stage(n);
    HIPCHECK(hipMemcpyAsync(s.stage, s.host_ptr(), n * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    KCHECK(launch_nchw_to_nhwc(s.stage, s.dev, s.esize, s.dim(0), s.dim(1), s.dim(2), s.dim(3), s.cp(), stream));
  } else {
    if (s.esize != 4) throw DcError(DC_EUNSUP, "only 4-D blobs have a half-precision device image");
    HIPCHECK(hipMemcpyAsync(s.dev, s.host_ptr(), n * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
  }
  if (wait) HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  s.head = SYNCED;
}

void storage_mutable_device(Storage& s, void* stream) {  // syncedmem.cpp:130-139
  storage_to_device(s, stream);
  s.head = HEAD_AT_GPU;
}

// SyncedMemory::to_cpu (syncedmem.cpp:25-47)
void storage_to_host(Storage& s, void* stream, Storage* base) {
  s.host_touched = true;
  if (s.head != HEAD_AT_GPU) {
    s.host_ptr();
    if (s.head == UNINITIALIZED) s.head = HEAD_AT_CPU;
    return;
  }
  s.host_wanted = true;  // read on demand once: the next forwards deliver it (Net::forward)
  storage_download_enqueue(s, stream, base);
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  s.head = SYNCED;
}

void storage_download_enqueue(Storage& s, void* stream, Storage* base) {
  size_t n = s.count();
  float* h = s.host_ptr();
  if (base) {  // channel slice of a concatenated tensor
    s.ensure_stage(n);
    KCHECK(launch_nhwc_to_nchw(base->dev, s.stage, base->esize, s.dim(0), s.dim(1), s.dim(2), s.dim(3), base->cp(), s.view_c0, stream));
    HIPCHECK(hipMemcpyAsync(h, s.stage, n * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  } else if (s.shape.size() == 4) {
    s.ensure_stage(n);
    KCHECK(launch_nhwc_to_nchw(s.dev, s.stage, s.esize, s.dim(0), s.dim(1), s.dim(2), s.dim(3), s.cp(), 0, stream));
    HIPCHECK(hipMemcpyAsync(h, s.stage, n * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  } else {
    if (s.esize != 4) throw DcError(DC_EUNSUP, "only 4-D blobs have a half-precision device image");
    HIPCHECK(hipMemcpyAsync(h, s.dev, n * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  }
}

void storage_copy(Storage& dst, Storage& src, Storage* src_base, void* stream) {
  if (dst.count() != src.count()) throw DcError(DC_ESHAPE, "Trying to copy blobs of different sizes.");  // blob.cpp:437-443
  if (&dst == &src) return;
  const size_t n = src.count();
  if (src.head == HEAD_AT_GPU && dst.is_param) {
    // A parameter's authoritative image is its HOST copy: filter packing reads it and the weights generation is driven by
    // its content hash.  A device-to-device copy would leave that copy stale and the forward would keep the old weights
    // (layer->blobs()[0]->CopyFrom(gpu_blob) silently ignored).  Bring the source to the host and copy there.
    storage_to_host(src, stream, src_base);
  }
  if (src.head == HEAD_AT_GPU) {  // device -> device; the two images may differ in channel pitch / element type
    dst.ensure_dev(dst.dev_count());
    if (src.shape.size() == 4 && dst.shape.size() == 4) {
      if (dst.shape != src.shape) throw DcError(DC_ESHAPE, "device copy needs equal 4-D shapes");
      dst.ensure_stage(n);
      Storage& img = src_base ? *src_base : src;
      KCHECK(launch_nhwc_to_nchw(img.dev, dst.stage, img.esize, src.dim(0), src.dim(1), src.dim(2), src.dim(3), img.cp(),
                                 src_base ? src.view_c0 : 0, stream));
      KCHECK(launch_nchw_to_nhwc(dst.stage, dst.dev, dst.esize, dst.dim(0), dst.dim(1), dst.dim(2), dst.dim(3), dst.cp(), stream));
    } else if (src.shape.size() != 4 && dst.shape.size() != 4) {
      HIPCHECK(hipMemcpyAsync(dst.dev, src.dev, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    } else {
      throw DcError(DC_ESHAPE, "device copy between a 4-D and a non-4-D blob");
    }
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    dst.head = HEAD_AT_GPU;
    return;
  }
  std::memcpy(dst.host_ptr(), src.host_ptr(), n * sizeof(float));  // UNINITIALIZED source: zeros (first touch zero-fills)
  dst.head = HEAD_AT_CPU;
}

void Net::sync_to_device(Storage& s) {
  if (s.head == HEAD_AT_GPU || s.head == SYNCED) return;
  ensure_device();
  storage_to_device(s, stream);
}

void Net::sync_to_host(Storage& s) {
  if (s.head == HEAD_AT_GPU) ensure_device();
  storage_to_host(s, stream, s.view_of >= 0 ? storages[s.view_of].get() : nullptr);
}

// Layer<Dtype>::SetUp for one reference layer: the layer's bottoms become the inputs of a one-layer net (same names, the
// given shapes), so that Layer::Reshape / Forward_gpu are Net::reshape / Net::forward of that net with no fusion.
}  // namespace dc
