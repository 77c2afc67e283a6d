// multi_gpu.cpp — the in-process multi-GPU forward of the C ABI (dc_comm_*, dc_forward_batch; round 5), torch-free.
//
// North-star: "batched multi-scale / multi-person inference is embarrassingly parallel and shards images across the 8 GPUs of one
// node with a RCCL-over-xGMI gather of score maps".  Rounds 1-4 had that only in Python on torch.distributed (one process per GPU,
// deepcut_tools/shard.py); a C / C++ caller of the library (the tools/caffe.cpp-style consumer SURVEY 8(b) names,
// /root/reference/tools/caffe.cpp:302-388) had no way to use more than one GPU, and the design's one exchange lived outside the
// product.  Here it is inside: one host thread per executor (mode and device are per THREAD, as in the reference:
// src/caffe/common.cpp:13-20), the images dealt by longest-processing-time-first over H*W (what deepcut_tools/shard.py does across
// ranks), same-shape images of an executor forwarded as one batch, and ONE exchange: the maps of every executor gathered into
// buffers on the root executor's device —
//   RCCL    ncclGroupStart / ncclRecv x (n-1) on the root / ncclSend on every peer / ncclGroupEnd, librccl.so opened with dlopen (no
//           link-time dependency: a 1-GPU box without RCCL still loads the library), one communicator per device from ncclCommInitAll;
//           every peer's payload travels over its own xGMI link into the root (no ring, no reduction — the reference's P2PSync,
//           src/caffe/parallel.cpp:287-322, is a training-time tree of gradient sums and not a model for this);
//   PEER    hipMemcpyPeerAsync from each executor's device (also the loop-back transport when several executors share ONE device: the
//           way the 8-executor path is tested on a 1-GPU box);
// on a communication stream per executor, ordered behind that executor's forwards by events.  Inside an executor's share the work is a
// PIPELINE (round 6): every same-shape group of 8 (float16: 16) or more images is cut into two sub-batches, and while sub-batch j computes, the
// images of j + 1 are staged into the other half of the pinned staging buffer and the maps of j - 1 travel to the host on a copy
// stream and are scattered into the caller's arrays; with the PEER transport each sub-batch's maps also leave for the root device as
// soon as they exist (behind the sub-batch's event), with RCCL the exchange stays ONE grouped send / recv behind the last forward (all
// RCCL calls of a call come from one thread, in one group).  The gathered maps stay on the root device until the next call
// (dc_comm_root_maps: the decode kernels can consume them there); the caller's host arrays, if it gave any, are filled by every
// executor for its own share, from its own device over its own host link, on its own thread.
#include <dlfcn.h>

#include <condition_variable>
#include <thread>

#include "net_internal.h"

namespace dc {

namespace {
// ---- the few RCCL entry points, by name (nccl.h is not needed: opaque communicator, ints, a byte payload) --------------------
struct Rccl {
  void* so = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream) = nullptr;
  int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, void* stream) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
  bool ok() const { return so != nullptr; }
};
constexpr int kNcclUint8 = 1;  // ncclDataType_t: ncclInt8 0, ncclUint8 1, ... (payloads travel as bytes)
Rccl& rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl();
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      x->so = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (x->so) break;
    }
    if (!x->so) {
      x->why = std::string("librccl.so could not be opened: ") + (dlerror() ? dlerror() : "?");
      return x;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(x->so, n);
      if (!p && x->why.empty()) x->why = std::string("librccl.so lacks ") + n;
      return p;
    };
    x->CommInitAll = reinterpret_cast<int (*)(void**, int, const int*)>(sym("ncclCommInitAll"));
    x->CommDestroy = reinterpret_cast<int (*)(void*)>(sym("ncclCommDestroy"));
    x->GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
    x->GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
    x->Send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, void*)>(sym("ncclSend"));
    x->Recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, void*)>(sym("ncclRecv"));
    x->GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
    if (!x->why.empty()) {
      dlclose(x->so);
      x->so = nullptr;
    }
    return x;
  }();
  return *r;
}
#define NCCLCHECK(expr)                                                                                              \
  do {                                                                                                               \
    int e_ = (expr);                                                                                                 \
    if (e_ != 0)                                                                                                     \
      throw DcError(DC_EDEVICE, std::string(#expr) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(e_) : "?")); \
  } while (0)

// a persistent worker: runs closures on its own thread (GPU mode, its device), one at a time
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, quit = false;
  std::string error;
  int code = 0;
  explicit Worker(int device) {
    th = std::thread([this, device] {
      Context::get().mode = DC_MODE_GPU;
      Context::get().device = device;
      for (;;) {
        std::function<void()> j;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return has_job || quit; });
          if (quit) return;
          j = std::move(job);
          has_job = false;
        }
        try {
          j();
        } catch (const DcError& e) {
          code = e.code, error = e.what();
        } catch (const std::exception& e) {
          code = DC_EINVAL, error = e.what();
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          done = true;
        }
        cv.notify_all();
      }
    });
  }
  void start(std::function<void()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true, done = false, code = 0, error.clear();
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
  ~Worker() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};
}  // namespace

// Longest-processing-time-first: items by cost descending (ties: lower index first), each to the least-loaded executor (ties: the
// lower executor).  Deterministic, so every party can recompute it (the payload sizes of the gather follow from it).
std::vector<std::vector<int>> lpt_schedule(const std::vector<double>& cost, int nexec) {
  std::vector<int> order(cost.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
  std::vector<std::vector<int>> out((size_t)nexec);
  std::vector<double> load((size_t)nexec, 0.0);
  for (int i : order) {
    int best = 0;
    for (int k = 1; k < nexec; ++k)
      if (load[(size_t)k] < load[(size_t)best]) best = k;
    out[(size_t)best].push_back(i);
    load[(size_t)best] += cost[(size_t)i];
  }
  for (auto& o : out) std::sort(o.begin(), o.end());  // a share in image order (deepcut_tools/shard.py:lpt_shards gives the same lists)
  return out;
}

struct Comm {
  int nexec = 0;
  std::vector<int> devices;
  int transport = DC_COMM_PEER;
  std::vector<void*> nccl;  // one communicator per executor (RCCL transport)
  std::vector<std::unique_ptr<Worker>> workers;
  std::vector<void*> comm_stream, fwd_done;  // per executor: its communication stream, the event behind its forwards
  std::vector<void*> copy_stream;            // per executor: the stream its maps travel to the host on, beside the next sub-batch's forward
  std::vector<std::vector<void*>> sub_ev;    // per executor: two events per sub-batch (forward + emit done, download done), reused across calls
  struct Buf {
    unsigned char* p = nullptr;
    size_t cap = 0;
  };
  std::vector<Buf> send, recv;  // send[k] on executor k's device; recv[k] on the root's (recv[0] unused: the root's maps are in send[0])
  std::vector<Buf> stage;       // pinned host: executor k's input batch
  std::vector<Buf> hout;        // pinned host: executor k's maps on their way to the caller's arrays
  // the last forward: what every executor ran (its same-shape groups, in order) and where every image's maps are
  struct Group {
    std::vector<int> idx;  // images of the group, batch position = position here
    int h = 0, w = 0;
    size_t off = 0;        // byte offset of the group's maps inside the executor's payload: [prob nb][loc nb][next nb], NCHW float32
    int pc = 0, lc = 0, nc = 0, mh = 0, mw = 0;
  };
  std::vector<std::vector<Group>> plan;  // per executor
  struct Item {
    int exec = -1, group = 0, pos = 0;
  };
  std::vector<Item> items;
  std::mutex call_mu;  // one forward at a time per communicator (its buffers and workers are the call's)

  ~Comm();
  void grow_dev(Buf& b, size_t bytes, int device);
  void grow_host(Buf& b, size_t bytes);
  void forward(Net* const* nets, const float* const* inputs, const int (*hw)[2], int n, float* const* prob, float* const* loc, float* const* next);
  const unsigned char* root_base(int k) const { return k == 0 ? send[0].p : recv[(size_t)k].p; }
};

Comm::~Comm() {
  workers.clear();  // joins the threads
  for (size_t k = 0; k < devices.size(); ++k) {
    (void)hipSetDevice(devices[k]);
    if (k < comm_stream.size() && comm_stream[k]) (void)hipStreamDestroy((hipStream_t)comm_stream[k]);
    if (k < copy_stream.size() && copy_stream[k]) (void)hipStreamDestroy((hipStream_t)copy_stream[k]);
    if (k < sub_ev.size())
      for (void* e : sub_ev[k])
        if (e) (void)hipEventDestroy((hipEvent_t)e);
    if (k < fwd_done.size() && fwd_done[k]) (void)hipEventDestroy((hipEvent_t)fwd_done[k]);
    if (k < send.size()) dev_free(send[k].p);
    if (k < recv.size()) dev_free(recv[k].p);
    if (k < stage.size() && stage[k].p) {
      try {
        host_free_pinned(stage[k].p);
      } catch (...) {
      }
    }
    if (k < nccl.size() && nccl[k] && rccl().ok()) (void)rccl().CommDestroy(nccl[k]);
  }
  if (!devices.empty()) (void)hipSetDevice(devices[0]);
  for (auto& b : hout)
    if (b.p) {
      try {
        host_free_pinned(b.p);
      } catch (...) {
      }
    }
}
// device / pinned buffers grow only; allocation and release go through the runtime lock (they exclude graph captures, net_internal.h)
void Comm::grow_dev(Buf& b, size_t bytes, int device) {
  if (b.cap >= bytes) return;
  HIPCHECK(hipSetDevice(device));
  dev_free(b.p);
  b.p = nullptr, b.cap = 0;
  dev_alloc(reinterpret_cast<void**>(&b.p), bytes);
  b.cap = bytes;
}
void Comm::grow_host(Buf& b, size_t bytes) {
  if (b.cap >= bytes) return;
  if (b.p) host_free_pinned(b.p);
  b.p = nullptr, b.cap = 0;
  b.p = reinterpret_cast<unsigned char*>(host_alloc_pinned(bytes));
  b.cap = bytes;
}

namespace {
// the thread's current HIP device is the caller's business: dc_comm_create / dc_forward_batch switch devices to reach every executor
// and put back what they found
struct DeviceRestore {
  int dev = -1;
  DeviceRestore() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1, (void)hipGetLastError();
  }
  ~DeviceRestore() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};
}  // namespace

// One byte from every peer to the root through the communicators just made: the whole transport (peer buffers, IPC handles, the
// grouped send / recv) exercised once before a forward depends on it.  Throws what RCCL or the runtime reports.
static void rccl_probe(Comm& c) {
  std::vector<unsigned char*> buf((size_t)c.nexec, nullptr);
  struct Free {
    std::vector<unsigned char*>& b;
    ~Free() {
      for (unsigned char* p : b) dev_free(p);
    }
  } guard{buf};
  for (int k = 0; k < c.nexec; ++k) {
    HIPCHECK(hipSetDevice(c.devices[(size_t)(k == 0 ? 0 : k)]));
    dev_alloc(reinterpret_cast<void**>(&buf[(size_t)k]), 256 * (size_t)(k == 0 ? c.nexec : 1));
  }
  NCCLCHECK(rccl().GroupStart());
  try {
    for (int k = 1; k < c.nexec; ++k) {
      HIPCHECK(hipSetDevice(c.devices[0]));
      NCCLCHECK(rccl().Recv(buf[0] + 256 * (size_t)k, 1, kNcclUint8, k, c.nccl[0], c.comm_stream[0]));
      HIPCHECK(hipSetDevice(c.devices[(size_t)k]));
      NCCLCHECK(rccl().Send(buf[(size_t)k], 1, kNcclUint8, 0, c.nccl[(size_t)k], c.comm_stream[(size_t)k]));
    }
  } catch (...) {
    (void)rccl().GroupEnd();
    throw;
  }
  NCCLCHECK(rccl().GroupEnd());
  for (int k = 0; k < c.nexec; ++k) {
    HIPCHECK(hipSetDevice(c.devices[(size_t)k]));
    HIPCHECK(hipStreamSynchronize((hipStream_t)c.comm_stream[(size_t)k]));
  }
}

Comm* comm_create(int nexec, const int* devices, int transport) {
  if (nexec < 1 || nexec > 64) throw DcError(DC_EINVAL, "dc_comm_create: 1..64 executors");
  const int ndev = device_count();
  if (ndev <= 0) throw DcError(DC_EDEVICE, "no HIP device visible: libdeepcut_hip has no CPU compute path");
  DeviceRestore restore;
  std::unique_ptr<Comm> c(new Comm());
  c->nexec = nexec;
  bool distinct = true;
  for (int k = 0; k < nexec; ++k) {
    const int d = devices ? devices[k] : k % ndev;
    if (d < 0 || d >= ndev) throw DcError(DC_EDEVICE, "dc_comm_create: device " + std::to_string(d) + " out of range (" + std::to_string(ndev) + " visible)");
    for (int o : c->devices) distinct = distinct && o != d;
    c->devices.push_back(d);
  }
  const bool automatic = transport == DC_COMM_AUTO;
  if (automatic) transport = (distinct && nexec > 1 && rccl().ok()) ? DC_COMM_RCCL : DC_COMM_PEER;
  if (transport != DC_COMM_RCCL && transport != DC_COMM_PEER) throw DcError(DC_EINVAL, "dc_comm_create: transport must be DC_COMM_AUTO, DC_COMM_RCCL or DC_COMM_PEER");
  if (transport == DC_COMM_RCCL) {
    if (!rccl().ok()) throw DcError(DC_EDEVICE, "DC_COMM_RCCL: " + rccl().why);
    if (!distinct) throw DcError(DC_EINVAL, "DC_COMM_RCCL needs one executor per device (executors sharing a device use DC_COMM_PEER)");
  }
  c->send.resize((size_t)nexec), c->recv.resize((size_t)nexec), c->stage.resize((size_t)nexec), c->hout.resize((size_t)nexec);
  c->comm_stream.assign((size_t)nexec, nullptr), c->fwd_done.assign((size_t)nexec, nullptr), c->copy_stream.assign((size_t)nexec, nullptr);
  c->sub_ev.assign((size_t)nexec, {});
  for (int k = 0; k < nexec; ++k) {
    HIPCHECK(hipSetDevice(c->devices[(size_t)k]));
    hipStream_t st;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->comm_stream[(size_t)k] = st;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->copy_stream[(size_t)k] = st;
    hipEvent_t ev;
    HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->fwd_done[(size_t)k] = ev;
  }
  if (transport == DC_COMM_RCCL) {
    // DC_COMM_AUTO must not leave the caller with a communicator that cannot move a byte: RCCL's peer buffers need dmabuf IPC on
    // hosts whose driver has no legacy IPC (HSA_ENABLE_IPC_MODE_LEGACY=0, include/deepcut_hip.h), and whatever else the node's RCCL
    // dislikes shows up at the first collective — so the communicators are made AND used once here; on any failure the automatic
    // choice falls back to peer copies, an explicit DC_COMM_RCCL reports the error
    try {
      c->nccl.assign((size_t)nexec, nullptr);
      NCCLCHECK(rccl().CommInitAll(c->nccl.data(), nexec, c->devices.data()));
      if (nexec > 1) rccl_probe(*c);
    } catch (const DcError& e) {
      for (void*& cm : c->nccl)
        if (cm) (void)rccl().CommDestroy(cm), cm = nullptr;
      c->nccl.clear();
      (void)hipGetLastError();
      if (!automatic) throw DcError(e.code, std::string("DC_COMM_RCCL: ") + e.what() + " (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported? see deepcut_hip.h)");
      transport = DC_COMM_PEER;
    }
  }
  c->transport = transport;
  for (int k = 0; k < nexec; ++k) {
    if (transport == DC_COMM_PEER && c->devices[(size_t)k] != c->devices[0]) {
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, c->devices[0], c->devices[(size_t)k]);
      if (can) {
        HIPCHECK(hipSetDevice(c->devices[0]));
        const hipError_t e = hipDeviceEnablePeerAccess(c->devices[(size_t)k], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
    c->workers.emplace_back(new Worker(c->devices[(size_t)k]));
  }
  return c.release();
}
void comm_destroy(Comm* c) { delete c; }
int comm_transport(const Comm* c) { return c->transport; }
int comm_nexec(const Comm* c) { return c->nexec; }

void Comm::forward(Net* const* nets, const float* const* inputs, const int (*hw)[2], int n, float* const* prob, float* const* loc, float* const* next) {
  if (n < 0) throw DcError(DC_EINVAL, "dc_forward_batch: negative image count");
  for (int k = 0; k < nexec; ++k) {
    if (!nets[k]) throw DcError(DC_EINVAL, "dc_forward_batch: null net");
    if (nets[k]->device >= 0 && nets[k]->device != devices[(size_t)k])
      throw DcError(DC_EINVAL, "dc_forward_batch: net " + std::to_string(k) + " lives on device " + std::to_string(nets[k]->device) + ", the communicator's executor " +
                                   std::to_string(k) + " on device " + std::to_string(devices[(size_t)k]));
    for (int o = 0; o < k; ++o)
      if (nets[o] == nets[k]) throw DcError(DC_EINVAL, "dc_forward_batch: one net given for two executors (a net is not re-entrant: clone it)");
  }
  for (int i = 0; i < n; ++i)
    if (!inputs || !inputs[i] || hw[i][0] <= 0 || hw[i][1] <= 0) throw DcError(DC_EINVAL, "dc_forward_batch: image " + std::to_string(i) + ": null input or empty shape");
  std::lock_guard<std::mutex> one_call(call_mu);
  DeviceRestore restore;
  items.clear();  // (a call that fails leaves no "last forward" behind: dc_comm_root_maps refuses)
  plan.assign((size_t)nexec, {});
  if (n == 0) return;
  std::vector<Item> placed((size_t)n);
  // ---- the schedule (host only, deterministic): LPT over H*W, then per executor its same-shape groups in order of first appearance,
  //      every group large enough cut into two sub-batches (the units of the executor's pipeline below)
  std::vector<double> cost((size_t)n);
  for (int i = 0; i < n; ++i) cost[(size_t)i] = (double)hw[i][0] * hw[i][1];
  const std::vector<std::vector<int>> share = lpt_schedule(cost, nexec);
  for (int k = 0; k < nexec; ++k) {
    std::vector<Group> whole;
    for (int i : share[(size_t)k]) {
      size_t g = 0;
      while (g < whole.size() && !(whole[g].h == hw[i][0] && whole[g].w == hw[i][1])) ++g;
      if (g == whole.size()) {
        whole.emplace_back();
        whole[g].h = hw[i][0], whole[g].w = hw[i][1];
      }
      whole[g].idx.push_back(i);
    }
    std::vector<Group>& gs = plan[(size_t)k];
    // a sub-batch must stay a batch the kernels run well on: halves of at least 4 images in float32, 8 in float16 (measured on one GPU,
    // 8 executors x 8 images of 544x736: float16 halves of 4 ran 624 images/s where whole batches of 8 run 803; float32 halves of 4
    // 357 against 333) — smaller groups go as one batch, still pipelined against the executor's other groups
    const size_t min_half = nets[k]->dtype == 1 ? 8 : 4;
    for (const Group& g : whole) {
      const size_t nb = g.idx.size(), first = nb >= 2 * min_half ? (nb + 1) / 2 : nb;
      for (size_t b0 = 0; b0 < nb; b0 += (b0 == 0 ? first : nb - first)) {
        Group sub;
        sub.h = g.h, sub.w = g.w;
        sub.idx.assign(g.idx.begin() + (long)b0, g.idx.begin() + (long)(b0 == 0 ? first : nb));
        for (size_t q = 0; q < sub.idx.size(); ++q) placed[(size_t)sub.idx[q]] = Item{k, (int)gs.size(), (int)q};
        gs.push_back(std::move(sub));
      }
    }
  }

  // ---- every executor on its own thread, its sub-batches as a pipeline: while sub-batch j computes (upload, forward and the maps as
  //      NCHW float32 into the send buffer, all enqueued on the net's own stream without a wait), the images of j + 1 are staged in
  //      the other half of the pinned staging buffer and the maps of j - 1 are downloaded on the copy stream and scattered into the
  //      caller's arrays.  PEER transport: each sub-batch's maps leave for the root device behind the sub-batch's event.
  std::vector<size_t> payload((size_t)nexec, 0);
  const bool to_host = prob || loc || next;
  for (int k = 0; k < nexec; ++k) {
    workers[(size_t)k]->start([this, k, nets, inputs, &payload, prob, loc, next, to_host] {
      Net* net = nets[k];
      std::vector<Group>& gs = plan[(size_t)k];
      if (gs.empty()) return;
      const int dev = devices[(size_t)k];
      HIPCHECK(hipSetDevice(dev));
      // sizes first (shape inference, host only): the buffers must not move once maps are being written into them
      size_t total = 0, stage_bytes = 0;
      const size_t in_c = (size_t)net->blobs[net->inputs[0]]->st->dim(1);
      for (Group& g : gs) {
        Storage& in = *net->blobs[net->inputs[0]]->st;
        in.reshape({(int)g.idx.size(), in.dim(1), g.h, g.w});
        net->reshape();
        auto dims = [&](const char* nm) -> const Storage& {
          auto it = net->blob_index.find(nm);
          if (it == net->blob_index.end()) throw DcError(DC_EINVAL, std::string("dc_forward_batch: net has no blob '") + nm + "'");
          return *net->blobs[it->second]->st;
        };
        const Storage &p = dims("prob"), &l = dims("loc_pred"), &x = dims("next_pred");
        g.pc = p.dim(1), g.lc = l.dim(1), g.nc = x.dim(1), g.mh = p.dim(2), g.mw = p.dim(3);
        g.off = total;
        total += g.idx.size() * (size_t)(g.pc + g.lc + g.nc) * g.mh * g.mw * sizeof(float);
        stage_bytes = std::max(stage_bytes, g.idx.size() * in_c * g.h * g.w * sizeof(float));
      }
      stage_bytes = (stage_bytes + 4095) & ~(size_t)4095;
      grow_dev(send[(size_t)k], total, dev);
      grow_host(stage[(size_t)k], 2 * stage_bytes);
      if (to_host) grow_host(hout[(size_t)k], total);
      const bool peer_now = transport == DC_COMM_PEER && k > 0;
      if (peer_now) {
        grow_dev(recv[(size_t)k], total, devices[0]);  // (allocated on the root device; this thread's device is put back)
        HIPCHECK(hipSetDevice(dev));
      }
      std::vector<void*>& ev = sub_ev[(size_t)k];
      while (ev.size() < 2 * gs.size()) {
        hipEvent_t e;
        HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev.push_back(e);
      }
      // pinned arrays of the caller (dc_host_alloc, hipHostMalloc, hipHostRegister) are moved by the DMA engines in place: no staging
      // copy in, no scatter copy out — on one GPU with eight executors the host copies of a 64-image call (0.3 GB in, 0.65 GB out) were
      // most of what the call took beyond its forwards
      auto pinned = [](const void* ptr) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
          (void)hipGetLastError();
          return false;
        }
        return a.type == hipMemoryTypeHost;
      };
      auto group_bytes = [](const Group& g) { return g.idx.size() * (size_t)(g.pc + g.lc + g.nc) * g.mh * g.mw * sizeof(float); };
      std::vector<char> direct_out(gs.size(), 0);
      auto scatter = [&](size_t j) {  // the caller's arrays of sub-batch j, once its download has landed
        const Group& g = gs[j];
        if (direct_out[j]) return;  // (downloaded straight into the caller's pinned arrays: waited for at the end)
        HIPCHECK(hipEventSynchronize((hipEvent_t)ev[2 * j + 1]));
        const int nb = (int)g.idx.size();
        const size_t cell = (size_t)g.mh * g.mw;
        const float* pp = reinterpret_cast<const float*>(hout[(size_t)k].p + g.off);
        const float* lp = pp + (size_t)nb * g.pc * cell;
        const float* np = lp + (size_t)nb * g.lc * cell;
        for (int b = 0; b < nb; ++b) {
          const int i = g.idx[(size_t)b];
          if (prob && prob[i]) std::memcpy(prob[i], pp + (size_t)b * g.pc * cell, (size_t)g.pc * cell * sizeof(float));
          if (loc && loc[i]) std::memcpy(loc[i], lp + (size_t)b * g.lc * cell, (size_t)g.lc * cell * sizeof(float));
          if (next && next[i]) std::memcpy(next[i], np + (size_t)b * g.nc * cell, (size_t)g.nc * cell * sizeof(float));
        }
      };
      for (size_t j = 0; j < gs.size(); ++j) {
        const Group& g = gs[j];
        const int nb = (int)g.idx.size();
        const size_t img = in_c * g.h * g.w, cell = (size_t)g.mh * g.mw;
        bool in_pinned = true, out_pinned = to_host;
        std::vector<const float*> srcs((size_t)nb);
        for (int b = 0; b < nb; ++b) {
          const int i = g.idx[(size_t)b];
          srcs[(size_t)b] = inputs[i];
          in_pinned = in_pinned && pinned(inputs[i]);
          for (float* const* arr : {prob, loc, next})
            if (arr && arr[i]) out_pinned = out_pinned && pinned(arr[i]);
        }
        direct_out[j] = out_pinned;
        float* pp = reinterpret_cast<float*>(send[(size_t)k].p + g.off);
        float* lp = pp + (size_t)nb * g.pc * cell;
        float* np = lp + (size_t)nb * g.lc * cell;
        if (in_pinned) {
          net->forward_host_images(srcs.data(), nb, g.h, g.w);  // uploads straight from the caller's arrays + forward, not waited for
        } else {
          // the staging half of sub-batch j - 2 is free once that sub-batch's work — its upload first of all — is done
          if (j >= 2) HIPCHECK(hipEventSynchronize((hipEvent_t)ev[2 * (j - 2)]));
          float* st = reinterpret_cast<float*>(stage[(size_t)k].p + (j & 1) * stage_bytes);
          for (int b = 0; b < nb; ++b) std::memcpy(st + b * img, srcs[(size_t)b], img * sizeof(float));
          net->forward_batch(st, nb, g.h, g.w, false, nullptr, nullptr, nullptr, nullptr, true);  // upload + forward enqueued on the net's own stream, not waited for
        }
        net->emit_last_maps(pp, lp, np, 0, true, (void*)-1);                                    // ... and the maps, same stream
        HIPCHECK(hipEventRecord((hipEvent_t)ev[2 * j], (hipStream_t)net->stream));
        if (to_host) {
          HIPCHECK(hipStreamWaitEvent((hipStream_t)copy_stream[(size_t)k], (hipEvent_t)ev[2 * j], 0));
          if (out_pinned) {
            for (int b = 0; b < nb; ++b) {
              const int i = g.idx[(size_t)b];
              if (prob && prob[i]) HIPCHECK(hipMemcpyAsync(prob[i], pp + (size_t)b * g.pc * cell, (size_t)g.pc * cell * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)copy_stream[(size_t)k]));
              if (loc && loc[i]) HIPCHECK(hipMemcpyAsync(loc[i], lp + (size_t)b * g.lc * cell, (size_t)g.lc * cell * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)copy_stream[(size_t)k]));
              if (next && next[i]) HIPCHECK(hipMemcpyAsync(next[i], np + (size_t)b * g.nc * cell, (size_t)g.nc * cell * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)copy_stream[(size_t)k]));
            }
          } else {
            HIPCHECK(hipMemcpyAsync(hout[(size_t)k].p + g.off, send[(size_t)k].p + g.off, group_bytes(g), hipMemcpyDeviceToHost, (hipStream_t)copy_stream[(size_t)k]));
          }
          HIPCHECK(hipEventRecord((hipEvent_t)ev[2 * j + 1], (hipStream_t)copy_stream[(size_t)k]));
        }
        if (peer_now) {
          HIPCHECK(hipStreamWaitEvent((hipStream_t)comm_stream[(size_t)k], (hipEvent_t)ev[2 * j], 0));
          if (dev == devices[0])
            HIPCHECK(hipMemcpyAsync(recv[(size_t)k].p + g.off, send[(size_t)k].p + g.off, group_bytes(g), hipMemcpyDeviceToDevice, (hipStream_t)comm_stream[(size_t)k]));
          else
            HIPCHECK(hipMemcpyPeerAsync(recv[(size_t)k].p + g.off, devices[0], send[(size_t)k].p + g.off, dev, group_bytes(g), (hipStream_t)comm_stream[(size_t)k]));
        }
        if (to_host && j >= 1) scatter(j - 1);
      }
      payload[(size_t)k] = total;
      HIPCHECK(hipEventRecord((hipEvent_t)fwd_done[(size_t)k], (hipStream_t)net->stream));
      if (to_host) scatter(gs.size() - 1);
      if (to_host) HIPCHECK(hipStreamSynchronize((hipStream_t)copy_stream[(size_t)k]));  // (the direct downloads, if any)
      HIPCHECK(hipStreamSynchronize((hipStream_t)net->stream));  // (the staging halves and the net are the next call's again)
    });
  }
  std::string err;
  int code = 0;
  for (int k = 0; k < nexec; ++k) {
    workers[(size_t)k]->wait();
    if (workers[(size_t)k]->code && !code) code = workers[(size_t)k]->code, err = "executor " + std::to_string(k) + ": " + workers[(size_t)k]->error;
  }
  if (code) throw DcError(code, err);

  // ---- RCCL: ONE exchange, every executor's payload -> the root's device behind its last forward (all RCCL calls of a call from this
  //      one thread, in one group).  PEER: the copies are already on their way, sub-batch by sub-batch.
  if (transport == DC_COMM_RCCL && nexec > 1) {
    for (int k = 1; k < nexec; ++k) grow_dev(recv[(size_t)k], payload[(size_t)k], devices[0]);
    for (int k = 0; k < nexec; ++k)
      if (payload[(size_t)k]) {
        HIPCHECK(hipSetDevice(devices[(size_t)k]));
        HIPCHECK(hipStreamWaitEvent((hipStream_t)comm_stream[(size_t)k], (hipEvent_t)fwd_done[(size_t)k], 0));
      }
    NCCLCHECK(rccl().GroupStart());
    try {
      for (int k = 1; k < nexec; ++k)
        if (payload[(size_t)k]) {  // (the device of the communicator a call is made on is made current first: one thread drives all of them)
          HIPCHECK(hipSetDevice(devices[0]));
          NCCLCHECK(rccl().Recv(recv[(size_t)k].p, payload[(size_t)k], kNcclUint8, k, nccl[0], comm_stream[0]));
          HIPCHECK(hipSetDevice(devices[(size_t)k]));
          NCCLCHECK(rccl().Send(send[(size_t)k].p, payload[(size_t)k], kNcclUint8, 0, nccl[(size_t)k], comm_stream[(size_t)k]));
        }
    } catch (...) {
      (void)rccl().GroupEnd();  // the group is never left open behind an error
      throw;
    }
    NCCLCHECK(rccl().GroupEnd());
  }
  for (int k = 0; k < nexec; ++k) {
    HIPCHECK(hipSetDevice(devices[(size_t)k]));
    HIPCHECK(hipStreamSynchronize((hipStream_t)comm_stream[(size_t)k]));
  }
  items = std::move(placed);  // only a forward that went through is "the last forward"
}

void comm_forward(Comm* c, Net* const* nets, int nexec, const float* const* inputs, const int (*hw)[2], int n, float* const* prob, float* const* loc,
                  float* const* next) {
  if (nexec != c->nexec) throw DcError(DC_EINVAL, "dc_forward_batch: " + std::to_string(nexec) + " nets for a communicator of " + std::to_string(c->nexec));
  if (n > 0 && !hw) throw DcError(DC_EINVAL, "dc_forward_batch: null shape array");
  c->forward(nets, inputs, hw, n, prob, loc, next);
}
// which executor forwarded image i of the last call (the LPT schedule made visible: tests, diagnostics)
int comm_item_executor(const Comm* c, int i) {
  if (i < 0 || (size_t)i >= c->items.size()) throw DcError(DC_EINVAL, "dc_comm_item_executor: no such image in the last forward");
  return c->items[(size_t)i].exec;
}
// the gathered maps of image i of the last call ON THE ROOT DEVICE (NCHW float32; valid until the next forward on this communicator)
void comm_root_maps(const Comm* c, int i, const void** prob, const void** loc, const void** next, int dims[5]) {
  if (i < 0 || (size_t)i >= c->items.size() || c->items[(size_t)i].exec < 0) throw DcError(DC_EINVAL, "dc_comm_root_maps: no such image in the last forward");
  const Comm::Item& it = c->items[(size_t)i];
  const Comm::Group& g = c->plan[(size_t)it.exec][(size_t)it.group];
  const int nb = (int)g.idx.size();
  const size_t cell = (size_t)g.mh * g.mw;
  const float* pp = reinterpret_cast<const float*>(c->root_base(it.exec) + g.off);
  const float* lp = pp + (size_t)nb * g.pc * cell;
  const float* np = lp + (size_t)nb * g.lc * cell;
  if (prob) *prob = pp + (size_t)it.pos * g.pc * cell;
  if (loc) *loc = lp + (size_t)it.pos * g.lc * cell;
  if (next) *next = np + (size_t)it.pos * g.nc * cell;
  if (dims) dims[0] = g.pc, dims[1] = g.lc, dims[2] = g.nc, dims[3] = g.mh, dims[4] = g.mw;
}

}  // namespace dc
