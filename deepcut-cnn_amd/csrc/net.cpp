// net.cpp — see net.h.  Host-side graph runtime: parse, InsertSplits, shape inference, weight
// loading, lowering to a fused launch plan, and execution on one HIP stream.
#include "net.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <sstream>

#include "../../include/deepcut_hip.h"

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
static std::recursive_mutex& runtime_mu() {
  static std::recursive_mutex* m = new std::recursive_mutex();  // never destroyed: blobs may be released at exit
  return *m;
}
struct RuntimeLock {
  std::lock_guard<std::recursive_mutex> lk;
  RuntimeLock() : lk(runtime_mu()) {}
};
// utility stream of the current device (non-blocking, never destroyed); caller holds the runtime lock
static hipStream_t util_stream() {
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
static void dev_zero(void* p, size_t bytes, void* s) {
  RuntimeLock rl;
  hipStream_t us = s ? (hipStream_t)s : util_stream();
  HIPCHECK(hipMemsetAsync(p, 0, bytes, us));
  HIPCHECK(hipStreamSynchronize(us));
}
static void dev_upload(void* dst, const void* src, size_t bytes, void* s) {
  RuntimeLock rl;
  hipStream_t us = s ? (hipStream_t)s : util_stream();
  HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, us));
  HIPCHECK(hipStreamSynchronize(us));
}
static void dev_free(void* p) {
  if (!p) return;
  RuntimeLock rl;
  (void)hipFree(p);
}
static void dev_alloc(void** p, size_t bytes) {
  RuntimeLock rl;
  HIPCHECK(hipMalloc(p, bytes));
}
// The launches `enqueue(cs)` puts on stream cs, as an executable graph.  Relaxed mode: cs is a non-blocking stream of the
// library's own and only kernel launches are recorded — nothing another thread does can belong to the capture.
template <class F>
static hipGraphExec_t capture_graph(void* cs, F&& enqueue) {
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

// ---- helpers ------------------------------------------------------------------------------------
namespace {
std::string split_layer_name(const std::string& layer, const std::string& blob, int idx) {
  return blob + "_" + layer + "_" + std::to_string(idx) + "_split";  // insert_splits.cpp:127-133
}
std::string split_blob_name(const std::string& layer, const std::string& blob, int idx, int k) {
  return split_layer_name(layer, blob, idx) + "_" + std::to_string(k);  // insert_splits.cpp:135-141
}

struct RawLayer {
  std::string name, type;
  std::vector<std::string> bottoms, tops;
  TextMsg def;
  bool is_split = false;
};

bool phase_included(const TextMsg& l, int phase) {
  // NetStateRule with a phase only (net.cpp:286-327).  Other rule kinds are not on this path.
  auto phase_of = [](const std::string& s) { return (s == "TEST" || s == "1") ? 1 : 0; };
  auto inc = l.subs("include"), exc = l.subs("exclude");
  if (!inc.empty()) {
    for (auto* r : inc)
      if (!r->has("phase") || phase_of(r->str("phase")) == phase) return true;
    return false;
  }
  for (auto* r : exc)
    if (r->has("phase") && phase_of(r->str("phase")) == phase) return false;
  return true;
}

// InsertSplits (src/caffe/util/insert_splits.cpp:12-101), loss weights ignored (none on this path)
std::vector<RawLayer> insert_splits(const std::vector<std::string>& inputs, const std::vector<RawLayer>& in) {
  typedef std::pair<int, int> P;
  std::map<std::string, P> last_top;
  std::map<P, P> bottom_src;
  std::map<P, int> top_count, split_idx;
  for (int i = 0; i < (int)inputs.size(); ++i) last_top[inputs[i]] = P(-1, i);
  for (int i = 0; i < (int)in.size(); ++i) {
    for (int j = 0; j < (int)in[i].bottoms.size(); ++j) {
      auto it = last_top.find(in[i].bottoms[j]);
      if (it == last_top.end())
        throw DcError(DC_EINVAL, "Unknown bottom blob '" + in[i].bottoms[j] + "' (layer '" + in[i].name +
                                     "', bottom index " + std::to_string(j) + ")");
      bottom_src[P(i, j)] = it->second;
      ++top_count[it->second];
    }
    for (int j = 0; j < (int)in[i].tops.size(); ++j) last_top[in[i].tops[j]] = P(i, j);
  }
  auto lname = [&](int i) { return i < 0 ? std::string("input") : in[i].name; };
  auto make_split = [&](const std::string& layer, const std::string& blob, int idx, int n) {
    RawLayer s;
    s.name = split_layer_name(layer, blob, idx);
    s.type = "Split";
    s.is_split = true;
    s.bottoms.push_back(blob);
    for (int k = 0; k < n; ++k) s.tops.push_back(split_blob_name(layer, blob, idx, k));
    return s;
  };
  std::vector<RawLayer> out;
  for (int i = 0; i < (int)inputs.size(); ++i)
    if (top_count[P(-1, i)] > 1) out.push_back(make_split("input", inputs[i], i, top_count[P(-1, i)]));
  for (int i = 0; i < (int)in.size(); ++i) {
    RawLayer l = in[i];
    for (int j = 0; j < (int)l.bottoms.size(); ++j) {
      P src = bottom_src[P(i, j)];
      if (top_count[src] > 1) l.bottoms[j] = split_blob_name(lname(src.first), l.bottoms[j], src.second, split_idx[src]++);
    }
    out.push_back(l);
    for (int j = 0; j < (int)l.tops.size(); ++j)
      if (top_count[P(i, j)] > 1) out.push_back(make_split(l.name, l.tops[j], j, top_count[P(i, j)]));
  }
  return out;
}

int pair_or(const TextMsg* m, const char* rep, const char* single, int idx, int def) {
  // ConvolutionParameter: repeated kernel_size/stride/pad/dilation or the _h/_w form
  // (base_conv_layer.cpp:23-99)
  if (!m) return def;
  if (m->has(single)) return (int)m->num(single, def);
  auto v = m->nums(rep);
  if (v.empty()) return def;
  return (int)(v.size() == 1 ? v[0] : v[std::min<size_t>(idx, v.size() - 1)]);
}
}  // namespace

// ---- Net: construction ----------------------------------------------------------------------------
DevVec::~DevVec() {
  dev_free(dev);
}

Net::~Net() {
  release_graph();
  for (auto& ps : parked_)
    if (ps->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)ps->graph_exec);
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
  dev_free(pose_dev);
  dev_free(scratch_dev_);
  dev_free(img_dev_);
  dev_free(tmp_dev_);
}

Net* Net::create(const std::string& text, int phase, const Net* clone_of) {
  std::unique_ptr<Net> n(new Net());
  n->phase = phase;
  n->proto_text = text;
  n->clone_src_ = clone_of;  // a clone adopts the source's parameter blobs layer by layer instead of allocating its own
  n->shared = clone_of ? clone_of->shared : std::make_shared<ModelShared>();
  TextMsg root = parse_text_proto(text);
  n->init_from(root);
  n->clone_src_ = nullptr;
  return n.release();
}

// A clone runs the same model concurrently with its parent (own activations, own stream, own graph) while
// sharing the parameter blobs and the packed filter images in HBM: this is how several independent forwards
// are kept in flight on one GPU without paying 263 MB per copy (deepcut_tools.Pipeline, bench.py).
// The parameter blobs, the packed images and the measured tile choices live in `shared` (ModelShared), owned jointly:
// either side may be destroyed first, and a parameter write through either side reaches both (weights_gen).
Net* Net::clone() {
  reshape();
  std::unique_ptr<Net> c(Net::create(proto_text, phase, this));
  if (c->layers.size() != layers.size()) throw DcError(DC_EINVAL, "clone: graph mismatch");
  for (size_t i = 0; i < inputs.size(); ++i) c->blobs[c->inputs[i]]->st->reshape(blobs[inputs[i]]->st->shape);
  c->fuse = fuse;
  c->use_graph = use_graph;
  if (dtype != c->dtype) {
    c->dtype = dtype;
    for (auto& st : c->storages)
      if (!st->is_param) st->esize = dtype == 1 ? 2 : 4;
  }
  c->device = device;
  c->reshape();
  return c.release();
}

// Device element type of activations and packed filters (host blobs stay float32 NCHW; accumulation and
// the epilogue stay float32).  Switching re-creates the device images and re-packs the filters.
void Net::set_dtype(int d) {
  if (d != 0 && d != 1) throw DcError(DC_EINVAL, "dtype must be 0 (float32) or 1 (float16)");
  if (d == dtype) return;
  dtype = d;
  for (auto& st : storages) {
    if (st->is_param) continue;
    if (st->head == HEAD_AT_GPU) sync_to_host(*st);  // keep what the user can still read
    if (st->head == SYNCED) st->head = HEAD_AT_CPU;
    if (st->dev) {
      dev_free(st->dev);
      st->dev = nullptr;
      st->dev_cap = 0;
    }
    st->esize = d == 1 ? 2 : 4;
  }
  invalidate_plans();  // packed images are keyed by dtype in the shared cache: no re-pack of the other type's images
}

void Net::synchronize() {
  if (stream) HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
}

int Net::layer_index(const std::string& nm) const {
  for (int i = 0; i < (int)layers.size(); ++i)
    if (layers[i].name == nm) return i;
  return -1;
}

void Net::init_from(const TextMsg& root) {
  name = root.str("name");
  // Deprecated V1 definitions (`layers { type: CONVOLUTION ... }`) are upgraded in place as the reference does on load
  // (UpgradeV1Net / UpgradeV1LayerParameter, upgrade_proto.cpp:647-850): the enum becomes the type string, the
  // train-only blobs_lr / weight_decay fields are dropped, every *_param message keeps its name.  V0 definitions
  // (a nested `layer { }` inside `layers`) are refused.
  std::vector<std::shared_ptr<TextMsg>> upgraded;
  if (!root.subs("layers").empty()) {
    if (!root.subs("layer").empty())
      throw DcError(DC_EINVAL, "prototxt mixes 'layer' and deprecated 'layers' entries");
    for (auto* l : root.subs("layers")) {
      if (l->sub("layer")) throw DcError(DC_EUNSUP, "V0 net definitions are not supported; upgrade with upgrade_net_proto_text");
      auto u = std::make_shared<TextMsg>();
      for (auto& f : l->fields) {
        if (f.key == "blobs_lr" || f.key == "weight_decay" || f.key == "blob_share_mode") continue;
        TextField g = f;
        if (f.key == "type" && !f.msg) {
          const char* nm = v1_layer_type_name(f.scalar);
          if (!*nm) throw DcError(DC_EUNSUP, "unknown V1 layer type '" + f.scalar + "'");
          g.scalar = nm;
          g.quoted = true;
        }
        u->fields.push_back(g);
      }
      upgraded.push_back(u);
    }
  }
  std::vector<std::string> in_names = root.strs("input");
  std::vector<std::vector<int>> in_shapes;
  {
    auto dims = root.nums("input_dim");
    auto shapes = root.subs("input_shape");
    if (!shapes.empty()) {
      for (auto* s : shapes) {
        std::vector<int> d;
        for (double v : s->nums("dim")) d.push_back((int)v);
        in_shapes.push_back(d);
      }
    } else {
      if (dims.size() != 4 * in_names.size())
        throw DcError(DC_EINVAL, "input_dim count must be 4 per input (net.cpp:84-90)");
      for (size_t i = 0; i < in_names.size(); ++i)
        in_shapes.push_back({(int)dims[4 * i], (int)dims[4 * i + 1], (int)dims[4 * i + 2], (int)dims[4 * i + 3]});
    }
    if (in_shapes.size() != in_names.size()) throw DcError(DC_EINVAL, "one input_shape per input required");
  }
  std::vector<RawLayer> raw;
  std::vector<const TextMsg*> layer_defs = root.subs("layer");
  for (auto& u : upgraded) layer_defs.push_back(u.get());
  for (auto* l : layer_defs) {
    if (!phase_included(*l, phase)) continue;
    RawLayer r;
    r.name = l->str("name");
    r.type = l->str("type");
    r.bottoms = l->strs("bottom");
    r.tops = l->strs("top");
    r.def = *l;
    raw.push_back(std::move(r));
  }
  std::vector<RawLayer> full = insert_splits(in_names, raw);

  std::set<std::string> available;
  auto new_blob = [&](const std::string& nm, std::shared_ptr<Storage> st) {
    auto b = std::make_shared<NetBlob>();
    b->name = nm;
    if (!st) {
      st = std::make_shared<Storage>();
      st->id = (int)storages.size();
      st->owner = this;
      storages.push_back(st);
    }
    b->st = st;
    blob_index[nm] = (int)blobs.size();
    blobs.push_back(b);
    return (int)blobs.size() - 1;
  };
  for (size_t i = 0; i < in_names.size(); ++i) {
    if (blob_index.count(in_names[i])) throw DcError(DC_EINVAL, "duplicate input '" + in_names[i] + "'");
    int bi = new_blob(in_names[i], nullptr);
    blobs[bi]->st->reshape(in_shapes[i]);
    inputs.push_back(bi);
    available.insert(in_names[i]);
  }
  for (auto& r : full) {
    LayerRec L;
    L.name = r.name;
    L.type = r.type;
    L.def = r.def;
    L.is_split = r.is_split;
    for (size_t j = 0; j < r.bottoms.size(); ++j) {  // Net::AppendBottom (net.cpp:440-467)
      auto it = blob_index.find(r.bottoms[j]);
      if (it == blob_index.end() || !available.count(r.bottoms[j]))
        throw DcError(DC_EINVAL, "Unknown bottom blob '" + r.bottoms[j] + "' (layer '" + r.name +
                                     "', bottom index " + std::to_string(j) + ")");
      L.bottoms.push_back(it->second);
      available.erase(r.bottoms[j]);
    }
    for (size_t j = 0; j < r.tops.size(); ++j) {  // Net::AppendTop (net.cpp:384-437)
      const std::string& tn = r.tops[j];
      if (j < r.bottoms.size() && tn == r.bottoms[j]) {
        L.tops.push_back(blob_index[tn]);  // in-place: same Blob
      } else if (blob_index.count(tn)) {
        throw DcError(DC_EINVAL, "Top blob '" + tn + "' produced by multiple sources.");
      } else if (r.is_split) {
        L.tops.push_back(new_blob(tn, blobs[L.bottoms[0]]->st));  // SplitLayer: ShareData (split_layer.cpp:26-31)
      } else {
        L.tops.push_back(new_blob(tn, nullptr));
      }
      available.insert(tn);
    }
    layers.push_back(std::move(L));
    setup_layer(layers.back());
    reshape_layer(layers.back());
  }
  for (auto& nm : available) outputs.push_back(blob_index[nm]);  // std::set order = alphabetical (net.cpp:268-273)
}

void Net::setup_layer(LayerRec& L) {
  const std::string& t = L.type;
  auto st_of = [&](int bi) -> Storage& { return *blobs[bi]->st; };
  const LayerRec* src = nullptr;  // clone: adopt the source net's parameter blobs of this layer
  if (clone_src_) {
    const size_t idx = (size_t)(&L - layers.data());
    if (idx >= clone_src_->layers.size() || clone_src_->layers[idx].name != L.name) throw DcError(DC_EINVAL, "clone: graph mismatch");
    src = &clone_src_->layers[idx];
  }
  auto add_param = [&](std::vector<int> shape, float fill) {
    if (src) {
      const size_t j = L.params.size();
      if (j >= src->params.size() || src->params[j]->st->shape != shape) throw DcError(DC_EINVAL, "clone: parameter mismatch");
      L.params.push_back(src->params[j]);
      return;
    }
    auto b = std::make_shared<NetBlob>();
    b->name = L.name;
    b->st = std::make_shared<Storage>();
    b->st->is_param = true;
    b->st->shared = shared;
    b->st->reshape(shape);
    float* p = b->st->host_ptr();
    size_t n = b->st->count();
    for (size_t i = 0; i < n; ++i) p[i] = fill;
    b->st->head = HEAD_AT_CPU;
    L.params.push_back(b);
  };
  auto need = [&](size_t nb, size_t nt) {
    if (L.bottoms.size() != nb || L.tops.size() != nt)
      throw DcError(DC_EINVAL, "layer '" + L.name + "' (" + t + ") needs " + std::to_string(nb) + " bottom(s) and " +
                                   std::to_string(nt) + " top(s)");
  };
  if (L.is_split) return;
  if (t == "Convolution" || t == "Deconvolution") {
    need(1, 1);
    const TextMsg* cp = L.def.sub("convolution_param");
    if (!cp) throw DcError(DC_EINVAL, "layer '" + L.name + "': convolution_param missing");
    ConvSpec& c = L.conv;
    c.num_output = (int)cp->num("num_output", 0);
    c.kh = pair_or(cp, "kernel_size", "kernel_h", 0, 0);
    c.kw = pair_or(cp, "kernel_size", "kernel_w", 1, 0);
    c.sh = pair_or(cp, "stride", "stride_h", 0, 1);
    c.sw = pair_or(cp, "stride", "stride_w", 1, 1);
    c.ph = pair_or(cp, "pad", "pad_h", 0, 0);
    c.pw = pair_or(cp, "pad", "pad_w", 1, 0);
    c.dh = pair_or(cp, "dilation", "", 0, 1);
    c.dw = pair_or(cp, "dilation", "", 1, 1);
    c.group = (int)cp->num("group", 1);
    c.bias = cp->boolean("bias_term", true);
    if (c.num_output <= 0 || c.kh <= 0 || c.kw <= 0 || c.sh <= 0 || c.sw <= 0 || c.dh <= 0 || c.dw <= 0)
      throw DcError(DC_EINVAL, "layer '" + L.name + "': bad convolution_param");
    if (c.group != 1) throw DcError(DC_EUNSUP, "layer '" + L.name + "': group != 1 is outside the DeeperCut path");
    int cin = st_of(L.bottoms[0]).dim(1);
    if (t == "Convolution") add_param({c.num_output, cin, c.kh, c.kw}, 0.f);
    else add_param({cin, c.num_output, c.kh, c.kw}, 0.f);  // reverse_dimensions (base_conv_layer.cpp:125-140)
    if (c.bias) add_param({c.num_output}, 0.f);
  } else if (t == "BatchNorm") {
    need(1, 1);
    const TextMsg* bp = L.def.sub("batch_norm_param");
    bool ugs = bp ? bp->boolean("use_global_stats", phase == DC_PHASE_TEST) : (phase == DC_PHASE_TEST);
    if (!ugs)
      throw DcError(DC_EUNSUP, "layer '" + L.name + "': BatchNorm with use_global_stats=false (batch statistics) is a "
                               "training mode outside the TEST-phase forward path");
    L.bn_eps = bp ? (float)bp->num("eps", 1e-5) : 1e-5f;
    int c = st_of(L.bottoms[0]).dim(1);
    add_param({c}, 0.f);
    add_param({c}, 0.f);
    add_param({1}, 0.f);
  } else if (t == "Scale") {
    const TextMsg* sp = L.def.sub("scale_param");
    if (L.bottoms.size() != 1 || L.tops.size() != 1)
      throw DcError(DC_EUNSUP, "layer '" + L.name + "': two-bottom Scale is outside the DeeperCut path");
    int axis = sp ? (int)sp->num("axis", 1) : 1, num_axes = sp ? (int)sp->num("num_axes", 1) : 1;
    if (axis != 1 || num_axes != 1)
      throw DcError(DC_EUNSUP, "layer '" + L.name + "': Scale only along the channel axis (axis 1, num_axes 1)");
    L.scale_bias = sp ? sp->boolean("bias_term", false) : false;
    float fill = 1.f;  // scale_layer.cpp:33-41: default filler is constant 1
    if (sp && sp->sub("filler")) fill = (float)sp->sub("filler")->num("value", 0.0);
    int c = st_of(L.bottoms[0]).dim(1);
    add_param({c}, fill);
    if (L.scale_bias) add_param({c}, 0.f);
  } else if (t == "ReLU") {
    need(1, 1);
    const TextMsg* rp = L.def.sub("relu_param");
    L.relu_slope = rp ? (float)rp->num("negative_slope", 0.0) : 0.f;
    if (L.relu_slope != 0.f) throw DcError(DC_EUNSUP, "layer '" + L.name + "': leaky ReLU is outside the DeeperCut path");
  } else if (t == "Sigmoid") {
    need(1, 1);
  } else if (t == "Pooling") {
    need(1, 1);
    const TextMsg* pp = L.def.sub("pooling_param");
    if (!pp) throw DcError(DC_EINVAL, "layer '" + L.name + "': pooling_param missing");
    std::string pool = pp->str("pool", "MAX");
    if (pool != "MAX" && pool != "0") throw DcError(DC_EUNSUP, "layer '" + L.name + "': only MAX pooling is on the path");
    if (pp->boolean("global_pooling", false)) throw DcError(DC_EUNSUP, "layer '" + L.name + "': global_pooling unsupported");
    L.pool_k = (int)pp->num("kernel_size", 0);
    L.pool_s = (int)pp->num("stride", 1);
    L.pool_p = (int)pp->num("pad", 0);
    if (pp->has("kernel_h") || pp->has("stride_h") || pp->has("pad_h"))
      throw DcError(DC_EUNSUP, "layer '" + L.name + "': rectangular pooling unsupported");
    if (L.pool_k <= 0 || L.pool_s <= 0 || L.pool_p >= L.pool_k) throw DcError(DC_EINVAL, "layer '" + L.name + "': bad pooling_param");
  } else if (t == "Eltwise") {
    if (L.bottoms.size() != 2 || L.tops.size() != 1)
      throw DcError(DC_EUNSUP, "layer '" + L.name + "': Eltwise needs exactly two bottoms on this path");
    const TextMsg* ep = L.def.sub("eltwise_param");
    if (ep) {
      std::string op = ep->str("operation", "SUM");
      if (op != "SUM" && op != "1") throw DcError(DC_EUNSUP, "layer '" + L.name + "': only Eltwise SUM is on the path");
      for (double c : ep->nums("coeff"))
        if (c != 1.0) throw DcError(DC_EUNSUP, "layer '" + L.name + "': Eltwise coeff != 1 unsupported");
    }
  } else if (t == "Crop") {
    need(2, 1);
    const TextMsg* cp = L.def.sub("crop_param");
    L.crop_oh = cp ? (int)cp->num("offset_height", 0) : 0;  // fork-specific CropParameter (caffe.proto:610-615)
    L.crop_ow = cp ? (int)cp->num("offset_width", 0) : 0;
  } else {
    throw DcError(DC_EUNSUP, "layer '" + L.name + "': type '" + t + "' is outside the DeeperCut forward path "
                             "(supported: Convolution, Deconvolution, BatchNorm, Scale, ReLU, Pooling, Eltwise, Crop, Sigmoid, Split)");
  }
}

void Net::reshape_layer(LayerRec& L) {
  auto st_of = [&](int bi) -> Storage& { return *blobs[bi]->st; };
  const std::string& t = L.type;
  if (L.is_split) return;  // shares the bottom's storage
  Storage& b0 = st_of(L.bottoms[0]);
  if (b0.shape.size() != 4) throw DcError(DC_ESHAPE, "layer '" + L.name + "': bottom must be 4-D");
  int N = b0.dim(0), C = b0.dim(1), H = b0.dim(2), W = b0.dim(3);
  Storage& top = st_of(L.tops[0]);
  if (t == "Convolution" || t == "Deconvolution") {
    const ConvSpec& c = L.conv;
    int cin_w = (t == "Convolution") ? L.params[0]->st->dim(1) : L.params[0]->st->dim(0);
    if (cin_w != C)
      throw DcError(DC_ESHAPE, "layer '" + L.name + "': input has " + std::to_string(C) + " channels, weights expect " +
                                   std::to_string(cin_w));
    int ekh = c.dh * (c.kh - 1) + 1, ekw = c.dw * (c.kw - 1) + 1;
    int OH, OW;
    if (t == "Convolution") {  // conv_layer.cpp:8-22
      OH = (H + 2 * c.ph - ekh) / c.sh + 1;
      OW = (W + 2 * c.pw - ekw) / c.sw + 1;
      if (H + 2 * c.ph < ekh || W + 2 * c.pw < ekw) OH = OW = 0;
    } else {  // deconv_layer.cpp:8-22
      OH = c.sh * (H - 1) + ekh - 2 * c.ph;
      OW = c.sw * (W - 1) + ekw - 2 * c.pw;
    }
    if (OH <= 0 || OW <= 0) throw DcError(DC_ESHAPE, "layer '" + L.name + "': input " + std::to_string(H) + "x" +
                                                         std::to_string(W) + " too small");
    b0.pad4 = true;
    top.reshape({N, c.num_output, OH, OW});
  } else if (t == "Pooling") {  // pooling_layer.cpp:79-123
    int k = L.pool_k, s = L.pool_s, p = L.pool_p;
    int OH = (int)std::ceil((float)(H + 2 * p - k) / s) + 1;
    int OW = (int)std::ceil((float)(W + 2 * p - k) / s) + 1;
    if (p) {
      if ((OH - 1) * s >= H + p) --OH;
      if ((OW - 1) * s >= W + p) --OW;
    }
    if (OH <= 0 || OW <= 0) throw DcError(DC_ESHAPE, "layer '" + L.name + "': input too small for pooling");
    top.reshape({N, C, OH, OW});
  } else if (t == "Eltwise") {
    Storage& b1 = st_of(L.bottoms[1]);
    if (b1.shape != b0.shape) {
      auto sh = [](const Storage& s) {
        std::string r;
        for (int d : s.shape) r += (r.empty() ? "" : "x") + std::to_string(d);
        return r;
      };
      throw DcError(DC_ESHAPE, "layer '" + L.name + "': Eltwise bottoms differ in shape (" + sh(b0) + " vs " + sh(b1) + ")");
    }
    top.reshape(b0.shape);
  } else if (t == "Crop") {  // crop_layer.cpp:25-34: strictly larger
    Storage& b1 = st_of(L.bottoms[1]);
    if (!(H - L.crop_oh > b1.dim(2)) || !(W - L.crop_ow > b1.dim(3)))
      throw DcError(DC_ESHAPE, "layer '" + L.name + "': invalid offset (Crop needs bottom[0] strictly larger than bottom[1])");
    top.reshape({N, C, b1.dim(2), b1.dim(3)});
  } else {  // BatchNorm, Scale, ReLU, Sigmoid
    if ((t == "BatchNorm" || t == "Scale") && L.params[0]->st->dim(0) != C)
      throw DcError(DC_ESHAPE, "layer '" + L.name + "': channel count changed");
    if (L.tops[0] != L.bottoms[0]) top.reshape(b0.shape);
  }
}

void Net::reshape() {
  for (auto& L : layers) reshape_layer(L);
}

// ---- weights --------------------------------------------------------------------------------------
void Net::copy_from(const std::string& path) {
  const bool h5 = is_hdf5_path(path);  // Net::CopyTrainedLayersFrom(string): ".h5" -> HDF5, else binaryproto (net.cpp:843-858)
  ModelFile m = h5 ? read_hdf5_weights(path) : read_caffemodel(path);
  for (auto& src : m.layers) {  // Net::CopyTrainedLayersFrom (net.cpp:805-840)
    int li = layer_index(src.name);
    if (li < 0) continue;  // "Ignoring source layer"
    LayerRec& L = layers[li];
    // binaryproto: the counts must agree (net.cpp:822-823); HDF5: the source may hold fewer only for shared
    // parameters (net.cpp:883-898), which this forward path does not have, so the same rule applies
    if (L.params.size() != src.blobs.size())
      throw DcError(DC_ESHAPE, "Incompatible number of blobs for layer " + src.name + ": net has " +
                                   std::to_string(L.params.size()) + ", file has " + std::to_string(src.blobs.size()));
    for (size_t j = 0; j < src.blobs.size(); ++j) {
      Storage& dst = *L.params[j]->st;
      const BlobData& sb = src.blobs[j];
      // Blob::ShapeEquals (blob.cpp:413-433): legacy 4-D shapes compare after left-padding with 1s
      std::vector<int> a = dst.shape, b = sb.shape;
      auto strip = [](std::vector<int> v) {
        while (v.size() > 1 && v.front() == 1) v.erase(v.begin());
        return v;
      };
      if (a != b && strip(a) != strip(b)) {
        auto sh = [](const std::vector<int>& s) {
          std::string r;
          for (int d : s) r += (r.empty() ? "" : " ") + std::to_string(d);
          return r;
        };
        throw DcError(DC_ESHAPE, "Cannot copy param " + std::to_string(j) + " weights from layer '" + src.name +
                                     "'; shape mismatch.  Source param shape is " + sh(b) + "; target param shape is " + sh(a));
      }
      if (sb.data.size() != dst.count())
        throw DcError(DC_ESHAPE, "layer '" + src.name + "' param " + std::to_string(j) + ": data length " +
                                     std::to_string(sb.data.size()) + " != " + std::to_string(dst.count()));
      std::memcpy(dst.host_ptr(), sb.data.data(), sb.data.size() * sizeof(float));
      dst.head = HEAD_AT_CPU;
    }
  }
  mark_weights_changed();
}

void Net::save(const std::string& path) {
  ModelFile m;
  m.name = name;
  for (auto& L : layers) {  // Net::ToProto writes every layer, with its blobs (net.cpp:910-925)
    LayerBlobs lb;
    lb.name = L.name;
    lb.type = L.type;
    for (int b : L.bottoms) lb.bottoms.push_back(blobs[b]->name);
    for (int t : L.tops) lb.tops.push_back(blobs[t]->name);
    for (auto& p : L.params) {
      BlobData bd;
      bd.shape = p->st->shape;
      bd.data.assign(p->st->host_ptr(), p->st->host_ptr() + p->st->count());
      lb.blobs.push_back(std::move(bd));
    }
    m.layers.push_back(std::move(lb));
  }
  write_caffemodel(path, m);
}

// ---- lowering ---------------------------------------------------------------------------------------
namespace {
struct LOp {
  enum Kind { CONV, DECONV, POOL, ELT, CROP } kind = CONV;
  std::vector<int> lids;
  int in = -1, in2 = -1, out = -1;
  int wl = -1;
  std::vector<int> wls;      // weight layers when several sibling layers are concatenated along Cout
  std::vector<double> a, b;  // folded per-channel affine (empty = identity)
  int sigmoid_ch = -1;       // >= 0: logistic on the first sigmoid_ch output channels only
  bool relu = false, sigmoid = false;
  int oh = 0, ow = 0;
  bool fused_crop = false;
  bool dead = false;
};

// cost model used to pick the tile variant (cycles; see DESIGN.md "Tile selection")
double variant_cost(const ConvGemmParams& p, int v) {
  const ConvVariant& cv = conv_variant(v);
  int bk = conv_variant_bk(v);
  int FM = cv.BM / cv.WR / 32, FN = cv.BN / cv.WC / 32;
  const double wps = cv.WR * cv.WC * cv.WK / 4.0;  // waves per SIMD of one workgroup
  double wgs = (double)conv_grid(p, v);
  // matrix-pipe cycles per k of one 32x32 fragment: 64/2 (v_mfma_f32_32x32x2_f32) or 32/16 (..._32x32x16_f16)
  const double cyc_per_k = p.esize == 2 ? 2.0 : 32.0;
  double mfma = (double)FM * FN * p.Ktot * cyc_per_k / cv.WK;
  double tiles = (double)p.Ktot / bk;
  // the matrix pipe serialises the MFMAs of co-resident waves; a second wave hides most per-tile overhead
  double per_wg = mfma * wps + tiles * (wps > 1 ? 60.0 : 220.0) + 2500.0;
  double rounds = std::ceil(wgs / 256.0);
  // the matrix pipe is shared by co-resident waves, so rounds serialise; partial last round costs a full one
  double t_mfma = rounds * per_wg;
  double bytes = wgs * (double)p.Ktot * (cv.BM + cv.BN) * (double)p.esize;
  double t_l2 = bytes / 4500.0;  // ~11 TB/s aggregate L2->LDS at 2.4 GHz
  return std::max(t_mfma, t_l2);
}
}  // namespace

// 64-bit content hash of a parameter blob (four independent multiply-xor lanes so that it runs at memory speed)
static uint64_t content_hash(const float* p, size_t n) {
  uint64_t h[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
  const uint32_t* u = reinterpret_cast<const uint32_t*>(p);
  size_t i = 0;
  for (; i + 4 <= n; i += 4)
    for (int k = 0; k < 4; ++k) h[k] = (h[k] ^ u[i + k]) * 0x100000001b3ull + (h[k] >> 29);
  for (; i < n; ++i) h[0] = (h[0] ^ u[i]) * 0x100000001b3ull + (h[0] >> 29);
  return (h[0] * 31 + h[1]) * 31 + (h[2] * 31 + h[3]) + n;
}

static int env_int(const char* k, int def) {
  const char* v = std::getenv(k);
  return v ? std::atoi(v) : def;
}

void Net::build_plan() {
  const int nL = (int)layers.size();
  auto sid = [&](int bi) { return blobs[bi]->st->id; };
  std::vector<LOp> ops;
  std::vector<char> absorbed(nL, 0);

  auto inplace_on = [&](int j, int storage) {
    const LayerRec& L = layers[j];
    return !L.is_split && L.bottoms.size() == 1 && L.tops.size() == 1 && L.bottoms[0] == L.tops[0] &&
           sid(L.tops[0]) == storage;
  };
  auto ensure_affine = [&](LOp& op, int C) {
    if (op.a.empty()) {
      op.a.assign(C, 1.0);
      op.b.assign(C, 0.0);
    }
  };
  auto fold_bn = [&](LOp& op, const LayerRec& L) {  // batch_norm_layer.cpp:86-93,138-149
    int C = L.params[0]->st->dim(0);
    ensure_affine(op, C);
    const float* mean = L.params[0]->st->host_ptr();
    const float* var = L.params[1]->st->host_ptr();
    float sfv = L.params[2]->st->host_ptr()[0];
    double sf = sfv == 0.f ? 0.0 : 1.0 / (double)sfv;
    for (int c = 0; c < C; ++c) {
      double s = 1.0 / std::sqrt((double)var[c] * sf + (double)L.bn_eps);
      op.a[c] = op.a[c] * s;
      op.b[c] = (op.b[c] - (double)mean[c] * sf) * s;
    }
  };
  auto fold_scale = [&](LOp& op, const LayerRec& L) {  // scale_layer.cpp:109-134, bias_layer.cpp:72-87
    int C = L.params[0]->st->dim(0);
    ensure_affine(op, C);
    const float* g = L.params[0]->st->host_ptr();
    const float* be = L.scale_bias ? L.params[1]->st->host_ptr() : nullptr;
    for (int c = 0; c < C; ++c) {
      op.a[c] = op.a[c] * (double)g[c];
      op.b[c] = op.b[c] * (double)g[c] + (be ? (double)be[c] : 0.0);
    }
  };
  // absorb the in-place BatchNorm / Scale / ReLU / Sigmoid layers that directly follow layer i on `op.out`
  auto absorb_chain = [&](LOp& op, int i, bool allow_affine) {
    int j = i + 1;
    while (j < nL && !op.relu && !op.sigmoid && inplace_on(j, op.out)) {
      const LayerRec& L = layers[j];
      if (L.type == "BatchNorm" && allow_affine) fold_bn(op, L);
      else if (L.type == "Scale" && allow_affine) fold_scale(op, L);
      else if (L.type == "ReLU") op.relu = true;
      else if (L.type == "Sigmoid") op.sigmoid = true;
      else break;
      absorbed[j] = 1;
      op.lids.push_back(j);
      ++j;
    }
  };

  // pass 1: one op per layer group
  for (int i = 0; i < nL; ++i) {
    if (absorbed[i] || layers[i].is_split) continue;
    const LayerRec& L = layers[i];
    LOp op;
    op.lids.push_back(i);
    op.in = sid(L.bottoms[0]);
    op.out = sid(L.tops[0]);
    if (L.type == "Convolution" || L.type == "Deconvolution") {
      op.kind = L.type == "Convolution" ? LOp::CONV : LOp::DECONV;
      op.wl = i;
      if (L.conv.bias) {
        const float* bias = L.params[1]->st->host_ptr();
        op.a.assign(L.conv.num_output, 1.0);
        op.b.assign(bias, bias + L.conv.num_output);
      }
      if (op.in != op.out) absorb_chain(op, i, true);
    } else if (L.type == "Pooling") {
      op.kind = LOp::POOL;
    } else if (L.type == "Eltwise") {
      op.kind = LOp::ELT;
      op.in2 = sid(L.bottoms[1]);
      absorb_chain(op, i, false);
    } else if (L.type == "Crop") {
      op.kind = LOp::CROP;
      op.oh = L.crop_oh;
      op.ow = L.crop_ow;
    } else {  // stand-alone BatchNorm / Scale / ReLU / Sigmoid
      op.kind = LOp::ELT;
      if (L.type == "BatchNorm") fold_bn(op, L);
      else if (L.type == "Scale") fold_scale(op, L);
      else if (L.type == "ReLU") op.relu = true;
      else if (L.type == "Sigmoid") op.sigmoid = true;
      if (!op.relu && !op.sigmoid) absorb_chain(op, i, true);
    }
    ops.push_back(std::move(op));
  }

  // pass 2: residual-add and deconvolution-head fusion
  if (fuse >= 1) {
    const int nS = (int)storages.size();
    auto analyse = [&](std::vector<int>& prod, std::vector<std::vector<int>>& cons) {
      prod.assign(nS, -1);
      cons.assign(nS, {});
      for (int k = 0; k < (int)ops.size(); ++k) {
        if (ops[k].dead) continue;
        cons[ops[k].in].push_back(k);
        if (ops[k].in2 >= 0) cons[ops[k].in2].push_back(k);
        prod[ops[k].out] = k;
      }
    };
    std::vector<int> prod;
    std::vector<std::vector<int>> cons;
    for (int e = 0; e < (int)ops.size(); ++e) {
      LOp& E = ops[e];
      if (E.dead || E.kind != LOp::ELT || E.in2 < 0 || !E.a.empty() || E.in == E.out || E.in2 == E.out) continue;
      analyse(prod, cons);
      int cand[2][2] = {{E.in2, E.in}, {E.in, E.in2}};
      if (prod[E.in] > prod[E.in2]) std::swap(cand[0], cand[1]);
      for (auto& c : cand) {
        int X = c[0], other = c[1];
        int pk = prod[X];
        if (pk < 0 || cons[X].size() != 1) continue;
        LOp& P = ops[pk];
        if (P.kind == LOp::CONV && !P.relu && !P.sigmoid && P.in2 < 0 && prod[other] < pk && P.in != P.out) {
          P.in2 = other;
          P.out = E.out;
          P.relu = E.relu;
          P.sigmoid = E.sigmoid;
          P.lids.insert(P.lids.end(), E.lids.begin(), E.lids.end());
          E.dead = true;
          break;
        }
        if (P.kind == LOp::CROP) {
          int dk = prod[P.in];
          if (dk < 0 || cons[P.in].size() != 1) continue;
          LOp& D = ops[dk];
          if (D.kind != LOp::DECONV || D.relu || D.sigmoid || D.in2 >= 0) continue;
          LOp F = D;
          F.in2 = other;
          F.out = E.out;
          F.oh = P.oh;
          F.ow = P.ow;
          F.fused_crop = true;
          F.relu = E.relu;
          F.sigmoid = E.sigmoid;
          F.lids.insert(F.lids.end(), P.lids.begin(), P.lids.end());
          F.lids.insert(F.lids.end(), E.lids.begin(), E.lids.end());
          D.dead = true;
          P.dead = true;
          ops[e] = F;  // executes at the Eltwise's position: both operands are ready there
          break;
        }
      }
    }
  }


  // pass 3 (DC_OPT_FUSE >= 2): sibling heads.  The three DeeperCut heads (part scores, location refinement,
  // pairwise regression) are the same Deconvolution on res5c + the same 1x1 skip convolution on res3's
  // last block, differing only in Cout (14 / 28 / 364): run them as ONE 406-channel skip GEMM and ONE
  // 406-channel deconvolution (res5c's 2048-deep rows are read once instead of three times, and the
  // 14/28-channel GEMMs no longer pad to 32-wide MFMA tiles).  The named output blobs become channel
  // views of the concatenated tensor; the Sigmoid of the score head moves into the epilogue.
  for (auto& st : storages) {
    st->view_of = -1;
    st->view_c0 = 0;
    st->view_cp = 0;
  }
  plan_views_.clear();
  if (fuse >= 2) {
    const int nS0 = (int)storages.size();
    std::vector<int> prod(nS0, -1);
    std::vector<std::vector<int>> cons(nS0);
    for (int k = 0; k < (int)ops.size(); ++k) {
      if (ops[k].dead) continue;
      cons[ops[k].in].push_back(k);
      if (ops[k].in2 >= 0) cons[ops[k].in2].push_back(k);
      prod[ops[k].out] = k;
    }
    auto same_geom = [&](int la, int lb) {
      const ConvSpec &x = layers[la].conv, &y = layers[lb].conv;
      return x.kh == y.kh && x.kw == y.kw && x.sh == y.sh && x.sw == y.sw && x.ph == y.ph && x.pw == y.pw && x.dh == y.dh &&
             x.dw == y.dw;
    };
    std::vector<char> used(ops.size(), 0);
    for (int f0 = 0; f0 < (int)ops.size(); ++f0) {
      if (used[f0] || ops[f0].dead || ops[f0].kind != LOp::DECONV || !ops[f0].fused_crop || ops[f0].in2 < 0 || ops[f0].relu ||
          ops[f0].sigmoid || !ops[f0].wls.empty())
        continue;
      std::vector<int> grp;  // deconv ops
      for (int f = f0; f < (int)ops.size(); ++f) {
        const LOp& F = ops[f];
        if (used[f] || F.dead || F.kind != LOp::DECONV || !F.fused_crop || F.in2 < 0 || F.relu || F.sigmoid || !F.wls.empty()) continue;
        if (F.in != ops[f0].in || F.oh != ops[f0].oh || F.ow != ops[f0].ow || !same_geom(F.wl, ops[f0].wl)) continue;
        const int ck = prod[F.in2];
        if (ck < 0 || cons[F.in2].size() != 1) continue;
        const LOp& Cq = ops[ck];
        const int c0k = prod[ops[f0].in2];
        if (Cq.kind != LOp::CONV || Cq.relu || Cq.sigmoid || Cq.in2 >= 0 || !Cq.wls.empty() || c0k < 0 || Cq.in != ops[c0k].in ||
            !same_geom(Cq.wl, ops[c0k].wl))
          continue;
        if (storages[F.out]->shape[2] != storages[ops[f0].out]->shape[2] || storages[F.out]->shape[3] != storages[ops[f0].out]->shape[3])
          continue;
        grp.push_back(f);
      }
      if (grp.size() < 2) continue;
      // sigmoid folding: a head whose only consumer is an out-of-place Sigmoid
      struct Member {
        int f, c, sig_elt, final_out;
      };
      std::vector<Member> mem;
      for (int f : grp) {
        Member m{f, prod[ops[f].in2], -1, ops[f].out};
        const auto& cs = cons[ops[f].out];
        if (cs.size() == 1) {
          const LOp& E = ops[cs[0]];
          if (E.kind == LOp::ELT && E.in2 < 0 && E.a.empty() && E.sigmoid && !E.relu && E.in != E.out) {
            m.sig_elt = cs[0];
            m.final_out = E.out;
          }
        }
        if (m.sig_elt < 0 && !cs.empty()) {  // some other kernel reads this head: it cannot become a strided view
          m.f = -1;
        }
        mem.push_back(m);
      }
      mem.erase(std::remove_if(mem.begin(), mem.end(), [](const Member& m) { return m.f < 0; }), mem.end());
      if (mem.size() < 2) continue;
      std::stable_sort(mem.begin(), mem.end(), [](const Member& x, const Member& y) { return (x.sig_elt >= 0) > (y.sig_elt >= 0); });
      int ctot = 0, sig_ch = 0;
      for (auto& m : mem) {
        const int c = layers[ops[m.f].wl].conv.num_output;
        if (m.sig_elt >= 0) sig_ch += c;
        ctot += c;
      }
      auto aux = [&](const std::string& key, std::vector<int> shape) {
        auto it = aux_index_.find(key);
        int id;
        if (it == aux_index_.end()) {
          auto st = std::make_shared<Storage>();
          st->id = (int)storages.size();
          st->owner = this;
          st->esize = dtype == 1 ? 2 : 4;
          storages.push_back(st);
          id = st->id;
          aux_index_[key] = id;
        } else {
          id = it->second;
        }
        storages[id]->reshape(shape);
        return id;
      };
      const Storage& o0 = *storages[ops[mem[0].f].out];
      const std::string gkey = std::to_string(ops[mem[0].f].lids.front());
      const int T1 = aux("heads_skip:" + gkey, {o0.dim(0), ctot, o0.dim(2), o0.dim(3)});
      const int T2 = aux("heads_out:" + gkey, {o0.dim(0), ctot, o0.dim(2), o0.dim(3)});
      LOp MC = ops[mem[0].c], MF = ops[mem[0].f];
      MC.wls.clear();
      MF.wls.clear();
      MC.lids.clear();
      MF.lids.clear();
      MC.a.clear();
      MC.b.clear();
      MF.a.clear();
      MF.b.clear();
      int c0 = 0;
      for (auto& m : mem) {
        const LOp &Cm = ops[m.c], &Fm = ops[m.f];
        const int c = layers[Fm.wl].conv.num_output;
        MC.wls.push_back(Cm.wl);
        MF.wls.push_back(Fm.wl);
        MC.lids.insert(MC.lids.end(), Cm.lids.begin(), Cm.lids.end());
        MF.lids.insert(MF.lids.end(), Fm.lids.begin(), Fm.lids.end());
        if (m.sig_elt >= 0) MF.lids.insert(MF.lids.end(), ops[m.sig_elt].lids.begin(), ops[m.sig_elt].lids.end());
        for (int k = 0; k < c; ++k) {
          MC.a.push_back(Cm.a.empty() ? 1.0 : Cm.a[k]);
          MC.b.push_back(Cm.b.empty() ? 0.0 : Cm.b[k]);
          MF.a.push_back(Fm.a.empty() ? 1.0 : Fm.a[k]);
          MF.b.push_back(Fm.b.empty() ? 0.0 : Fm.b[k]);
        }
        Storage& v = *storages[m.final_out];
        v.view_of = T2;
        v.view_c0 = c0;
        plan_views_.push_back(m.final_out);
        c0 += c;
      }
      MC.out = T1;
      MF.in2 = T1;
      MF.out = T2;
      MF.sigmoid_ch = sig_ch;
      // the merged ops execute where the LAST member deconvolution stood (all operands are ready there)
      int last_f = 0;
      for (auto& m : mem) {
        last_f = std::max(last_f, m.f);
        ops[m.c].dead = true;
        ops[m.f].dead = true;
        used[m.f] = 1;
        if (m.sig_elt >= 0) ops[m.sig_elt].dead = true;
      }
      for (auto& m : mem)
        if (m.sig_elt >= 0 && m.sig_elt < last_f) { /* sigmoid stood before the last head: fine, it is folded */ }
      ops[last_f] = MF;
      ops[last_f].dead = false;
      used[last_f] = 1;
      ops.insert(ops.begin() + last_f, MC);  // skip GEMM right before it
      used.insert(used.begin() + last_f, 1);
      break;  // one head group per net is all the path has; indices moved, stop scanning
    }
  }

  // tensors combined element-wise / pooled / cropped must agree on channel pitch: propagate before any
  // launch parameters are derived from cp()
  for (bool changed = true; changed;) {
    changed = false;
    for (auto& op : ops) {
      if (op.dead || op.kind == LOp::CONV || op.kind == LOp::DECONV) continue;
      bool p4 = storages[op.in]->pad4 || storages[op.out]->pad4 || (op.in2 >= 0 && storages[op.in2]->pad4);
      if (!p4) continue;
      for (int sx : {op.in, op.in2, op.out})
        if (sx >= 0 && !storages[sx]->pad4) storages[sx]->pad4 = true, changed = true;
    }
  }
  {
    std::vector<char> live(storages.size(), 0);
    for (int bi : inputs) live[blobs[bi]->st->id] = 1;
    for (auto& op : ops)
      if (!op.dead) live[op.out] = 1;
    for (auto& st : storages) st->elided = !live[st->id] && st->view_of < 0;
  }

  // finalize: launches
  plan.clear();
  plan_flops = 0;
  ++stats.lowerings;
  // the packed-image cache is shared with the clones: the first executor to lower after a parameter change empties it
  // (images still referenced by another executor's plans stay alive until that executor re-lowers too)
  std::lock_guard<std::mutex> pack_lock(shared->mu);
  if (shared->packed_gen != shared->weights_gen) {
    shared->vec_by_key.clear();
    shared->packed_gen = shared->weights_gen;
    for (auto& L : layers)
      for (auto& pb : L.params) pb->st->packed_hash = content_hash(pb->st->host_ptr(), pb->st->count());
    ++stats.repacks;
  }
  auto get_vec = [&](const std::string& key, const std::function<void(std::vector<float>&)>& fill) {
    auto it = shared->vec_by_key.find(key);
    if (it != shared->vec_by_key.end()) return it->second;
    auto v = std::make_shared<DevVec>();
    fill(v->host);
    shared->vec_by_key[key] = v;
    return v;
  };
  auto label_of = [&](const LOp& op) {
    std::string s;
    for (size_t k = 0; k < op.lids.size(); ++k) {
      if (k) s += "+";
      s += layers[op.lids[k]].name;
    }
    return s;
  };
  const int force_variant = env_int("DC_CONV_VARIANT", -1);
  const int es = dtype == 1 ? 2 : 4;          // bytes per activation / filter element
  const int kmin = dtype == 1 ? 64 : 32;      // smallest K tile of the dtype's variants (one 128-byte line)
  const std::string dkey = dtype == 1 ? "h:" : "";

  auto affine_vecs = [&](const LOp& op, Launch& l, int C) {
    if (op.a.empty()) return;
    std::string key = std::to_string(op.lids.front()) + ":" + std::to_string(op.lids.size()) + ":" + std::to_string(op.wls.size());
    l.scale = get_vec("a:" + key, [&](std::vector<float>& h) {
      h.resize(C);
      for (int c = 0; c < C; ++c) h[c] = (float)op.a[c];
    });
    l.shift = get_vec("b:" + key, [&](std::vector<float>& h) {
      h.resize(C);
      for (int c = 0; c < C; ++c) h[c] = (float)op.b[c];
    });
  };
  // float16 filter images: per-output-channel power-of-two pre-scaling (exact in fp32; undone by the epilogue's fp32 scale), see
  // DevVec::row_scale.  Called right after the image of a launch is made / found; replaces the launch's scale vector by
  // a[c] * 2^-k(c) (keyed by the image, since k depends on the image's rows).
  static const bool half_rowscale = env_int("DC_HALF_ROWSCALE", 1) != 0;
  auto half_row_scale = [&](Launch& l, const LOp& op, int OC) {
    if (dtype != 1 || !half_rowscale || !l.w) return;
    DevVec& Wv = *l.w;
    if (Wv.row_scale.empty()) {
      if (Wv.host.empty()) return;  // an image uploaded before this feature existed in the process: leave it
      struct Seg { size_t off; int K; };
      std::vector<Seg> segs;
      if (l.cg.ncls > 1)
        for (int q = 0; q < l.cg.ncls; ++q) segs.push_back({(size_t)l.cg.cls[q].w_off, l.cg.cls[q].Ktot});
      else
        segs.push_back({0, l.cg.Ktot});
      for (const Seg& sg : segs)
        if (sg.off + (size_t)OC * sg.K > Wv.host.size())
          throw DcError(DC_EINVAL, "launch '" + l.label + "': filter image of " + std::to_string(Wv.host.size()) + " elements is smaller than " +
                                       std::to_string(OC) + " rows of " + std::to_string(sg.K));
      Wv.row_scale.assign(OC, 1.f);
      for (int c = 0; c < OC; ++c) {
        float mx = 0.f;
        for (const Seg& sg : segs) {
          const float* r = Wv.host.data() + sg.off + (size_t)c * sg.K;
          for (int k = 0; k < sg.K; ++k) mx = std::max(mx, std::fabs(r[k]));
        }
        if (!(mx > 0.f) || !std::isfinite(mx)) continue;
        int k = 13 - std::ilogb(mx);
        k = std::max(-60, std::min(60, k));
        if (k == 0) continue;
        const float f = std::ldexp(1.f, k);
        for (const Seg& sg : segs) {
          float* r = Wv.host.data() + sg.off + (size_t)c * sg.K;
          for (int q = 0; q < sg.K; ++q) r[q] *= f;
        }
        Wv.row_scale[c] = std::ldexp(1.f, -k);
      }
    }
    std::shared_ptr<DevVec> rs = l.w;  // keeps row_scale alive inside the fill
    char wkey[40];  // the image's identity: its address (images and these vectors live and die together in vec_by_key)
    std::snprintf(wkey, sizeof wkey, "%p", (void*)l.w.get());
    l.scale = get_vec(std::string("ha:") + wkey + ":" + std::to_string(op.lids.front()) + ":" + std::to_string(op.lids.size()), [&](std::vector<float>& h) {
      h.resize(OC);
      for (int c = 0; c < OC; ++c) h[c] = (float)((op.a.empty() ? 1.0 : op.a[c]) * (double)rs->row_scale[c]);
    });
  };
  const int wino_mode = env_int("DC_WINOGRAD", -1);  // -1: where measured faster (autotune); 0: never; 1: wherever eligible
  auto choose_variant = [&](Launch& l, int kgcd) {
    int best = -1;
    double bc = 0;
    const bool mc = l.cg.ncls > 1;  // multi-class launches need a tile with a multi-class instantiation
    for (int v = 0; v < conv_num_variants(); ++v) {
      if (kgcd % conv_variant_bk(v) != 0 || conv_variant_esize(v) != es || (mc && !conv_variant_multiclass(v))) continue;
      if (force_variant >= 0 && v != force_variant) continue;
      double c = variant_cost(l.cg, v);
      if (best < 0 || c < bc) best = v, bc = c;
    }
    if (best < 0)
      for (int v = 0; v < conv_num_variants(); ++v) {
        if (kgcd % conv_variant_bk(v) != 0 || conv_variant_esize(v) != es || (mc && !conv_variant_multiclass(v))) continue;
        double c = variant_cost(l.cg, v);
        if (best < 0 || c < bc) best = v, bc = c;
      }
    if (best < 0) throw DcError(DC_EUNSUP, "launch '" + l.label + "': no tile variant takes K segments of " + std::to_string(kgcd) + " elements");
    l.variant = best;
    l.kernel = std::string("conv_gemm<") + conv_variant(best).name + ">";
    l.grid = conv_grid(l.cg, best);
  };
  auto use_wino = [&](Launch& l) {
    l.variant = kWinoVariant;
    l.kernel = "wino_f23<4x8x16>";
    l.grid = wino_grid(l.cg);
  };

  // Channel split of a wide-but-ragged GEMM (the merged heads: N = 406 = 3 x 128 + 22).  On 128-wide tiles a quarter of the
  // fourth column block is padding (26 % of the launch's MFMA work and filter fetches for nothing); as two launches — the
  // first floor(N / 128) * 128 channels, then the tail on a narrow tile — the padding is 22 -> 32/64 channels.  Host-side only:
  // the second launch is the same kernel on offset filter rows / epilogue constants / output channels.
  // DC_HEAD_SPLIT=1 switches it on.  OFF by default — measured (round 4, EXPERIMENTS.md): the tail launch re-reads all of res5c's
  // 2048-deep rows for 22 channels (float16 grouped pyramid: 900 us -> 720 + 177 us; float32 batch 1: 457 -> 459 images/s in
  // flight, 337 -> 333 alone): the padding it removes is paid back as operand traffic.  Kept as a switch with its parity test.
  const int head_split = env_int("DC_HEAD_SPLIT", 0);
  auto push_split = [&](Launch&& l, int kgcd) {
    const int OC = l.cg.Cout;
    const bool want = head_split == 1;
    const int c0 = OC / 128 * 128;
    if (!want || OC < 256 || c0 == OC || OC - c0 > 64 || force_variant >= 0) {
      plan.push_back(std::move(l));
      return;
    }
    Launch a = l, b = l;
    const double fa = (double)c0 / OC;
    a.cg.Cout = c0;
    a.flops = l.flops * fa;
    a.label += " [ch 0-" + std::to_string(c0 - 1) + "]";
    a.cg.sigmoid_ch = std::min(l.cg.sigmoid_ch, c0);
    b.cg.Cout = OC - c0;
    b.flops = l.flops * (1.0 - fa);
    b.label += " [ch " + std::to_string(c0) + "-" + std::to_string(OC - 1) + "]";
    b.cg.sigmoid_ch = std::max(0, l.cg.sigmoid_ch - c0);
    b.y_off = l.y_off + c0;
    b.c_off = l.c_off + c0;
    if (l.cg.ncls > 1) {
      for (int q = 0; q < l.cg.ncls; ++q) b.cg.cls[q].w_off = l.cg.cls[q].w_off + (long)c0 * l.cg.cls[q].Ktot;
    } else {
      b.w_off = l.w_off + (long)c0 * l.cg.Ktot;
    }
    choose_variant(a, kgcd);
    choose_variant(b, kgcd);
    plan.push_back(std::move(a));
    plan.push_back(std::move(b));
  };

  for (auto& op : ops) {
    if (op.dead) continue;
    Launch base;
    base.label = label_of(op);
    base.first_layer = *std::min_element(op.lids.begin(), op.lids.end());
    base.last_layer = *std::max_element(op.lids.begin(), op.lids.end());
    base.in = op.in;
    base.in2 = op.in2;
    base.out = op.out;
    base.relu = op.relu;
    base.sigmoid = op.sigmoid;
    Storage& X = *storages[op.in];
    Storage& Y = *storages[op.out];
    const int N = X.dim(0), C = X.dim(1), H = X.dim(2), W = X.dim(3);
    const int CP = X.cp();
    const int OC = Y.dim(1), OHt = Y.dim(2), OWt = Y.dim(3), OCP = Y.cp();
    if (op.kind == LOp::CONV) {
      const LayerRec& L = layers[op.wl];
      const ConvSpec& c = L.conv;
      Launch l = base;
      l.kind = Launch::CONV;
      ConvGemmParams& g = l.cg;
      g.esize = es;
      g.x_img_stride = (long)H * W * CP;
      g.x_row_stride = W * CP;
      g.x_rows = H;
      g.x_rowlen = W * CP;
      g.sy = c.sh;
      g.sx = c.sw * CP;
      int kgcd;
      const bool rowtap = (CP % kmin) != 0;
      if (rowtap) {
        // small-channel input (the 3->4 channel stem): one tap per kernel ROW, the kw adjacent pixels of
        // that row being contiguous in NHWC; K per tap = kw*CP rounded up to 32 with zero weights
        if (c.dw != 1 || CP % (16 / es) != 0)
          throw DcError(DC_EUNSUP, "layer '" + L.name + "': convolution over " + std::to_string(C) +
                                       " channels needs dilation_w 1 (row-tap path) or a multiple of " + std::to_string(kmin) + " channels");
        int klen = (c.kw * CP + kmin - 1) / kmin * kmin;
        if (c.kh > kMaxTaps) throw DcError(DC_EUNSUP, "layer '" + L.name + "': kernel too tall");
        g.nty = c.kh;
        g.ntx = 1;
        g.dy0 = -c.ph;
        g.ddy = c.dh;
        g.x0 = -c.pw * CP;
        g.ddx = 0;
        g.klen = klen;
        g.Ktot = c.kh * klen;
        kgcd = klen;
        // sibling layers merged into one launch (the skip convolutions of the heads) are concatenated along Cout here too:
        // round 2 packed only the first member in this path — a float16 net whose skip level has fewer than 64 channels
        // ran its second and third head on rows beyond the image (found by the row scaling's bounds check in round 3)
        const std::vector<int> members = op.wls.empty() ? std::vector<int>{op.wl} : op.wls;
        l.w = get_vec(dkey + "w:" + std::to_string(members.front()) + "x" + std::to_string(members.size()) + "r", [&](std::vector<float>& h) {
          h.assign((size_t)OC * g.Ktot, 0.f);
          int cbase = 0;
          for (int ml : members) {
            const float* w = layers[ml].params[0]->st->host_ptr();  // [Cout][Cin][kh][kw]
            const int cm = layers[ml].conv.num_output;
            for (int co = 0; co < cm; ++co)
              for (int ci = 0; ci < C; ++ci)
                for (int ky = 0; ky < c.kh; ++ky)
                  for (int kx = 0; kx < c.kw; ++kx)
                    h[(size_t)(cbase + co) * g.Ktot + ky * klen + kx * CP + ci] = w[(((size_t)co * C + ci) * c.kh + ky) * c.kw + kx];
            cbase += cm;
          }
        });
      } else {
        if (c.kh * c.kw > kMaxTaps)
          throw DcError(DC_EUNSUP, "layer '" + L.name + "': more than " + std::to_string(kMaxTaps) + " kernel taps");
        g.nty = c.kh;
        g.ntx = c.kw;
        g.dy0 = -c.ph;
        g.ddy = c.dh;
        g.x0 = -c.pw * CP;
        g.ddx = c.dw * CP;
        g.klen = CP;
        g.Ktot = c.kh * c.kw * CP;
        kgcd = CP;
        const std::vector<int> members = op.wls.empty() ? std::vector<int>{op.wl} : op.wls;
        l.w = get_vec(dkey + "w:" + std::to_string(members.front()) + "x" + std::to_string(members.size()), [&](std::vector<float>& h) {
          h.assign((size_t)OC * g.Ktot, 0.f);
          const int taps = c.kh * c.kw;
          int cbase = 0;
          for (int ml : members) {  // sibling layers concatenated along Cout
            const float* w = layers[ml].params[0]->st->host_ptr();
            const int cm = layers[ml].conv.num_output;
            for (int co = 0; co < cm; ++co)
              for (int ci = 0; ci < C; ++ci) {
                const float* src = w + ((size_t)co * C + ci) * taps;
                float* dst = h.data() + (size_t)(cbase + co) * g.Ktot + ci;
                for (int tp = 0; tp < taps; ++tp) dst[(size_t)tp * CP] = src[tp];
              }
            cbase += cm;
          }
        });
      }
      g.NB = N;
      g.OH = OHt;
      g.OW = OWt;
      g.M = N * OHt * OWt;
      g.Cout = OC;
      g.y_img_stride = (long)OHt * OWt * OCP;
      g.y_row_stride = OWt * OCP;
      g.y_pix_stride = OCP;
      g.relu = op.relu;
      g.sigmoid_ch = op.sigmoid ? OC : 0;
      affine_vecs(op, l, OC);
      l.flops = 2.0 * g.M * (double)OC * C * c.kh * c.kw;
      plan_flops += l.flops;
      l.w->as_half = dtype == 1;
          half_row_scale(l, op, OC);
      choose_variant(l, kgcd);
      // stride-1 3x3 layers can also run as Winograd F(2x2,3x3): keep the transformed filters next to the direct ones
      // and let the per-shape timing decide (kernels.hip, wino_f23_kernel)
      if (!rowtap && wino_mode != 0 && op.wls.empty() && wino_eligible(g)) {
        l.wino_w = get_vec(dkey + "wino:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          h.assign(wino_packed_floats(c.num_output, C), 0.f);
          wino_pack_filters(L.params[0]->st->host_ptr(), c.num_output, C, h.data());
        });
        if (wino_mode == 1 && (force_variant < 0 || force_variant == kWinoVariant)) use_wino(l);
      }
      if (l.wino_w || rowtap) plan.push_back(std::move(l));
      else push_split(std::move(l), kgcd);
    } else if (op.kind == LOp::DECONV) {
      // stride-s transposed convolution = s*s ordinary gather-GEMMs, one per output residue class
      // (Y mod s, X mod s): output pixel (s*i + r) receives tap k iff (r + p - k*d) % s == 0, from input
      // row i + (r + p - k*d)/s  (col2im_cpu, im2col.cpp:163-197, inverted: output-stationary).
      const LayerRec& L = layers[op.wl];
      const ConvSpec& c = L.conv;
      if (CP % kmin != 0)
        throw DcError(DC_EUNSUP, "layer '" + L.name + "': deconvolution input channels must be a multiple of " + std::to_string(kmin));
      const int DH = c.sh * (H - 1) + c.dh * (c.kh - 1) + 1 - 2 * c.ph;  // full deconv output
      const int DW = c.sw * (W - 1) + c.dw * (c.kw - 1) + 1 - 2 * c.pw;
      const int oh = op.fused_crop ? op.oh : 0, ow = op.fused_crop ? op.ow : 0;
      plan_flops += 2.0 * (double)C * H * W * N * OC * c.kh * c.kw;  // SURVEY §8(d) definition (OC = all member heads)
      // one record per residue class; they become ONE multi-class launch (kernels.h ConvClass) when a multi-class
      // instantiation of the chosen tile exists, else one launch each
      struct ClassRec {
        ConvGemmParams g;
        long y_off;
        int ry, rx;
        std::vector<std::pair<int, int>> tky, tkx;
      };
      std::vector<ClassRec> recs;
      for (int ry = 0; ry < c.sh; ++ry)
        for (int rx = 0; rx < c.sw; ++rx) {
          // rows of this class inside the (cropped) output window: Y = s*i + ry, y = Y - oh in [0, OHt)
          auto range = [](int r, int s, int off, int outn, int full, int& i0, int& cnt) {
            int lo = off - r;  // s*i >= lo
            i0 = lo <= 0 ? 0 : (lo + s - 1) / s;
            int hiY = std::min(full, off + outn) - 1;  // last Y
            int i1 = (hiY - r) >= 0 ? (hiY - r) / s : -1;
            cnt = i1 - i0 + 1;
          };
          int i0, nh, j0, nw;
          range(ry, c.sh, oh, OHt, DH, i0, nh);
          range(rx, c.sw, ow, OWt, DW, j0, nw);
          if (nh <= 0 || nw <= 0) continue;
          ClassRec rec;
          rec.ry = ry, rec.rx = rx;
          auto &tky = rec.tky, &tkx = rec.tkx;  // (k, source offset)
          for (int k = 0; k < c.kh; ++k)
            if ((ry + c.ph - k * c.dh) % c.sh == 0) tky.push_back({k, (ry + c.ph - k * c.dh) / c.sh});
          for (int k = 0; k < c.kw; ++k)
            if ((rx + c.pw - k * c.dw) % c.sw == 0) tkx.push_back({k, (rx + c.pw - k * c.dw) / c.sw});
          ConvGemmParams& g = rec.g;
          g = ConvGemmParams{};
          g.esize = es;
          g.x_img_stride = (long)H * W * CP;
          g.x_row_stride = W * CP;
          g.x_rows = H;
          g.x_rowlen = W * CP;
          g.sy = 1;
          g.sx = CP;
          const int ntaps = (int)(tky.size() * tkx.size());
          if (ntaps > kMaxTaps) throw DcError(DC_EUNSUP, "layer '" + L.name + "': too many taps");
          if (ntaps == 0) throw DcError(DC_EUNSUP, "layer '" + L.name + "': deconvolution with kernel smaller than stride");
          // the taps of a residue class form an arithmetic grid (k advances by s/gcd(s,d))
          g.nty = (int)tky.size();
          g.ntx = (int)tkx.size();
          g.dy0 = tky[0].second + i0;
          g.ddy = tky.size() > 1 ? tky[1].second - tky[0].second : 0;
          g.x0 = (tkx[0].second + j0) * CP;
          g.ddx = tkx.size() > 1 ? (tkx[1].second - tkx[0].second) * CP : 0;
          for (size_t q = 1; q < tky.size(); ++q)
            if (tky[q].second - tky[q - 1].second != g.ddy) throw DcError(DC_EUNSUP, "layer '" + L.name + "': irregular tap grid");
          for (size_t q = 1; q < tkx.size(); ++q)
            if ((tkx[q].second - tkx[q - 1].second) * CP != g.ddx) throw DcError(DC_EUNSUP, "layer '" + L.name + "': irregular tap grid");
          g.klen = CP;
          g.Ktot = ntaps * CP;
          g.NB = N;
          g.OH = nh;
          g.OW = nw;
          g.M = N * nh * nw;
          g.Cout = OC;
          g.y_img_stride = (long)OHt * OWt * OCP;
          g.y_row_stride = c.sh * OWt * OCP;
          g.y_pix_stride = c.sw * OCP;
          rec.y_off = ((long)(c.sh * i0 + ry - oh) * OWt + (c.sw * j0 + rx - ow)) * OCP;
          g.relu = op.relu;
          g.sigmoid_ch = op.sigmoid_ch >= 0 ? op.sigmoid_ch : (op.sigmoid ? OC : 0);
          recs.push_back(std::move(rec));
        }
      if (recs.empty()) throw DcError(DC_ESHAPE, "layer '" + L.name + "': empty deconvolution output");
      const std::vector<int> members = op.wls.empty() ? std::vector<int>{op.wl} : op.wls;
      // filter image of one class: [OC][taps of the class][CP], sibling layers concatenated along Cout
      auto fill_class = [&](const ClassRec& rec, float* h) {
        int cbase = 0;
        for (int ml : members) {
          const float* w = layers[ml].params[0]->st->host_ptr();  // [Cin][Cout][kh][kw]
          const int cm = layers[ml].conv.num_output;
          int t2 = 0;
          for (auto& a : rec.tky)
            for (auto& b : rec.tkx) {
              for (int co = 0; co < cm; ++co)
                for (int ci = 0; ci < C; ++ci)
                  h[(size_t)(cbase + co) * rec.g.Ktot + (size_t)t2 * CP + ci] = w[(((size_t)ci * cm + co) * c.kh + a.first) * c.kw + b.first];
              ++t2;
            }
          cbase += cm;
        }
      };
      const std::string wkey = dkey + "w:" + std::to_string(members.front()) + "x" + std::to_string(members.size());
      bool merged = false;
      if (recs.size() > 1 && (int)recs.size() <= kMaxClasses && env_int("DC_DECONV_MERGE", 1) != 0) {
        // heaviest class first: its workgroups are dispatched first, the light classes fill the tail
        std::stable_sort(recs.begin(), recs.end(), [](const ClassRec& a, const ClassRec& b) { return a.g.Ktot > b.g.Ktot; });
        Launch l = base;
        l.kind = Launch::CONV;
        l.label += " [" + std::to_string(recs.size()) + " classes]";
        l.cg = recs[0].g;
        l.cg.ncls = (int)recs.size();
        long woff = 0;
        for (size_t q = 0; q < recs.size(); ++q) {
          const ConvGemmParams& g = recs[q].g;
          ConvClass& k = l.cg.cls[q];
          k.nty = g.nty, k.ntx = g.ntx, k.dy0 = g.dy0, k.ddy = g.ddy, k.x0 = g.x0, k.ddx = g.ddx, k.Ktot = g.Ktot;
          k.OH = g.OH, k.OW = g.OW, k.M = g.M;
          k.w_off = woff;
          k.y_off = recs[q].y_off;
          woff += (long)OC * g.Ktot;
          l.flops += 2.0 * g.M * (double)OC * C * g.nty * g.ntx;
        }
        affine_vecs(op, l, OC);
        // a multi-class tile must exist among the candidates of this K granularity (or be the forced one)
        bool have_mc = false;
        for (int v = 0; v < conv_num_variants(); ++v)
          if (CP % conv_variant_bk(v) == 0 && conv_variant_esize(v) == es && conv_variant_multiclass(v) &&
              (force_variant < 0 || force_variant == v))
            have_mc = true;
        if (have_mc) {
          std::string key = wkey + ":mc";
          for (auto& r : recs) key += ":" + std::to_string(r.ry) + "," + std::to_string(r.rx);
          l.w = get_vec(key, [&](std::vector<float>& h) {
            h.assign((size_t)woff, 0.f);
            for (size_t q = 0; q < recs.size(); ++q) fill_class(recs[q], h.data() + l.cg.cls[q].w_off);
          });
          l.w->as_half = dtype == 1;
          half_row_scale(l, op, OC);
          choose_variant(l, CP);
          push_split(std::move(l), CP);
          merged = true;
        }
      }
      if (!merged)
        for (auto& rec : recs) {
          Launch l = base;
          l.kind = Launch::CONV;
          l.label += " [class " + std::to_string(rec.ry) + "," + std::to_string(rec.rx) + "]";
          l.cg = rec.g;
          l.y_off = rec.y_off;
          affine_vecs(op, l, OC);
          l.w = get_vec(wkey + ":" + std::to_string(rec.ry) + "," + std::to_string(rec.rx), [&](std::vector<float>& h) {
            h.assign((size_t)OC * rec.g.Ktot, 0.f);
            fill_class(rec, h.data());
          });
          l.flops = 2.0 * rec.g.M * (double)OC * C * rec.g.nty * rec.g.ntx;
          l.w->as_half = dtype == 1;
          half_row_scale(l, op, OC);
          choose_variant(l, CP);
          plan.push_back(std::move(l));
        }
    } else if (op.kind == LOp::POOL) {
      const LayerRec& L = layers[op.lids[0]];
      Launch l = base;
      l.kind = Launch::POOL;
      l.kernel = "maxpool";
      l.pk = L.pool_k;
      l.ps = L.pool_s;
      l.pp = L.pool_p;
      plan.push_back(std::move(l));
    } else if (op.kind == LOp::ELT) {
      Launch l = base;
      l.kind = Launch::ELT;
      l.kernel = "eltwise";
      affine_vecs(op, l, C);
      plan.push_back(std::move(l));
    } else {
      Launch l = base;
      l.kind = Launch::CROP;
      l.kernel = "crop";
      l.oh = op.oh;
      l.ow = op.ow;
      plan.push_back(std::move(l));
    }
  }
  plan_valid = true;
  tuned = false;
  plan_input_shape.clear();
  for (int bi : inputs)
    for (int d : blobs[bi]->st->shape) plan_input_shape.push_back(d);
  cur_last_use_ = ++use_clock_;
  release_graph();
}

// ---- per-shape plan cache -----------------------------------------------------------------------------
static std::vector<int> input_signature(const Net& n) {
  std::vector<int> sig;
  for (int bi : n.inputs)
    for (int d : n.blobs[bi]->st->shape) sig.push_back(d);
  return sig;
}

void Net::mark_weights_changed() {
  std::lock_guard<std::mutex> lk(shared->mu);
  ++shared->weights_gen;
}

// Parameters are handed out writable on every access (pycaffe's Blob.data is mutable_cpu_data, _caffe.cpp:273), so an
// access alone says nothing: the blobs touched since the last run are re-hashed here and only a CONTENT change moves
// the shared generation.  Every executor of the model compares that generation with the one its plans came from.
void Net::check_weights() {
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lk(shared->mu);
    if (!shared->touched.empty()) {
      bool changed = shared->packed_gen != shared->weights_gen;  // nothing packed yet: hashes are not meaningful
      for (auto& w : shared->touched)
        if (auto st = w.lock()) {
          st->touch_listed = false;
          if (!changed && content_hash(st->host_ptr(), st->count()) != st->packed_hash) changed = true;
        }
      shared->touched.clear();
      if (changed && shared->packed_gen == shared->weights_gen) ++shared->weights_gen;
    }
    gen = shared->weights_gen;
  }
  if (gen != seen_weights_gen) {
    invalidate_plans();
    seen_weights_gen = gen;
  }
}

void Net::invalidate_plans() {
  if (stream && (plan_valid || !parked_.empty())) (void)hipStreamSynchronize((hipStream_t)stream);  // nothing in flight reads them
  release_graph();
  for (auto& ps : parked_)
    if (ps->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)ps->graph_exec);
  parked_.clear();
  plan.clear();
  plan_valid = false;
  tuned = false;
}

void Net::park_current() {
  if (!plan_valid) return;
  std::unique_ptr<PlanState> ps(new PlanState());
  ps->input_shape = plan_input_shape;
  ps->plan.swap(plan);
  ps->flops = plan_flops;
  ps->views.swap(plan_views_);
  for (auto& st : storages) ps->sstate.push_back({st->id, st->view_of, st->view_c0, st->elided});
  for (auto& kv : aux_index_) ps->aux_shapes.push_back({kv.second, storages[kv.second]->shape});
  ps->graph_exec = graph_exec;
  ps->graph_buf_gen = graph_buf_gen;
  ps->tuned = tuned;
  ps->last_use = cur_last_use_;
  graph_exec = nullptr;
  plan_valid = false;
  parked_.push_back(std::move(ps));
  static const int cap = std::max(1, env_int("DC_PLAN_CACHE", 16));
  while ((int)parked_.size() > cap) {  // least recently used shape goes
    size_t lru = 0;
    for (size_t i = 1; i < parked_.size(); ++i)
      if (parked_[i]->last_use < parked_[lru]->last_use) lru = i;
    if (parked_[lru]->graph_exec) {
      if (stream) (void)hipStreamSynchronize((hipStream_t)stream);
      (void)hipGraphExecDestroy((hipGraphExec_t)parked_[lru]->graph_exec);
    }
    parked_.erase(parked_.begin() + lru);
  }
}

// Make the plan of the CURRENT input shape the active one.  Shapes of every blob are re-derived first (Layer::Forward
// calls Reshape on every forward, layer.hpp:451-456); a shape met before costs that walk and a swap, nothing else.
void Net::ensure_plan() {
  check_weights();
  reshape();
  const std::vector<int> sig = input_signature(*this);
  if (plan_valid && sig == plan_input_shape) {
    cur_last_use_ = ++use_clock_;
    return;
  }
  for (size_t i = 0; i < parked_.size(); ++i) {
    if (parked_[i]->input_shape != sig) continue;
    std::unique_ptr<PlanState> ps = std::move(parked_[i]);
    parked_.erase(parked_.begin() + i);
    park_current();
    plan_input_shape = ps->input_shape;
    plan.swap(ps->plan);
    plan_flops = ps->flops;
    plan_views_.swap(ps->views);
    for (auto& ss : ps->sstate) {
      Storage& st = *storages[ss.id];
      st.view_of = ss.view_of, st.view_c0 = ss.view_c0, st.elided = ss.elided;
    }
    for (auto& as : ps->aux_shapes) storages[as.first]->reshape(as.second);
    graph_exec = ps->graph_exec;
    graph_buf_gen = ps->graph_buf_gen;
    tuned = ps->tuned;
    plan_valid = true;
    cur_last_use_ = ++use_clock_;
    ++stats.plan_hits;
    return;
  }
  park_current();
  build_plan();
}

void Net::reserve(int n, int h, int w) { begin_batch(n, h, w); }

// ---- execution ----------------------------------------------------------------------------------------
void Net::ensure_device() {
  if (device_count() <= 0)
    throw DcError(DC_EDEVICE, "no HIP device visible: libdeepcut_hip has no CPU compute path (the CPU restatement of the "
                              "reference is test-only, under oracle/)");
  Context& c = Context::get();
  if (device < 0) device = c.device;
  HIPCHECK(hipSetDevice(device));
  if (!stream) {
    hipStream_t s;
    HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    stream = s;
  }
}

void Net::upload_vecs() {
  std::lock_guard<std::mutex> lk(shared->mu);  // an image may be shared with a clone that uploads at the same moment
  std::vector<DevVec*> todo;
  for (auto& l : plan)
    for (const std::shared_ptr<DevVec>* vp : {&l.w, &l.scale, &l.shift, &l.wino_w})
      if (*vp && !(*vp)->dev && !(*vp)->host.empty()) todo.push_back(vp->get());
  for (DevVec* vq : todo) {
    DevVec& v = *vq;
    if (!v.dev && !v.host.empty()) {
      if (v.as_half) {  // filter image of an fp16 net: upload as float, convert on the device, keep the half copy
        float* tmp = nullptr;
        dev_alloc((void**)&tmp, v.host.size() * sizeof(float));
        struct TmpGuard {
          float* p;
          ~TmpGuard() { dev_free(p); }
        } tmp_guard{tmp};
        dev_upload(tmp, v.host.data(), v.host.size() * sizeof(float), stream);
        dev_alloc((void**)&v.dev, v.host.size() * 2);
        KCHECK(launch_f32_to_f16(tmp, v.dev, (long)v.host.size(), stream));
        HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
      } else {
        dev_alloc((void**)&v.dev, v.host.size() * sizeof(float));
        dev_upload(v.dev, v.host.data(), v.host.size() * sizeof(float), stream);
      }
      v.uploaded = v.host.size();
      std::vector<float>().swap(v.host);  // the packed image lives in HBM only
    }
  }
}

// Per-shape tile selection by measurement ("benchmark mode"): every distinct GEMM signature of the plan
// is timed once per process with each eligible tile variant on the real buffers (all variants compute
// the same values up to fp32 summation order) and the fastest is kept.  DC_AUTOTUNE=0 keeps the cost
// model's choice; DC_CONV_VARIANT forces one variant.
// DC_TUNE_CACHE=<file>: "signature tile-name" per line.  Caller holds shared.mu.
static void write_tune_cache_locked(const ModelShared& shared) {
  const char* cache_path = std::getenv("DC_TUNE_CACHE");
  if (!cache_path || !*cache_path) return;
  if (FILE* f = std::fopen(cache_path, "w")) {
    for (auto& kv : shared.tune_cache)
      std::fprintf(f, "%s %s\n", kv.first.c_str(), kv.second == kWinoVariant ? "wino_f23" : conv_variant(kv.second).name);
    std::fclose(f);
  }
}

void Net::autotune() {
  tuned = true;
  if (env_int("DC_AUTOTUNE", 1) == 0 || env_int("DC_CONV_VARIANT", -1) >= 0) return;
  // DC_TUNE_CACHE=<file>: tuning results persist across processes ("signature variant-name" per line), so
  // a service (or a profiling run) starts without the timing launches
  std::lock_guard<std::mutex> tune_lock(shared->mu);  // one executor times a shape, the clones reuse its choices
  std::map<std::string, int>& tune_cache_ = shared->tune_cache;
  const char* cache_path = std::getenv("DC_TUNE_CACHE");
  if (cache_path && !shared->tune_file_loaded) {
    shared->tune_file_loaded = true;
    if (FILE* f = std::fopen(cache_path, "r")) {
      // one "<signature> <tile>" per line; group signatures (NetGroup) concatenate their members' and run to several hundred
      // characters, so lines are read whole and cut at the LAST blank
      std::string line;
      int ch;
      auto take = [&]() {
        const size_t sp = line.find_last_of(' ');
        if (sp != std::string::npos && sp > 0 && sp + 1 < line.size()) {
          const std::string key = line.substr(0, sp), vname = line.substr(sp + 1);
          if (vname == "wino_f23") tune_cache_[key] = kWinoVariant;
          for (int v = 0; v < conv_num_variants(); ++v)
            if (vname == conv_variant(v).name) tune_cache_[key] = v;
        }
        line.clear();
      };
      while ((ch = std::fgetc(f)) != EOF) {
        if (ch == '\n' || ch == '\r') take();
        else line.push_back((char)ch);
      }
      take();
      std::fclose(f);
    }
  }
  size_t cached_before = tune_cache_.size();
  bool timed_any = false;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } ev_guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  const int reps = 5;
  auto key_of = [&](const Launch& l) { return tune_key(l); };
  auto burst_ms = [&](const Launch& trial) {  // best of two timed bursts: a single burst is noisy at 10-20 us per launch
    run_launch(trial, stream);  // warm
    float ms = 1e30f;
    for (int t2 = 0; t2 < 2; ++t2) {
      HIPCHECK(hipEventRecord(e0, (hipStream_t)stream));
      for (int r = 0; r < reps; ++r) run_launch(trial, stream);
      HIPCHECK(hipEventRecord(e1, (hipStream_t)stream));
      HIPCHECK(hipEventSynchronize(e1));
      float m2 = 0;
      HIPCHECK(hipEventElapsedTime(&m2, e0, e1));
      ms = std::min(ms, m2);
    }
    return ms;
  };
  // (1) every distinct signature not in the cache: each eligible tile (and the Winograd form) timed alone, back to back
  std::map<std::string, std::vector<std::pair<float, int>>> timed;  // signature -> (ms, variant) of this pass
  for (auto& l : plan) {
    if (l.kind != Launch::CONV) continue;
    const ConvGemmParams& g = l.cg;
    const std::string key = key_of(l);
    if (tune_cache_.count(key)) continue;
    timed_any = true;
    std::vector<std::pair<float, int>>& c = timed[key];
    for (int v = 0; v < conv_num_variants(); ++v) {
      if (g.klen % conv_variant_bk(v) != 0 || conv_variant_esize(v) != g.esize) continue;
      if (g.ncls > 1 && !conv_variant_multiclass(v)) continue;
      Launch trial = l;
      trial.variant = v;
      c.push_back({burst_ms(trial), v});
    }
    if (l.wino_w) {  // the Winograd form of this layer competes with the direct tiles
      Launch trial = l;
      trial.variant = kWinoVariant;
      c.push_back({burst_ms(trial), kWinoVariant});
    }
    std::sort(c.begin(), c.end());
    tune_cache_[key] = c.empty() ? l.variant : c.front().second;
    shared->tune_timings[key] = c;
  }
  // (2) in situ: a launch timed alone re-reads warm filters and starts on an idle chip; inside a forward it follows another
  // kernel's tail and finds its filters wherever the 263 MB sweep of the forward left them.  The candidates within 12 % of a
  // signature's best (at most 4) are therefore compared once more inside whole passes over the plan (hipEvents around every
  // launch of the signature, summed; best of 3 passes per candidate): measured on the float16 batch-8 forward, the isolated
  // timing took the 256x128 tile for the merged heads on two boxes of three where the 128-wide ones are 9 % faster in the
  // network.  (Like pass 1 this runs before the inputs of the forward are brought to the device: outputs are scratch here.)
  // These are latency choices: for the float16 batch-8 forward they are 1-3 % faster one forward at a time (the 4-scale pyramid:
  // 535 -> 560 image-pyramids/s) and 1-2 % slower with two forwards in flight (they lean to the one-workgroup-per-CU tiles,
  // which leave the second forward no room) — a service that keeps forwards in flight re-tunes for its load (set_tile,
  // deepcut_tools.tune_in_flight).
  if (timed_any && env_int("DC_TUNE_INSITU", 1) != 0) {
    std::map<std::string, std::vector<int>> shortlist;
    size_t rounds = 0;
    for (auto& kv : timed) {
      std::vector<int> sl;
      for (auto& c : kv.second)  // (25 % / six candidates were tried: same choices, three times the passes)
        if (sl.size() < 4 && c.first <= kv.second.front().first * 1.12f) sl.push_back(c.second);
      if (sl.size() >= 2) {
        rounds = std::max(rounds, sl.size());
        shortlist[kv.first] = sl;
      }
    }
    if (rounds) {
      std::vector<int> idx;  // plan indices of the launches under comparison
      std::vector<std::string> keys;
      for (size_t i = 0; i < plan.size(); ++i) {
        if (plan[i].kind != Launch::CONV) continue;
        std::string k = key_of(plan[i]);
        if (shortlist.count(k)) idx.push_back((int)i), keys.push_back(k);
      }
      std::vector<hipEvent_t> ev(2 * idx.size(), nullptr);
      std::vector<Launch> saved = plan;
      // the passes below overwrite plan[].variant with trial tiles: whatever throws in there, the executor must get its
      // plan back (labels and grids of `saved` match its variants) and the events must not leak
      struct Restore {
        std::vector<Launch>& plan;
        std::vector<Launch>& saved;
        std::vector<hipEvent_t>& ev;
        bool armed = true;
        ~Restore() {
          if (armed) plan = saved;
          for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        }
      } restore{plan, saved, ev};
      for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
      std::map<std::string, std::vector<float>> best;  // signature -> per shortlist entry, ms summed over its launches
      for (auto& kv : shortlist) best[kv.first].assign(kv.second.size(), 1e30f);
      for (size_t r = 0; r < rounds; ++r)
        for (int pass = 0; pass < 3; ++pass) {
          for (size_t j = 0; j < idx.size(); ++j) {
            const std::vector<int>& sl = shortlist[keys[j]];
            plan[idx[j]].variant = sl[std::min(r, sl.size() - 1)];
          }
          size_t j = 0;
          for (size_t i = 0; i < plan.size(); ++i) {
            const bool watched = j < idx.size() && idx[j] == (int)i;
            if (watched) HIPCHECK(hipEventRecord(ev[2 * j], (hipStream_t)stream));
            run_launch(plan[i], stream);
            if (watched) {
              HIPCHECK(hipEventRecord(ev[2 * j + 1], (hipStream_t)stream));
              ++j;
            }
          }
          HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
          std::map<std::string, float> sum;
          for (size_t q = 0; q < idx.size(); ++q) {
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ev[2 * q], ev[2 * q + 1]));
            sum[keys[q]] += ms;
          }
          for (auto& kv : sum) {
            const size_t e = std::min(r, shortlist[kv.first].size() - 1);
            best[kv.first][e] = std::min(best[kv.first][e], kv.second);
          }
        }
      plan = saved;
      restore.armed = false;  // (its destructor still destroys the events)
      for (auto& kv : best) {
        size_t arg = 0;
        for (size_t e = 1; e < kv.second.size(); ++e)
          if (kv.second[e] < kv.second[arg]) arg = e;
        tune_cache_[kv.first] = shortlist[kv.first][arg];
      }
    }
  }
  // (3) the choices go into the plan
  for (auto& l : plan) {
    if (l.kind != Launch::CONV) continue;
    const ConvGemmParams& g = l.cg;
    auto it = tune_cache_.find(key_of(l));
    // a cache line naming the Winograd form while it is switched off (or not eligible any more): keep the cost model's tile
    if (it != tune_cache_.end() && !(it->second == kWinoVariant && !l.wino_w) &&
        !(g.ncls > 1 && (it->second == kWinoVariant || !conv_variant_multiclass(it->second))))
      l.variant = it->second;
    if (l.variant == kWinoVariant) {
      l.kernel = "wino_f23<4x8x16>";
      l.grid = wino_grid(l.cg);
    } else {
      l.kernel = std::string("conv_gemm<") + conv_variant(l.variant).name + ">";
      l.grid = conv_grid(l.cg, l.variant);
    }
  }
  if (timed_any) ++stats.autotune_runs;
  if (cache_path && (tune_cache_.size() != cached_before || timed_any)) write_tune_cache_locked(*shared);
  ++tile_gen_;
  release_graph();
}

// GEMM signature of a launch: the key of the tile choice ("h" prefix: float16; "+w": the Winograd form competes for this layer —
// a different candidate set than with DC_WINOGRAD=0 —; "+mcN:K..": a multi-class launch, N classes with these K and M)
std::string Net::tune_key(const Launch& l) const {
  const ConvGemmParams& g = l.cg;
  char key[200];
  std::string mck;
  if (g.ncls > 1) {
    mck = "+mc" + std::to_string(g.ncls);
    for (int c = 0; c < g.ncls; ++c) mck += ":" + std::to_string(g.cls[c].Ktot) + "m" + std::to_string(g.cls[c].M);
  }
  std::snprintf(key, sizeof key, "%s%d/%d/%d/%d/%dx%d/%d,%d/%d/%d%s%s", g.esize == 2 ? "h" : "", g.M, g.Cout, g.Ktot, g.klen, g.nty, g.ntx,
                g.sy, g.sx, l.in2 >= 0 ? 1 : 0, g.OW, l.wino_w ? "+w" : "", mck.c_str());
  return key;
}

// One line per GEMM signature of the current plan, in plan order:
//   <signature> \t <tile in use> \t <launches with it> \t <tile>:<us per launch, timed alone> ...   (fastest first; empty if the
// choice came from a DC_TUNE_CACHE file).  What deepcut_tools.tune_in_flight walks.
std::string Net::tune_report_text() {
  std::lock_guard<std::mutex> lk(shared->mu);
  std::vector<std::string> order;
  std::map<std::string, std::pair<int, int>> seen;  // key -> (variant in use, launches)
  for (auto& l : plan) {
    if (l.kind != Launch::CONV) continue;
    const std::string k = tune_key(l);
    auto it = seen.find(k);
    if (it == seen.end()) order.push_back(k), seen[k] = {l.variant, 1};
    else ++it->second.second;
  }
  auto vname = [](int v) { return std::string(v == kWinoVariant ? "wino_f23" : conv_variant(v).name); };
  std::string out;
  for (auto& k : order) {
    out += k + "\t" + vname(seen[k].first) + "\t" + std::to_string(seen[k].second) + "\t";
    auto t = shared->tune_timings.find(k);
    if (t != shared->tune_timings.end())
      for (size_t i = 0; i < t->second.size(); ++i) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "%s%s:%.2f", i ? " " : "", vname(t->second[i].second).c_str(), t->second[i].first * 1000.f / 5.f);
        out += buf;
      }
    out += "\n";
  }
  return out;
}

// The tile of one signature, chosen by the caller (a tuner working under its own load): checked against every launch of the
// current plan that has the signature, recorded in the shared choice table (clones pick it up at their next lowering; call
// set_tile on each executor to change their current plans), and the captured graph is dropped.
void Net::set_tile(const std::string& key, const std::string& tile) {
  int v = -1;
  if (tile == "wino_f23") v = kWinoVariant;
  for (int i = 0; v < 0 && i < conv_num_variants(); ++i)
    if (tile == conv_variant(i).name) v = i;
  if (v < 0) throw DcError(DC_EINVAL, "no tile variant named '" + tile + "'");
  bool any = false;
  for (auto& l : plan) {
    if (l.kind != Launch::CONV || tune_key(l) != key) continue;
    const ConvGemmParams& g = l.cg;
    const bool ok = v == kWinoVariant ? (bool)l.wino_w && g.ncls <= 1
                                      : g.klen % conv_variant_bk(v) == 0 && conv_variant_esize(v) == g.esize && (g.ncls <= 1 || conv_variant_multiclass(v));
    if (!ok) throw DcError(DC_EUNSUP, "tile '" + tile + "' cannot take launch '" + l.label + "' (" + key + ")");
    any = true;
  }
  if (!any) throw DcError(DC_EINVAL, "the current plan has no launch with signature '" + key + "'");
  for (auto& l : plan) {
    if (l.kind != Launch::CONV || tune_key(l) != key) continue;
    l.variant = v;
    if (v == kWinoVariant) {
      l.kernel = "wino_f23<4x8x16>";
      l.grid = wino_grid(l.cg);
    } else {
      l.kernel = std::string("conv_gemm<") + conv_variant(v).name + ">";
      l.grid = conv_grid(l.cg, v);
    }
  }
  {
    std::lock_guard<std::mutex> lk(shared->mu);
    auto it = shared->tune_cache.find(key);
    if (it == shared->tune_cache.end() || it->second != v) {
      shared->tune_cache[key] = v;
      write_tune_cache_locked(*shared);  // an override changes a value, not the size of the table: persist it too
    }
  }
  ++tile_gen_;
  release_graph();
}

void Net::release_graph() {
  if (graph_exec) {
    (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    graph_exec = nullptr;
  }
}

// SyncedMemory::to_gpu (syncedmem.cpp:49-77): UNINITIALIZED -> a zeroed device image that is authoritative
// (HEAD_AT_GPU); HEAD_AT_CPU -> upload, SYNCED.  The device image of a 4-D blob is channels-last.
static void storage_to_device_impl(Storage& s, void* stream, bool wait);
void storage_to_device(Storage& s, void* stream) { storage_to_device_impl(s, stream, true); }
// wait = false: the upload is enqueued and the caller synchronises the stream before the host copy can change again
// (Net::forward: its own final synchronisation covers the inputs it sent up — one round trip less per forward)
static void storage_to_device_impl(Storage& s, void* stream, bool wait) {
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
Net* Net::create_for_layer(const std::string& layer_text, int phase, const std::vector<std::vector<int>>& bottom_shapes) {
  TextMsg m = parse_text_proto(layer_text);
  const TextMsg* L = &m;
  if (m.sub("layer") && !m.has("type")) L = m.sub("layer");  // given with the enclosing `layer { }`
  std::vector<std::string> bottoms = L->strs("bottom");
  if (bottoms.size() != bottom_shapes.size())
    throw DcError(DC_EINVAL, "layer '" + L->str("name") + "' declares " + std::to_string(bottoms.size()) + " bottom(s), " +
                                 std::to_string(bottom_shapes.size()) + " given");
  std::string text = "name: \"" + L->str("name") + "\"\n";
  std::set<std::string> seen;
  for (size_t i = 0; i < bottoms.size(); ++i) {
    if (!seen.insert(bottoms[i]).second) throw DcError(DC_EUNSUP, "layer '" + L->str("name") + "': the same bottom twice");
    text += "input: \"" + bottoms[i] + "\"\ninput_shape {";
    for (int d : bottom_shapes[i]) text += " dim: " + std::to_string(d);
    text += " }\n";
  }
  text += L == &m ? "layer {\n" + layer_text + "\n}\n" : layer_text + "\n";
  std::unique_ptr<Net> n(Net::create(text, phase));
  n->fuse = 0;
  return n.release();
}

void Net::run_launch(const Launch& l, void* s) {
  Storage& X = *storages[l.in];
  Storage& Y = *storages[l.out];
  switch (l.kind) {
    case Launch::CONV: {
      ConvGemmParams g = l.cg;
      g.dbg = nullptr;
      g.x = X.dev;
      g.y = Y.dev_at(l.y_off);
      g.resid = l.in2 >= 0 ? storages[l.in2]->dev_at(l.y_off) : nullptr;
      g.w = reinterpret_cast<const unsigned char*>(l.w->dev) + (size_t)l.w_off * (size_t)g.esize;
      g.scale = l.scale ? l.scale->dev + l.c_off : nullptr;
      g.shift = l.shift ? l.shift->dev + l.c_off : nullptr;
      const bool wino = l.variant == kWinoVariant;  // Winograd F(2x2,3x3) form of a stride-1 3x3 layer
      if (wino) {
        if (!l.wino_w) throw DcError(DC_EINVAL, "launch '" + l.label + "' has no Winograd filter image");
        g.w = l.wino_w->dev;
      }
      static const int dbg_idx = env_int("DC_DEBUG_TIMING", -1);
      // index of this launch in the plan (autotuning passes copies, which have none)
      const bool in_plan = !plan.empty() && std::greater_equal<const Launch*>()(&l, plan.data()) &&
                           std::less<const Launch*>()(&l, plan.data() + plan.size());
      const int my_idx = in_plan ? (int)(&l - plan.data()) : -1;
      if (dbg_idx >= 0 && my_idx == dbg_idx) {
        // device-side phase timestamps of ONE launch (diagnostics only): per wave the shader cycle counter at up to 8 phase
        // boundaries (slots 0..7) and the chip-wide 100 MHz clock at start / end (slots 8, 9)
        const int nwv = wino ? 8 : conv_variant(l.variant).WR * conv_variant(l.variant).WC * conv_variant(l.variant).WK;
        const long n = (l.grid * 2 + 64) * nwv * 12;  // the XCD-aware maps pad the grid (at most 8 x the longest XCD list)
        long long* d = nullptr;
        dev_alloc((void**)&d, n * sizeof(long long));
        dev_zero(d, n * sizeof(long long), s);
        for (int rep = 0; rep < 3; ++rep) {
          g.dbg = d;
          HIPCHECK(hipStreamSynchronize((hipStream_t)s));
          if (wino) KCHECK(launch_wino_conv(g, s));
          else KCHECK(launch_conv_gemm(g, l.variant, s));
          HIPCHECK(hipStreamSynchronize((hipStream_t)s));
        }
        std::vector<long long> h(n);
        HIPCHECK(hipMemcpyAsync(h.data(), d, n * sizeof(long long), hipMemcpyDeviceToHost, (hipStream_t)s));
        HIPCHECK(hipStreamSynchronize((hipStream_t)s));
        dev_free(d);
        double dsum[8] = {0}, karg = 0;
        long cnt = 0;
        long long t0min = 0, t0max = 0, t7max = 0;
        for (long i = 0; i < (l.grid * 2 + 64) * nwv; ++i) {
          const long long* w = &h[i * 12];
          if (w[7] == 0 || w[0] == 0) continue;  // workgroup of the padded XCD grid that exited at once
          for (int k = 1; k < 8; ++k) dsum[k] += (double)(w[k] - w[k - 1]);
          karg += (double)(w[10] - w[8]);
          if (!cnt || w[8] < t0min) t0min = w[8];
          if (!cnt || w[8] > t0max) t0max = w[8];
          if (!cnt || w[9] > t7max) t7max = w[9];
          ++cnt;
        }
        std::fprintf(stderr, "[dc timing] first wave start -> last wave start %.2f us | first start -> last end %.2f us | wave entry -> kernel "
                     "arguments there %.2f us | waves %ld\n",
                     (t0max - t0min) / 100.0, (t7max - t0min) / 100.0, karg / std::max(cnt, 1L) / 100.0, cnt);
        // conv_gemm slots: 0 start, 1 filter loads + epilogue constants issued, 2 rows decoded + activation loads issued,
        // 3 output offsets in LDS, 4 first tile staged (K-loop entry), 5 K-loop exit, 6 split-K exchange done, 7 stores issued
        static const char* kGemm[7] = {"filter-load issue", "row decode + activation-load issue", "output offsets to LDS", "wait+stage+barrier",
                                       "K loop", "split-K exchange", "epilogue math+stores"};
        static const char* kWino[7] = {"index setup", "first loads issued", "two stages in LDS", "K loop", "partials to LDS + barrier",
                                       "inverse transform + epilogue constants", "shortcut + stores"};
        std::string line;
        for (int k = 1; k < 8; ++k) {
          char buf[96];
          std::snprintf(buf, sizeof buf, "%s%s %.0f", k > 1 ? " | " : "", (wino ? kWino : kGemm)[k - 1], dsum[k] / std::max(cnt, 1L));
          line += buf;
        }
        std::fprintf(stderr, "[dc timing] launch %d %s %s\n  mean cycles per wave: %s\n", my_idx, l.kernel.c_str(), l.label.c_str(), line.c_str());
        g.dbg = nullptr;
      }
      if (wino) {
        KCHECK(launch_wino_conv(g, s));
        break;
      }
      {
        const int rc = launch_conv_gemm(g, l.variant, s);
        if (rc == (int)hipErrorInvalidValue)
          throw DcError(DC_EUNSUP, "launch '" + l.label + "': unsupported geometry (a tensor of 2 GiB or more per launch — "
                                   "split the batch — or a tap / K layout this variant cannot take)");
        KCHECK(rc);
      }
      break;
    }
    case Launch::POOL:
      KCHECK(launch_maxpool(X.dev, Y.dev, X.esize, X.dim(0), X.dim(2), X.dim(3), X.cp(), Y.dim(2), Y.dim(3), l.pk, l.ps, l.pp, s));
      break;
    case Launch::ELT:
      KCHECK(launch_eltwise(X.dev, l.in2 >= 0 ? storages[l.in2]->dev : nullptr, l.scale ? l.scale->dev : nullptr,
                            l.shift ? l.shift->dev : nullptr, Y.dev, Y.esize, (long)Y.dev_count(), Y.cp(), l.relu,
                            l.sigmoid, s));
      break;
    case Launch::CROP:
      KCHECK(launch_crop(X.dev, Y.dev, X.esize, X.dim(0), X.dim(2), X.dim(3), X.cp(), l.oh, l.ow, Y.dim(2), Y.dim(3), s));
      break;
  }
}

void Net::run_plan(int start, int end, void* s) {
  for (auto& l : plan) {
    if (l.last_layer < start || l.first_layer > end) continue;
    if (l.first_layer < start || l.last_layer > end)
      throw DcError(DC_EINVAL, "forward range [" + std::to_string(start) + "," + std::to_string(end) +
                                   "] cuts through the fused group '" + l.label + "'; use DC_OPT_FUSE 0 for partial ranges");
    run_launch(l, s);
  }
}

static void prepare_buffers(Net& n, bool& grew) {
  grew = false;
  auto prep = [&](int sidx) {
    Storage& s = *n.storages[sidx];
    size_t need = s.dev_count();
    if (!s.dev || s.dev_cap < std::max<size_t>(need, 8) * (size_t)s.esize) {
      s.ensure_dev(need);
      grew = true;
    }
  };
  for (auto& l : n.plan) {
    prep(l.in);
    prep(l.out);
    if (l.in2 >= 0) prep(l.in2);
  }
}

void Net::forward(int start, int end) {
  if (Context::get().mode != DC_MODE_GPU)
    throw DcError(DC_ENOCPU, "forward() in CPU mode: libdeepcut_hip provides the MI355X path only — call set_mode_gpu() "
                             "(the CPU restatement of the reference is test infrastructure under oracle/)");
  ensure_plan();
  ensure_device();
  upload_vecs();
  bool grew;
  prepare_buffers(*this, grew);
  const bool whole = start <= 0 && end >= (int)layers.size() - 1;
  // The timing passes replay launches of the WHOLE plan (in-place and Eltwise ones included) before the inputs are synced:
  // harmless while every output they overwrite is scratch, i.e. for a full forward.  A partial range may start from
  // intermediate blobs the caller placed on the device (mutable_gpu_data): those must not be clobbered, so a partial
  // forward runs with the tiles the plan has (cost model / tune cache) and leaves the tuning to the next full forward.
  if (!tuned && whole) autotune();
  // inputs of the executed range whose host copy is authoritative go up first (SyncedMemory::to_gpu)
  for (auto& l : plan) {
    if (l.last_layer < start || l.first_layer > end) continue;
    for (int sidx : {l.in, l.in2})
      if (sidx >= 0) {
        Storage& s = *storages[sidx];
        if (s.head == HEAD_AT_CPU || s.head == UNINITIALIZED) {
          bool produced_earlier = false;
          for (auto& m : plan) {
            if (&m == &l) break;
            if (m.last_layer < start || m.first_layer > end) continue;
            if (m.out == sidx) produced_earlier = true;
          }
          if (!produced_earlier) {
            ensure_device();
            // pinned host copy (non-parameter blobs): the copy engine reads it behind our back until the stream is drained —
            // which forward() does before it returns; pageable memory is staged by the runtime at enqueue time
            storage_to_device_impl(s, stream, false);
          }
        }
      }
  }
  if (use_graph && whole) {
    if (graph_exec && graph_buf_gen != buf_gen_) release_graph();  // a buffer it addresses was reallocated since
    if (!graph_exec) {
      graph_exec = capture_graph(stream, [&](void* cs) { run_plan(start, end, cs); });
      graph_buf_gen = buf_gen_;
      ++stats.graph_instantiations;
    }
    HIPCHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  } else {
    run_plan(start, end, stream);
  }
  for (auto& l : plan) {
    if (l.last_layer < start || l.first_layer > end) continue;
    storages[l.out]->head = HEAD_AT_GPU;
  }
  for (int v : plan_views_) storages[v]->head = HEAD_AT_GPU;
  // the outputs the caller has been reading through host pointers travel now, behind the last launch (Storage::host_wanted)
  std::vector<Storage*> delivered;
  if (whole)
    for (int bi : outputs) {
      Storage& st = *blobs[bi]->st;
      if (!st.host_wanted || st.head != HEAD_AT_GPU || st.elided) continue;
      if (!st.host_touched) {  // not read since the last delivery: stop sending it
        st.host_wanted = false;
        continue;
      }
      st.host_touched = false;
      storage_download_enqueue(st, stream, st.view_of >= 0 ? storages[st.view_of].get() : nullptr);
      delivered.push_back(&st);
    }
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  for (Storage* st : delivered) st->head = SYNCED;
}

// Common front half of the batched entries: shape the input blob, (re)build the plan, make the device state ready.
Storage& Net::begin_batch(int n, int h, int w) {
  if (Context::get().mode != DC_MODE_GPU)
    throw DcError(DC_ENOCPU, "forward_batch() in CPU mode: libdeepcut_hip provides the MI355X path only");
  if (inputs.size() != 1) throw DcError(DC_EINVAL, "forward_batch needs a single-input net");
  Storage& in = *blobs[inputs[0]]->st;
  int C = in.dim(1);
  in.reshape({n, C, h, w});
  ensure_plan();
  ensure_device();
  upload_vecs();
  bool grew;
  prepare_buffers(*this, grew);
  if (!tuned) autotune();
  return in;
}

// Enqueue every launch of the plan on stream s (the input image is already in HBM).
void Net::enqueue_plan(void* s) {
  const int last = (int)layers.size() - 1;
  if (use_graph) {
    // the launch sequence is captured once on the net's own stream and replayed on whichever stream
    // the caller works on (a graph is not tied to its capture stream)
    if (graph_exec && graph_buf_gen != buf_gen_) release_graph();  // a buffer it addresses was reallocated since
    if (!graph_exec) {
      graph_exec = capture_graph(stream, [&](void* cs) { run_plan(0, last, cs); });
      graph_buf_gen = buf_gen_;
      ++stats.graph_instantiations;
    }
    HIPCHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)s));
  } else {
    run_plan(0, last, s);
  }
  for (auto& l : plan) storages[l.out]->head = HEAD_AT_GPU;
  for (int v : plan_views_) storages[v]->head = HEAD_AT_GPU;
}

// Copy the three output maps out as NCHW (host or device destination), enqueued on s: float32, or — dst_esize 2, fp16
// nets only — the half values as they are in HBM (half the gather payload).
void Net::emit_maps(void* prob, void* loc, void* next, bool is_device, void* s, int dst_esize) {
  struct Out {
    const char* name;
    void* dst;
  } outs[3] = {{"prob", prob}, {"loc_pred", loc}, {"next_pred", next}};
  if (dst_esize != 4 && !(dst_esize == 2 && dtype == 1))
    throw DcError(DC_EINVAL, "maps are emitted as float32, or as float16 from a float16 net (DC_OPT_DTYPE 1)");
  for (auto& o : outs) {
    if (!o.dst) continue;
    auto it = blob_index.find(o.name);
    if (it == blob_index.end()) throw DcError(DC_EINVAL, std::string("net has no blob '") + o.name + "'");
    Storage& st = *blobs[it->second]->st;
    size_t m = st.count();
    const void* src = st.view_of >= 0 ? storages[st.view_of]->dev : st.dev;
    const int ses = st.view_of >= 0 ? storages[st.view_of]->esize : st.esize;
    const int scp = st.view_of >= 0 ? storages[st.view_of]->cp() : st.cp();
    const int sc0 = st.view_of >= 0 ? st.view_c0 : 0;
    if (st.head == UNINITIALIZED) throw DcError(DC_EINVAL, std::string("'") + o.name + "': run a forward first");
    if (is_device) {
      KCHECK(launch_nhwc_to_nchw(src, o.dst, ses, st.dim(0), st.dim(1), st.dim(2), st.dim(3), scp, sc0, s, dst_esize));
    } else {
      st.ensure_stage(m);  // sized in floats: large enough for either element type
      KCHECK(launch_nhwc_to_nchw(src, st.stage, ses, st.dim(0), st.dim(1), st.dim(2), st.dim(3), scp, sc0, s, dst_esize));
      HIPCHECK(hipMemcpyAsync(o.dst, st.stage, m * (size_t)dst_esize, hipMemcpyDeviceToHost, (hipStream_t)s));
    }
  }
}

void Net::emit_last_maps(void* prob, void* loc, void* next, int elem, bool is_device, void* user_stream) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "emit_maps() in CPU mode");
  if (elem != 0 && elem != 1) throw DcError(DC_EINVAL, "element type must be 0 (float32) or 1 (float16)");
  ensure_device();
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  void* s = user_stream ? user_stream : stream;
  emit_maps(prob, loc, next, is_device, s, elem == 1 ? 2 : 4);
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void Net::forward_batch(const float* input, int n, int h, int w, bool is_device, float* prob, float* loc, float* next,
                        void* user_stream) {
  Storage& in = begin_batch(n, h, w);
  const int C = in.dim(1);
  const bool own_async = user_stream == (void*)-1;  // DC_STREAM_OWN: the net's stream, no final sync
  if (own_async) user_stream = nullptr;
  void* s = user_stream ? user_stream : stream;
  size_t cnt = in.count();
  if (is_device) {
    KCHECK(launch_nchw_to_nhwc(input, in.dev, in.esize, n, C, h, w, in.cp(), s));
  } else {
    in.ensure_stage(cnt);
    HIPCHECK(hipMemcpyAsync(in.stage, input, cnt * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)s));
    KCHECK(launch_nchw_to_nhwc(in.stage, in.dev, in.esize, n, C, h, w, in.cp(), s));
  }
  in.head = HEAD_AT_GPU;
  enqueue_plan(s);
  emit_maps(prob, loc, next, is_device, s);
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

// n independent requests of one image each -> one batch-n launch plan: at batch 1 a res4 layer is 196 workgroups on 256 CUs
// and a third of its time is fixed cost; the same layers at batch 2-4 fill the chip and pay the fixed cost once.  The
// per-request NCHW device buffers are gathered into / scattered from the batch image by the layout kernels themselves.
void Net::forward_requests(int n, const float* const* inputs, int h, int w, float* const* prob, float* const* loc, float* const* next,
                           void* user_stream) {
  Storage& in = begin_batch(n, h, w);
  const int C = in.dim(1);
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  void* s = user_stream ? user_stream : stream;
  const long img = (long)h * w * in.cp();
  for (int i = 0; i < n; ++i) KCHECK(launch_nchw_to_nhwc(inputs[i], in.dev_at(i * img), in.esize, 1, C, h, w, in.cp(), s));
  in.head = HEAD_AT_GPU;
  enqueue_plan(s);
  struct Out {
    const char* name;
    float* const* dst;
  } outs[3] = {{"prob", prob}, {"loc_pred", loc}, {"next_pred", next}};
  for (auto& o : outs) {
    if (!o.dst) continue;
    const MapRef m = map_ref(o.name);
    const long per = (long)m.H * m.W * m.cp;
    for (int i = 0; i < n; ++i)
      if (o.dst[i])
        KCHECK(launch_nhwc_to_nchw((const unsigned char*)m.ptr + (size_t)i * per * m.es, o.dst[i], m.es, 1, m.C, m.H, m.W, m.cp, m.c0, s));
  }
  if (!(user_stream || own_async)) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

// ---- image entry: the demo's pre-processing on the device ------------------------------------------------------------
// python/pose/estimate_pose.py:83-103: replicate the last row/column 64 px, scipy.misc.imresize(.., scale, 'bilinear')
// (= Pillow's 8-bit two-pass resample to (int(W*s), int(H*s)); identity when the size does not change), subtract the
// BGR mean, paste on a zero canvas whose sides are rounded up to the stride.  The resample is integer arithmetic with
// 22-bit fixed-point weights; the weights are computed here on the host in double precision exactly as
// Pillow's precompute_coeffs / normalize_coeffs_8bpc do, so the device result is bit-identical to the reference's.
#pragma clang fp contract(off)
ResampleTable::~ResampleTable() {
  dev_free(dev_bounds);
  dev_free(dev_coeffs);
}

void resample_coeffs(int in_size, int out_size, int& ksize, std::vector<int>& bounds, std::vector<int>& coeffs) {
  const int kPrecisionBits = 32 - 8 - 2;
  double filterscale, scale;
  filterscale = scale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;  // bilinear: support 1
  ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  coeffs.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      const double w = v < 1.0 ? 1.0 - v : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      coeffs[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
    }
    bounds[(size_t)xx * 2] = xmin;
    bounds[(size_t)xx * 2 + 1] = xmax;
  }
}

std::shared_ptr<ResampleTable> Net::resample_table(int in_size, int out_size) {
  auto key = std::make_pair(in_size, out_size);
  auto it = resample_.find(key);
  if (it != resample_.end()) return it->second;
  auto t = std::make_shared<ResampleTable>();
  std::vector<int> b, c;
  resample_coeffs(in_size, out_size, t->ksize, b, c);
  t->bounds = b;
  dev_alloc((void**)&t->dev_bounds, b.size() * sizeof(int));
  dev_alloc((void**)&t->dev_coeffs, c.size() * sizeof(int));
  dev_upload(t->dev_bounds, b.data(), b.size() * sizeof(int), stream);
  dev_upload(t->dev_coeffs, c.data(), c.size() * sizeof(int), stream);
  if (resample_.size() > 64) resample_.clear();  // a pyramid uses a handful; bound the cache anyway
  resample_[key] = t;
  return t;
}

void image_canvas_size(int h, int w, double scale, int& out_h, int& out_w, int& new_h, int& new_w) {
  const int kStride = 8, kPad = 64;
  out_w = (int)(std::ceil((double)w * scale / kStride) * kStride);  // estimate_pose.py:85-88
  out_h = (int)(std::ceil((double)h * scale / kStride) * kStride);
  new_w = (int)((double)(w + kPad) * scale);  // scipy.misc.imresize: (array(im.size) * scale).astype(int)
  new_h = (int)((double)(h + kPad) * scale);
}

void Net::forward_images(const unsigned char* bgr, int n, int h, int w, double scale, bool is_device, float* prob, float* loc,
                         float* next, double* pose, void* user_stream) {
  if (Context::get().mode != DC_MODE_GPU)
    throw DcError(DC_ENOCPU, "forward_images() in CPU mode: libdeepcut_hip provides the MI355X path only");
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  // (the net's own stream exists only after ensure_device(): prep_images resolves a null `s` to it)
  prep_images(bgr, n, h, w, scale, is_device, user_stream);
  void* s = user_stream ? user_stream : stream;
  enqueue_plan(s);
  emit_maps(prob, loc, next, is_device, s);
  if (pose) {
    decode_pose(scale, pose, is_device, (user_stream || own_async) ? s : nullptr);
  }
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void Net::prep_images(const unsigned char* bgr, int n, int h, int w, double scale, bool is_device, void* s) {
  if (n <= 0 || h <= 0 || w <= 0 || !(scale > 0)) throw DcError(DC_EINVAL, "forward_images: n, height, width and scale must be positive");
  int out_h, out_w, new_h, new_w;
  image_canvas_size(h, w, scale, out_h, out_w, new_h, new_w);
  if (new_h < 1 || new_w < 1 || out_h < 8 || out_w < 8)
    throw DcError(DC_ESHAPE, "forward_images: scale " + std::to_string(scale) + " leaves no pixels of a " + std::to_string(h) + "x" +
                                 std::to_string(w) + " image");
  Storage& in = begin_batch(n, out_h, out_w);
  if (in.dim(1) != 3) throw DcError(DC_ESHAPE, "forward_images needs a 3-channel input blob");
  if (!s) s = stream;
  const int kPad = 64;
  const int ph = h + kPad, pw = w + kPad;              // the replicate-padded image (never materialised)
  const int use_h = std::min(out_h, new_h), use_w = std::min(out_w, new_w);  // part of the resized image on the canvas
  const unsigned char* src = bgr;
  const size_t bytes = (size_t)n * h * w * 3;
  if (!is_device) {
    if (bytes > img_cap_) {
      dev_free(img_dev_);
      img_dev_ = nullptr;
      dev_alloc((void**)&img_dev_, bytes);
      img_cap_ = bytes;
    }
    HIPCHECK(hipMemcpyAsync(img_dev_, bgr, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    src = img_dev_;
  }
  const bool need_x = new_w != pw, need_y = new_h != ph;
  ImagePrepParams q{};
  q.src = src;
  q.n = n, q.h = h, q.w = w;
  q.out_h = out_h, q.out_w = out_w, q.use_h = use_h, q.use_w = use_w;
  q.dst = in.dev, q.dst_esize = in.esize, q.dst_cp = in.cp();
  q.mean[0] = 104.f, q.mean[1] = 117.f, q.mean[2] = 123.f;  // _MEAN, estimate_pose.py:26
  std::shared_ptr<ResampleTable> hold_y, hold_x;  // the tables outlive a cache flush until the launches are enqueued
  if (need_y) {
    hold_y = resample_table(ph, new_h);
    const ResampleTable& ty = *hold_y;
    q.y_bounds = ty.dev_bounds, q.y_coeffs = ty.dev_coeffs, q.y_ksize = ty.ksize;
    // rows of the (padded, horizontally resampled) image the kept output rows read
    q.row0 = ty.bounds[0];
    q.rows = ty.bounds[(size_t)(use_h - 1) * 2] + ty.bounds[(size_t)(use_h - 1) * 2 + 1] - q.row0;
  } else {
    q.row0 = 0, q.rows = use_h;
  }
  if (need_x) {
    hold_x = resample_table(pw, new_w);
    const ResampleTable& tx = *hold_x;
    q.x_bounds = tx.dev_bounds, q.x_coeffs = tx.dev_coeffs, q.x_ksize = tx.ksize;
    const size_t tb = (size_t)n * q.rows * use_w * 4;
    if (tb > tmp_cap_) {
      dev_free(tmp_dev_);
      tmp_dev_ = nullptr;
      dev_alloc((void**)&tmp_dev_, tb);
      tmp_cap_ = tb;
    }
    q.tmp = tmp_dev_;
  }
  KCHECK(launch_image_prep(q, s));
  in.head = HEAD_AT_GPU;
}

// _pose_from_mats (python/pose/estimate_pose.py:131-143) on the device: reads the `prob` and `loc_pred`
// images of the last forward where they live (channel views of the merged head tensor included) and
// returns 5 x J doubles per image — the 10 MB of maps need not cross PCIe for single-person decoding.
void Net::decode_pose(double scale, double* out, bool is_device, void* user_stream) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "decode_pose() in CPU mode");
  auto ip = blob_index.find("prob"), il = blob_index.find("loc_pred");
  if (ip == blob_index.end() || il == blob_index.end()) throw DcError(DC_EINVAL, "net has no 'prob' / 'loc_pred' blobs");
  Storage& P = *blobs[ip->second]->st;
  Storage& L = *blobs[il->second]->st;
  if (P.head == UNINITIALIZED || L.head == UNINITIALIZED) throw DcError(DC_EINVAL, "decode_pose: run forward() first");
  if (L.dim(1) != 2 * P.dim(1) || L.dim(2) != P.dim(2) || L.dim(3) != P.dim(3) || L.dim(0) != P.dim(0))
    throw DcError(DC_ESHAPE, "decode_pose: loc_pred must have 2 channels per joint and the score map's size");
  ensure_device();
  auto img = [&](Storage& s, const void*& ptr, int& cp, int& c0) {
    if (s.view_of >= 0) {
      ptr = storages[s.view_of]->dev;
      cp = storages[s.view_of]->cp();
      c0 = s.view_c0;
    } else {
      if (s.head == HEAD_AT_CPU || s.head == UNINITIALIZED) sync_to_device(s);
      ptr = s.dev;
      cp = s.cp();
      c0 = 0;
    }
  };
  const void *pp, *lp;
  int pcp, pc0, lcp, lc0;
  const int pes = P.view_of >= 0 ? storages[P.view_of]->esize : P.esize;
  img(P, pp, pcp, pc0);
  img(L, lp, lcp, lc0);
  const int NB = P.dim(0), J = P.dim(1), H = P.dim(2), W = P.dim(3);
  void* s = user_stream ? user_stream : stream;
  const size_t cnt = (size_t)NB * 5 * J;
  if (is_device) {
    KCHECK(launch_pose_decode(pp, pcp, pc0, lp, lcp, lc0, pes, NB, H, W, J, scale, out, s));
    if (!user_stream) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
    return;
  }
  if (cnt > pose_cap) {
    dev_free(pose_dev);
    pose_dev = nullptr;
    dev_alloc((void**)&pose_dev, cnt * sizeof(double));
    pose_cap = cnt;
  }
  KCHECK(launch_pose_decode(pp, pcp, pc0, lp, lcp, lc0, pes, NB, H, W, J, scale, pose_dev, s));
  HIPCHECK(hipMemcpyAsync(out, pose_dev, cnt * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)s));
  HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

Net::MapRef Net::map_ref(const char* blob_name) {
  auto it = blob_index.find(blob_name);
  if (it == blob_index.end()) throw DcError(DC_EINVAL, std::string("net has no '") + blob_name + "' blob");
  Storage& s = *blobs[it->second]->st;
  if (s.head == UNINITIALIZED) throw DcError(DC_EINVAL, std::string("'") + blob_name + "': run forward() first");
  if (s.shape.size() != 4) throw DcError(DC_ESHAPE, std::string("'") + blob_name + "' is not a 4-D map");
  MapRef r{};
  if (s.view_of >= 0) {
    Storage& b = *storages[s.view_of];
    r.ptr = b.dev, r.cp = b.cp(), r.c0 = s.view_c0, r.es = b.esize;
  } else {
    if (s.head == HEAD_AT_CPU) sync_to_device(s);
    r.ptr = s.dev, r.cp = s.cp(), r.c0 = 0, r.es = s.esize;
  }
  r.NB = s.dim(0), r.C = s.dim(1), r.H = s.dim(2), r.W = s.dim(3);
  return r;
}

void* Net::scratch(size_t bytes) {
  if (bytes > scratch_cap_) {
    dev_free(scratch_dev_);
    scratch_dev_ = nullptr;
    dev_alloc((void**)&scratch_dev_, bytes);
    scratch_cap_ = bytes;
  }
  return scratch_dev_;
}

// Part candidates: non-maximum suppression of every score map + location refinement, on the device.
void Net::detect_parts(double scale, float thr, int radius, int max_det, int* counts, double* dets) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "detect_parts() in CPU mode");
  if (!(scale > 0) || !(thr >= 0.f) || radius < 0 || radius > 64 || max_det < 1 || max_det > 4096)
    throw DcError(DC_EINVAL, "detect_parts: scale > 0, threshold >= 0, 0 <= radius <= 64, 1 <= max_det <= 4096");
  ensure_device();
  const MapRef P = map_ref("prob"), L = map_ref("loc_pred");
  if (L.C != 2 * P.C || L.H != P.H || L.W != P.W || L.NB != P.NB || L.es != P.es)
    throw DcError(DC_ESHAPE, "detect_parts: loc_pred must have 2 channels per joint and the score map's size");
  const int lists = P.NB * P.C;
  const size_t cnt_b = ((size_t)lists * sizeof(int) + 255) / 256 * 256;
  const size_t spill_b = (size_t)lists * P.H * P.W * sizeof(unsigned long long);  // every cell may be a local maximum
  const size_t out_b = (size_t)lists * max_det * 5 * sizeof(double);
  unsigned char* base = (unsigned char*)scratch(cnt_b + spill_b + out_b);
  int* cnt = (int*)base;
  unsigned long long* spill = (unsigned long long*)(base + cnt_b);
  double* out = (double*)(base + cnt_b + spill_b);
  KCHECK(launch_part_select(P.ptr, P.cp, P.c0, L.ptr, L.cp, L.c0, P.es, P.NB, P.H, P.W, P.C, thr, radius, scale, max_det, spill, cnt, out,
                            stream));
  HIPCHECK(hipMemcpyAsync(counts, cnt, (size_t)lists * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHECK(hipMemcpyAsync(dets, out, out_b, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
}

// Pairwise regression of the next joint from a set of detections (cells), on the device.
void Net::decode_pairwise(double scale, int ndet, const int* det, const double* mean, const double* stdev, double* out) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "decode_pairwise() in CPU mode");
  if (!(scale > 0) || ndet < 0) throw DcError(DC_EINVAL, "decode_pairwise: scale > 0, ndet >= 0");
  if (ndet == 0) return;
  ensure_device();
  const MapRef N = map_ref("next_pred");
  if (N.C % 2) throw DcError(DC_ESHAPE, "decode_pairwise: next_pred must have 2 channels per regression edge");
  const int E = N.C / 2;
  const size_t det_b = ((size_t)ndet * 3 * sizeof(int) + 255) / 256 * 256, st_b = (size_t)E * 2 * sizeof(double);
  const size_t out_b = (size_t)ndet * E * 2 * sizeof(double);
  unsigned char* base = (unsigned char*)scratch(det_b + 2 * st_b + out_b);
  int* ddet = (int*)base;
  double* dmean = (double*)(base + det_b);
  double* dstd = dmean + (size_t)E * 2;
  double* dout = (double*)(base + det_b + 2 * st_b);
  HIPCHECK(hipMemcpyAsync(ddet, det, (size_t)ndet * 3 * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
  if (mean) HIPCHECK(hipMemcpyAsync(dmean, mean, st_b, hipMemcpyHostToDevice, (hipStream_t)stream));
  if (stdev) HIPCHECK(hipMemcpyAsync(dstd, stdev, st_b, hipMemcpyHostToDevice, (hipStream_t)stream));
  KCHECK(launch_pairwise_decode(N.ptr, N.cp, N.c0, N.es, N.NB, N.H, N.W, E, scale, ndet, ddet, mean ? dmean : nullptr,
                                stdev ? dstd : nullptr, dout, stream));
  HIPCHECK(hipMemcpyAsync(out, dout, out_b, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
}

std::string Net::plan_text() {
  ensure_plan();
  std::ostringstream os;
  os << "# plan for input";
  for (int d : plan_input_shape) os << " " << d;
  os << ": " << plan.size() << " launches, " << plan_flops / 1e9 << " GFLOP algorithmic, fuse=" << fuse
     << (dtype == 1 ? ", dtype=f16" : ", dtype=f32") << "\n";
  for (size_t i = 0; i < plan.size(); ++i) {
    const Launch& l = plan[i];
    os << i << "\t" << l.kernel << "\t";
    if (l.kind == Launch::CONV)
      os << "M=" << l.cg.M << " N=" << l.cg.Cout << " K=" << l.cg.Ktot << " taps=" << l.cg.nty * l.cg.ntx
         << (l.cg.ncls > 1 ? " classes=" + std::to_string(l.cg.ncls) : std::string()) << " grid=" << l.grid
         << (l.in2 >= 0 ? " +resid" : "") << (l.relu ? " +relu" : "") << (l.cg.sigmoid_ch ? " +sigmoid" : "");
    os << "\t" << l.label << "\n";
  }
  return os.str();
}

std::string Net::profile_text(int iters) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "profile in CPU mode");
  if (!plan_valid) throw DcError(DC_EINVAL, "profile_text: run forward() first");
  ensure_device();
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  std::ostringstream os;
  os << "idx\tkernel\tus\tGFLOP\tTFLOP/s\tgrid\tlabel\n";
  double total_us = 0;
  for (size_t i = 0; i < plan.size(); ++i) {
    const Launch& l = plan[i];
    run_launch(l, stream);  // warm
    HIPCHECK(hipEventRecord(e0, (hipStream_t)stream));
    for (int k = 0; k < iters; ++k) run_launch(l, stream);
    HIPCHECK(hipEventRecord(e1, (hipStream_t)stream));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1000.0 / iters;
    total_us += us;
    char buf[512];
    std::snprintf(buf, sizeof buf, "%zu\t%s\t%.2f\t%.3f\t%.2f\t%ld\t%s\n", i, l.kernel.c_str(), us, l.flops / 1e9,
                  us > 0 ? l.flops / us / 1e6 : 0.0, l.grid, l.label.c_str());
    os << buf;
  }
  os << "# sum of per-launch times: " << total_us << " us\n";
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return os.str();
}


// Net::ForwardDebugInfo / InputDebugInfo (net.cpp:648-681 of the reference, `debug_info: true`): the mean absolute value of
// every top blob and every parameter blob, in the reference's own line format, from the blobs of the LAST forward — what
// somebody bisecting a mismatch against a Caffe debug_info log needs.  Differences that follow from the lowering: an
// in-place layer chain (conv -> BatchNorm -> Scale -> ReLU on one blob) runs as one kernel, so the blob is only ever
// seen after the LAST layer of the chain: the line is printed for that layer, the earlier in-place layers of the chain get
// a `(folded into ...)` note; a blob swallowed by residual / head fusion (DC_OPT_FUSE >= 1) is reported as elided — run with
// DC_OPT_FUSE 0 to materialise all 220 Caffe-visible blobs.
std::string Net::debug_info_text() {
  if (!plan_valid) throw DcError(DC_EINVAL, "debug_info: run forward() first");
  std::ostringstream os;
  char buf[384];
  auto mean_abs = [&](Storage& st) -> double {
    sync_to_host(st);
    const float* h = st.host_ptr();
    const size_t n = st.count();
    double a = 0;
    for (size_t i = 0; i < n; ++i) a += std::fabs((double)h[i]);
    return n ? a / (double)n : 0.0;
  };
  for (int b : inputs) {
    std::snprintf(buf, sizeof buf, "    [Forward] Input %s data: %g\n", blobs[b]->name.c_str(), mean_abs(*blobs[b]->st));
    os << buf;
  }
  // the last layer (in file order) that writes each blob: the only moment its value exists here
  std::vector<int> last_writer(blobs.size(), -1);
  for (size_t i = 0; i < layers.size(); ++i)
    for (int t : layers[i].tops) last_writer[t] = (int)i;
  for (size_t i = 0; i < layers.size(); ++i) {
    const LayerRec& L = layers[i];
    for (int t : L.tops) {
      Storage& st = *blobs[t]->st;
      if (last_writer[t] != (int)i) {
        std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, top blob %s data: (in place: folded into layer %s)\n", L.name.c_str(),
                      blobs[t]->name.c_str(), layers[last_writer[t]].name.c_str());
      } else if (st.elided) {
        std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, top blob %s data: (elided by fusion; DC_OPT_FUSE 0 materialises it)\n",
                      L.name.c_str(), blobs[t]->name.c_str());
      } else {
        std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, top blob %s data: %g\n", L.name.c_str(), blobs[t]->name.c_str(), mean_abs(st));
      }
      os << buf;
    }
    for (size_t k = 0; k < L.params.size(); ++k) {
      Storage& st = *L.params[k]->st;
      const float* h = st.host_ptr();
      const size_t n = st.count();
      double a = 0;
      for (size_t q = 0; q < n; ++q) a += std::fabs((double)h[q]);
      // Net::AppendParam (net.cpp:469-482): the ParamSpec's name when it has one, else the index
      const auto specs = L.def.subs("param");
      const std::string pname = k < specs.size() && !specs[k]->str("name").empty() ? specs[k]->str("name") : std::to_string(k);
      std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, param blob %s data: %g\n", L.name.c_str(), pname.c_str(), n ? a / (double)n : 0.0);
      os << buf;
    }
  }
  return os.str();
}

// ---- NetGroup: the same model over several tensors as ONE launch sequence (net.h) ---------------------------------------
// Candidate streams for the lanes of groups (per device, process-wide, never destroyed: a destroyed stream would hand its
// hardware-queue slot to the next one created).  WHICH of them a group's lanes run on is decided by measurement
// (NetGroup::choose_lane_streams): a HIP process has a handful of hardware queues, the runtime binds a stream to one of them at
// creation, and whether two streams really run side by side cannot be asked.
static std::mutex g_lane_mu;
static std::map<void*, int> g_lane_users;  // candidate stream -> groups whose lanes run on it (two groups in flight must not share one)
static std::vector<void*>& lane_stream_candidates(int device, size_t want) {
  static std::map<int, std::vector<void*>> pool;
  std::lock_guard<std::mutex> lk(g_lane_mu);
  std::vector<void*>& p = pool[device];
  while (p.size() < want) {
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    p.push_back(st);
  }
  return p;
}
static void lane_streams_release(const std::vector<void*>& side) {
  std::lock_guard<std::mutex> lk(g_lane_mu);
  for (void* st : side)
    if (st && g_lane_users[st] > 0) --g_lane_users[st];
}

NetGroup* NetGroup::create(const std::vector<Net*>& members) {
  if (members.empty()) throw DcError(DC_EINVAL, "a group needs at least one net");
  for (Net* n : members) {
    if (!n) throw DcError(DC_EINVAL, "null net in group");
    if (n->shared != members[0]->shared)
      throw DcError(DC_EINVAL, "the members of a group must be executors of ONE model: a net and its clones (dc_net_clone)");
    if (n->dtype != members[0]->dtype || n->fuse != members[0]->fuse)
      throw DcError(DC_EINVAL, "the members of a group must agree on DC_OPT_DTYPE and DC_OPT_FUSE");
    if (n->inputs.size() != 1) throw DcError(DC_EINVAL, "group members must be single-input nets");
  }
  for (size_t i = 0; i < members.size(); ++i)
    for (size_t j = i + 1; j < members.size(); ++j)
      if (members[i] == members[j]) throw DcError(DC_EINVAL, "the same net twice in a group (every member needs its own activations: clone it)");
  std::unique_ptr<NetGroup> g(new NetGroup());
  g->nets = members;
  return g.release();
}

void* NetGroup::stream() { return nets[0]->stream; }

void NetGroup::drop_plan(GroupPlan& gp) {
  // nothing enqueued may still replay the graphs.  The device-wide wait covers the members' streams, the lanes' and the caller's
  // without touching a member: a group may be destroyed AFTER its nets (a garbage collector finalises a cycle in any order)
  {
    RuntimeLock rl;  // (no capture of ours is open while the device is waited for)
    (void)hipDeviceSynchronize();
  }
  gp.drop_graphs();
}

NetGroup::~NetGroup() {
  for (auto& gp : plans_) drop_plan(*gp);
  for (void* e : lane_events_) (void)hipEventDestroy((hipEvent_t)e);
  if (fork_event_) (void)hipEventDestroy((hipEvent_t)fork_event_);
  for (auto& kv : lane_choice_) lane_streams_release(kv.second);  // (the streams themselves belong to the process-wide list)
}

void GroupPlan::drop_graphs() {
  for (void*& g : lane_graphs)
    if (g) (void)hipGraphExecDestroy((hipGraphExec_t)g), g = nullptr;
}

void NetGroup::set_lanes(int n) {
  if (n < 0) throw DcError(DC_EINVAL, "lanes must be 0 (automatic) or positive");
  if (n == lanes_opt_) return;
  lanes_opt_ = n;
  for (auto& kv : lane_choice_) lane_streams_release(kv.second);
  lane_choice_.clear();
  for (auto& gp : plans_) drop_plan(*gp);  // every merged plan was cut for the old lane count
  plans_.clear();
  cur_ = nullptr;
}

// The merged plan of the members' CURRENT shapes (every member has been through begin_batch: its plan is active, its
// buffers allocated, its filter images uploaded, its own tiles chosen).
GroupPlan& NetGroup::ensure_plan() {
  std::vector<std::vector<int>> shapes;
  for (Net* n : nets) shapes.push_back(n->plan_input_shape);
  GroupPlan* hit = nullptr;
  for (auto& gp : plans_)
    if (gp->shapes == shapes) hit = gp.get();
  if (hit) {
    if (plan_current(*hit)) {
      hit->last_use = ++use_clock_;
      ++stats.plan_hits;
      return *hit;
    }
    drop_plan(*hit);  // a member re-lowered (weights, options), reallocated a buffer or changed a tile: merge again (choices are cached)
    hit->launches.clear();
    hit->tuned = false;
    try {
      merge(*hit);
    } catch (...) {  // a half-merged plan must not be found again
      forget_plan(hit);
      throw;
    }
    hit->last_use = ++use_clock_;
    return *hit;
  }
  static const size_t cap = (size_t)std::max(1, env_int("DC_GROUP_PLAN_CACHE", 8));
  while (plans_.size() >= cap) {
    size_t lru = 0;
    for (size_t i = 1; i < plans_.size(); ++i)
      if (plans_[i]->last_use < plans_[lru]->last_use) lru = i;
    drop_plan(*plans_[lru]);
    forget_plan(plans_[lru].get());
  }
  plans_.emplace_back(new GroupPlan());
  GroupPlan& gp = *plans_.back();
  gp.shapes = shapes;
  try {
    merge(gp);
  } catch (...) {
    forget_plan(&gp);
    throw;
  }
  gp.last_use = ++use_clock_;
  return gp;
}

// does the merged plan still describe its members (their lowering, buffers, filter images, tiles)?
bool NetGroup::plan_current(const GroupPlan& gp) const {
  for (size_t c = 0; c < nets.size(); ++c)
    if (gp.shapes[c] != nets[c]->plan_input_shape || gp.lowerings[c] != (uint64_t)nets[c]->stats.lowerings || gp.buf_gens[c] != nets[c]->buf_gen_ ||
        gp.weight_gens[c] != nets[c]->seen_weights_gen || gp.tile_gens[c] != nets[c]->tile_gen_)
      return false;
  return true;
}

// the plan of the last forward, for the calls that launch from it outside a forward: a member that has been reshaped, re-lowered or
// re-tiled on its own since then has moved the buffers the prepared launches point at
GroupPlan& NetGroup::current_plan() {
  if (!cur_) throw DcError(DC_EINVAL, "group: run a forward first");
  if (!plan_current(*cur_)) throw DcError(DC_EINVAL, "group: a member changed (shape, weights, tiles) since the group's last forward: run a forward first");
  return *cur_;
}

void NetGroup::forget_plan(GroupPlan* gp) {
  if (cur_ == gp) cur_ = nullptr;
  for (size_t i = 0; i < plans_.size(); ++i)
    if (plans_[i].get() == gp) {
      plans_.erase(plans_.begin() + (long)i);
      return;
    }
}

void NetGroup::merge(GroupPlan& gp) {
  const size_t NM = nets.size();
  gp.lowerings.resize(NM), gp.buf_gens.resize(NM), gp.weight_gens.resize(NM), gp.tile_gens.resize(NM);
  for (size_t c = 0; c < NM; ++c) {
    gp.tile_gens[c] = nets[c]->tile_gen_;
    gp.lowerings[c] = (uint64_t)nets[c]->stats.lowerings;
    gp.buf_gens[c] = nets[c]->buf_gen_;
    gp.weight_gens[c] = nets[c]->seen_weights_gen;
  }
  ++stats.merges;
  gp.flops = 0;
  for (Net* n : nets) gp.flops += n->plan_flops;
  const size_t NL = nets[0]->plan.size();
  for (Net* n : nets)
    if (n->plan.size() != NL) throw DcError(DC_EINVAL, "group: the members' plans differ in length (different fusion options or graphs?)");
  const bool grouping = env_int("DC_GROUP", 1) != 0;  // 0: every launch member by member (A/B of the merge itself)
  // LANES.  The members are dealt to `nlanes` lanes — snake order over their sizes: largest with smallest — and every lane is
  // merged on its own and runs on a stream of its own, concurrently with the others: the launches of one lane fill the
  // dispatch ramps and the tails of the other's (a grouped 4-scale float16 pyramid batch: 12.06 ms as one lane, 10.61 ms as two
  // lanes of two scales — what two independent groups in flight reach, for ONE request), at the price of fetching a layer's
  // filters once per lane.  Default (measured on float16 batch-8 members, tools/group_profile.py --scales): TWO members run as two
  // lanes — nothing merged, plain concurrency: 8.26 against 9.27 ms merged (544x736 + 680x920), 5.44 against 6.10 (408x552 + 544x736) —,
  // THREE as one lane (one merged launch per layer: 11.15 against 12.06 ms for the lop-sided {A, C} | {B}), FOUR or more as two lanes
  // of merged members.  dc_group_set_lanes / DC_GROUP_LANES override.
  int nl = lanes_opt_ > 0 ? lanes_opt_ : env_int("DC_GROUP_LANES", 0);
  if (nl <= 0) nl = NM == 3 ? 1 : 2;
  nl = std::max(1, std::min<int>(nl, (int)NM));
  gp.nlanes = nl;
  gp.lane_members.assign(nl, {});
  {
    std::vector<size_t> order(NM);
    for (size_t c = 0; c < NM; ++c) order[c] = c;
    auto rows = [&](size_t c) {
      long r = 1;
      for (int d : nets[c]->plan_input_shape) r *= d;
      return r;
    };
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return rows(x) > rows(y); });
    for (size_t j = 0; j < NM; ++j) {
      const size_t r = j / nl, c = j % nl;
      gp.lane_members[r % 2 == 0 ? c : nl - 1 - c].push_back((int)order[j]);
    }
    for (auto& lm : gp.lane_members) std::sort(lm.begin(), lm.end());
  }
  for (int lane = 0; lane < nl; ++lane) {
  const std::vector<int>& mem = gp.lane_members[lane];
  const size_t NMl = mem.size();
  for (size_t i = 0; i < NL; ++i) {
    const Launch& l0 = nets[mem[0]]->plan[i];
    // (a lane with a single member runs that member's own launches: a one-problem multi-problem launch is the same work behind a
    //  longer prologue — measured 6 % slower at float16 batch 8)
    bool mergeable = grouping && l0.kind == Launch::CONV && NMl >= 2;
    for (size_t cc = 0; cc < NMl && mergeable; ++cc) {
      const size_t c = (size_t)mem[cc];
      const Launch& l = nets[c]->plan[i];
      const ConvGemmParams &g = l.cg, &g0 = l0.cg;
      if (l.kind != Launch::CONV || l.variant == kWinoVariant || l.w != l0.w || l.scale != l0.scale || l.shift != l0.shift || l.c_off != l0.c_off ||
          l.w_off != l0.w_off ||
          (l.in2 >= 0) != (l0.in2 >= 0) || g.esize != g0.esize || g.klen != g0.klen || g.sy != g0.sy || g.sx != g0.sx || g.Cout != g0.Cout ||
          g.relu != g0.relu || g.sigmoid_ch != g0.sigmoid_ch)
        mergeable = false;
    }
    if (mergeable) {
      // a multi-problem tile must exist for this K granularity
      bool have = false;
      for (int v = 0; v < conv_num_variants(); ++v)
        if (conv_variant_multiproblem(v) && l0.cg.klen % conv_variant_bk(v) == 0 && conv_variant_esize(v) == l0.cg.esize) have = true;
      mergeable = have;
    }
    if (!mergeable) {
      for (size_t cc = 0; cc < NMl; ++cc) {
        const size_t c = (size_t)mem[cc];
        if (nets[c]->plan[i].kind != l0.kind) throw DcError(DC_EINVAL, "group: the members' plans differ at launch " + std::to_string(i));
        GroupLaunch gl;
        gl.lane = lane;
        gl.multi = false;
        gl.index = (int)i;
        gl.member = (int)c;
        gl.label = nets[c]->plan[i].label;
        gp.launches.push_back(std::move(gl));
      }
      continue;
    }
    // the problems: per member, its single problem or its deconvolution classes; heaviest K first, then the most pixels
    struct Rec {
      ConvProblem q;
      int member;
      std::string key;
    };
    std::vector<Rec> recs;
    std::string keys;
    for (size_t cc = 0; cc < NMl; ++cc) {
      const size_t c = (size_t)mem[cc];
      Net& n = *nets[c];
      const Launch& l = n.plan[i];
      const ConvGemmParams& g = l.cg;
      Storage& X = *n.storages[l.in];
      Storage& Y = *n.storages[l.out];
      const int nc = g.ncls > 1 ? g.ncls : 1;
      for (int k = 0; k < nc; ++k) {
        ConvProblem q{};
        const long yo = g.ncls > 1 ? l.y_off + g.cls[k].y_off : l.y_off;
        q.x = X.dev;
        q.y = Y.dev_at(yo);
        q.resid = l.in2 >= 0 ? n.storages[l.in2]->dev_at(yo) : nullptr;
        q.x_img_stride = g.x_img_stride, q.y_img_stride = g.y_img_stride;
        q.x_row_stride = g.x_row_stride, q.x_rows = g.x_rows, q.x_rowlen = g.x_rowlen;
        q.y_row_stride = g.y_row_stride, q.y_pix_stride = g.y_pix_stride;
        q.NB = g.NB;
        if (g.ncls > 1) {
          const ConvClass& cl = g.cls[k];
          q.w_off = cl.w_off;
          q.nty = cl.nty, q.ntx = cl.ntx, q.dy0 = cl.dy0, q.ddy = cl.ddy, q.x0 = cl.x0, q.ddx = cl.ddx, q.Ktot = cl.Ktot;
          q.OH = cl.OH, q.OW = cl.OW, q.M = cl.M;
        } else {
          q.w_off = l.w_off;
          q.nty = g.nty, q.ntx = g.ntx, q.dy0 = g.dy0, q.ddy = g.ddy, q.x0 = g.x0, q.ddx = g.ddx, q.Ktot = g.Ktot;
          q.OH = g.OH, q.OW = g.OW, q.M = g.M;
        }
        recs.push_back({q, (int)c, std::string()});
      }
      keys += (cc ? "|" : "") + n.tune_key(l);
    }
    // order of the problems = order in which every XCD walks them.  Default: tensor after tensor (the residue classes of ONE
    // member's deconvolution next to each other: they read the same 2048-deep input rows through different taps, which the
    // memory-side cache then still holds), biggest tensor first, inside a tensor the heaviest class first.
    // DC_GROUP_ORDER=1: heaviest K first across all members (class-major).
    if (env_int("DC_GROUP_ORDER", 0) == 1)
      std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.q.Ktot != b.q.Ktot ? a.q.Ktot > b.q.Ktot : a.q.M > b.q.M; });
    else {
      std::vector<long> msum(NM, 0);
      for (auto& r : recs) msum[r.member] += r.q.M;
      std::stable_sort(recs.begin(), recs.end(), [&](const Rec& a, const Rec& b) {
        if (a.member != b.member) return msum[a.member] != msum[b.member] ? msum[a.member] > msum[b.member] : a.member < b.member;
        return a.q.Ktot > b.q.Ktot;
      });
    }
    for (size_t r0 = 0, part = 0; r0 < recs.size(); r0 += kMaxProblems, ++part) {
      GroupLaunch gl;
      gl.lane = lane;
      gl.multi = true;
      gl.index = (int)i;
      gl.nprob = (int)std::min<size_t>(kMaxProblems, recs.size() - r0);
      gl.p = l0.cg;  // the layer's common fields: esize, klen, sy, sx, Cout, relu, sigmoid_ch
      gl.p.ncls = 0;
      gl.p.dbg = nullptr;
      gl.p.x = nullptr, gl.p.y = nullptr, gl.p.resid = nullptr;
      gl.p.w = l0.w->dev;
      gl.p.scale = l0.scale ? l0.scale->dev + l0.c_off : nullptr;
      gl.p.shift = l0.shift ? l0.shift->dev + l0.c_off : nullptr;
      for (int k = 0; k < gl.nprob; ++k) {
        gl.table.prob[k] = recs[r0 + k].q;
        gl.prob_member.push_back(recs[r0 + k].member);
      }
      {
        double fl = 0;
        for (int c : mem) fl += nets[c]->plan[i].flops;
        gl.flops = fl * gl.nprob / (double)recs.size();
      }
      gl.key = "G" + std::to_string(gl.nprob) + (recs.size() > (size_t)kMaxProblems ? "p" + std::to_string(part) : "") + ":" + keys;
      gl.label = l0.label + " x" + std::to_string(NMl) + (gl.nprob != (int)NMl ? " [" + std::to_string(gl.nprob) + " problems]" : "");
      gp.launches.push_back(std::move(gl));
    }
  }
  }  // lane
  // tile of every merged launch: the shared choice table, else (until the group is timed) the widest member's own tile
  {
    std::lock_guard<std::mutex> lk(nets[0]->shared->mu);
    for (auto& gl : gp.launches) {
      if (!gl.multi) continue;
      auto it = nets[0]->shared->tune_cache.find(gl.key);
      int v = it != nets[0]->shared->tune_cache.end() ? it->second : -1;
      const int forced = env_int("DC_CONV_VARIANT", -1);
      if (forced >= 0 && forced < conv_num_variants() && conv_variant_multiproblem(forced) && gl.p.klen % conv_variant_bk(forced) == 0 &&
          conv_variant_esize(forced) == gl.p.esize)
        v = forced;
      gl.variant = v;
    }
  }
  for (auto& gl : gp.launches) {
    if (!gl.multi) continue;
    int v = gl.variant;
    auto usable = [&](int cand) {
      if (cand < 0 || cand >= conv_num_variants() || cand == kWinoVariant) return false;
      ConvGemmParams p = gl.p;
      ConvMultiTable t = gl.table;
      return prepare_conv_multi(p, t, gl.nprob, cand) > 0;
    };
    if (!usable(v)) {
      v = -1;
      size_t big = (size_t)gp.lane_members[gl.lane][0];  // the lane's member with the most pixels
      for (int c : gp.lane_members[gl.lane])
        if (nets[c]->plan[gl.index].cg.M > nets[big]->plan[gl.index].cg.M) big = (size_t)c;
      if (usable(nets[big]->plan[gl.index].variant)) v = nets[big]->plan[gl.index].variant;
      for (int cand = 0; v < 0 && cand < conv_num_variants(); ++cand)
        if (usable(cand)) v = cand;
      if (v < 0) throw DcError(DC_EUNSUP, "group launch '" + gl.label + "': no multi-problem tile takes it");
    }
    apply_variant(gp, gl, v);
  }
}

// prepare the launch (common block + problem table = its kernel arguments) for a tile
void NetGroup::apply_variant(GroupPlan&, GroupLaunch& gl, int variant) {
  ConvMultiArgs a;
  a.p = gl.p;
  a.t = gl.table;
  const long grid = prepare_conv_multi(a.p, a.t, gl.nprob, variant);
  if (grid <= 0) throw DcError(DC_EUNSUP, "group launch '" + gl.label + "': tile " + conv_variant(variant).name + " cannot take it");
  gl.args = a;
  gl.variant = variant;
  gl.grid = grid;
}

// Tile of every merged launch by measurement, once per distinct signature (shared with every group of the model through
// the model's choice table, persisted with DC_TUNE_CACHE like the single-problem choices).
void NetGroup::autotune(GroupPlan& gp) {
  gp.tuned = true;
  if (env_int("DC_AUTOTUNE", 1) == 0 || env_int("DC_CONV_VARIANT", -1) >= 0) return;
  Net& n0 = *nets[0];
  std::lock_guard<std::mutex> lk(n0.shared->mu);
  std::map<std::string, int>& cache = n0.shared->tune_cache;
  bool timed_any = false;
  std::set<std::string> fresh;  // the signatures timed by THIS call: a choice that is in the table stays (other plans run it)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  void* s = stream();
  for (auto& gl : gp.launches) {
    if (!gl.multi || cache.count(gl.key)) continue;
    timed_any = true;
    fresh.insert(gl.key);
    std::vector<std::pair<float, int>> c;
    for (int v = 0; v < conv_num_variants(); ++v) {
      if (!conv_variant_multiproblem(v) || gl.p.klen % conv_variant_bk(v) != 0 || conv_variant_esize(v) != gl.p.esize) continue;
      ConvMultiArgs p;
      p.p = gl.p;
      p.t = gl.table;
      const long grid = prepare_conv_multi(p.p, p.t, gl.nprob, v);
      if (grid <= 0) continue;
      KCHECK(launch_conv_multi(p, v, grid, s));  // warm
      float ms = 1e30f;
      for (int t2 = 0; t2 < 2; ++t2) {
        HIPCHECK(hipEventRecord(e0, (hipStream_t)s));
        for (int r = 0; r < 3; ++r) KCHECK(launch_conv_multi(p, v, grid, s));
        HIPCHECK(hipEventRecord(e1, (hipStream_t)s));
        HIPCHECK(hipEventSynchronize(e1));
        float m2 = 0;
        HIPCHECK(hipEventElapsedTime(&m2, e0, e1));
        ms = std::min(ms, m2);
      }
      c.push_back({ms, v});
    }
    std::sort(c.begin(), c.end());
    if (!c.empty()) {
      cache[gl.key] = c.front().second;
      for (auto& tm : c) tm.first *= 5.f / 3.f;  // the report prints "ms of a 5-launch burst"
      n0.shared->tune_timings[gl.key] = c;
    }
  }
  // (2) in situ, as Net::autotune does: the candidates within 15 % of a signature's best (at most 4) once more inside whole
  // passes over the GROUP plan (events around every launch of the signature, best of 3 passes per candidate).  Timed alone a
  // launch re-reads warm filters and meets an idle chip; in the sequence it follows another kernel's tail — on the two-pyramid
  // group the isolated pass took a 128x128 tile for the 256->1024+shortcut layers that is 15 % slower there than the 64x128 one.
  if (timed_any && env_int("DC_TUNE_INSITU", 1) != 0) {
    std::map<std::string, std::vector<int>> shortlist;
    size_t rounds = 0;
    for (auto& gl : gp.launches) {
      if (!gl.multi || shortlist.count(gl.key) || !fresh.count(gl.key)) continue;
      auto t = n0.shared->tune_timings.find(gl.key);
      if (t == n0.shared->tune_timings.end()) continue;
      std::vector<int> sl;
      for (auto& c : t->second)
        if (sl.size() < 4 && c.first <= t->second.front().first * 1.15f) sl.push_back(c.second);
      if (sl.size() >= 2) rounds = std::max(rounds, sl.size()), shortlist[gl.key] = sl;
    }
    if (rounds) {
      std::vector<size_t> idx;
      for (size_t i = 0; i < gp.launches.size(); ++i)
        if (gp.launches[i].multi && shortlist.count(gp.launches[i].key)) idx.push_back(i);
      std::vector<hipEvent_t> ev(2 * idx.size(), nullptr);
      struct EvList {
        std::vector<hipEvent_t>& ev;
        ~EvList() {
          for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        }
      } ev_list{ev};
      for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
      std::map<std::string, std::vector<float>> best;
      for (auto& kv : shortlist) best[kv.first].assign(kv.second.size(), 1e30f);
      for (size_t r = 0; r < rounds; ++r) {
        for (size_t i : idx) {
          const std::vector<int>& sl = shortlist[gp.launches[i].key];
          const int v = sl[std::min(r, sl.size() - 1)];
          if (gp.launches[i].variant != v) apply_variant(gp, gp.launches[i], v);
        }
        for (int pass = 0; pass < 3; ++pass) {
          size_t j = 0;
          for (size_t i = 0; i < gp.launches.size(); ++i) {
            const GroupLaunch& gl = gp.launches[i];
            const bool watched = j < idx.size() && idx[j] == i;
            if (watched) HIPCHECK(hipEventRecord(ev[2 * j], (hipStream_t)s));
            if (gl.multi) KCHECK(launch_conv_multi(gl.args, gl.variant, gl.grid, s));
            else nets[gl.member]->run_launch(nets[gl.member]->plan[gl.index], s);
            if (watched) {
              HIPCHECK(hipEventRecord(ev[2 * j + 1], (hipStream_t)s));
              ++j;
            }
          }
          HIPCHECK(hipStreamSynchronize((hipStream_t)s));
          std::map<std::string, float> sum;
          for (size_t q = 0; q < idx.size(); ++q) {
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ev[2 * q], ev[2 * q + 1]));
            sum[gp.launches[idx[q]].key] += ms;
          }
          for (auto& kv : sum) {
            const size_t e = std::min(r, shortlist[kv.first].size() - 1);
            best[kv.first][e] = std::min(best[kv.first][e], kv.second);
          }
        }
      }
      for (auto& kv : best) {
        size_t arg = 0;
        for (size_t e = 1; e < kv.second.size(); ++e)
          if (kv.second[e] < kv.second[arg]) arg = e;
        cache[kv.first] = shortlist[kv.first][arg];
      }
    }
  }
  for (auto& gl : gp.launches) {
    if (!gl.multi) continue;
    auto it = cache.find(gl.key);
    if (it != cache.end() && it->second != gl.variant) apply_variant(gp, gl, it->second);
  }
  if (timed_any) {
    ++stats.autotune_runs;
    write_tune_cache_locked(*n0.shared);
  }
  gp.drop_graphs();
}

void NetGroup::run(GroupPlan& gp, int lane, void* s) {
  for (auto& gl : gp.launches) {
    if (lane >= 0 && gl.lane != lane) continue;
    if (gl.multi) {
      const int rc = launch_conv_multi(gl.args, gl.variant, gl.grid, s);
      if (rc != 0) throw DcError(DC_EDEVICE, "group launch '" + gl.label + "' failed: " + hipGetErrorString((hipError_t)rc));
    } else {
      Net& n = *nets[gl.member];
      n.run_launch(n.plan[gl.index], s);
    }
  }
}

void NetGroup::enqueue(void* s) {
  GroupPlan& gp = ensure_plan();
  cur_ = &gp;
  if (!gp.tuned) {
    HIPCHECK(hipStreamSynchronize((hipStream_t)s));  // the inputs are in place; the timing launches run on the group's own stream
    autotune(gp);
  }
  bool use_graph = true;
  for (Net* n : nets) use_graph = use_graph && n->use_graph;
  const int nl = gp.nlanes;
  if (use_graph && (int)gp.lane_graphs.size() != nl) gp.lane_graphs.assign(nl, nullptr);
  // With more than one lane, lane 0 runs on the caller's stream and every other lane on a stream of the device's candidate list,
  // forked from and joined back into the caller's stream by events — WHICH candidate is measured (choose_lane_streams).
  if (nl > 1) {
    while ((int)lane_events_.size() < nl) {
      hipEvent_t ev;
      HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      lane_events_.push_back(ev);
    }
    if (!fork_event_) {
      hipEvent_t ev;
      HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      fork_event_ = ev;
    }
  }
  // (graphs first: the measurement below replays them)
  if (use_graph)
    for (int lane = 0; lane < nl; ++lane) {
      if (gp.lane_graphs[lane]) continue;
      gp.lane_graphs[lane] = capture_graph(stream(), [&](void* cs) { run(gp, lane, cs); });
      ++stats.graph_instantiations;
    }
  if (nl > 1) {
    auto it = lane_choice_.find(s);
    if (it == lane_choice_.end() || (int)it->second.size() != nl) {
      if (lane_choice_.size() >= 8) {  // a caller that keeps changing streams: start over rather than grow
        for (auto& kv : lane_choice_) lane_streams_release(kv.second);
        lane_choice_.clear();
      }
      choose_lane_streams(gp, s, use_graph);
      it = lane_choice_.find(s);
    }
    launch_lanes(gp, s, it->second, use_graph);
  } else {
    launch_lanes(gp, s, {}, use_graph);
  }
  for (Net* n : nets) {
    for (auto& l : n->plan) n->storages[l.out]->head = HEAD_AT_GPU;
    for (int v : n->plan_views_) n->storages[v]->head = HEAD_AT_GPU;
  }
}

// one grouped forward: lane 0 on s, lane k on side[k] (side[0] unused), fork / join by events
void NetGroup::launch_lanes(GroupPlan& gp, void* s, const std::vector<void*>& side, bool use_graph) {
  const int nl = gp.nlanes;
  if (nl > 1) HIPCHECK(hipEventRecord((hipEvent_t)fork_event_, (hipStream_t)s));
  for (int lane = 0; lane < nl; ++lane) {
    void* ls = lane == 0 ? s : side[lane];
    if (lane > 0) HIPCHECK(hipStreamWaitEvent((hipStream_t)ls, (hipEvent_t)fork_event_, 0));
    if (use_graph) HIPCHECK(hipGraphLaunch((hipGraphExec_t)gp.lane_graphs[lane], (hipStream_t)ls));
    else run(gp, lane, ls);
    if (lane > 0) {
      HIPCHECK(hipEventRecord((hipEvent_t)lane_events_[lane], (hipStream_t)ls));
      HIPCHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)lane_events_[lane], 0));
    }
  }
}

// Which streams do the lanes beyond the first run on?  Measured, per caller stream: the forward itself is timed (warm run + one
// timed run, events on s) with the side lanes on successive candidates, and the assignment with the shortest forward stays.  A
// side stream that shares a hardware queue with the caller's stream — or with another side lane's — runs its lane AFTER the other
// one (a grouped float16 pyramid batch: 12.0 instead of 10.7 ms); which candidate does depends on everything the process created
// before, so nothing but a measurement on the real launch sequence tells.  Costs a dozen forwards, once per (group, caller stream).
void NetGroup::choose_lane_streams(GroupPlan& gp, void* s, bool use_graph) {
  const int nl = gp.nlanes;
  const size_t ncand = 6;
  std::vector<void*>& cand = lane_stream_candidates(nets[0]->device, ncand + (size_t)nl);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  HIPCHECK(hipStreamSynchronize((hipStream_t)s));
  float best = 1e30f, best_free = 1e30f;
  std::vector<void*> best_set, best_free_set;
  for (size_t first = 0; first + (size_t)(nl - 1) <= cand.size(); ++first) {
    std::vector<void*> side(nl, nullptr);
    for (int k = 1; k < nl; ++k) side[k] = cand[first + (size_t)k - 1];
    launch_lanes(gp, s, side, use_graph);  // warm
    HIPCHECK(hipEventRecord(e0, (hipStream_t)s));
    launch_lanes(gp, s, side, use_graph);
    HIPCHECK(hipEventRecord(e1, (hipStream_t)s));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms, best_set = side;
    bool free_ = true;
    {
      std::lock_guard<std::mutex> lk(g_lane_mu);
      for (int k = 1; k < nl; ++k) free_ = free_ && g_lane_users[side[k]] == 0;
    }
    if (free_ && ms < best_free) best_free = ms, best_free_set = side;
  }
  // a stream no other group's lanes run on, if one is (nearly) as good: two groups in flight whose side lanes share ONE stream
  // run those lanes one after the other (measured alone, both would pick the same winner)
  if (!best_free_set.empty() && best_free <= best * 1.04f) best_set = best_free_set;
  if (best_set.empty()) throw DcError(DC_EDEVICE, "group: no stream could be created for the lanes beyond the first (dc_group_set_lanes(g, 1) runs one lane)");
  {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    for (int k = 1; k < nl; ++k) ++g_lane_users[best_set[k]];
  }
  auto old = lane_choice_.find(s);
  if (old != lane_choice_.end()) lane_streams_release(old->second);
  lane_choice_[s] = best_set;
}

void NetGroup::forward_batch(const float* const* inputs, const int* n, const int* h, const int* w, bool is_device, float* const* prob,
                             float* const* loc, float* const* next, void* user_stream) {
  if (!inputs || !n || !h || !w) throw DcError(DC_EINVAL, "group forward: null argument");
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  std::vector<Storage*> ins;
  for (size_t c = 0; c < nets.size(); ++c) ins.push_back(&nets[c]->begin_batch(n[c], h[c], w[c]));
  void* s = user_stream ? user_stream : stream();
  for (size_t c = 0; c < nets.size(); ++c) {
    Storage& in = *ins[c];
    const int C = in.dim(1);
    if (is_device) {
      KCHECK(launch_nchw_to_nhwc(inputs[c], in.dev, in.esize, n[c], C, h[c], w[c], in.cp(), s));
    } else {
      in.ensure_stage(in.count());
      HIPCHECK(hipMemcpyAsync(in.stage, inputs[c], in.count() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)s));
      KCHECK(launch_nchw_to_nhwc(in.stage, in.dev, in.esize, n[c], C, h[c], w[c], in.cp(), s));
    }
    in.head = HEAD_AT_GPU;
  }
  enqueue(s);
  for (size_t c = 0; c < nets.size(); ++c)
    nets[c]->emit_maps(prob ? prob[c] : nullptr, loc ? loc[c] : nullptr, next ? next[c] : nullptr, is_device, s);
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void NetGroup::forward_images(const unsigned char* const* bgr, const int* n, const int* h, const int* w, const double* scale, bool is_device,
                              float* const* prob, float* const* loc, float* const* next, double* const* pose, void* user_stream) {
  if (!bgr || !n || !h || !w || !scale) throw DcError(DC_EINVAL, "group forward_images: null argument");
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "forward_images() in CPU mode: libdeepcut_hip provides the MI355X path only");
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  nets[0]->ensure_device();
  void* s = user_stream ? user_stream : stream();
  for (size_t c = 0; c < nets.size(); ++c) nets[c]->prep_images(bgr[c], n[c], h[c], w[c], scale[c], is_device, s);
  enqueue(s);
  for (size_t c = 0; c < nets.size(); ++c) {
    nets[c]->emit_maps(prob ? prob[c] : nullptr, loc ? loc[c] : nullptr, next ? next[c] : nullptr, is_device, s);
    // decode on the group's stream; a host destination synchronises inside (the maps are complete there: same stream)
    if (pose && pose[c]) nets[c]->decode_pose(scale[c], pose[c], is_device, s);
  }
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

int NetGroup::num_launches() { return cur_ ? (int)cur_->launches.size() : 0; }
int NetGroup::num_multi_launches() {
  int m = 0;
  if (cur_)
    for (auto& gl : cur_->launches) m += gl.multi ? 1 : 0;
  return m;
}
double NetGroup::flops() { return cur_ ? cur_->flops : 0.0; }

// same line formats as Net::plan_text / Net::profile_text (tools/breakdown.py aggregates both)
std::string NetGroup::plan_text() {
  if (!cur_) throw DcError(DC_EINVAL, "group: run a forward first");
  std::ostringstream os;
  os << "# group of " << nets.size() << " executors in " << cur_->nlanes << " lane" << (cur_->nlanes > 1 ? "s" : "") << ": " << cur_->launches.size() << " launches (" << num_multi_launches() << " multi-problem), "
     << cur_->flops / 1e9 << " GFLOP algorithmic" << (nets[0]->dtype == 1 ? ", dtype=f16" : ", dtype=f32") << "\n";
  for (size_t i = 0; i < cur_->launches.size(); ++i) {
    const GroupLaunch& gl = cur_->launches[i];
    os << i << "\t";
    if (gl.multi) {
      long M = 0;
      int kmax = 0;
      for (int k = 0; k < gl.nprob; ++k) M += gl.table.prob[k].M, kmax = std::max(kmax, gl.table.prob[k].Ktot);
      os << "conv_gemm_mp<" << conv_variant(gl.variant).name << ">\tM=" << M << " N=" << gl.p.Cout << " K=" << kmax << " problems=" << gl.nprob
         << " grid=" << gl.grid << (cur_->nlanes > 1 ? " lane=" + std::to_string(gl.lane) : std::string()) << (gl.table.prob[0].resid ? " +resid" : "")
         << (gl.p.relu ? " +relu" : "") << (gl.p.sigmoid_ch ? " +sigmoid" : "");
    } else {
      os << nets[gl.member]->plan[gl.index].kernel << "\tmember " << gl.member;
    }
    os << "\t" << gl.label << "\n";
  }
  return os.str();
}

std::string NetGroup::tune_report_text() {
  if (!cur_) throw DcError(DC_EINVAL, "group: run a forward first");
  std::lock_guard<std::mutex> lk(nets[0]->shared->mu);
  std::vector<std::string> order;
  std::map<std::string, std::pair<int, int>> seen;
  for (auto& gl : cur_->launches) {
    if (!gl.multi) continue;
    auto it = seen.find(gl.key);
    if (it == seen.end()) order.push_back(gl.key), seen[gl.key] = {gl.variant, 1};
    else ++it->second.second;
  }
  std::string out;
  for (auto& k : order) {
    out += k + "\t" + conv_variant(seen[k].first).name + "\t" + std::to_string(seen[k].second) + "\t";
    auto t = nets[0]->shared->tune_timings.find(k);
    if (t != nets[0]->shared->tune_timings.end())
      for (size_t i = 0; i < t->second.size(); ++i) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "%s%s:%.2f", i ? " " : "", conv_variant(t->second[i].second).name, t->second[i].first * 1000.f / 5.f);
        out += buf;
      }
    out += "\n";
  }
  return out;
}

void NetGroup::set_tile(const std::string& key, const std::string& tile) {
  current_plan();
  int v = -1;
  for (int i = 0; v < 0 && i < conv_num_variants(); ++i)
    if (tile == conv_variant(i).name) v = i;
  if (v < 0) throw DcError(DC_EINVAL, "no tile variant named '" + tile + "'");
  bool any = false;
  for (auto& gl : cur_->launches) {
    if (!gl.multi || gl.key != key) continue;
    ConvMultiArgs a;
    a.p = gl.p;
    a.t = gl.table;
    if (!conv_variant_multiproblem(v) || prepare_conv_multi(a.p, a.t, gl.nprob, v) <= 0)
      throw DcError(DC_EUNSUP, "tile '" + tile + "' cannot take group launch '" + gl.label + "'");
    any = true;
  }
  if (!any) throw DcError(DC_EINVAL, "the group's current plan has no launch with signature '" + key + "'");
  // nothing of this plan may be in flight while its launches change (the forward may have run on a caller's stream and on lane
  // streams: the device-wide wait covers them all)
  {
    RuntimeLock rl;
    (void)hipDeviceSynchronize();
  }
  for (auto& gl : cur_->launches)
    if (gl.multi && gl.key == key) apply_variant(*cur_, gl, v);
  {
    std::lock_guard<std::mutex> lk(nets[0]->shared->mu);
    auto it = nets[0]->shared->tune_cache.find(key);
    if (it == nets[0]->shared->tune_cache.end() || it->second != v) {
      nets[0]->shared->tune_cache[key] = v;
      write_tune_cache_locked(*nets[0]->shared);
    }
  }
  cur_->drop_graphs();
}

std::string NetGroup::profile_text(int iters) {
  GroupPlan& gp = current_plan();
  void* s = stream();
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  std::ostringstream os;
  os << "idx\tkernel\tus\tGFLOP\tTFLOP/s\tgrid\tlabel\n";
  double total_us = 0;
  auto one = [&](const GroupLaunch& gl) {
    if (gl.multi) KCHECK(launch_conv_multi(gl.args, gl.variant, gl.grid, s));
    else nets[gl.member]->run_launch(nets[gl.member]->plan[gl.index], s);
  };
  for (size_t i = 0; i < gp.launches.size(); ++i) {
    const GroupLaunch& gl = gp.launches[i];
    one(gl);
    HIPCHECK(hipEventRecord(e0, (hipStream_t)s));
    for (int k = 0; k < iters; ++k) one(gl);
    HIPCHECK(hipEventRecord(e1, (hipStream_t)s));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    total_us += us;
    const double fl = gl.multi ? gl.flops : nets[gl.member]->plan[gl.index].flops;
    const std::string kn = gl.multi ? std::string("conv_gemm_mp<") + conv_variant(gl.variant).name + ">" : nets[gl.member]->plan[gl.index].kernel;
    char buf[640];
    std::snprintf(buf, sizeof buf, "%zu\t%s\t%.2f\t%.3f\t%.2f\t%ld\t%s\n", i, kn.c_str(), us, fl / 1e9, us > 0 ? fl / us / 1e6 : 0.0,
                  gl.multi ? gl.grid : nets[gl.member]->plan[gl.index].grid, gl.label.c_str());
    os << buf;
  }
  os << "# sum of per-launch times: " << total_us << " us\n";
  return os.str();
}

}  // namespace dc
