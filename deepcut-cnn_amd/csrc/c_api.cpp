// c_api.cpp — the extern "C" boundary declared in include/deepcut_hip.h.  No C++ exception and no
// C++ type crosses it.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/deepcut_hip.h"
#include "net.h"

using namespace dc;

// dc_net* / dc_blob* are dc::Net* / dc::NetBlob* behind opaque C names

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
template <typename F>
int guard(F&& f) {
  try {
    f();
    return DC_OK;
  } catch (const DcError& e) {
    return fail(e.code, e.what());
  } catch (const std::bad_alloc&) {
    return fail(DC_EDEVICE, "out of host memory");
  } catch (const std::exception& e) {
    return fail(DC_EINVAL, e.what());
  }
}
// stand-alone blobs (dc_blob_create) move on the default stream of the calling thread's device (Caffe::SetDevice)
void standalone_device() {
  if (device_count() <= 0) throw DcError(DC_EDEVICE, "no HIP device visible");
  if (hipSetDevice(Context::get().device) != hipSuccess) throw DcError(DC_EDEVICE, "hipSetDevice failed");
}
inline Net* N(dc_net* n) { return reinterpret_cast<Net*>(n); }
inline NetBlob* B(dc_blob* b) { return reinterpret_cast<NetBlob*>(b); }
#define REQUIRE(p)                                                   \
  if (!(p)) return fail(DC_EINVAL, "null argument: " #p)
}  // namespace

extern "C" {

const char* dc_last_error(void) { return g_err.c_str(); }
const char* dc_version(void) { return "deepcut_hip 0.1 (gfx950)"; }

int dc_set_mode(int mode) {
  if (mode != DC_MODE_CPU && mode != DC_MODE_GPU) return fail(DC_EINVAL, "mode must be DC_MODE_CPU or DC_MODE_GPU");
  Context::get().mode = mode;
  return DC_OK;
}
int dc_get_mode(void) { return Context::get().mode; }
int dc_set_device(int id) {
  int n = device_count();
  if (id < 0 || (n > 0 && id >= n))
    return fail(DC_EDEVICE, "device " + std::to_string(id) + " out of range (" + std::to_string(n) + " visible)");
  Context::get().device = id;
  if (n > 0 && hipSetDevice(id) != hipSuccess) return fail(DC_EDEVICE, "hipSetDevice failed");
  return DC_OK;
}
int dc_get_device(void) { return Context::get().device; }
int dc_device_count(void) { return device_count(); }

int dc_net_create_from_text(const char* text, const char* model, int phase, dc_net** out) {
  REQUIRE(text);
  REQUIRE(out);
  *out = nullptr;
  return guard([&] {
    std::unique_ptr<Net> n(Net::create(text, phase));
    if (model && *model) n->copy_from(model);
    *out = reinterpret_cast<dc_net*>(n.release());
  });
}
int dc_net_create(const char* proto, const char* model, int phase, dc_net** out) {
  REQUIRE(proto);
  REQUIRE(out);
  *out = nullptr;
  return guard([&] {
    std::string text = read_file(proto);
    if (model && *model) {  // CheckFile (_caffe.cpp:45-52): fail before building the net
      FILE* f = std::fopen(model, "rb");
      if (!f) throw DcError(DC_EIO, std::string("Could not open file ") + model);
      std::fclose(f);
    }
    std::unique_ptr<Net> n(Net::create(text, phase));
    if (model && *model) n->copy_from(model);
    *out = reinterpret_cast<dc_net*>(n.release());
  });
}
int dc_net_clone(dc_net* net, dc_net** out) {
  REQUIRE(net);
  REQUIRE(out);
  *out = nullptr;
  return guard([&] { *out = reinterpret_cast<dc_net*>(N(net)->clone()); });
}
int dc_net_forward_host_async(dc_net* net, const float* input, int n, int h, int w, float* prob, float* loc, float* next) {
  REQUIRE(net);
  REQUIRE(input);
  if (n <= 0 || h <= 0 || w <= 0) return fail(DC_EINVAL, "forward_host_async: n, h, w must be positive");
  return guard([&] { N(net)->forward_batch(input, n, h, w, false, prob, loc, next, nullptr, true); });
}
int dc_host_alloc(size_t bytes, void** out) {
  REQUIRE(out);
  *out = nullptr;
  if (bytes == 0) return fail(DC_EINVAL, "dc_host_alloc: zero bytes");
  return guard([&] {
    standalone_device();
    *out = host_alloc_pinned(bytes);
  });
}
int dc_host_free(void* p) {
  if (!p) return DC_OK;
  return guard([&] { host_free_pinned(p); });
}
int dc_nets_choose_streams(dc_net* const* nets, int n, int candidates, int reps, double* rate_chosen, double* rate_first) {
  REQUIRE(nets);
  if (n < 1 || n > 64) return fail(DC_EINVAL, "dc_nets_choose_streams: 1..64 executors");
  std::vector<Net*> v;
  for (int i = 0; i < n; ++i) {
    if (!nets[i]) return fail(DC_EINVAL, "dc_nets_choose_streams: null net");
    for (Net* o : v)
      if (o == N(nets[i])) return fail(DC_EINVAL, "dc_nets_choose_streams: the same net twice");
    v.push_back(N(nets[i]));
  }
  return guard([&] { Net::choose_streams(v, candidates, reps <= 0 ? 3 : reps, rate_chosen, rate_first); });
}
int dc_net_stream(dc_net* net, void** out) {
  REQUIRE(net);
  REQUIRE(out);
  *out = nullptr;
  return guard([&] {
    if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "dc_net_stream in CPU mode: libdeepcut_hip provides the MI355X path only");
    *out = N(net)->own_stream();
  });
}
int dc_net_busy(dc_net* net, int* busy) {
  REQUIRE(net);
  REQUIRE(busy);
  *busy = 0;
  if (!N(net)->stream) return DC_OK;
  hipError_t e = hipStreamQuery((hipStream_t)N(net)->stream);
  if (e == hipErrorNotReady) {
    *busy = 1;
    return DC_OK;
  }
  if (e != hipSuccess) return fail(DC_EDEVICE, std::string("hipStreamQuery failed: ") + hipGetErrorString(e));
  return DC_OK;
}
int dc_net_synchronize(dc_net* net) {
  REQUIRE(net);
  return guard([&] { N(net)->synchronize(); });
}
int dc_net_destroy(dc_net* net) {
  if (net) delete N(net);
  return DC_OK;
}
int dc_net_set_option(dc_net* net, int key, int value) {
  REQUIRE(net);
  return guard([&] {
    Net* n = N(net);
    if (key == DC_OPT_FUSE) {
      if (n->fuse != value) n->invalidate_plans();
      n->fuse = value;
    } else if (key == DC_OPT_HIPGRAPH) {
      n->use_graph = value;
    } else if (key == DC_OPT_DTYPE) {
      n->set_dtype(value);
    } else if (key == DC_OPT_OUTPUTS) {
      n->set_outputs_mask(value);
    } else {
      throw DcError(DC_EINVAL, "unknown option " + std::to_string(key));
    }
  });
}
int dc_net_get_option(dc_net* net, int key, int* value) {
  REQUIRE(net);
  REQUIRE(value);
  Net* n = N(net);
  if (key == DC_OPT_FUSE) *value = n->fuse;
  else if (key == DC_OPT_HIPGRAPH) *value = n->use_graph;
  else if (key == DC_OPT_DTYPE) *value = n->dtype;
  else if (key == DC_OPT_OUTPUTS) *value = n->outputs_mask;
  else return fail(DC_EINVAL, "unknown option " + std::to_string(key));
  return DC_OK;
}
int dc_net_copy_from(dc_net* net, const char* path) {
  REQUIRE(net);
  REQUIRE(path);
  return guard([&] { N(net)->copy_from(path); });
}
int dc_net_save(dc_net* net, const char* path) {
  REQUIRE(net);
  REQUIRE(path);
  return guard([&] { N(net)->save(path); });
}
const char* dc_net_name(dc_net* net) { return net ? N(net)->name.c_str() : ""; }

int dc_net_num_layers(dc_net* net) { return net ? (int)N(net)->layers.size() : 0; }
const char* dc_net_layer_name(dc_net* net, int i) {
  if (!net || i < 0 || i >= (int)N(net)->layers.size()) return nullptr;
  return N(net)->layers[i].name.c_str();
}
const char* dc_net_layer_type(dc_net* net, int i) {
  if (!net || i < 0 || i >= (int)N(net)->layers.size()) return nullptr;
  return N(net)->layers[i].type.c_str();
}
int dc_net_num_blobs(dc_net* net) { return net ? (int)N(net)->blobs.size() : 0; }
const char* dc_net_blob_name(dc_net* net, int i) {
  if (!net || i < 0 || i >= (int)N(net)->blobs.size()) return nullptr;
  return N(net)->blobs[i]->name.c_str();
}
int dc_net_blob(dc_net* net, const char* name, dc_blob** out) {
  REQUIRE(net);
  REQUIRE(name);
  REQUIRE(out);
  auto it = N(net)->blob_index.find(name);
  if (it == N(net)->blob_index.end()) return fail(DC_EINVAL, std::string("Unknown blob name ") + name);
  *out = reinterpret_cast<dc_blob*>(N(net)->blobs[it->second].get());
  return DC_OK;
}
int dc_net_num_inputs(dc_net* net) { return net ? (int)N(net)->inputs.size() : 0; }
const char* dc_net_input_name(dc_net* net, int i) {
  if (!net || i < 0 || i >= (int)N(net)->inputs.size()) return nullptr;
  return N(net)->blobs[N(net)->inputs[i]]->name.c_str();
}
int dc_net_num_outputs(dc_net* net) { return net ? (int)N(net)->outputs.size() : 0; }
const char* dc_net_output_name(dc_net* net, int i) {
  if (!net || i < 0 || i >= (int)N(net)->outputs.size()) return nullptr;
  return N(net)->blobs[N(net)->outputs[i]]->name.c_str();
}
int dc_net_layer_num_params(dc_net* net, const char* layer) {
  if (!net || !layer) return 0;
  int li = N(net)->layer_index(layer);
  return li < 0 ? 0 : (int)N(net)->layers[li].params.size();
}
int dc_net_param(dc_net* net, const char* layer, int idx, dc_blob** out) {
  REQUIRE(net);
  REQUIRE(layer);
  REQUIRE(out);
  int li = N(net)->layer_index(layer);
  if (li < 0) return fail(DC_EINVAL, std::string("Unknown layer name ") + layer);
  auto& ps = N(net)->layers[li].params;
  if (idx < 0 || idx >= (int)ps.size()) return fail(DC_EINVAL, "param index out of range");
  *out = reinterpret_cast<dc_blob*>(ps[idx].get());
  return DC_OK;
}

int dc_net_reshape(dc_net* net) {
  REQUIRE(net);
  return guard([&] { N(net)->reshape(); });
}
int dc_net_forward(dc_net* net, int start, int end, float* loss) {
  REQUIRE(net);
  if (loss) *loss = 0.f;
  int nl = (int)N(net)->layers.size();
  if (start < 0 || end >= nl || start > end + 1)
    return fail(DC_EINVAL, "forward range out of bounds (net.cpp:566-567)");
  return guard([&] { N(net)->forward(start, end); });
}
int dc_net_forward_all(dc_net* net) {
  REQUIRE(net);
  return guard([&] { N(net)->forward(0, (int)N(net)->layers.size() - 1); });
}

int dc_blob_num_axes(dc_blob* b) { return b ? (int)B(b)->st->shape.size() : 0; }
int dc_blob_shape(dc_blob* b, int* ndim, int* dims) {
  REQUIRE(b);
  REQUIRE(ndim);
  REQUIRE(dims);
  auto& s = B(b)->st->shape;
  if (s.size() > 8) return fail(DC_EINVAL, "too many axes");
  *ndim = (int)s.size();
  for (size_t i = 0; i < s.size(); ++i) dims[i] = s[i];
  return DC_OK;
}
int dc_blob_count(dc_blob* b) { return b ? (int)B(b)->st->count() : 0; }
int dc_blob_reshape(dc_blob* b, int ndim, const int* dims) {
  REQUIRE(b);
  if (ndim < 0 || ndim > 8) return fail(DC_EINVAL, "bad number of axes");
  if (ndim > 0) REQUIRE(dims);  // a 0-axis (scalar) shape has no dimension list, as in dc_blob_create
  return guard([&] {
    Storage& s = *B(b)->st;
    if (s.is_param) throw DcError(DC_EINVAL, "parameter blobs cannot be reshaped");
    s.reshape(ndim > 0 ? std::vector<int>(dims, dims + ndim) : std::vector<int>());
  });
}
static int blob_host(dc_blob* b, float** out, bool mut) {
  return guard([&] {
    Storage& s = *B(b)->st;
    if (s.elided && !s.is_param)
      throw DcError(DC_EUNSUP, "blob '" + B(b)->name + "' is folded into a fused kernel in the current plan (or feeds only outputs that "
                                "DC_OPT_OUTPUTS leaves out) and never materialised; create the net with DC_OPT_FUSE 0 / all outputs to observe it");
    if (s.head == HEAD_AT_GPU) {
      if (s.owner) s.owner->sync_to_host(s);
      else standalone_device(), storage_to_host(s, nullptr, nullptr);
    }
    *out = s.host_ptr();
    s.host_touched = true;
    if (s.head == UNINITIALIZED) s.head = HEAD_AT_CPU;
    if (mut) {
      s.head = HEAD_AT_CPU;
      if (s.is_param && s.shared) {  // a writable view was handed out: its content is re-checked before the next run
        std::lock_guard<std::mutex> lk(s.shared->mu);
        if (!s.touch_listed) s.shared->touched.push_back(B(b)->st), s.touch_listed = true;
      }
    }
  });
}
int dc_blob_cpu_data(dc_blob* b, const float** out) {
  REQUIRE(b);
  REQUIRE(out);
  float* p = nullptr;
  int r = blob_host(b, &p, false);
  *out = p;
  return r;
}
int dc_blob_mutable_cpu_data(dc_blob* b, float** out) {
  REQUIRE(b);
  REQUIRE(out);
  return blob_host(b, out, true);
}
int dc_blob_head(dc_blob* b) { return b ? B(b)->st->head : 0; }
int dc_blob_gpu_data(dc_blob* b, const void** dev, int* pitch) {
  REQUIRE(b);
  REQUIRE(dev);
  return guard([&] {
    Storage& s = *B(b)->st;
    if (s.is_param) throw DcError(DC_EUNSUP, "parameters are packed per kernel on the device; no NHWC image exists");
    if (s.view_of < 0 && (s.head == HEAD_AT_CPU || s.head == UNINITIALIZED)) {
      if (s.owner) s.owner->sync_to_device(s);
      else standalone_device(), storage_to_device(s, nullptr);
    }
    if (s.view_of >= 0) {  // channel slice of a concatenated head tensor
      Storage& base = *s.owner->storages[s.view_of];
      *dev = base.dev_at(s.view_c0);
      if (pitch) *pitch = base.cp();
    } else {
      *dev = s.dev;
      if (pitch) *pitch = s.cp();
    }
  });
}

int dc_blob_create(int ndim, const int* dims, dc_blob** out) {
  REQUIRE(out);
  *out = nullptr;
  if (ndim < 0 || ndim > 8 || (ndim > 0 && !dims)) return fail(DC_EINVAL, "bad number of axes");
  return guard([&] {
    std::unique_ptr<NetBlob> b(new NetBlob());
    b->standalone = true;
    b->st = std::make_shared<Storage>();
    b->st->reshape(std::vector<int>(dims, dims + ndim));
    *out = reinterpret_cast<dc_blob*>(b.release());
  });
}
int dc_blob_destroy(dc_blob* b) {
  if (!b) return DC_OK;
  if (!B(b)->standalone) return fail(DC_EINVAL, "this blob belongs to a net");
  delete B(b);
  return DC_OK;
}
int dc_blob_mutable_gpu_data(dc_blob* b, void** dev, int* pitch) {
  REQUIRE(b);
  REQUIRE(dev);
  return guard([&] {
    Storage& s = *B(b)->st;
    if (s.is_param) throw DcError(DC_EUNSUP, "parameters are packed per kernel on the device; no NHWC image exists");
    if (s.view_of >= 0 || s.elided)
      throw DcError(DC_EUNSUP, "blob '" + B(b)->name + "' is fused into another tensor in the current plan; use DC_OPT_FUSE 0");
    if (s.owner) {
      s.owner->sync_to_device(s);
      s.head = HEAD_AT_GPU;
    } else {
      standalone_device();
      storage_mutable_device(s, nullptr);
    }
    *dev = s.dev;
    if (pitch) *pitch = s.cp();
  });
}
int dc_blob_copy_from(dc_blob* dst, dc_blob* src, int reshape) {
  REQUIRE(dst);
  REQUIRE(src);
  return guard([&] {
    Storage &d = *B(dst)->st, &s = *B(src)->st;
    if (s.elided && !s.is_param) throw DcError(DC_EUNSUP, "source blob '" + B(src)->name + "' is never materialised in the current plan");
    if (d.shape != s.shape) {  // Blob::CopyFrom (blob.cpp:435-443)
      if (!reshape) throw DcError(DC_ESHAPE, "Trying to copy blobs of different sizes.");
      if (d.is_param) throw DcError(DC_EINVAL, "parameter blobs cannot be reshaped");
      d.reshape(s.shape);
    }
    Net* own = d.owner ? d.owner : s.owner;
    void* stream = nullptr;
    if (s.head == HEAD_AT_GPU) {
      if (s.owner && s.owner != own) s.owner->synchronize();  // the source may still be being written on its own net's stream
      if (own) {
        own->synchronize();
        stream = own->stream;
        if (hipSetDevice(own->device >= 0 ? own->device : Context::get().device) != hipSuccess) throw DcError(DC_EDEVICE, "hipSetDevice failed");
      } else {
        standalone_device();
      }
    }
    storage_copy(d, s, s.view_of >= 0 && s.owner ? s.owner->storages[s.view_of].get() : nullptr, stream);
    if (d.is_param && d.shared) {
      std::lock_guard<std::mutex> lk(d.shared->mu);
      if (!d.touch_listed) d.shared->touched.push_back(B(dst)->st), d.touch_listed = true;
    }
  });
}
int dc_net_create_for_layer(const char* layer_text, int phase, int nbottom, dc_blob* const* bottoms, dc_net** out) {
  REQUIRE(layer_text);
  REQUIRE(out);
  *out = nullptr;
  if (nbottom < 0 || (nbottom > 0 && !bottoms)) return fail(DC_EINVAL, "bad bottom list");
  return guard([&] {
    std::vector<std::vector<int>> shapes;
    for (int i = 0; i < nbottom; ++i) {
      if (!bottoms[i]) throw DcError(DC_EINVAL, "null bottom blob");
      shapes.push_back(B(bottoms[i])->st->shape);
    }
    *out = reinterpret_cast<dc_net*>(Net::create_for_layer(layer_text, phase, shapes));
  });
}

int dc_net_forward_batch(dc_net* net, const float* input, int n, int h, int w, int is_device, float* prob,
                         float* loc_pred, float* next_pred, void* stream) {
  REQUIRE(net);
  REQUIRE(input);
  if (n <= 0 || h <= 0 || w <= 0) return fail(DC_EINVAL, "bad batch shape");
  return guard([&] { N(net)->forward_batch(input, n, h, w, is_device != 0, prob, loc_pred, next_pred, stream); });
}

int dc_net_forward_requests(dc_net* net, int n, const float* const* inputs, int h, int w, float* const* prob,
                            float* const* loc_pred, float* const* next_pred, void* stream) {
  REQUIRE(net);
  REQUIRE(inputs);
  if (n <= 0 || h <= 0 || w <= 0) return fail(DC_EINVAL, "bad batch shape");
  for (int i = 0; i < n; ++i)
    if (!inputs[i]) return fail(DC_EINVAL, "null request input");
  return guard([&] { N(net)->forward_requests(n, inputs, h, w, prob, loc_pred, next_pred, stream); });
}

int dc_net_decode_pose(dc_net* net, double scale, double* pose, int is_device, void* stream) {
  REQUIRE(net);
  REQUIRE(pose);
  if (!(scale > 0)) return fail(DC_EINVAL, "scale must be positive");
  return guard([&] { N(net)->decode_pose(scale, pose, is_device != 0, stream); });
}

int dc_net_emit_maps(dc_net* net, void* prob, void* loc_pred, void* next_pred, int elem, int is_device, void* stream) {
  REQUIRE(net);
  return guard([&] { N(net)->emit_last_maps(prob, loc_pred, next_pred, elem, is_device != 0, stream); });
}

int dc_net_forward_images(dc_net* net, const unsigned char* images, int n, int height, int width, double scale,
                          int is_device, float* prob, float* loc_pred, float* next_pred, double* pose, void* stream) {
  REQUIRE(net);
  REQUIRE(images);
  if (n <= 0 || height <= 0 || width <= 0) return fail(DC_EINVAL, "n, height and width must be positive");
  if (!(scale > 0)) return fail(DC_EINVAL, "scale must be positive");
  return guard([&] {
    N(net)->forward_images(images, n, height, width, scale, is_device != 0, prob, loc_pred, next_pred, pose, stream);
  });
}

int dc_image_canvas_size(int height, int width, double scale, int* canvas_h, int* canvas_w) {
  REQUIRE(canvas_h);
  REQUIRE(canvas_w);
  if (height <= 0 || width <= 0 || !(scale > 0)) return fail(DC_EINVAL, "height, width and scale must be positive");
  int nh, nw;
  dc::image_canvas_size(height, width, scale, *canvas_h, *canvas_w, nh, nw);
  return DC_OK;
}

int dc_net_detect_parts(dc_net* net, double scale, float threshold, int radius, int max_det, int* counts, double* dets) {
  REQUIRE(net);
  REQUIRE(counts);
  REQUIRE(dets);
  return guard([&] { N(net)->detect_parts(scale, threshold, radius, max_det, counts, dets); });
}

int dc_net_decode_pairwise(dc_net* net, double scale, int ndet, const int* detections, const double* mean,
                           const double* stdev, double* out) {
  REQUIRE(net);
  if (ndet > 0) {
    REQUIRE(detections);
    REQUIRE(out);
  }
  return guard([&] { N(net)->decode_pairwise(scale, ndet, detections, mean, stdev, out); });
}

int dc_net_flops(dc_net* net, double* flops) {
  REQUIRE(net);
  REQUIRE(flops);
  return guard([&] {
    Net* n = N(net);
    n->ensure_plan();
    *flops = n->plan_flops;
  });
}
int dc_net_num_launches(dc_net* net) {
  if (!net) return 0;
  int r = 0;
  guard([&] {
    N(net)->plan_text();
    r = (int)N(net)->plan.size();
  });
  return r;
}
const char* dc_net_plan_text(dc_net* net) {
  if (!net) return nullptr;
  Net* n = N(net);
  int rc = guard([&] { n->text_buf = n->plan_text(); });
  return rc == DC_OK ? n->text_buf.c_str() : nullptr;
}
int dc_net_stats(dc_net* net, long long* out, int n) {
  REQUIRE(net);
  REQUIRE(out);
  const NetStats& st = N(net)->stats;
  const long long v[DC_NUM_STATS] = {st.lowerings,      st.graph_instantiations, st.plan_hits, st.autotune_runs,
                                     st.buffer_growths, st.repacks,              (long long)N(net)->parked_.size() + (N(net)->plan_valid ? 1 : 0)};
  for (int i = 0; i < n && i < DC_NUM_STATS; ++i) out[i] = v[i];
  return DC_OK;
}
int dc_net_reserve(dc_net* net, int n, int h, int w) {
  REQUIRE(net);
  if (n <= 0 || h <= 0 || w <= 0) return fail(DC_EINVAL, "bad batch shape");
  return guard([&] { N(net)->reserve(n, h, w); });
}
int dc_net_device(dc_net* net) { return net ? N(net)->device : -1; }

const char* dc_net_profile_text(dc_net* net, int iters) {
  if (!net) return nullptr;
  Net* n = N(net);
  int rc = guard([&] { n->text_buf = n->profile_text(iters > 0 ? iters : 10); });
  return rc == DC_OK ? n->text_buf.c_str() : nullptr;
}

const char* dc_net_debug_info(dc_net* net) {
  if (!net) return nullptr;
  Net* n = N(net);
  int rc = guard([&] { n->text_buf = n->debug_info_text(); });
  return rc == DC_OK ? n->text_buf.c_str() : nullptr;
}

const char* dc_net_tune_report(dc_net* net) {
  if (!net) return nullptr;
  Net* n = N(net);
  int rc = guard([&] { n->text_buf = n->tune_report_text(); });
  return rc == DC_OK ? n->text_buf.c_str() : nullptr;
}
int dc_net_set_tile(dc_net* net, const char* signature, const char* tile) {
  REQUIRE(net);
  REQUIRE(signature);
  REQUIRE(tile);
  return guard([&] { N(net)->set_tile(signature, tile); });
}

// tile variants of the gather-GEMM (diagnostics / tests: DC_CONV_VARIANT takes an index into this table)
int dc_conv_variant_count(void) { return dc::conv_num_variants(); }
const char* dc_conv_variant_name(int i) { return i >= 0 && i < dc::conv_num_variants() ? dc::conv_variant(i).name : nullptr; }
int dc_conv_variant_esize(int i) { return i >= 0 && i < dc::conv_num_variants() ? dc::conv_variant_esize(i) : 0; }
int dc_wino_half_pack(const float* g, int cout, int cin, int rowscale, float* out, float* row_scale) {
  REQUIRE(g);
  REQUIRE(out);
  REQUIRE(row_scale);
  if (cout <= 0 || cin <= 0 || cout % 32 || cin % 16) return fail(DC_EINVAL, "dc_wino_half_pack: cout must be a multiple of 32, cin of 16");
  return guard([&] { dc::wino_half_pack_filters(g, cout, cin, rowscale != 0, out, row_scale); });
}

int dc_stream1x1_pack(const float* g, int cout, int k, float* out) {
  if (!g || !out) return fail(DC_EINVAL, "dc_stream1x1_pack: null pointer");
  if (cout <= 0 || k <= 0 || cout % 32 || k % 16) return fail(DC_EINVAL, "dc_stream1x1_pack: cout must be a multiple of 32, k of 16");
  return guard([&] { dc::stream1x1_pack_filters(g, cout, k, out); });
}

int dc_stream1x1f_pack(const float* g, int cout, int k, float* out) {
  if (!g || !out) return fail(DC_EINVAL, "dc_stream1x1f_pack: null pointer");
  if (cout <= 0 || k <= 0 || cout % 16 || k % 16) return fail(DC_EINVAL, "dc_stream1x1f_pack: cout and k must be multiples of 16");
  return guard([&] { dc::stream1x1f_pack_filters(g, cout, k, out); });
}

int dc_stem7x7_pack(const float* g, int c, float* out) {
  if (!g || !out) return fail(DC_EINVAL, "dc_stem7x7_pack: null pointer");
  if (c < 1 || c > 4) return fail(DC_EINVAL, "dc_stem7x7_pack: 1 to 4 input channels");
  return guard([&] { dc::stem7x7_pack_filters(g, c, out); });
}

// ---- pyramid-grouped execution ---------------------------------------------------------------------------------------
namespace {
inline NetGroup* G(dc_group* g) { return reinterpret_cast<NetGroup*>(g); }
}
int dc_group_create(dc_net* const* nets, int n, dc_group** out) {
  REQUIRE(nets);
  REQUIRE(out);
  *out = nullptr;
  if (n <= 0) return fail(DC_EINVAL, "a group needs at least one net");
  return guard([&] {
    std::vector<Net*> m;
    for (int i = 0; i < n; ++i) m.push_back(N(nets[i]));
    *out = reinterpret_cast<dc_group*>(NetGroup::create(m));
  });
}
int dc_group_destroy(dc_group* group) {
  delete G(group);
  return DC_OK;
}
int dc_group_size(dc_group* group) { return group ? (int)G(group)->nets.size() : 0; }
int dc_group_set_lanes(dc_group* group, int lanes) {
  REQUIRE(group);
  return guard([&] { G(group)->set_lanes(lanes); });
}
int dc_group_forward_batch(dc_group* group, const float* const* inputs, const int* n, const int* h, const int* w, int is_device,
                           float* const* prob, float* const* loc_pred, float* const* next_pred, void* stream) {
  REQUIRE(group);
  REQUIRE(inputs);
  REQUIRE(n);
  REQUIRE(h);
  REQUIRE(w);
  for (size_t c = 0; c < G(group)->nets.size(); ++c) {
    if (!inputs[c]) return fail(DC_EINVAL, "null input for group member " + std::to_string(c));
    if (n[c] <= 0 || h[c] <= 0 || w[c] <= 0) return fail(DC_EINVAL, "bad batch shape for group member " + std::to_string(c));
  }
  return guard([&] { G(group)->forward_batch(inputs, n, h, w, is_device != 0, prob, loc_pred, next_pred, stream); });
}
int dc_group_forward_images(dc_group* group, const unsigned char* const* images, const int* n, const int* height, const int* width,
                            const double* scale, int is_device, float* const* prob, float* const* loc_pred, float* const* next_pred,
                            double* const* pose, void* stream) {
  REQUIRE(group);
  REQUIRE(images);
  REQUIRE(n);
  REQUIRE(height);
  REQUIRE(width);
  REQUIRE(scale);
  for (size_t c = 0; c < G(group)->nets.size(); ++c) {
    if (!images[c]) return fail(DC_EINVAL, "null images for group member " + std::to_string(c));
    if (n[c] <= 0 || height[c] <= 0 || width[c] <= 0 || !(scale[c] > 0)) return fail(DC_EINVAL, "bad image shape / scale for group member " + std::to_string(c));
  }
  return guard([&] { G(group)->forward_images(images, n, height, width, scale, is_device != 0, prob, loc_pred, next_pred, pose, stream); });
}
int dc_comm_create(int nexec, const int* devices, int transport, dc_comm** out) {
  REQUIRE(out);
  *out = nullptr;
  return guard([&] { *out = reinterpret_cast<dc_comm*>(comm_create(nexec, devices, transport)); });
}
int dc_comm_destroy(dc_comm* comm) {
  if (!comm) return DC_OK;
  return guard([&] { comm_destroy(reinterpret_cast<Comm*>(comm)); });
}
int dc_comm_transport(dc_comm* comm) {
  if (!comm) return fail(DC_EINVAL, "null argument: comm");
  return comm_transport(reinterpret_cast<Comm*>(comm));
}
int dc_forward_batch(dc_comm* comm, dc_net* const* nets, int nexec, const float* const* inputs, const int (*hw)[2], int n,
                     float* const* prob, float* const* loc, float* const* next) {
  REQUIRE(comm);
  REQUIRE(nets);
  if (nexec < 1) return fail(DC_EINVAL, "dc_forward_batch: no executors");
  return guard([&] {
    std::vector<Net*> v;
    for (int k = 0; k < nexec; ++k) v.push_back(N(nets[k]));
    comm_forward(reinterpret_cast<Comm*>(comm), v.data(), nexec, inputs, hw, n, prob, loc, next);
  });
}
int dc_comm_item_executor(dc_comm* comm, int i) {
  if (!comm) return fail(DC_EINVAL, "null argument: comm");
  int r = -1;
  int rc = guard([&] { r = comm_item_executor(reinterpret_cast<Comm*>(comm), i); });
  return rc == DC_OK ? r : rc;
}
int dc_comm_root_maps(dc_comm* comm, int i, const void** prob, const void** loc, const void** next, int dims[5]) {
  REQUIRE(comm);
  return guard([&] { comm_root_maps(reinterpret_cast<Comm*>(comm), i, prob, loc, next, dims); });
}
int dc_lpt_schedule(const double* cost, int n, int nexec, int* exec_of_item) {
  if (n < 0 || nexec < 1) return fail(DC_EINVAL, "dc_lpt_schedule: n >= 0 items on nexec >= 1 executors");
  if (n == 0) return DC_OK;
  REQUIRE(cost);
  REQUIRE(exec_of_item);
  return guard([&] {
    const auto share = lpt_schedule(std::vector<double>(cost, cost + n), nexec);
    for (int k = 0; k < nexec; ++k)
      for (int i : share[(size_t)k]) exec_of_item[i] = k;
  });
}

const char* dc_group_plan_text(dc_group* group) {
  if (!group) return nullptr;
  NetGroup* g = G(group);
  int rc = guard([&] { g->text_buf = g->plan_text(); });
  return rc == DC_OK ? g->text_buf.c_str() : nullptr;
}
const char* dc_group_profile_text(dc_group* group, int iters) {
  if (!group) return nullptr;
  NetGroup* g = G(group);
  int rc = guard([&] { g->text_buf = g->profile_text(iters > 0 ? iters : 10); });
  return rc == DC_OK ? g->text_buf.c_str() : nullptr;
}
const char* dc_group_tune_report(dc_group* group) {
  if (!group) return nullptr;
  NetGroup* g = G(group);
  int rc = guard([&] { g->text_buf = g->tune_report_text(); });
  return rc == DC_OK ? g->text_buf.c_str() : nullptr;
}
int dc_group_set_tile(dc_group* group, const char* signature, const char* tile) {
  REQUIRE(group);
  REQUIRE(signature);
  REQUIRE(tile);
  return guard([&] { G(group)->set_tile(signature, tile); });
}
int dc_group_stats(dc_group* group, long long* out, int n) {
  REQUIRE(group);
  REQUIRE(out);
  NetGroup* g = G(group);
  const long long v[DC_NUM_GSTATS] = {g->stats.merges, g->stats.graph_instantiations, g->stats.autotune_runs, g->stats.plan_hits,
                                      (long long)g->num_launches(), (long long)g->num_multi_launches(), (long long)g->lanes()};
  for (int i = 0; i < n && i < DC_NUM_GSTATS; ++i) out[i] = v[i];
  return DC_OK;
}
int dc_group_flops(dc_group* group, double* out) {
  REQUIRE(group);
  REQUIRE(out);
  *out = G(group)->flops();
  return DC_OK;
}

}  // extern "C"
