// net_init.cpp — see net_internal.h: Net::Init semantics (InsertSplits, set-up, shapes) and weight files.
#include "net_internal.h"

namespace dc {

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
  if (stream && stream_borrowed_) pool_stream_release(stream);
  else if (stream) (void)hipStreamDestroy((hipStream_t)stream);
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
  c->outputs_mask = outputs_mask;
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
void Net::set_outputs_mask(int mask) {
  const int n = (int)outputs.size();
  const int all = n >= 31 ? -1 : (1 << n) - 1;
  if (mask != -1 && n < 31) {
    if (mask & ~all) throw DcError(DC_EINVAL, "DC_OPT_OUTPUTS: the net has " + std::to_string(n) + " outputs");
    if (mask == 0) throw DcError(DC_EINVAL, "DC_OPT_OUTPUTS: at least one output must be wanted");
    if (mask == all) mask = -1;
  }
  if (mask == outputs_mask) return;
  outputs_mask = mask;
  invalidate_plans();
}

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

}  // namespace dc
