// caffe_facade.hpp — header-only C++ facade over the C ABI (deepcut_hip.h) with the reference's names, for
// non-Python callers written against caffe::Net / caffe::Blob / caffe::Caffe (tools/caffe.cpp style).
// Float instantiation only (pycaffe is float-only too, python/caffe/_caffe.cpp:34).  The reference uses
// boost::shared_ptr inside namespace caffe (include/caffe/common.hpp:79); std::shared_ptr here.
//
//   reference                                   this facade
//   caffe::Caffe::set_mode / SetDevice          dc_set_mode / dc_set_device          (common.hpp:148, common.cpp:140-158)
//   caffe::Net<float>(file, phase)              dc_net_create                        (net.cpp:31-37)
//   Net::CopyTrainedLayersFrom(file)            dc_net_copy_from                     (net.cpp:843-858)
//   Net::ForwardFromTo / ForwardPrefilled       dc_net_forward / dc_net_forward_all  (net.cpp:565-581)
//   Net::Reshape, blob_by_name, has_blob        dc_net_reshape, dc_net_blob          (net.cpp:744-749,947-957)
//   Net::blob_names / layer_names / input_blobs / output_blobs
//   Blob<float>::Reshape / shape / count / offset / cpu_data / mutable_cpu_data / gpu_data / mutable_gpu_data / CopyFrom
//                                                                                    (blob.hpp:52-164, blob.cpp:82-139,435-474)
//   Blob<float>() / Blob<float>(shape)          dc_blob_create (a blob of its own)   (blob.hpp:26-33)
//   SyncedMemory(size) cpu_data / gpu_data / mutable_* / head / size                 (syncedmem.hpp:45-83, syncedmem.cpp:25-139)
//   Layer<float>(LayerParameter) SetUp / Reshape / Forward / blobs, protected virtual Forward_cpu / Forward_gpu
//                                               dc_net_create_for_layer              (layer.hpp:40-74,131-151,335-345,451-487)
//   LayerRegistry<float>::CreateLayer(param)                                         (layer_factory.hpp:75-84)
//   LayerParameter: the reference's is a protobuf message (caffe.proto:311-334); here it carries the TEXT form of that
//   message (what TextFormat would parse) — libprotobuf is not a dependency of this path.
// Errors: the reference LOG(FATAL)s; the facade throws std::runtime_error with the library's message.
#ifndef CAFFE_FACADE_HPP_
#define CAFFE_FACADE_HPP_

#include <array>
#include <cctype>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" {
#include "deepcut_hip.h"
}

namespace caffe {

using std::shared_ptr;
using std::string;
using std::vector;

enum Phase { TRAIN = 0, TEST = 1 };  // caffe.proto:253-256

inline void dc_check_(int rc) {
  if (rc != DC_OK) throw std::runtime_error(dc_last_error());
}

class Caffe {
 public:
  enum Brew { CPU = DC_MODE_CPU, GPU = DC_MODE_GPU };
  static void set_mode(Brew mode) { dc_check_(dc_set_mode(mode)); }
  static Brew mode() { return static_cast<Brew>(dc_get_mode()); }
  static void SetDevice(const int device_id) { dc_check_(dc_set_device(device_id)); }
};

template <typename Dtype>
class Blob;

// SyncedMemory (syncedmem.hpp:45-83): the 4-state lazily synchronised host/device buffer.  Backed by a 1-axis blob of
// ceil(size/4) floats; the device side of a non-4-D blob is a plain copy of the host bytes.  set_cpu_data/set_gpu_data
// (adopting foreign pointers: data layers only) are not offered.
class SyncedMemory {
 public:
  enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };
  SyncedMemory() : SyncedMemory(0) {}
  explicit SyncedMemory(size_t size) : size_(size) {
    int n = (int)((size + 3) / 4);
    dc_check_(dc_blob_create(1, &n, &h_));
  }
  ~SyncedMemory() { dc_blob_destroy(h_); }
  SyncedMemory(const SyncedMemory&) = delete;
  SyncedMemory& operator=(const SyncedMemory&) = delete;
  const void* cpu_data() {
    const float* p = nullptr;
    dc_check_(dc_blob_cpu_data(h_, &p));
    return p;
  }
  void* mutable_cpu_data() {
    float* p = nullptr;
    dc_check_(dc_blob_mutable_cpu_data(h_, &p));
    return p;
  }
  const void* gpu_data() {
    const void* p = nullptr;
    dc_check_(dc_blob_gpu_data(h_, &p, nullptr));
    return p;
  }
  void* mutable_gpu_data() {
    void* p = nullptr;
    dc_check_(dc_blob_mutable_gpu_data(h_, &p, nullptr));
    return p;
  }
  SyncedHead head() { return static_cast<SyncedHead>(dc_blob_head(h_)); }
  size_t size() { return size_; }

 private:
  dc_blob* h_ = nullptr;
  size_t size_;
};

template <>
class Blob<float> {
 public:
  explicit Blob(dc_blob* h) : h_(h) {}  // a net's blob (owned by the net)
  Blob() : Blob(vector<int>()) {}       // a blob of its own (blob.hpp:26-33)
  explicit Blob(const vector<int>& shape) : own_(true) { dc_check_(dc_blob_create((int)shape.size(), shape.data(), &h_)); }
  Blob(int num, int channels, int height, int width) : Blob(vector<int>{num, channels, height, width}) {}
  ~Blob() {
    if (own_) dc_blob_destroy(h_);
  }
  Blob(const Blob&) = delete;
  Blob& operator=(const Blob&) = delete;
  void Reshape(const vector<int>& shape) { dc_check_(dc_blob_reshape(h_, (int)shape.size(), shape.data())); }
  void Reshape(int num, int channels, int height, int width) { Reshape(vector<int>{num, channels, height, width}); }
  vector<int> shape() const {
    int n = 0, d[8];
    dc_check_(dc_blob_shape(h_, &n, d));
    return vector<int>(d, d + n);
  }
  int num_axes() const { return dc_blob_num_axes(h_); }
  int count() const { return dc_blob_count(h_); }
  int LegacyShape(int index) const {  // blob.hpp:118-134
    vector<int> s = shape();
    int pad = 4 - (int)s.size();
    if (index < 0) index += 4;
    return index < pad ? 1 : s[index - pad];
  }
  int num() const { return LegacyShape(0); }
  int channels() const { return LegacyShape(1); }
  int height() const { return LegacyShape(2); }
  int width() const { return LegacyShape(3); }
  int offset(int n, int c = 0, int h = 0, int w = 0) const {  // blob.hpp:153-164
    return ((n * channels() + c) * height() + h) * width() + w;
  }
  const float* cpu_data() const {
    const float* p = nullptr;
    dc_check_(dc_blob_cpu_data(h_, &p));
    return p;
  }
  float* mutable_cpu_data() {
    float* p = nullptr;
    dc_check_(dc_blob_mutable_cpu_data(h_, &p));
    return p;
  }
  // device image: channels-last (NHWC) float32 with the returned channel pitch — NOT the reference's NCHW
  const float* gpu_data(int* channel_pitch = nullptr) const {
    const void* p = nullptr;
    dc_check_(dc_blob_gpu_data(h_, &p, channel_pitch));
    return static_cast<const float*>(p);
  }
  // Blob::mutable_gpu_data (blob.cpp:111-115): the device image becomes authoritative
  float* mutable_gpu_data(int* channel_pitch = nullptr) {
    void* p = nullptr;
    dc_check_(dc_blob_mutable_gpu_data(h_, &p, channel_pitch));
    return static_cast<float*>(p);
  }
  SyncedMemory::SyncedHead head() const { return static_cast<SyncedMemory::SyncedHead>(dc_blob_head(h_)); }
  // Blob::CopyFrom (blob.cpp:435-474), data only
  void CopyFrom(const Blob& source, bool copy_diff = false, bool reshape = false) {
    if (copy_diff) throw std::runtime_error("no diff side on the inference path");
    dc_check_(dc_blob_copy_from(h_, source.h_, reshape ? 1 : 0));
  }
  void ReshapeLike(const Blob& other) { Reshape(other.shape()); }
  dc_blob* handle() const { return h_; }

 private:
  dc_blob* h_ = nullptr;
  bool own_ = false;
};

// LayerParameter (caffe.proto:311-334) in text form: `name: "c" type: "Convolution" bottom: "x" top: "y" convolution_param { ... }`
class LayerParameter {
 public:
  LayerParameter() {}
  explicit LayerParameter(const string& text_format) : text_(text_format) {}
  const string& text() const { return text_; }
  string name() const { return field_("name"); }
  string type() const { return field_("type"); }

 private:
  string field_(const char* key) const {  // first top-level `key: "value"`
    int depth = 0;
    const string k(key);
    for (size_t i = 0; i < text_.size(); ++i) {
      const char c = text_[i];
      if (c == '{') ++depth;
      else if (c == '}') --depth;
      else if (c == '#') while (i < text_.size() && text_[i] != '\n') ++i;
      else if (c == '"') { for (++i; i < text_.size() && text_[i] != '"'; ++i) {} }
      else if (depth == 0 && text_.compare(i, k.size(), k) == 0 && (i == 0 || !(isalnum((unsigned char)text_[i - 1]) || text_[i - 1] == '_'))) {
        size_t j = i + k.size();
        while (j < text_.size() && (text_[j] == ' ' || text_[j] == ':')) ++j;
        if (j < text_.size() && text_[j] == '"') {
          size_t e = text_.find('"', j + 1);
          return text_.substr(j + 1, e == string::npos ? string::npos : e - j - 1);
        }
      }
    }
    return string();
  }
  string text_;
};

// Layer<float> (layer.hpp): one reference layer on its own.  SetUp builds the layer for the given bottoms, Reshape follows
// the bottoms' shapes, Forward reshapes then dispatches on Caffe::mode() to the protected virtual Forward_cpu / Forward_gpu
// exactly like layer.hpp:451-487.  Forward_gpu runs the layer's CDNA4 kernels (the counterpart of
// src/caffe/layers/*_layer.cu Forward_gpu); Forward_cpu refuses: this library has no CPU compute path.
template <typename Dtype>
class Layer;

template <>
class Layer<float> {
 public:
  explicit Layer(const LayerParameter& param) : layer_param_(param) {}
  virtual ~Layer() {
    if (net_) dc_net_destroy(net_);
  }
  Layer(const Layer&) = delete;
  Layer& operator=(const Layer&) = delete;

  void SetUp(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {  // layer.hpp:67-74
    LayerSetUp(bottom, top);
    Reshape(bottom, top);
  }
  virtual void LayerSetUp(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
    (void)top;
    vector<dc_blob*> hs;
    for (auto* b : bottom) hs.push_back(b->handle());
    if (net_) dc_net_destroy(net_), net_ = nullptr;
    dc_check_(dc_net_create_for_layer(layer_param_.text().c_str(), DC_PHASE_TEST, (int)hs.size(), hs.data(), &net_));
    layer_index_ = dc_net_num_layers(net_) - 1;
    for (int i = 0; i < dc_net_num_layers(net_); ++i)
      if (layer_param_.name() == dc_net_layer_name(net_, i)) layer_index_ = i;
    blobs_.clear();
    const char* lname = dc_net_layer_name(net_, layer_index_);
    for (int i = 0; i < dc_net_layer_num_params(net_, lname); ++i) {
      dc_blob* pb = nullptr;
      dc_check_(dc_net_param(net_, lname, i, &pb));
      blobs_.push_back(std::make_shared<Blob<float> >(pb));
    }
  }
  virtual void Reshape(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
    require_setup_();
    if ((int)bottom.size() != dc_net_num_inputs(net_)) throw std::runtime_error("Layer::Reshape: bottom count changed since SetUp");
    for (size_t i = 0; i < bottom.size(); ++i) {
      const vector<int> s = bottom[i]->shape();
      dc_check_(dc_blob_reshape(in_(i), (int)s.size(), s.data()));
    }
    dc_check_(dc_net_reshape(net_));
    for (size_t i = 0; i < top.size(); ++i) {
      int n = 0, d[8];
      dc_check_(dc_blob_shape(out_(i), &n, d));
      top[i]->Reshape(vector<int>(d, d + n));
    }
  }
  inline float Forward(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {  // layer.hpp:451-487
    Reshape(bottom, top);
    switch (Caffe::mode()) {
      case Caffe::CPU: Forward_cpu(bottom, top); break;
      case Caffe::GPU: Forward_gpu(bottom, top); break;
    }
    return 0.f;  // no loss layers on the inference path
  }
  // The parameter blobs are handles INTO this layer's private net: they are valid while the Layer lives and until its next
  // LayerSetUp.  (The reference's shared_ptr<Blob> owns its memory, so weight sharing by keeping the pointer beyond the
  // layer works there and does not here: copy with Blob::CopyFrom instead.)
  vector<shared_ptr<Blob<float> > >& blobs() { return blobs_; }
  const LayerParameter& layer_param() const { return layer_param_; }
  virtual inline const char* type() const { return net_ ? dc_net_layer_type(net_, layer_index_) : ""; }

 protected:
  virtual void Forward_cpu(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
    (void)bottom, (void)top;
    throw std::runtime_error("Layer::Forward_cpu: libdeepcut_hip provides the MI355X path only (DC_ENOCPU); set Caffe::GPU");
  }
  virtual void Forward_gpu(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
    require_setup_();
    for (size_t i = 0; i < bottom.size(); ++i) dc_check_(dc_blob_copy_from(in_(i), bottom[i]->handle(), 0));  // device side if authoritative there
    dc_check_(dc_net_forward_all(net_));
    for (size_t i = 0; i < top.size(); ++i) dc_check_(dc_blob_copy_from(top[i]->handle(), out_(i), 1));  // stays on the device (HEAD_AT_GPU)
  }
  LayerParameter layer_param_;
  vector<shared_ptr<Blob<float> > > blobs_;

 private:
  void require_setup_() const {
    if (!net_) throw std::runtime_error("Layer used before SetUp");
  }
  dc_blob* in_(size_t i) const {
    dc_blob* b = nullptr;
    dc_check_(dc_net_blob(net_, dc_net_input_name(net_, (int)i), &b));
    return b;
  }
  dc_blob* out_(size_t i) const {  // tops = the blobs the one-layer net leaves unconsumed (an in-place top is its bottom's blob)
    const char* name = dc_net_output_name(net_, (int)i);
    if (!name) throw std::runtime_error("Layer: top index out of range");
    dc_blob* b = nullptr;
    dc_check_(dc_net_blob(net_, name, &b));
    return b;
  }
  dc_net* net_ = nullptr;
  int layer_index_ = 0;
};

// LayerRegistry<float>::CreateLayer (layer_factory.hpp:75-84): every layer type of the path is served by the same class
template <typename Dtype>
class LayerRegistry;
template <>
class LayerRegistry<float> {
 public:
  static shared_ptr<Layer<float> > CreateLayer(const LayerParameter& param) { return std::make_shared<Layer<float> >(param); }
};

template <typename Dtype>
class Net;

template <>
class Net<float> {
 public:
  Net(const string& param_file, Phase phase) { dc_check_(dc_net_create(param_file.c_str(), nullptr, phase, &h_)); }
  ~Net() { dc_net_destroy(h_); }
  Net(const Net&) = delete;
  Net& operator=(const Net&) = delete;

  void CopyTrainedLayersFrom(const string trained_filename) { dc_check_(dc_net_copy_from(h_, trained_filename.c_str())); }
  float ForwardFromTo(int start, int end) {
    float loss = 0;
    dc_check_(dc_net_forward(h_, start, end, &loss));
    return loss;
  }
  const vector<Blob<float>*>& ForwardPrefilled(float* loss = nullptr) {
    float l = ForwardFromTo(0, dc_net_num_layers(h_) - 1);
    if (loss) *loss = l;
    return output_blobs();
  }
  void Reshape() { dc_check_(dc_net_reshape(h_)); }
  const string name() const { return dc_net_name(h_); }
  vector<string> layer_names() const {
    vector<string> v;
    for (int i = 0; i < dc_net_num_layers(h_); ++i) v.push_back(dc_net_layer_name(h_, i));
    return v;
  }
  vector<string> blob_names() const {
    vector<string> v;
    for (int i = 0; i < dc_net_num_blobs(h_); ++i) v.push_back(dc_net_blob_name(h_, i));
    return v;
  }
  bool has_blob(const string& blob_name) const {
    dc_blob* b = nullptr;
    return dc_net_blob(h_, blob_name.c_str(), &b) == DC_OK;
  }
  const shared_ptr<Blob<float> > blob_by_name(const string& blob_name) const {
    dc_blob* b = nullptr;
    dc_check_(dc_net_blob(h_, blob_name.c_str(), &b));
    return std::make_shared<Blob<float> >(b);
  }
  const vector<Blob<float>*>& input_blobs() { return collect_(in_, true); }
  const vector<Blob<float>*>& output_blobs() { return collect_(out_, false); }
  int num_inputs() const { return dc_net_num_inputs(h_); }
  int num_outputs() const { return dc_net_num_outputs(h_); }
  dc_net* handle() const { return h_; }
  // A second executor of this model on the same device (dc_net_clone: own activations and stream, SHARED parameters and packed
  // filters).  No reference counterpart; Net::ShareTrainedLayersWith (net.cpp:751-769) is the nearest idea.
  shared_ptr<Net<float> > Clone() const {
    dc_net* c = nullptr;
    dc_check_(dc_net_clone(h_, &c));
    return shared_ptr<Net<float> >(new Net<float>(c));
  }

 private:
  explicit Net(dc_net* adopted) : h_(adopted) {}
  const vector<Blob<float>*>& collect_(vector<Blob<float>*>& cache, bool inputs) {
    if (cache.empty()) {
      int n = inputs ? dc_net_num_inputs(h_) : dc_net_num_outputs(h_);
      for (int i = 0; i < n; ++i) {
        dc_blob* b = nullptr;
        dc_check_(dc_net_blob(h_, inputs ? dc_net_input_name(h_, i) : dc_net_output_name(h_, i), &b));
        owned_.emplace_back(new Blob<float>(b));
        cache.push_back(owned_.back().get());
      }
    }
    return cache;
  }
  dc_net* h_ = nullptr;
  vector<std::unique_ptr<Blob<float> > > owned_;
  vector<Blob<float>*> in_, out_;
};

// ---- in-process multi-GPU forward (dc_comm_create / dc_forward_batch) for C++ callers -----------------------------------------
// No reference counterpart: the reference's only multi-GPU code is the training-time P2PSync tree (src/caffe/parallel.cpp:287-322).
// nets[k] runs on devices[k] (empty: k modulo the visible devices) on a host thread of its own inside the library; Forward() deals the
// images longest-processing-time-first over H*W, batches the same-shape images of an executor, gathers the maps on the root
// executor's device (RCCL send/recv, or peer copies) and returns them as host vectors.  A tools/caffe.cpp-style program
// (tools/caffe.cpp:302-388) creates one Net per device under Caffe::SetDevice(k) — or clones where executors share a device.
class ForwardPool {
 public:
  struct Maps {
    std::vector<float> prob, loc_pred, next_pred;  // [C][h][w] each
    int map_h = 0, map_w = 0;
  };
  explicit ForwardPool(const vector<Net<float>*>& nets, const vector<int>& devices = vector<int>(), int transport = DC_COMM_AUTO) {
    for (Net<float>* n : nets) nets_.push_back(n->handle());
    dc_check_(dc_comm_create((int)nets_.size(), devices.empty() ? nullptr : devices.data(), transport, &comm_));
    shared_ptr<Blob<float> > p = nets[0]->blob_by_name("prob"), l = nets[0]->blob_by_name("loc_pred"), x = nets[0]->blob_by_name("next_pred");
    pc_ = p->channels(), lc_ = l->channels(), nc_ = x->channels();
  }
  ~ForwardPool() { dc_comm_destroy(comm_); }
  ForwardPool(const ForwardPool&) = delete;
  ForwardPool& operator=(const ForwardPool&) = delete;
  // images[i]: 3 x hw[i].first x hw[i].second float32 (NCHW, host)
  std::vector<Maps> Forward(const std::vector<const float*>& images, const std::vector<std::pair<int, int> >& hw) {
    const int n = (int)images.size();
    std::vector<Maps> out((size_t)n);
    std::vector<int> shape((size_t)n * 2 + 2);
    std::vector<float*> pp((size_t)n + 1), lp((size_t)n + 1), np((size_t)n + 1);
    for (int i = 0; i < n; ++i) {
      shape[2 * i] = hw[i].first, shape[2 * i + 1] = hw[i].second;
      Maps& m = out[i];
      map_dims_(hw[i].first, hw[i].second, &m.map_h, &m.map_w);
      m.prob.resize((size_t)pc_ * m.map_h * m.map_w), m.loc_pred.resize((size_t)lc_ * m.map_h * m.map_w), m.next_pred.resize((size_t)nc_ * m.map_h * m.map_w);
      pp[i] = m.prob.data(), lp[i] = m.loc_pred.data(), np[i] = m.next_pred.data();
    }
    dc_check_(dc_forward_batch(comm_, nets_.data(), (int)nets_.size(), images.data(), reinterpret_cast<const int(*)[2]>(shape.data()), n, pp.data(),
                               lp.data(), np.data()));
    return out;
  }
  int transport() const { return dc_comm_transport(comm_); }
  int executor_of(int i) const { return dc_comm_item_executor(comm_, i); }

 private:
  // the maps' height / width for an h x w input: the net's own shape inference (host only), once per distinct shape — the
  // library writes C x map_h x map_w floats per map, so the vectors must be exactly that
  void map_dims_(int h, int w, int* mh, int* mw) {
    for (size_t i = 0; i < seen_.size(); ++i)
      if (seen_[i][0] == h && seen_[i][1] == w) {
        *mh = seen_[i][2], *mw = seen_[i][3];
        return;
      }
    dc_blob *in = nullptr, *p = nullptr;
    dc_check_(dc_net_blob(nets_[0], dc_net_input_name(nets_[0], 0), &in));
    const int shape[4] = {1, 3, h, w};
    dc_check_(dc_blob_reshape(in, 4, shape));
    dc_check_(dc_net_reshape(nets_[0]));
    dc_check_(dc_net_blob(nets_[0], "prob", &p));
    int nd = 0, d[32];
    dc_check_(dc_blob_shape(p, &nd, d));
    *mh = d[2], *mw = d[3];
    seen_.push_back({{h, w, d[2], d[3]}});
  }
  dc_comm* comm_ = nullptr;
  std::vector<dc_net*> nets_;
  std::vector<std::array<int, 4> > seen_;
  int pc_ = 0, lc_ = 0, nc_ = 0;
};

}  // namespace caffe
#endif  // CAFFE_FACADE_HPP_
