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
//   Blob<float>::Reshape / shape / count / offset / cpu_data / mutable_cpu_data / gpu_data
//                                                                                    (blob.hpp:52-164, blob.cpp:82-139)
// Errors: the reference LOG(FATAL)s; the facade throws std::runtime_error with the library's message.
#ifndef CAFFE_FACADE_HPP_
#define CAFFE_FACADE_HPP_

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

template <>
class Blob<float> {
 public:
  explicit Blob(dc_blob* h) : h_(h) {}
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
  dc_blob* handle() const { return h_; }

 private:
  dc_blob* h_;
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

 private:
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

}  // namespace caffe
#endif  // CAFFE_FACADE_HPP_
