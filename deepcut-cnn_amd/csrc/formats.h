// formats.h — model-definition and weight-file formats of the DeeperCut path.
//
//  * text prototxt  (reference: ReadProtoFromTextFile, src/caffe/util/io.cpp:34-43, driven by
//    libprotobuf's TextFormat).  libprotobuf is not available, so this is a small hand-written
//    recursive-descent parser of the protobuf text format into a generic field tree.
//  * binary .caffemodel (reference: ReadProtoFromBinaryFile io.cpp:52-65 / WriteProtoToBinaryFile
//    io.cpp:67-70).  Hand-written protobuf wire-format reader/writer for the subset
//    NetParameter{name=1, layer=100{name=1,type=2,bottom=3,top=4,blobs=7{shape=7{dim=1},data=5,
//    double_data=8, num/channels/height/width=1..4}}} (src/caffe/proto/caffe.proto:6-22,64-96,311-334).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace dc {

struct DcError : std::runtime_error {
  int code;
  DcError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- protobuf text format -----------------------------------------------------------------
struct TextMsg;
struct TextField {
  std::string key;
  std::string scalar;              // raw token (unquoted) when !msg
  bool quoted = false;             // scalar was a string literal
  std::shared_ptr<TextMsg> msg;    // nested message when set
};
struct TextMsg {
  std::vector<TextField> fields;
  // first scalar with this key or default
  std::string str(const std::string& key, const std::string& def = "") const;
  bool has(const std::string& key) const;
  double num(const std::string& key, double def) const;
  bool boolean(const std::string& key, bool def) const;
  std::vector<std::string> strs(const std::string& key) const;
  std::vector<double> nums(const std::string& key) const;
  const TextMsg* sub(const std::string& key) const;
  std::vector<const TextMsg*> subs(const std::string& key) const;
};
TextMsg parse_text_proto(const std::string& text);

// ---- .caffemodel --------------------------------------------------------------------------
struct BlobData {
  std::vector<int> shape;
  std::vector<float> data;
  size_t count() const {
    size_t c = 1;
    for (int d : shape) c *= (size_t)d;
    return c;
  }
};
struct LayerBlobs {
  std::string name, type;
  std::vector<std::string> bottoms, tops;
  std::vector<BlobData> blobs;
};
struct ModelFile {
  std::string name;
  std::vector<LayerBlobs> layers;
};
// Reads the current format (NetParameter.layer, field 100) and, like the reference's UpgradeNetAsNeeded
// (src/caffe/util/upgrade_proto.cpp:19-78), the deprecated V1 `layers` (field 2, V1LayerParameter) including V0 files
// whose V1 entries wrap a V0LayerParameter (`layer`, field 1): name / type / bottoms / tops / blobs are carried over,
// the V1 type enum becomes the current type string.
ModelFile read_caffemodel(const std::string& path);
// V1LayerParameter.LayerType (caffe.proto:1211-1252) -> current type string (upgrade_proto.cpp:852-940); "" if unknown
const char* v1_layer_type_name(int enum_value);
const char* v1_layer_type_name(const std::string& enum_identifier);  // "CONVOLUTION" -> "Convolution"
void write_caffemodel(const std::string& path, const ModelFile& m);
// HDF5 weights (Net::ToHDF5 / CopyTrainedLayersFromHDF5, net.cpp:861-975): /data/<layer>/<param index> float datasets.
// Decoded from the HDF5 file format directly (hdf5_reader.cpp); no HDF5 library involved.
bool is_hdf5_path(const std::string& path);  // the reference's rule: the name ends in ".h5" (net.cpp:843-850)
ModelFile read_hdf5_weights(const std::string& path);

std::string read_file(const std::string& path);  // throws DcError(DC_EIO, "Could not open file ...")

}  // namespace dc
