// formats.cpp — see formats.h.
#include "formats.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../include/deepcut_hip.h"

namespace dc {

std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) throw DcError(DC_EIO, "Could not open file " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// ============================ text format ====================================================
namespace {
struct Lexer {
  const std::string& s;
  size_t p = 0;
  int line = 1;
  explicit Lexer(const std::string& t) : s(t) {}
  void skip() {
    for (;;) {
      while (p < s.size() && std::isspace((unsigned char)s[p])) {
        if (s[p] == '\n') ++line;
        ++p;
      }
      if (p < s.size() && s[p] == '#') {
        while (p < s.size() && s[p] != '\n') ++p;
        continue;
      }
      break;
    }
  }
  bool eof() {
    skip();
    return p >= s.size();
  }
  char peek() {
    skip();
    return p < s.size() ? s[p] : '\0';
  }
  [[noreturn]] void fail(const std::string& m) {
    throw DcError(DC_EINVAL, "prototxt line " + std::to_string(line) + ": " + m);
  }
  std::string ident() {
    skip();
    size_t b = p;
    while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '.')) ++p;
    if (b == p) fail(std::string("expected identifier, got '") + (p < s.size() ? s[p] : '?') + "'");
    return s.substr(b, p - b);
  }
  // a scalar value: quoted string (either quote kind, adjacent literals concatenate) or bare token
  std::string scalar(bool* quoted) {
    skip();
    *quoted = false;
    if (p < s.size() && (s[p] == '"' || s[p] == '\'')) {
      *quoted = true;
      std::string out;
      while (p < s.size() && (s[p] == '"' || s[p] == '\'')) {
        char q = s[p++];
        while (p < s.size() && s[p] != q) {
          if (s[p] == '\\' && p + 1 < s.size()) {
            char c = s[++p];
            switch (c) {
              case 'n': out += '\n'; break;
              case 't': out += '\t'; break;
              case 'r': out += '\r'; break;
              default: out += c;
            }
            ++p;
          } else {
            if (s[p] == '\n') ++line;
            out += s[p++];
          }
        }
        if (p >= s.size()) fail("unterminated string literal");
        ++p;
        skip();
      }
      return out;
    }
    size_t b = p;
    while (p < s.size() && !std::isspace((unsigned char)s[p]) && s[p] != '{' && s[p] != '}' &&
           s[p] != '#' && s[p] != ',' && s[p] != ';' && s[p] != ']')
      ++p;
    if (b == p) fail("expected a value");
    return s.substr(b, p - b);
  }
};

void parse_fields(Lexer& lx, TextMsg& m, char closer) {
  for (;;) {
    if (lx.eof()) {
      if (closer) lx.fail("missing closing brace");
      return;
    }
    char c = lx.peek();
    if (closer && c == closer) {
      ++lx.p;
      return;
    }
    if (c == '}' || c == '>') lx.fail("unbalanced closing brace");
    TextField f;
    f.key = lx.ident();
    c = lx.peek();
    bool colon = false;
    if (c == ':') {
      ++lx.p;
      colon = true;
      c = lx.peek();
    }
    if (c == '{' || c == '<') {
      ++lx.p;
      f.msg = std::make_shared<TextMsg>();
      parse_fields(lx, *f.msg, c == '{' ? '}' : '>');
      m.fields.push_back(std::move(f));
    } else if (colon && c == '[') {  // short repeated form  key: [a, b]
      ++lx.p;
      for (;;) {
        c = lx.peek();
        if (c == ']') {
          ++lx.p;
          break;
        }
        TextField e;
        e.key = f.key;
        e.scalar = lx.scalar(&e.quoted);
        m.fields.push_back(std::move(e));
        if (lx.peek() == ',') ++lx.p;
      }
    } else {
      if (!colon) lx.fail("expected ':' or '{' after field name '" + f.key + "'");
      f.scalar = lx.scalar(&f.quoted);
      m.fields.push_back(std::move(f));
    }
    c = lx.peek();
    if (c == ',' || c == ';') ++lx.p;
  }
}
}  // namespace

TextMsg parse_text_proto(const std::string& text) {
  Lexer lx(text);
  TextMsg m;
  parse_fields(lx, m, '\0');
  return m;
}

bool TextMsg::has(const std::string& key) const {
  for (auto& f : fields)
    if (f.key == key) return true;
  return false;
}
std::string TextMsg::str(const std::string& key, const std::string& def) const {
  for (auto& f : fields)
    if (f.key == key && !f.msg) return f.scalar;
  return def;
}
double TextMsg::num(const std::string& key, double def) const {
  for (auto& f : fields)
    if (f.key == key && !f.msg) {
      char* e = nullptr;
      double v = std::strtod(f.scalar.c_str(), &e);
      if (e == f.scalar.c_str()) throw DcError(DC_EINVAL, "field '" + key + "': '" + f.scalar + "' is not a number");
      return v;
    }
  return def;
}
bool TextMsg::boolean(const std::string& key, bool def) const {
  for (auto& f : fields)
    if (f.key == key && !f.msg) {
      if (f.scalar == "true" || f.scalar == "True" || f.scalar == "t" || f.scalar == "1") return true;
      if (f.scalar == "false" || f.scalar == "False" || f.scalar == "f" || f.scalar == "0") return false;
      throw DcError(DC_EINVAL, "field '" + key + "': '" + f.scalar + "' is not a bool");
    }
  return def;
}
std::vector<std::string> TextMsg::strs(const std::string& key) const {
  std::vector<std::string> v;
  for (auto& f : fields)
    if (f.key == key && !f.msg) v.push_back(f.scalar);
  return v;
}
std::vector<double> TextMsg::nums(const std::string& key) const {
  std::vector<double> v;
  for (auto& f : fields)
    if (f.key == key && !f.msg) v.push_back(std::strtod(f.scalar.c_str(), nullptr));
  return v;
}
const TextMsg* TextMsg::sub(const std::string& key) const {
  for (auto& f : fields)
    if (f.key == key && f.msg) return f.msg.get();
  return nullptr;
}
std::vector<const TextMsg*> TextMsg::subs(const std::string& key) const {
  std::vector<const TextMsg*> v;
  for (auto& f : fields)
    if (f.key == key && f.msg) v.push_back(f.msg.get());
  return v;
}

// ============================ wire format ====================================================
namespace {
struct Reader {
  const uint8_t* p;
  const uint8_t* e;
  bool done() const { return p >= e; }
  uint64_t varint() {
    uint64_t v = 0;
    int sh = 0;
    while (p < e) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) return v;
      sh += 7;
      if (sh > 63) break;
    }
    throw DcError(DC_EINVAL, "caffemodel: truncated or malformed varint");
  }
  Reader sub() {
    uint64_t n = varint();
    if (n > (uint64_t)(e - p)) throw DcError(DC_EINVAL, "caffemodel: length-delimited field overruns buffer");
    Reader r{p, p + n};
    p += n;
    return r;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: need(8); p += 8; break;
      case 2: sub(); break;
      case 5: need(4); p += 4; break;
      default: throw DcError(DC_EINVAL, "caffemodel: unsupported wire type " + std::to_string(wt));
    }
  }
  void need(size_t n) {
    if ((size_t)(e - p) < n) throw DcError(DC_EINVAL, "caffemodel: truncated fixed field");
  }
};

BlobData parse_blob(Reader r) {
  BlobData b;
  int legacy[4] = {0, 0, 0, 0};
  bool has_legacy = false, has_shape = false;
  std::vector<double> dd;
  while (!r.done()) {
    uint64_t tag = r.varint();
    int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn >= 1 && fn <= 4 && wt == 0) {
      legacy[fn - 1] = (int)r.varint();
      has_legacy = true;
    } else if (fn == 7 && wt == 2) {  // BlobShape
      Reader s = r.sub();
      has_shape = true;
      while (!s.done()) {
        uint64_t t2 = s.varint();
        int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
        if (f2 == 1 && w2 == 2) {
          Reader d = s.sub();
          while (!d.done()) b.shape.push_back((int)d.varint());
        } else if (f2 == 1 && w2 == 0) {
          b.shape.push_back((int)s.varint());
        } else {
          s.skip(w2);
        }
      }
    } else if (fn == 5 && wt == 2) {  // packed float data
      Reader d = r.sub();
      size_t n = (size_t)(d.e - d.p) / 4;
      size_t old = b.data.size();
      b.data.resize(old + n);
      if (n) std::memcpy(b.data.data() + old, d.p, n * 4);  // (an empty packed field: no copy from / to a null vector)
    } else if (fn == 5 && wt == 5) {  // unpacked float
      r.need(4);
      float f;
      std::memcpy(&f, r.p, 4);
      r.p += 4;
      b.data.push_back(f);
    } else if (fn == 8 && wt == 2) {  // packed double_data
      Reader d = r.sub();
      size_t n = (size_t)(d.e - d.p) / 8;
      for (size_t i = 0; i < n; ++i) {
        double v;
        std::memcpy(&v, d.p + 8 * i, 8);
        dd.push_back(v);
      }
    } else if (fn == 8 && wt == 1) {
      r.need(8);
      double v;
      std::memcpy(&v, r.p, 8);
      r.p += 8;
      dd.push_back(v);
    } else {
      r.skip(wt);
    }
  }
  if (b.data.empty() && !dd.empty()) {  // Blob::FromProto: double_data -> Dtype (blob.cpp:478-483)
    b.data.resize(dd.size());
    for (size_t i = 0; i < dd.size(); ++i) b.data[i] = (float)dd[i];
  }
  if (!has_shape && has_legacy) {  // legacy 4-D (blob.cpp:449-457)
    b.shape = {legacy[0], legacy[1], legacy[2], legacy[3]};
  }
  return b;
}

LayerBlobs parse_layer(Reader r) {
  LayerBlobs L;
  while (!r.done()) {
    uint64_t tag = r.varint();
    int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (wt == 2 && (fn == 1 || fn == 2 || fn == 3 || fn == 4)) {
      Reader s = r.sub();
      std::string v((const char*)s.p, (size_t)(s.e - s.p));
      if (fn == 1) L.name = v;
      else if (fn == 2) L.type = v;
      else if (fn == 3) L.bottoms.push_back(v);
      else L.tops.push_back(v);
    } else if (fn == 7 && wt == 2) {
      L.blobs.push_back(parse_blob(r.sub()));
    } else {
      r.skip(wt);
    }
  }
  return L;
}

// V1LayerParameter (caffe.proto:1205-1296): bottom 2, top 3, name 4, type 5 (enum), blobs 6, and the V0 wrapper
// `layer` 1 (V0LayerParameter, caffe.proto:1299-1341: name 1, type 2 (string), blobs 50).
LayerBlobs parse_v1_layer(Reader r) {
  LayerBlobs L;
  while (!r.done()) {
    uint64_t tag = r.varint();
    int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (wt == 2 && (fn == 2 || fn == 3 || fn == 4)) {
      Reader s = r.sub();
      std::string v((const char*)s.p, (size_t)(s.e - s.p));
      if (fn == 4) L.name = v;
      else if (fn == 2) L.bottoms.push_back(v);
      else L.tops.push_back(v);
    } else if (fn == 5 && wt == 0) {
      L.type = v1_layer_type_name((int)r.varint());
    } else if (fn == 6 && wt == 2) {
      L.blobs.push_back(parse_blob(r.sub()));
    } else if (fn == 1 && wt == 2) {  // V0: the connectivity is outside, name / type / blobs inside
      Reader v0 = r.sub();
      while (!v0.done()) {
        uint64_t t0 = v0.varint();
        int f0 = (int)(t0 >> 3), w0 = (int)(t0 & 7);
        if (w0 == 2 && (f0 == 1 || f0 == 2)) {
          Reader s = v0.sub();
          std::string v((const char*)s.p, (size_t)(s.e - s.p));
          if (f0 == 1) L.name = v;
          else {
            // V0 type strings are lower-case ("conv", "relu", ...; upgrade_proto.cpp:540-600); only the ones this
            // path can hold weights for are named, the rest keep their V0 spelling
            static const std::pair<const char*, const char*> kV0[] = {
                {"conv", "Convolution"}, {"innerproduct", "InnerProduct"}, {"pool", "Pooling"}, {"relu", "ReLU"},
                {"sigmoid", "Sigmoid"},  {"split", "Split"}};
            L.type = v;
            for (auto& e : kV0)
              if (v == e.first) L.type = e.second;
          }
        } else if (f0 == 50 && w0 == 2) {
          L.blobs.push_back(parse_blob(v0.sub()));
        } else {
          v0.skip(w0);
        }
      }
    } else {
      r.skip(wt);
    }
  }
  return L;
}

struct Writer {
  std::string out;
  void varint(uint64_t v) {
    while (v >= 0x80) {
      out.push_back((char)((v & 0x7f) | 0x80));
      v >>= 7;
    }
    out.push_back((char)v);
  }
  void tag(int fn, int wt) { varint(((uint64_t)fn << 3) | (uint64_t)wt); }
  void bytes(int fn, const std::string& s) {
    tag(fn, 2);
    varint(s.size());
    out += s;
  }
};
}  // namespace

ModelFile read_caffemodel(const std::string& path) {
  std::string buf = read_file(path);
  ModelFile m;
  Reader r{(const uint8_t*)buf.data(), (const uint8_t*)buf.data() + buf.size()};
  std::vector<LayerBlobs> v1;
  while (!r.done()) {
    uint64_t tag = r.varint();
    int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 1 && wt == 2) {
      Reader s = r.sub();
      m.name.assign((const char*)s.p, (size_t)(s.e - s.p));
    } else if (fn == 100 && wt == 2) {
      m.layers.push_back(parse_layer(r.sub()));
    } else if (fn == 2 && wt == 2) {  // V1LayerParameter `layers` (caffe.proto:95)
      v1.push_back(parse_v1_layer(r.sub()));
    } else {
      r.skip(wt);
    }
  }
  // UpgradeV1Net (upgrade_proto.cpp:647-664): when a file carries V1 `layers`, they are the model and any `layer`
  // entries beside them are ignored
  if (!v1.empty()) m.layers = std::move(v1);
  return m;
}

namespace {
struct V1Type {
  int id;
  const char* ident;
  const char* name;
};
// enum values: caffe.proto:1211-1252; names: upgrade_proto.cpp:852-940
const V1Type kV1Types[] = {
    {0, "NONE", ""}, {35, "ABSVAL", "AbsVal"}, {1, "ACCURACY", "Accuracy"}, {30, "ARGMAX", "ArgMax"}, {2, "BNLL", "BNLL"},
    {3, "CONCAT", "Concat"}, {37, "CONTRASTIVE_LOSS", "ContrastiveLoss"}, {4, "CONVOLUTION", "Convolution"},
    {5, "DATA", "Data"}, {39, "DECONVOLUTION", "Deconvolution"}, {6, "DROPOUT", "Dropout"},
    {32, "DUMMY_DATA", "DummyData"}, {7, "EUCLIDEAN_LOSS", "EuclideanLoss"}, {25, "ELTWISE", "Eltwise"},
    {38, "EXP", "Exp"}, {8, "FLATTEN", "Flatten"}, {9, "HDF5_DATA", "HDF5Data"}, {10, "HDF5_OUTPUT", "HDF5Output"},
    {28, "HINGE_LOSS", "HingeLoss"}, {11, "IM2COL", "Im2col"}, {12, "IMAGE_DATA", "ImageData"},
    {13, "INFOGAIN_LOSS", "InfogainLoss"}, {14, "INNER_PRODUCT", "InnerProduct"}, {15, "LRN", "LRN"},
    {29, "MEMORY_DATA", "MemoryData"}, {16, "MULTINOMIAL_LOGISTIC_LOSS", "MultinomialLogisticLoss"}, {34, "MVN", "MVN"},
    {17, "POOLING", "Pooling"}, {26, "POWER", "Power"}, {18, "RELU", "ReLU"}, {19, "SIGMOID", "Sigmoid"},
    {27, "SIGMOID_CROSS_ENTROPY_LOSS", "SigmoidCrossEntropyLoss"}, {36, "SILENCE", "Silence"}, {20, "SOFTMAX", "Softmax"},
    {21, "SOFTMAX_LOSS", "SoftmaxWithLoss"}, {22, "SPLIT", "Split"}, {33, "SLICE", "Slice"}, {23, "TANH", "TanH"},
    {24, "WINDOW_DATA", "WindowData"}, {31, "THRESHOLD", "Threshold"}};
}  // namespace

const char* v1_layer_type_name(int enum_value) {
  for (auto& t : kV1Types)
    if (t.id == enum_value) return t.name;
  return "";
}
const char* v1_layer_type_name(const std::string& ident) {
  for (auto& t : kV1Types)
    if (ident == t.ident) return t.name;
  return "";
}

void write_caffemodel(const std::string& path, const ModelFile& m) {
  std::ofstream f(path, std::ios::binary);
  if (!f.good()) throw DcError(DC_EIO, "Could not open file " + path);
  Writer top;
  top.bytes(1, m.name);
  f.write(top.out.data(), (std::streamsize)top.out.size());
  for (auto& L : m.layers) {
    Writer lw;
    lw.bytes(1, L.name);
    lw.bytes(2, L.type);
    for (auto& b : L.bottoms) lw.bytes(3, b);
    for (auto& t : L.tops) lw.bytes(4, t);
    for (auto& b : L.blobs) {
      Writer bw;
      {  // shape = 7 { dim = 1 packed }
        Writer dims;
        for (int d : b.shape) dims.varint((uint64_t)d);
        Writer sh;
        sh.bytes(1, dims.out);
        bw.bytes(7, sh.out);
      }
      bw.tag(5, 2);
      bw.varint(b.data.size() * 4);
      bw.out.append((const char*)b.data.data(), b.data.size() * 4);
      lw.bytes(7, bw.out);
    }
    Writer hdr;
    hdr.tag(100, 2);
    hdr.varint(lw.out.size());
    f.write(hdr.out.data(), (std::streamsize)hdr.out.size());
    f.write(lw.out.data(), (std::streamsize)lw.out.size());
  }
  if (!f.good()) throw DcError(DC_EIO, "write failed for " + path);
}

}  // namespace dc
