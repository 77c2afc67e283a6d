// hdf5_reader.cpp — read-only parser for the HDF5 weight files Caffe writes (Net::ToHDF5, src/caffe/net.cpp:926-975;
// read back by Net::CopyTrainedLayersFromHDF5, net.cpp:861-909, chosen when the file name ends in ".h5", net.cpp:843-850).
//
// The HDF5 C library and its headers are not a dependency of this package, so the few on-disk structures such files
// consist of are decoded directly, following the published HDF5 File Format Specification (version 1.1 / 2.0):
//   superblock v0/v1 (and v2/v3)   -> root group object header
//   object header v1 (+ continuation blocks), messages: Symbol Table 0x11, Dataspace 0x01, Datatype 0x03, Layout 0x08
//   old-style groups: v1 B-tree of type 0 -> symbol-table nodes "SNOD" -> names in the group's local heap "HEAP"
//   datasets: IEEE little-endian float32 / float64, contiguous or compact layout (what H5LTmake_dataset_float/double
//   produce, src/caffe/util/hdf5.cpp:85-125)
// Layout of a weights file:  /data/<layer name>/<param index>  (net.cpp:931-963).
// Anything else (new-style dense groups, chunked / filtered datasets, other element types) is refused by name.
#include <cstdint>
#include <cstring>

#include "../../include/deepcut_hip.h"
#include "formats.h"

namespace dc {
namespace {

struct H5 {
  const uint8_t* b;
  size_t n;
  int so = 8, sl = 8;  // size of offsets / lengths
  uint64_t base = 0;
  std::string path;

  [[noreturn]] void bad(const std::string& what) const { throw DcError(DC_EUNSUP, "HDF5 file " + path + ": " + what); }
  void need(uint64_t off, uint64_t len) const {
    if (off > n || len > n - off) bad("truncated (offset " + std::to_string(off) + " + " + std::to_string(len) + " beyond the file)");
  }
  uint64_t u(uint64_t off, int bytes) const {
    need(off, (uint64_t)bytes);
    uint64_t v = 0;
    for (int i = bytes - 1; i >= 0; --i) v = (v << 8) | b[off + i];
    return v;
  }
  uint64_t addr(uint64_t off) const { return u(off, so); }  // file address field (relative to the base address)
  bool undef(uint64_t a) const { return so == 8 ? a == ~0ull : a == ((1ull << (8 * so)) - 1); }
  bool sig(uint64_t off, const char* s) const {
    need(off, 4);
    return std::memcmp(b + off, s, 4) == 0;
  }
};

struct Msg {
  int type;
  uint64_t off, size;
};

// Object header v1: 16-byte prefix, then messages {type:2, size:2, flags:1, reserved:3, data}, continued through
// Object Header Continuation messages (0x0010: offset, length).
std::vector<Msg> object_messages(const H5& f, uint64_t at) {
  at += f.base;
  const int ver = (int)f.u(at, 1);
  if (ver != 1) {
    if (f.sig(at, "OHDR")) f.bad("version-2 object headers (a file written with a 'latest' library format bound) are not supported");
    f.bad("unknown object header version " + std::to_string(ver));
  }
  const int nmsg = (int)f.u(at + 2, 2);
  const uint64_t hsize = f.u(at + 8, 4);
  std::vector<Msg> out;
  std::vector<std::pair<uint64_t, uint64_t>> blocks = {{at + 16, hsize}};
  for (size_t bi = 0; bi < blocks.size() && (int)out.size() < nmsg; ++bi) {
    uint64_t p = blocks[bi].first;
    const uint64_t end = p + blocks[bi].second;
    while (p + 8 <= end && (int)out.size() < nmsg) {
      const int type = (int)f.u(p, 2);
      const uint64_t size = f.u(p + 2, 2);
      f.need(p + 8, size);
      out.push_back({type, p + 8, size});
      if (type == 0x10) blocks.push_back({f.base + f.addr(p + 8), f.u(p + 8 + f.so, f.sl)});
      p += 8 + size;
    }
  }
  return out;
}

// Old-style group: (name, object header address) of every link, in the B-tree's (name-sorted) order.
void walk_btree(const H5& f, uint64_t node, uint64_t heap_data, std::vector<std::pair<std::string, uint64_t>>& out, int depth) {
  if (depth > 16) f.bad("group B-tree too deep");
  node += f.base;
  if (!f.sig(node, "TREE")) f.bad("group B-tree node signature missing");
  if (f.u(node + 4, 1) != 0) f.bad("B-tree node is not a group node");
  const int level = (int)f.u(node + 5, 1), used = (int)f.u(node + 6, 2);
  uint64_t p = node + 8 + 2 * (uint64_t)f.so;  // skip the sibling addresses
  for (int i = 0; i < used; ++i) {
    p += f.sl;  // key i
    const uint64_t child = f.addr(p);
    p += f.so;
    if (level > 0) {
      walk_btree(f, child, heap_data, out, depth + 1);
      continue;
    }
    const uint64_t sn = child + f.base;
    if (!f.sig(sn, "SNOD")) f.bad("symbol table node signature missing");
    const int nsym = (int)f.u(sn + 6, 2);
    uint64_t e = sn + 8;
    for (int s = 0; s < nsym; ++s) {
      const uint64_t name_off = f.addr(e), obj = f.addr(e + f.so);
      const uint64_t np = heap_data + name_off;
      f.need(np, 1);
      const void* z = std::memchr(f.b + np, 0, f.n - np);
      if (!z) f.bad("unterminated link name");
      out.emplace_back(std::string((const char*)f.b + np, (const char*)z), obj);
      e += 2 * (uint64_t)f.so + 4 + 4 + 16;
    }
  }
}

std::vector<std::pair<std::string, uint64_t>> group_links(const H5& f, uint64_t header, const std::string& what) {
  for (const Msg& m : object_messages(f, header)) {
    if (m.type == 0x11) {
      const uint64_t btree = f.addr(m.off), heap = f.addr(m.off + f.so) + f.base;
      if (!f.sig(heap, "HEAP")) f.bad("local heap signature missing in " + what);
      const uint64_t data = f.addr(heap + 8 + 2 * (uint64_t)f.sl) + f.base;
      std::vector<std::pair<std::string, uint64_t>> out;
      walk_btree(f, btree, data, out, 0);
      return out;
    }
    if (m.type == 0x02 || m.type == 0x06)
      f.bad(what + " is a new-style group (link messages / dense storage); only symbol-table groups, as written by the "
                   "reference with the default library format, are supported");
  }
  f.bad(what + " is not a group");
}

BlobData read_dataset(const H5& f, uint64_t header, const std::string& what) {
  BlobData out;
  int elem = 0;
  bool have_space = false, have_type = false, have_layout = false;
  uint64_t data_at = 0, data_len = 0;
  for (const Msg& m : object_messages(f, header)) {
    if (m.type == 0x01) {  // dataspace
      const int ver = (int)f.u(m.off, 1), rank = (int)f.u(m.off + 1, 1);
      uint64_t p = m.off + (ver == 1 ? 8 : 4);
      if (ver != 1 && ver != 2) f.bad(what + ": dataspace version " + std::to_string(ver));
      if (rank > 32) f.bad(what + ": rank " + std::to_string(rank));  // kMaxBlobAxes (blob.hpp:20)
      for (int d = 0; d < rank; ++d, p += f.sl) {
        const uint64_t dim = f.u(p, f.sl);
        if (dim > 0x7fffffffull) f.bad(what + ": dimension too large");
        out.shape.push_back((int)dim);
      }
      have_space = true;
    } else if (m.type == 0x03) {  // datatype
      const int cls = (int)(f.u(m.off, 1) & 0x0f), bits0 = (int)f.u(m.off + 1, 1);
      elem = (int)f.u(m.off + 4, 4);
      if (cls != 1 || (bits0 & 1) || (elem != 4 && elem != 8))
        f.bad(what + ": only little-endian IEEE float32 / float64 datasets are weights (class " + std::to_string(cls) + ", size " +
              std::to_string(elem) + ")");
      have_type = true;
    } else if (m.type == 0x08) {  // data layout
      const int ver = (int)f.u(m.off, 1);
      if (ver != 3) f.bad(what + ": data layout message version " + std::to_string(ver) + " (expected 3)");
      const int cls = (int)f.u(m.off + 1, 1);
      if (cls == 1) {
        const uint64_t a = f.addr(m.off + 2);
        data_len = f.u(m.off + 2 + f.so, f.sl);
        data_at = f.undef(a) ? 0 : a + f.base;
        if (f.undef(a)) data_len = 0;  // never written: all zeros
      } else if (cls == 0) {
        data_len = f.u(m.off + 2, 2);
        data_at = m.off + 4;
      } else {
        f.bad(what + ": chunked datasets are not supported (Caffe writes contiguous ones)");
      }
      have_layout = true;
    } else if (m.type == 0x0b) {
      f.bad(what + ": filtered (compressed) datasets are not supported");
    }
  }
  if (!have_space || !have_type || !have_layout) f.bad(what + " is not a simple dataset");
  // the element count: 32 dimensions of up to 2^31 - 1 overflow any integer; a blob holds at most INT_MAX elements (blob.cpp:33)
  uint64_t cnt64 = 1;
  for (int d : out.shape) {
    cnt64 *= (uint64_t)d;
    if (cnt64 > 0x7fffffffull) f.bad(what + ": more than INT_MAX elements");
  }
  const size_t cnt = (size_t)cnt64;
  if (data_len == 0) {
    // never written: all zeros — from no bytes at all, so the size is capped (2^29 elements = 2 GiB; the largest layer
    // blob of the model zoo, VGG's fc6, has 1.0e8) instead of letting a 1-KB file ask for 8 GB
    if (cnt > ((size_t)1 << 29)) f.bad(what + ": an unwritten dataset of " + std::to_string(cnt) + " elements");
    out.data.assign(cnt, 0.f);
    return out;
  }
  if (data_len < (uint64_t)cnt * elem) f.bad(what + ": stored size smaller than the dataspace");
  f.need(data_at, (uint64_t)cnt * elem);  // (the bytes are there: the allocation below is justified by the file's size)
  out.data.assign(cnt, 0.f);
  if (cnt == 0) return out;
  if (elem == 4) {
    std::memcpy(out.data.data(), f.b + data_at, cnt * 4);
  } else {  // double -> Dtype, as hdf5_load_nd_dataset<float> does through H5LTread_dataset_float
    for (size_t i = 0; i < cnt; ++i) {
      double v;
      std::memcpy(&v, f.b + data_at + 8 * i, 8);
      out.data[i] = (float)v;
    }
  }
  return out;
}

}  // namespace

bool is_hdf5_path(const std::string& path) { return path.size() >= 3 && path.compare(path.size() - 3, 3, ".h5") == 0; }

ModelFile read_hdf5_weights(const std::string& path) {
  const std::string buf = read_file(path);
  H5 f{(const uint8_t*)buf.data(), buf.size()};
  f.path = path;
  static const uint8_t kSig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  uint64_t sb = 0;
  bool found = false;
  for (uint64_t off = 0; off + 8 <= buf.size(); off = off ? off * 2 : 512) {  // 0, 512, 1024, ...
    if (std::memcmp(buf.data() + off, kSig, 8) == 0) {
      sb = off;
      found = true;
      break;
    }
  }
  if (!found) throw DcError(DC_EINVAL, "Couldn't open " + path + ": not an HDF5 file");
  const int ver = (int)f.u(sb + 8, 1);
  uint64_t root = 0, eof = 0;
  if (ver == 0 || ver == 1) {
    f.so = (int)f.u(sb + 13, 1);
    f.sl = (int)f.u(sb + 14, 1);
    if ((f.so != 4 && f.so != 8) || (f.sl != 4 && f.sl != 8)) f.bad("offset / length sizes other than 4 or 8 bytes");
    uint64_t p = sb + 24 + (ver == 1 ? 4 : 0);
    f.base = f.addr(p);
    eof = f.addr(p + 2 * (uint64_t)f.so);
    p += 4 * (uint64_t)f.so;          // base, free-space info, end of file, driver info
    root = f.addr(p + f.so);          // root symbol table entry: link name offset, OBJECT HEADER ADDRESS, ...
  } else if (ver == 2 || ver == 3) {
    f.so = (int)f.u(sb + 9, 1);
    f.sl = (int)f.u(sb + 10, 1);
    if ((f.so != 4 && f.so != 8) || (f.sl != 4 && f.sl != 8)) f.bad("offset / length sizes other than 4 or 8 bytes");
    f.base = f.addr(sb + 12);
    eof = f.addr(sb + 12 + 2 * (uint64_t)f.so);
    root = f.addr(sb + 12 + 3 * (uint64_t)f.so);
  } else {
    f.bad("superblock version " + std::to_string(ver));
  }
  if (f.undef(eof) || f.base + eof > buf.size())  // the library's own check (H5F: "truncated file")
    f.bad("truncated: the superblock records " + std::to_string(f.base + eof) + " bytes, the file has " + std::to_string(buf.size()));
  ModelFile m;
  uint64_t data_group = 0;
  bool have = false;
  for (auto& l : group_links(f, root, "the root group"))
    if (l.first == "data") data_group = l.second, have = true;
  if (!have) throw DcError(DC_EINVAL, "Error reading weights from " + path + ": no /data group (net.cpp:865-866)");
  for (auto& layer : group_links(f, data_group, "/data")) {
    LayerBlobs L;
    L.name = layer.first;
    // datasets are named by parameter index: "0", "1", ...; a gap ends the list as H5Lexists does in the reference
    std::vector<std::pair<std::string, uint64_t>> links = group_links(f, layer.second, "/data/" + layer.first);
    for (int j = 0;; ++j) {
      const std::string want = std::to_string(j);
      bool hit = false;
      for (auto& d : links)
        if (d.first == want) {
          L.blobs.push_back(read_dataset(f, d.second, "/data/" + layer.first + "/" + want));
          hit = true;
        }
      if (!hit) break;
    }
    if (L.blobs.size() != links.size())
      f.bad("/data/" + layer.first + " holds links other than consecutive parameter indices");
    m.layers.push_back(std::move(L));
  }
  return m;
}

}  // namespace dc
