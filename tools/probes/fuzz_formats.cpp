// fuzz_formats.cpp — the three file parsers of the library (protobuf text format, .caffemodel wire format, the from-scratch HDF5
// decoder) under AddressSanitizer + UBSan, no GPU involved: a seed file is mutated (byte flips, inserted extremes, truncation,
// spliced chunks) a few thousand times; every mutant must either parse or throw std::exception — no crash, no sanitizer
// report, no allocation beyond what the file can justify, no parse that takes seconds.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I include \
//       tools/probes/fuzz_formats.cpp deepcut-cnn_amd/csrc/formats.cpp deepcut-cnn_amd/csrc/hdf5_reader.cpp -o /tmp/fuzz_formats
//   /tmp/fuzz_formats text|model|hdf5 <seed file> <iterations> [rng seed]      (run by tests/test_format_fuzz.py)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../deepcut-cnn_amd/csrc/formats.h"

static unsigned long long rng_state = 88172645463325252ull;
static unsigned long long rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static std::string mutate(const std::string& seed) {
  std::string s = seed;
  const int ops = 1 + (int)(rnd() % 4);
  for (int o = 0; o < ops && !s.empty(); ++o) {
    const size_t pos = rnd() % s.size();
    switch (rnd() % 7) {
      case 0: s[pos] = (char)(rnd() & 0xff); break;                       // a random byte
      case 1: s[pos] = (char)(s[pos] ^ (1 << (rnd() % 8))); break;        // one bit
      case 2: {                                                           // an extreme little-endian word (lengths, offsets, counts)
        static const unsigned long long ext[] = {0ull, 1ull, 0x7fffffffull, 0x80000000ull, 0xffffffffull, 0x7fffffffffffffffull, ~0ull, 0xffffull};
        const unsigned long long v = ext[rnd() % 8];
        const size_t w = (rnd() & 1) ? 8 : 4;
        for (size_t b = 0; b < w && pos + b < s.size(); ++b) s[pos + b] = (char)((v >> (8 * b)) & 0xff);
        break;
      }
      case 3: s.resize(pos); break;                                       // truncation
      case 4: {                                                           // a chunk copied over another place
        const size_t len = 1 + rnd() % 64, from = rnd() % s.size();
        for (size_t b = 0; b < len && pos + b < s.size() && from + b < s.size(); ++b) s[pos + b] = s[from + b];
        break;
      }
      case 5: s.insert(pos, std::string(1 + rnd() % 16, (char)(rnd() & 0xff))); break;  // inserted bytes
      case 6: s.erase(pos, 1 + rnd() % 16); break;                        // removed bytes
    }
  }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: fuzz_formats text|model|hdf5 <seed file> <iterations> [rng seed]\n");
    return 2;
  }
  const std::string kind = argv[1];
  const std::string seed = dc::read_file(argv[2]);
  const long iters = std::atol(argv[3]);
  if (argc > 4) rng_state ^= std::strtoull(argv[4], nullptr, 10) * 0x9e3779b97f4a7c15ull;
  const std::string tmp = std::string("/tmp/fuzz_formats_") + std::to_string((long)getpid()) + (kind == "hdf5" ? ".h5" : ".bin");
  long parsed = 0, refused = 0;
  double worst_ms = 0;
  for (long i = 0; i < iters; ++i) {
    const std::string m = i == 0 ? seed : mutate(seed);  // (iteration 0: the seed itself must parse)
    const auto t0 = std::chrono::steady_clock::now();
    try {
      if (kind == "text") {
        dc::TextMsg msg = dc::parse_text_proto(m);
        (void)msg.subs("layer");
      } else {
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return 3;
        std::fwrite(m.data(), 1, m.size(), f);
        std::fclose(f);
        dc::ModelFile mf = kind == "hdf5" ? dc::read_hdf5_weights(tmp) : dc::read_caffemodel(tmp);
        size_t total = 0;
        for (auto& l : mf.layers)
          for (auto& b : l.blobs) total += b.data.size();
        // what comes out cannot be more than what went in (4 bytes per float; the V0 / double forms are no denser)
        if (total * 2 > m.size() + 64) {
          std::fprintf(stderr, "iteration %ld: %zu floats out of a %zu-byte file\n", i, total, m.size());
          return 1;
        }
      }
      ++parsed;
    } catch (const std::exception&) {
      ++refused;
      if (i == 0) {
        std::fprintf(stderr, "the seed itself does not parse\n");
        return 1;
      }
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > worst_ms) worst_ms = ms;
    if (ms > 2000.0) {
      std::fprintf(stderr, "iteration %ld: a %zu-byte mutant took %.0f ms\n", i, m.size(), ms);
      return 1;
    }
  }
  std::remove(tmp.c_str());
  std::printf("%s: %ld mutants, %ld parsed, %ld refused, slowest %.1f ms\n", kind.c_str(), iters, parsed, refused, worst_ms);
  return 0;
}
