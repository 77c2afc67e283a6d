// net_tune.cpp — see net_internal.h: tile variants chosen by measurement, the tune-cache file, set_tile / reports.
#include <unistd.h>

#include "net_internal.h"

namespace dc {

// DC_TUNE_CACHE=<file>: "<signature> <tile>" per line; group signatures (NetGroup) concatenate their members' and run to several
// hundred characters, so lines are read whole and cut at the LAST blank.  Entries of the file never overwrite what this process
// measured or was told (set_tile) since: the in-memory table wins.  Caller holds shared.mu.
void load_tune_cache_locked(ModelShared& shared) {
  if (shared.tune_file_loaded) return;
  shared.tune_file_loaded = true;
  const char* cache_path = std::getenv("DC_TUNE_CACHE");
  if (!cache_path || !*cache_path) return;
  FILE* f = std::fopen(cache_path, "r");
  if (!f) return;
  std::string line;
  int ch;
  auto take = [&]() {
    const size_t sp = line.find_last_of(' ');
    if (sp != std::string::npos && sp > 0 && sp + 1 < line.size()) {
      const std::string key = line.substr(0, sp), vname = line.substr(sp + 1);
      int v = -1;
      v = wino_variant_by_name(vname.c_str());
      for (int i = 0; v < 0 && i < conv_num_variants(); ++i)
        if (vname == conv_variant(i).name) v = i;
      if (v >= 0 && !shared.tune_cache.count(key)) shared.tune_cache[key] = v;
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

// The file is always the union of what it held and what this process knows: it is LOADED first (a process that only ever calls
// set_tile, or runs with DC_AUTOTUNE=0, used to truncate an existing cache to its one override: round-4 advice), and it is replaced
// atomically (temporary file + rename: several ranks share one file, a reader must never see half of it).
void write_tune_cache_locked(ModelShared& shared) {
  const char* cache_path = std::getenv("DC_TUNE_CACHE");
  if (!cache_path || !*cache_path) return;
  load_tune_cache_locked(shared);
  const std::string tmp = std::string(cache_path) + ".tmp." + std::to_string((long)getpid());
  FILE* f = std::fopen(tmp.c_str(), "w");
  if (!f) return;
  for (auto& kv : shared.tune_cache)
    std::fprintf(f, "%s %s\n", kv.first.c_str(), is_wino_variant(kv.second) ? wino_variant_name(kv.second) : conv_variant(kv.second).name);
  const bool ok = std::fflush(f) == 0;
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), cache_path) != 0) std::remove(tmp.c_str());
}

void Net::autotune() {
  tuned = true;
  if (env_int("DC_AUTOTUNE", 1) == 0 || env_int("DC_CONV_VARIANT", -1) >= 0) return;
  // DC_TUNE_CACHE=<file>: tuning results persist across processes ("signature variant-name" per line), so
  // a service (or a profiling run) starts without the timing launches
  std::lock_guard<std::mutex> tune_lock(shared->mu);  // one executor times a shape, the clones reuse its choices
  std::map<std::string, int>& tune_cache_ = shared->tune_cache;
  load_tune_cache_locked(*shared);
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
    if (l.wino_w)  // the Winograd forms of this layer (8 and 16 waves per workgroup) compete with the direct tiles
      for (int wv : {kWinoVariant, kWinoVariant16, kWinoHalf, kStreamHalf, kStemHalf, kStreamFloat, kStemFloat}) {
        if (!l.takes_wino(wv)) continue;
        Launch trial = l;
        trial.variant = wv;
        c.push_back({burst_ms(trial), wv});
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
  // (2b) the two Winograd forms against each other, by whole passes over the plan.  They differ in how well a workgroup hides
  // its own latencies (16 waves per workgroup: four per SIMD on launches of at most one workgroup per CU), which shows where a
  // launch starts behind another kernel's tail on cold caches and hardly at all in a burst of identical launches or between
  // hipEvents (res4 3x3 at 544x736, batch 1: 15.6 against 15.8 us timed alone, 1.25 us per launch inside the forward): where one
  // of them was chosen and the other is eligible, both run in whole passes (no events inside) and the faster pass stays.
  if (timed_any && env_int("DC_TUNE_INSITU", 1) != 0) {
    std::vector<Launch> saved = plan;
    struct Restore {
      std::vector<Launch>& plan;
      std::vector<Launch>& saved;
      ~Restore() { plan = saved; }
    } restore{plan, saved};
    for (auto& l : plan) {  // the passes run with the tiles chosen so far (what (3) will put into the plan)
      if (l.kind != Launch::CONV) continue;
      auto it = tune_cache_.find(key_of(l));
      if (it != tune_cache_.end() && !(is_wino_variant(it->second) && !l.takes_wino(it->second)) &&
          !(l.cg.ncls > 1 && (is_wino_variant(it->second) || !conv_variant_multiclass(it->second))))
        l.variant = it->second;
    }
    auto pass_ms = [&]() {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {  // (the first pass warms)
        HIPCHECK(hipEventRecord(e0, (hipStream_t)stream));
        for (auto& l : plan) run_launch(l, stream);
        HIPCHECK(hipEventRecord(e1, (hipStream_t)stream));
        HIPCHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
      }
      return best;
    };
    for (auto& kv : timed) {
      auto it = tune_cache_.find(kv.first);
      if (it == tune_cache_.end() || !is_wino_variant(it->second) || it->second == kWinoHalf || it->second == kStreamHalf || it->second == kStemHalf || it->second == kStreamFloat || it->second == kStemFloat) continue;
      bool both = false;
      for (auto& c : kv.second) both = both || (is_wino_variant(c.second) && c.second != it->second);
      if (!both) continue;
      float ms[2];
      const int forms[2] = {kWinoVariant, kWinoVariant16};
      for (int f = 0; f < 2; ++f) {
        for (auto& l : plan)
          if (l.kind == Launch::CONV && l.wino_w && key_of(l) == kv.first) l.variant = forms[f];
        ms[f] = pass_ms();
      }
      it->second = ms[1] < ms[0] ? kWinoVariant16 : kWinoVariant;
      for (auto& l : plan)  // (later signatures are compared with this one's choice in place)
        if (l.kind == Launch::CONV && l.wino_w && key_of(l) == kv.first) l.variant = it->second;
    }
  }
  // (3) the choices go into the plan
  for (auto& l : plan) {
    if (l.kind != Launch::CONV) continue;
    const ConvGemmParams& g = l.cg;
    auto it = tune_cache_.find(key_of(l));
    // a cache line naming the Winograd form while it is switched off (or not eligible any more): keep the cost model's tile
    if (it != tune_cache_.end() && !(is_wino_variant(it->second) && !l.takes_wino(it->second)) &&
        !(g.ncls > 1 && (is_wino_variant(it->second) || !conv_variant_multiclass(it->second))))
      l.variant = it->second;
    if (is_wino_variant(l.variant)) {
      l.kernel = wino_kernel_label(l.variant);
      l.grid = wino_grid(l.cg);
    } else {
      l.kernel = std::string("conv_gemm<") + conv_variant(l.variant).name + ">";
      l.grid = conv_grid(l.cg, l.variant);
    }
  }
  if (timed_any) ++stats.autotune_runs;
  if (tune_cache_.size() != cached_before || timed_any) write_tune_cache_locked(*shared);
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
  auto vname = [](int v) { return std::string(is_wino_variant(v) ? wino_variant_name(v) : conv_variant(v).name); };
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
  int v = wino_variant_by_name(tile.c_str());
  for (int i = 0; v < 0 && i < conv_num_variants(); ++i)
    if (tile == conv_variant(i).name) v = i;
  if (v < 0) throw DcError(DC_EINVAL, "no tile variant named '" + tile + "'");
  bool any = false;
  for (auto& l : plan) {
    if (l.kind != Launch::CONV || tune_key(l) != key) continue;
    const ConvGemmParams& g = l.cg;
    const bool ok = is_wino_variant(v) ? l.takes_wino(v)
                                      : g.klen % conv_variant_bk(v) == 0 && conv_variant_esize(v) == g.esize && (g.ncls <= 1 || conv_variant_multiclass(v));
    if (!ok) throw DcError(DC_EUNSUP, "tile '" + tile + "' cannot take launch '" + l.label + "' (" + key + ")");
    any = true;
  }
  if (!any) throw DcError(DC_EINVAL, "the current plan has no launch with signature '" + key + "'");
  for (auto& l : plan) {
    if (l.kind != Launch::CONV || tune_key(l) != key) continue;
    l.variant = v;
    if (is_wino_variant(v)) {
      l.kernel = wino_kernel_label(v);
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

}  // namespace dc
