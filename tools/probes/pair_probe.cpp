// pair_probe — stand-alone check + timing of the bottleneck-pair kernel (tools/probes/pair_kernel.hip):
// res<i>_branch2c + shortcut + ReLU -> res<i+1>_branch2a + ReLU as ONE launch, the 4*WD-wide intermediate fed to the second
// GEMM from registers.  No Python, no torch: one GPU call checks ragged and multi-tensor cases and times the layer shapes.
//
//   tools/probes/build_pair_probe.sh
//   tools/probes/bin/pair_probe [--wd 256|128|all] [--m 12512,3128,...] [--reps 30] [--check-rows 96]
//
// Check: for sampled pixel rows (first / last rows of every tile boundary included) Y and Z are recomputed on the host in float
// from the SAME float16 inputs (Y rounded to float16 before the second product, as the kernel and the two-launch path do).
// Timing: hipEvents over `reps` launches rotating through 4 operand sets (activations come from HBM / the memory-side cache).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "pair_kernel.h"

using namespace dc;

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                          \
    }                                                                                        \
  } while (0)

static unsigned short f2h(float f) {
  _Float16 h = (_Float16)f;
  unsigned short u;
  std::memcpy(&u, &h, 2);
  return u;
}
static float h2f(unsigned short u) {
  _Float16 h;
  std::memcpy(&h, &u, 2);
  return (float)h;
}
static float rh(float f) { return h2f(f2h(f)); }

struct Set {
  unsigned short *b, *s, *y, *z;
};

static std::vector<int> parse_list(const std::string& s) {
  std::vector<int> v;
  size_t p = 0;
  while (p < s.size()) {
    size_t q = s.find(',', p);
    if (q == std::string::npos) q = s.size();
    v.push_back(std::atoi(s.substr(p, q - p).c_str()));
    p = q + 1;
  }
  return v;
}

// one case: `Ms` = the tensors of the launch (one problem each)
static int g_stamps = 0, g_nocheck = 0, g_p2c = 0;
static int run_case(int WD, const std::vector<int>& Ms, int reps, int check_rows, bool timing) {
  const int NC = 4 * WD;
  std::mt19937 rng(1234 + WD + (int)Ms.size());
  std::normal_distribution<float> nd(0.f, 1.f);
  // filters and constants
  std::vector<float> w1((size_t)NC * WD), w2((size_t)WD * NC), ab1(2 * NC), ab2(2 * WD);
  for (auto& v : w1) v = rh(nd(rng) * std::sqrt(2.f / WD));
  for (auto& v : w2) v = rh(nd(rng) * std::sqrt(2.f / NC));
  for (int c = 0; c < NC; ++c) ab1[c] = 0.2f * (1.f + 0.1f * nd(rng)), ab1[NC + c] = 0.05f * nd(rng);
  for (int c = 0; c < WD; ++c) ab2[c] = 1.f + 0.1f * nd(rng), ab2[WD + c] = 0.05f * nd(rng);
  std::vector<unsigned short> packed(pair_packed_halves(WD));
  pair_pack_filters(w1.data(), w2.data(), WD, packed.data());
  unsigned short* d_w;
  float *d_ab1, *d_ab2;
  CK(hipMalloc(&d_w, packed.size() * 2));
  CK(hipMalloc(&d_ab1, ab1.size() * 4));
  CK(hipMalloc(&d_ab2, ab2.size() * 4));
  CK(hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ab1, ab1.data(), ab1.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ab2, ab2.data(), ab2.size() * 4, hipMemcpyHostToDevice));

  const int nsets = timing ? 4 : 1;
  long Mtot = 0;
  for (int m : Ms) Mtot += m;
  // host operands of set 0 (the checked one), per problem
  std::vector<std::vector<unsigned short>> hb(Ms.size()), hs(Ms.size());
  for (size_t k = 0; k < Ms.size(); ++k) {
    hb[k].resize((size_t)Ms[k] * WD);
    hs[k].resize((size_t)Ms[k] * NC);
    for (auto& v : hb[k]) v = f2h(std::max(0.f, nd(rng)));  // branch2b's output is post-ReLU
    for (auto& v : hs[k]) v = f2h(std::max(0.f, nd(rng)));
  }
  std::vector<std::vector<Set>> sets(nsets, std::vector<Set>(Ms.size()));
  for (int t = 0; t < nsets; ++t)
    for (size_t k = 0; k < Ms.size(); ++k) {
      Set& q = sets[t][k];
      CK(hipMalloc(&q.b, hb[k].size() * 2 + 16));
      CK(hipMalloc(&q.s, hs[k].size() * 2 + 16));
      CK(hipMalloc(&q.y, hs[k].size() * 2 + 16));
      CK(hipMalloc(&q.z, hb[k].size() * 2 + 16));
      CK(hipMemcpy(q.b, hb[k].data(), hb[k].size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(q.s, hs[k].data(), hs[k].size() * 2, hipMemcpyHostToDevice));
      CK(hipMemset(q.y, 0xff, hs[k].size() * 2 + 16));  // NaN pattern: an unwritten output shows
      CK(hipMemset(q.z, 0xff, hb[k].size() * 2 + 16));
    }
  std::vector<PairArgs> args(nsets);
  long grid = 0;
  for (int t = 0; t < nsets; ++t) {
    PairArgs& a = args[t];
    std::memset(&a, 0, sizeof(a));
    a.w = d_w;
    a.ab1 = d_ab1;
    a.ab2 = d_ab2;
    a.WD = WD;
    a.relu2 = 1;
    a.nprob = (int)Ms.size();
    for (size_t k = 0; k < Ms.size(); ++k) {
      a.prob[k].b = sets[t][k].b;
      a.prob[k].s = sets[t][k].s;
      a.prob[k].y = sets[t][k].y;
      a.prob[k].z = sets[t][k].z;
      a.prob[k].M = Ms[k];
    }
    grid = pair_grid(a);
  }
  int rc = launch_pair_gemm(args[0], grid, nullptr);
  if (rc) {
    std::printf("WD %d: launch failed: %s\n", WD, hipGetErrorString((hipError_t)rc));
    return 1;
  }
  CK(hipDeviceSynchronize());

  // ---- check
  int bad = 0;
  double worst_y = 0, worst_z = 0;
  for (size_t k = 0; k < Ms.size(); ++k) {
    const int M = Ms[k];
    std::vector<unsigned short> gy((size_t)M * NC + 8), gz((size_t)M * WD + 8);
    CK(hipMemcpy(gy.data(), sets[0][k].y, gy.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gz.data(), sets[0][k].z, gz.size() * 2, hipMemcpyDeviceToHost));
    for (int e = 0; e < 8; ++e)  // the guard bytes behind the tensors must be untouched
      if (gy[(size_t)M * NC + e] != 0xffff || gz[(size_t)M * WD + e] != 0xffff) {
        if (bad++ < 5) std::printf("  WRONG: problem %zu wrote behind its output\n", k);
      }
    std::vector<int> rows;
    for (int r = 0; r < M && r < 3; ++r) rows.push_back(r);
    for (int r = std::max(0, M - 3); r < M; ++r) rows.push_back(r);
    for (int t = 128; t < M; t += 128 * std::max(1, M / 128 / 8)) rows.push_back(t - 1), rows.push_back(t);
    for (int i = 0; i < check_rows; ++i) rows.push_back((int)(rng() % (unsigned)M));
    std::vector<float> yr(NC), bf(WD);
    for (int r : rows) {
      for (int c = 0; c < WD; ++c) bf[c] = h2f(hb[k][(size_t)r * WD + c]);
      for (int n = 0; n < NC; ++n) {
        float acc = 0.f;
        const float* wr = &w1[(size_t)n * WD];
        for (int c = 0; c < WD; ++c) acc += bf[c] * wr[c];
        const float v = std::max(0.f, acc * ab1[n] + ab1[NC + n] + h2f(hs[k][(size_t)r * NC + n]));
        yr[n] = rh(v);
        const float got = h2f(gy[(size_t)r * NC + n]);
        const double err = std::fabs((double)got - v);
        worst_y = std::max(worst_y, err / std::max(1.0, (double)std::fabs(v)));
        if (!(err <= 4e-3 * std::max(1.0, (double)std::fabs(v)))) {
          if (bad++ < 8) std::printf("  WRONG Y: problem %zu row %d ch %d: got %g want %g\n", k, r, n, got, v);
        }
      }
      // the second product from the KERNEL's own Y (what it fed GEMM 2 with), so that one rounding flip in Y is not counted twice
      for (int n = 0; n < NC; ++n) yr[n] = h2f(gy[(size_t)r * NC + n]);
      for (int n = 0; n < WD; ++n) {
        float acc = 0.f;
        const float* wr = &w2[(size_t)n * NC];
        for (int c = 0; c < NC; ++c) acc += yr[c] * wr[c];
        const float v = std::max(0.f, acc * ab2[n] + ab2[WD + n]);
        const float got = h2f(gz[(size_t)r * WD + n]);
        const double err = std::fabs((double)got - v);
        worst_z = std::max(worst_z, err / std::max(1.0, (double)std::fabs(v)));
        if (!(err <= 4e-3 * std::max(1.0, (double)std::fabs(v)))) {
          if (bad++ < 8) std::printf("  WRONG Z: problem %zu row %d ch %d: got %g want %g\n", k, r, n, got, v);
        }
      }
    }
  }
  std::string ms;
  for (int m : Ms) ms += (ms.empty() ? "" : "+") + std::to_string(m);
  std::printf("WD %3d  M %-28s grid %5ld  check: worst rel err Y %.2e Z %.2e  %s\n", WD, ms.c_str(), grid, worst_y, worst_z, bad ? "WRONG" : "ok");

  if (g_nocheck) bad = 0;
  if (timing && !bad) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) launch_pair_gemm(args[i % nsets], grid, nullptr);
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) launch_pair_gemm(args[i % nsets], grid, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float msec = 0;
    CK(hipEventElapsedTime(&msec, e0, e1));
    const double us = msec * 1e3 / reps;
    const double flops = 2.0 * 2.0 * (double)Mtot * NC * WD;
    const double bytes = (double)Mtot * (WD + NC + NC + WD) * 2.0;
    std::printf("        %.2f us per launch   %.0f TFLOP/s (%.3f of 2.5 PF)   %.2f TB/s of algorithmic bytes (%.1f MB)\n", us, flops / us / 1e6,
                flops / us / 1e6 / 2500.0, bytes / us / 1e6, bytes / 1e6);
  }
  if (timing && !bad && g_p2c) {
    // the first layer alone in the persistent form: Y must equal the pair kernel's Y (same arithmetic), then timing
    std::vector<std::vector<unsigned short>> ypair(Ms.size());
    for (size_t k = 0; k < Ms.size(); ++k) {
      ypair[k].resize((size_t)Ms[k] * NC);
      CK(hipMemcpy(ypair[k].data(), sets[0][k].y, ypair[k].size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemset(sets[0][k].y, 0xff, ypair[k].size() * 2 + 16));
    }
    int rc2 = launch_p2c_gemm(args[0], grid, nullptr);
    CK(hipDeviceSynchronize());
    long diff = 0;
    for (size_t k = 0; k < Ms.size() && !rc2; ++k) {
      std::vector<unsigned short> y2((size_t)Ms[k] * NC + 8);
      CK(hipMemcpy(y2.data(), sets[0][k].y, y2.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ypair[k].size(); ++i) diff += y2[i] != ypair[k][i];
      for (int e = 0; e < 8; ++e) diff += y2[ypair[k].size() + e] != 0xffff;
    }
    if (rc2 || diff) {
      std::printf("        p2c: %s (%ld elements differ from the pair kernel's Y)  WRONG\n", rc2 ? hipGetErrorString((hipError_t)rc2) : "ran", diff);
      bad++;
    } else {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int i = 0; i < 4; ++i) launch_p2c_gemm(args[i % nsets], grid, nullptr);
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < reps; ++i) launch_p2c_gemm(args[i % nsets], grid, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float msec = 0;
      CK(hipEventElapsedTime(&msec, e0, e1));
      const double us = msec * 1e3 / reps;
      const double bytes = (double)Mtot * (WD + NC + NC) * 2.0;
      std::printf("        p2c (first layer alone, persistent form): Y identical; %.2f us per launch   %.0f TFLOP/s   %.2f TB/s of algorithmic bytes (%.1f MB)\n", us,
                  2.0 * Mtot * NC * WD / us / 1e6, bytes / us / 1e6, bytes / 1e6);
    }
  }
  if (timing && !bad && g_stamps) {
    long long* d_dbg;
    CK(hipMalloc(&d_dbg, (size_t)grid * 256 * 8));
    CK(hipMemset(d_dbg, 0, (size_t)grid * 256 * 8));
    PairArgs a = args[0];
    a.dbg = d_dbg;
    launch_pair_gemm(a, grid, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<long long> st((size_t)grid * 256);
    CK(hipMemcpy(st.data(), d_dbg, st.size() * 8, hipMemcpyDeviceToHost));
    for (long b : {0L, grid / 2, grid - 1}) {
      const long long* v = &st[(size_t)b * 256];
      int n = 0;
      while (n < 256 && v[n]) ++n;
      std::printf("        stamps of workgroup %ld (%d): total %lld cycles; per period [wait-vm | barrier | work]:\n         ", b, n, n ? v[n - 1] - v[0] : 0);
      // v[0] = start, then triples (before wait, after wait, after barrier) per period, last = end
      for (int i = 1; i + 3 < n; i += 3)
        std::printf(" %lld|%lld|%lld", v[i + 1] - v[i], v[i + 2] - v[i + 1], v[i + 3] - v[i + 2]);
      // averages: periods alternate A (GEMM 1) and B (GEMM 2 || epilogue 1)
      double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
      int na = 0, nb = 0;
      for (int i = 1, t = 0; i + 3 < n; i += 3, ++t) {
        double* d = (t & 1) ? sb : sa;
        d[0] += v[i + 1] - v[i], d[1] += v[i + 2] - v[i + 1], d[2] += v[i + 3] - v[i + 2];
        (t & 1) ? ++nb : ++na;
      }
      if (na && nb)
        std::printf("\n         average A period: wait %.0f barrier %.0f work %.0f | B period: wait %.0f barrier %.0f work %.0f | per chunk %.0f",
                    sa[0] / na, sa[1] / na, sa[2] / na, sb[0] / nb, sb[1] / nb, sb[2] / nb, (sa[0] + sa[1] + sa[2]) / na + (sb[0] + sb[1] + sb[2]) / nb);
      std::printf("\n         prologue (start -> first stamped period): %lld\n", n > 1 ? v[1] - v[0] : 0);
    }
    CK(hipFree(d_dbg));
  }
  for (auto& t : sets)
    for (auto& q : t) {
      CK(hipFree(q.b));
      CK(hipFree(q.s));
      CK(hipFree(q.y));
      CK(hipFree(q.z));
    }
  CK(hipFree(d_w));
  CK(hipFree(d_ab1));
  CK(hipFree(d_ab2));
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  std::string wd = "all", mlist = "";
  int reps = 30, check_rows = 64;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "--wd") wd = next();
    else if (a == "--m") mlist = next();
    else if (a == "--reps") reps = std::atoi(next().c_str());
    else if (a == "--check-rows") check_rows = std::atoi(next().c_str());
    else if (a == "--stamps") g_stamps = 1;
    else if (a == "--no-check") g_nocheck = 1;
    else if (a == "--p2c") g_p2c = 1;
  }
  int bad = 0;
  for (int WD : {256, 128}) {
    if (wd != "all" && std::atoi(wd.c_str()) != WD) continue;
    // correctness first, on ragged / tiny / multi-tensor cases (short: a barrier or counting bug must show before the long runs)
    bad += run_case(WD, {1}, 1, 4, false);
    bad += run_case(WD, {127}, 1, 16, false);
    bad += run_case(WD, {129}, 1, 16, false);
    bad += run_case(WD, {1000}, 1, 32, false);
    bad += run_case(WD, {391, 77, 128, 910}, 1, 32, false);
    if (bad && !g_nocheck) {
      std::printf("WRONG results: not timing\n");
      return 1;
    }
    if (!mlist.empty()) {
      bad += run_case(WD, parse_list(mlist), reps, check_rows, true);
      continue;
    }
    // the shapes of the float16 batch-8 forwards: M = 8 * (H/16) * (W/16) for conv4_x, 8 * (H/8) * (W/8) for conv3_x
    const int s4[4] = {8 * 17 * 23, 8 * 26 * 35, 8 * 34 * 46, 8 * 43 * 58};
    const int s3[4] = {8 * 34 * 46, 8 * 51 * 69, 8 * 68 * 92, 8 * 85 * 115};
    const int* sc = WD == 256 ? s4 : s3;
    for (int i = 0; i < 4; ++i) bad += run_case(WD, {sc[i]}, reps, check_rows, true);
    bad += run_case(WD, {sc[3], sc[0]}, reps, check_rows, true);            // the two lanes of a grouped pyramid
    bad += run_case(WD, {sc[2], sc[1]}, reps, check_rows, true);
    bad += run_case(WD, {sc[3], sc[2], sc[1], sc[0]}, reps, check_rows, true);  // one lane: the whole pyramid
  }
  return bad ? 1 : 0;
}
