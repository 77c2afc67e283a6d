// conv_probe — stand-alone A/B harness for the gather-GEMM tile variants of kernels.hip (no Python, no torch:
// starts in milliseconds, so one GPU call can time every variant on every layer shape).
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probes/conv_probe.cpp deepcut-cnn_amd/lib/kernels.hip.o -o tools/probes/bin/conv_probe
//   tools/probes/bin/conv_probe [--dtype h|f] [--batch 8] [--shapes res4_3x3,res4_c,...] [--variants all|d|name,name] [--reps 30]
//
// For every shape: each eligible variant is (a) checked against a float64 CPU evaluation of 4096 sampled outputs, (b) compared
// element-wise with the first variant that ran, (c) timed with hipEvents over `reps` launches that rotate through several
// operand sets (so that the activations come from HBM / the memory-side cache as they do inside a forward, not from L2).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../deepcut-cnn_amd/csrc/kernels.h"

using namespace dc;

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                    \
    }                                                                                  \
  } while (0)

struct Shape {
  const char* name;
  int H, W, Cin, Cout, k, dil, resid, relu;
};
// the layer shapes of the 544x736 forward (stride-1 forms; H, W = the layer's map)
static const Shape kShapes[] = {
    {"res4_3x3", 34, 46, 256, 256, 3, 1, 0, 1},    // 36x, 21 % of the float16 batch-8 forward
    {"res4_c", 34, 46, 256, 1024, 1, 1, 1, 1},     // 36x, + shortcut: bandwidth class
    {"res4_a", 34, 46, 1024, 256, 1, 1, 0, 1},     // 35x
    {"res3_3x3", 68, 92, 128, 128, 3, 1, 0, 1},    // 8x
    {"res3_c", 68, 92, 128, 512, 1, 1, 1, 1},      // 8x
    {"res3_a", 68, 92, 512, 128, 1, 1, 0, 1},      // 7x
    {"res5_3x3", 34, 46, 512, 512, 3, 2, 0, 1},    // 3x, dilation 2
    {"res5_c", 34, 46, 512, 2048, 1, 1, 1, 1},     // 3x
    {"res5_a", 34, 46, 2048, 512, 1, 1, 0, 1},     // 2x
    {"res2_3x3", 136, 184, 64, 64, 3, 1, 0, 1},    // 3x
    {"res2_c", 136, 184, 64, 256, 1, 1, 1, 1},     // 3x
    {"res2_a", 136, 184, 256, 64, 1, 1, 0, 1},     // 2x
    {"res5_b1", 34, 46, 1024, 2048, 1, 1, 0, 0},   // projection shortcut
    {"res4_c_nr", 34, 46, 256, 1024, 1, 1, 0, 1},  // res4_c without the shortcut (ablation: what the shortcut read costs)
    {"res3_c_nr", 68, 92, 128, 512, 1, 1, 0, 1},
    {"tiny_3x3", 9, 11, 64, 64, 3, 1, 1, 1},       // ragged everything: M = 99*NB, exercises the zero padding
    {"tiny_d2", 7, 13, 128, 96, 3, 2, 0, 0},       // dilation 2, Cout not a multiple of 64
    {"tiny_odd", 9, 11, 128, 70, 1, 1, 1, 1},      // Cout not a multiple of 8: element-wise epilogue
    {"tiny_w1", 9, 11, 64, 320, 1, 1, 1, 1},       // 3 n tiles of 128 (the last one half full), one K tile each
    {"tiny_w3", 9, 11, 64, 328, 3, 1, 1, 1},       // ... 3x3, 9 K tiles, ragged Cout
};

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

int main(int argc, char** argv) {
  std::string dtype = "h", shapes = "all", variants = "all";
  int batch = 8, reps = 30, nsets = 4, stamps = 0;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "--dtype") dtype = next();
    else if (a == "--batch") batch = std::atoi(next().c_str());
    else if (a == "--shapes") shapes = next();
    else if (a == "--variants") variants = next();
    else if (a == "--reps") reps = std::atoi(next().c_str());
    else if (a == "--sets") nsets = std::atoi(next().c_str());
    else if (a == "--stamps") stamps = 1;  // device-side phase stamps (ConvGemmParams::dbg) of every variant run
  }
  const int es = dtype == "h" ? 2 : 4;
  auto want = [](const std::string& list, const std::string& name, bool prefix_ok) {
    if (list == "all") return true;
    size_t pos = 0;
    while (pos <= list.size()) {
      size_t e = list.find(',', pos);
      if (e == std::string::npos) e = list.size();
      const std::string tok = list.substr(pos, e - pos);
      if (tok == name || (prefix_ok && !tok.empty() && name.compare(0, tok.size(), tok) == 0)) return true;
      pos = e + 1;
    }
    return false;
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);

  for (const Shape& sh : kShapes) {
    if (!want(shapes, sh.name, false)) continue;
    const int NB = batch, H = sh.H, W = sh.W, C = sh.Cin, OC = sh.Cout, k = sh.k, d = sh.dil;
    const int taps = k * k, Ktot = taps * C, M = NB * H * W, pad = (k / 2) * d;
    const size_t xn = (size_t)M * C, wn = (size_t)OC * Ktot, yn = (size_t)M * OC;
    // host data (values kept as floats already rounded to the device element type)
    std::vector<float> hx(xn), hw(wn), hr(sh.resid ? yn : 0), hs(OC), hb(OC);
    const float wsc = 1.f / std::sqrt((float)Ktot);
    auto rnd = [&](float s) { float v = U(rng) * s; return es == 2 ? h2f(f2h(v)) : v; };
    for (auto& v : hx) v = rnd(1.f);
    for (auto& v : hw) v = rnd(wsc * 1.7f);
    for (auto& v : hr) v = rnd(1.f);
    for (int c = 0; c < OC; ++c) hs[c] = 0.5f + 0.5f * (U(rng) + 1.f), hb[c] = U(rng);
    auto upload = [&](const std::vector<float>& h) -> void* {
      void* dptr = nullptr;
      if (h.empty()) return dptr;
      CK(hipMalloc(&dptr, h.size() * es));
      if (es == 4) CK(hipMemcpy(dptr, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      else {
        std::vector<unsigned short> t(h.size());
        for (size_t i = 0; i < h.size(); ++i) t[i] = f2h(h[i]);
        CK(hipMemcpy(dptr, t.data(), t.size() * 2, hipMemcpyHostToDevice));
      }
      return dptr;
    };
    std::vector<void*> dx(nsets), dr(nsets), dy(nsets);
    for (int s = 0; s < nsets; ++s) {
      dx[s] = upload(hx);
      dr[s] = upload(hr);
      CK(hipMalloc(&dy[s], yn * es));
    }
    void* dw = upload(hw);
    float *dsc, *dsh;
    CK(hipMalloc(&dsc, OC * 4));
    CK(hipMalloc(&dsh, OC * 4));
    CK(hipMemcpy(dsc, hs.data(), OC * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsh, hb.data(), OC * 4, hipMemcpyHostToDevice));

    ConvGemmParams g;
    std::memset(&g, 0, sizeof g);
    g.esize = es;
    g.x_img_stride = (long)H * W * C, g.x_row_stride = W * C, g.x_rows = H, g.x_rowlen = W * C;
    g.sy = 1, g.sx = C;
    g.nty = k, g.ntx = k, g.dy0 = -pad, g.ddy = d, g.x0 = -pad * C, g.ddx = d * C;
    g.klen = C, g.Ktot = Ktot;
    g.w = dw;
    g.NB = NB, g.OH = H, g.OW = W, g.M = M, g.Cout = OC;
    g.y_img_stride = (long)H * W * OC, g.y_row_stride = W * OC, g.y_pix_stride = OC;
    g.scale = dsc, g.shift = dsh, g.relu = sh.relu, g.sigmoid_ch = 0;
    g.ncls = 1;

    // sampled float64 reference
    const int NS = 4096;
    std::vector<int> sm(NS), sc(NS);
    std::vector<double> sref(NS);
    for (int i = 0; i < NS; ++i) {
      const int m = i < 64 ? (i & 1 ? M - 1 - i : i) : (int)(rng() % (unsigned)M), co = i < 8 ? OC - 1 - i % OC : (int)(rng() % (unsigned)OC);
      sm[i] = m, sc[i] = co;
      const int n = m / (H * W), oy = (m / W) % H, ox = m % W;
      double acc = 0;
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          const int iy = oy - pad + ky * d, ix = ox - pad + kx * d;
          if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
          const float* xp = &hx[((size_t)(n * H + iy) * W + ix) * C];
          const float* wp = &hw[(size_t)co * Ktot + (ky * k + kx) * C];
          for (int c = 0; c < C; ++c) acc += (double)xp[c] * wp[c];
        }
      double v = acc * hs[co] + hb[co] + (sh.resid ? hr[(size_t)m * OC + co] : 0.0);
      if (sh.relu && v < 0) v = 0;
      sref[i] = v;
    }

    const double flops = 2.0 * M * (double)OC * Ktot;
    std::printf("== %s  NB=%d %dx%d  %d->%d  k%d d%d%s%s   M=%d N=%d K=%d  %.2f GFLOP  min bytes %.1f MB\n", sh.name, NB, H, W, C, OC, k, d,
                sh.resid ? " +resid" : "", sh.relu ? " +relu" : "", M, OC, Ktot, flops * 1e-9,
                (xn + wn + yn * (sh.resid ? 2 : 1)) * (double)es * 1e-6);
    std::vector<unsigned short> yref16;
    std::vector<float> yref32;
    for (int v = 0; v < conv_num_variants(); ++v) {
      if (conv_variant_esize(v) != es || C % conv_variant_bk(v) != 0) continue;
      const std::string vn = conv_variant(v).name;
      if (!want(variants, vn, true)) continue;
      bool refused = false;
      auto run = [&](int s) {
        g.x = dx[s], g.resid = sh.resid ? dr[s] : nullptr, g.y = dy[s];
        const int rc = launch_conv_gemm(g, v, st);
        if (rc == (int)hipErrorInvalidValue) refused = true;  // the variant does not take this geometry (the library falls back)
        else if (rc != 0) {
          std::fprintf(stderr, "launch %s failed: %d\n", vn.c_str(), rc);
          std::exit(3);
        }
      };
      if (getenv("PROBE_TRACE")) std::fprintf(stderr, "[probe] %s %s\n", sh.name, vn.c_str());
      CK(hipMemsetAsync(dy[0], 0xff, yn * es, st));
      run(0);
      CK(hipStreamSynchronize(st));
      if (refused) {
        std::printf("  %-26s (refuses this geometry)\n", vn.c_str());
        continue;
      }
      // correctness
      double maxerr = 0, maxdiff = 0;
      size_t ndiff = 0;
      if (es == 2) {
        std::vector<unsigned short> y(yn);
        CK(hipMemcpy(y.data(), dy[0], yn * 2, hipMemcpyDeviceToHost));
        for (int i = 0; i < NS; ++i) {
          const double got = h2f(y[(size_t)sm[i] * OC + sc[i]]);
          const double err = std::fabs(got - sref[i]) / (1.0 + std::fabs(sref[i]));
          if (!(err <= maxerr)) maxerr = err;
        }
        if (yref16.empty()) yref16 = y;
        else
          for (size_t i = 0; i < yn; ++i)
            if (y[i] != yref16[i]) {
              ++ndiff;
              const double df = std::fabs((double)h2f(y[i]) - h2f(yref16[i]));
              if (!(df <= maxdiff)) maxdiff = df;
            }
      } else {
        std::vector<float> y(yn);
        CK(hipMemcpy(y.data(), dy[0], yn * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < NS; ++i) {
          const double err = std::fabs((double)y[(size_t)sm[i] * OC + sc[i]] - sref[i]) / (1.0 + std::fabs(sref[i]));
          if (!(err <= maxerr)) maxerr = err;
        }
        if (yref32.empty()) yref32 = y;
        else
          for (size_t i = 0; i < yn; ++i) {
            const double df = std::fabs((double)y[i] - yref32[i]);
            if (df > 1e-4) ++ndiff;
            if (!(df <= maxdiff)) maxdiff = df;
          }
      }
      // timing
      for (int r = 0; r < 3; ++r) run(r % nsets);
      float best = 1e30f;
      for (int burst = 0; burst < 2; ++burst) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) run(r % nsets);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      const double us = best * 1e3 / reps;
      const double tol = es == 2 ? 1.5e-3 : 2e-5;
      std::printf("  %-26s grid %5ld  %8.2f us  %7.1f TF/s   err_vs_f64 %.2e  vs_first: %zu differ, max %.3g  %s\n", vn.c_str(), conv_grid(g, v), us,
                  flops / us * 1e-6, maxerr, ndiff, maxdiff, maxerr <= tol ? "ok" : "**WRONG**");
      if (stamps) {
        // per wave: shader-cycle stamps at 8 phase boundaries (slots 0..7), chip-wide 100 MHz clock at entry / start / end (8, 10, 9)
        const ConvVariant& cv = conv_variant(v);
        const int nwv = cv.WR * cv.WC * cv.WK;
        const long nw = (conv_grid(g, v) * 2 + 64) * nwv;
        long long* d = nullptr;
        CK(hipMalloc((void**)&d, nw * 12 * sizeof(long long)));
        CK(hipMemset(d, 0, nw * 12 * sizeof(long long)));
        g.dbg = d;
        run(1 % nsets);
        run(2 % nsets);
        CK(hipStreamSynchronize(st));
        g.dbg = nullptr;
        std::vector<long long> h(nw * 12);
        CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        CK(hipFree(d));
        double dsum[8] = {0};
        long cnt = 0;
        long long t0min = 0, t0max = 0, t9max = 0;
        for (long i = 0; i < nw; ++i) {
          const long long* w = &h[i * 12];
          if (w[7] == 0 || w[0] == 0) continue;
          for (int k = 1; k < 8; ++k) dsum[k] += (double)(w[k] - w[k - 1]);
          if (!cnt || w[8] < t0min) t0min = w[8];
          if (!cnt || w[8] > t0max) t0max = w[8];
          if (!cnt || w[9] > t9max) t9max = w[9];
          ++cnt;
        }
        const double c = (double)std::max(cnt, 1L);
        std::printf("      stamps (mean cycles/wave): B-issue %.0f | decode+A-issue %.0f | rowinfo %.0f | wait tile0 %.0f | K loop %.0f | split-K %.0f | epilogue %.0f"
                    " || total %.0f | ramp %.2f us | span %.2f us\n",
                    dsum[1] / c, dsum[2] / c, dsum[3] / c, dsum[4] / c, dsum[5] / c, dsum[6] / c, dsum[7] / c,
                    (dsum[1] + dsum[2] + dsum[3] + dsum[4] + dsum[5] + dsum[6] + dsum[7]) / c, (t0max - t0min) / 100.0, (t9max - t0min) / 100.0);
      }
      std::fflush(stdout);
    }
    for (int s = 0; s < nsets; ++s) {
      CK(hipFree(dx[s]));
      if (dr[s]) CK(hipFree(dr[s]));
      CK(hipFree(dy[s]));
    }
    CK(hipFree(dw));
    CK(hipFree(dsc));
    CK(hipFree(dsh));
  }
  return 0;
}
