/*
 * oracle/caffe_cpu.c — CPU restatement of the reference's TEST-phase layer arithmetic for the
 * DeeperCut forward path.  TEST INFRASTRUCTURE ONLY: it is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under
 * deepcut-cnn_amd/ links, imports or calls it.
 *
 * It follows the reference's ALGORITHM, not its fused MI355X replacement: NCHW float32, one image at
 * a time, im2col -> SGEMM (conv) / SGEMM -> col2im (deconv), bias as a rank-1 update, and separate
 * BatchNorm / Scale / ReLU / Eltwise / Crop / Sigmoid / Pooling passes.  Each function cites the
 * reference lines it restates (paths relative to the reference tree).
 *
 * Pinning: the reference cannot be compiled in this image (every translation unit needs
 * boost/glog/gflags/protobuf-generated headers and an external BLAS, none present, and stand-ins are
 * not allowed), so this oracle is pinned against the known-answer vectors of the reference's own unit
 * tests (tests/test_oracle_reference_vectors.py: deconvolution closed form, max-pool literals, ...)
 * and cross-checked against torch-CPU float32/float64 functional ops.  The reference tests do not pin
 * BatchNorm(use_global_stats), the fork's Crop or the whole DeeperCut graph: for those this file is a
 * line-by-line restatement only ("parity unpinned" — see DESIGN.md).
 *
 * The SGEMM stands for the reference's external BLAS (cblas_sgemm via caffe_cpu_gemm,
 * src/caffe/util/math_functions.cpp:12-21): float32 multiply-add, accumulation order unspecified by
 * the reference.  oracle_set_double_acc(1) switches the GEMM to float64 accumulation (a tighter
 * reference used to bound the float32 noise of both sides).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
static int g_double_acc = 0;

void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int oracle_get_threads(void) { return g_threads; }
void oracle_set_double_acc(int on) { g_double_acc = on; }
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- SGEMM: C[M,N] = op(A)[M,K] * B[K,N] + beta*C, row-major (caffe_cpu_gemm, math_functions.cpp:12-21;
 * call sites base_conv_layer.cpp:267-270 (CblasNoTrans) and :290-293 (CblasTrans)).  Stands for the
 * reference's external BLAS: a register-blocked 6x16 micro-kernel (float32 FMA, k-ordered accumulation
 * per output) over MC x NC cache blocks, OpenMP over the blocks. ------------------------------------- */
typedef float v8f __attribute__((vector_size(32), aligned(4)));

#define GEMM_MR 6
#define GEMM_NR 16
#define GEMM_KC 384
#define GEMM_MC 48
#define GEMM_NC 256

/* one MC x NC block of C: B is packed per KC slab into NR-wide panels (contiguous per k), then the 6x16
 * micro-kernel runs with its accumulators in registers. */
__attribute__((target_clones("avx2,fma", "default")))
static void gemm_block_f32(int transA, int M, int N, int K, const float* A, const float* B, float beta, float* C, int i0,
                           int i1, int j0, int j1, float* bp) {
  const int npan = (j1 - j0 + GEMM_NR - 1) / GEMM_NR;
  for (int k0 = 0; k0 < K; k0 += GEMM_KC) {
    const int kc = K - k0 < GEMM_KC ? K - k0 : GEMM_KC;
    const float bk = k0 == 0 ? beta : 1.f;
    for (int pn = 0; pn < npan; ++pn) { /* pack */
      const int j = j0 + pn * GEMM_NR;
      const int nr = j1 - j < GEMM_NR ? j1 - j : GEMM_NR;
      float* dst = bp + (size_t)pn * GEMM_KC * GEMM_NR;
      for (int k = 0; k < kc; ++k) {
        const float* src = B + (size_t)(k0 + k) * N + j;
        for (int q = 0; q < nr; ++q) dst[k * GEMM_NR + q] = src[q];
        for (int q = nr; q < GEMM_NR; ++q) dst[k * GEMM_NR + q] = 0.f;
      }
    }
    for (int i = i0; i < i1; i += GEMM_MR) {
      const int mr = i1 - i < GEMM_MR ? i1 - i : GEMM_MR;
      for (int pn = 0; pn < npan; ++pn) {
        const int j = j0 + pn * GEMM_NR;
        const int nr = j1 - j < GEMM_NR ? j1 - j : GEMM_NR;
        const float* b = bp + (size_t)pn * GEMM_KC * GEMM_NR;
        v8f acc[GEMM_MR][2];
        for (int r = 0; r < GEMM_MR; ++r) acc[r][0] = acc[r][1] = (v8f){0, 0, 0, 0, 0, 0, 0, 0};
        if (mr == GEMM_MR) {
          for (int k = 0; k < kc; ++k) {
            const v8f b0 = *(const v8f*)(b + k * GEMM_NR), b1 = *(const v8f*)(b + k * GEMM_NR + 8);
            for (int r = 0; r < GEMM_MR; ++r) {
              const float a = transA ? A[(size_t)(k0 + k) * M + i + r] : A[(size_t)(i + r) * K + k0 + k];
              const v8f av = {a, a, a, a, a, a, a, a};
              acc[r][0] += av * b0;
              acc[r][1] += av * b1;
            }
          }
        } else {
          for (int k = 0; k < kc; ++k) {
            const v8f b0 = *(const v8f*)(b + k * GEMM_NR), b1 = *(const v8f*)(b + k * GEMM_NR + 8);
            for (int r = 0; r < mr; ++r) {
              const float a = transA ? A[(size_t)(k0 + k) * M + i + r] : A[(size_t)(i + r) * K + k0 + k];
              const v8f av = {a, a, a, a, a, a, a, a};
              acc[r][0] += av * b0;
              acc[r][1] += av * b1;
            }
          }
        }
        for (int r = 0; r < mr; ++r) {
          float* c = C + (size_t)(i + r) * N + j;
          float tmp[GEMM_NR];
          *(v8f*)tmp = acc[r][0];
          *(v8f*)(tmp + 8) = acc[r][1];
          if (bk == 0.f)
            for (int q = 0; q < nr; ++q) c[q] = tmp[q];
          else
            for (int q = 0; q < nr; ++q) c[q] = bk * c[q] + tmp[q];
        }
      }
    }
  }
}

static void gemm_block_f64(int transA, int M, int N, int K, const float* A, const float* B, float beta, float* C, int i0,
                           int i1, int j0, int j1) {
  for (int i = i0; i < i1; ++i)
    for (int j = j0; j < j1; ++j) {
      double acc = beta == 0.f ? 0.0 : (double)beta * C[(size_t)i * N + j];
      for (int k = 0; k < K; ++k)
        acc += (double)(transA ? A[(size_t)k * M + i] : A[(size_t)i * K + k]) * (double)B[(size_t)k * N + j];
      C[(size_t)i * N + j] = (float)acc;
    }
}

void oracle_sgemm(int transA, int M, int N, int K, const float* A, const float* B, float beta, float* C) {
  const int bm = (M + GEMM_MC - 1) / GEMM_MC, bn = (N + GEMM_NC - 1) / GEMM_NC;
  const int nblk = bm * bn;
  int nt = g_threads < nblk ? g_threads : nblk;
  if (nt < 1) nt = 1;
#pragma omp parallel num_threads(nt)
  {
    float* bp = g_double_acc ? NULL : (float*)aligned_alloc(64, (size_t)GEMM_KC * GEMM_NC * sizeof(float));
#pragma omp for schedule(dynamic, 1)
    for (int blk = 0; blk < nblk; ++blk) {
      const int i0 = (blk / bn) * GEMM_MC, j0 = (blk % bn) * GEMM_NC;
      const int i1 = i0 + GEMM_MC < M ? i0 + GEMM_MC : M, j1 = j0 + GEMM_NC < N ? j0 + GEMM_NC : N;
      if (g_double_acc) gemm_block_f64(transA, M, N, K, A, B, beta, C, i0, i1, j0, j1);
      else gemm_block_f32(transA, M, N, K, A, B, beta, C, i0, i1, j0, j1, bp);
    }
    free(bp);
  }
}

/* ---- im2col / col2im (src/caffe/util/im2col.cpp:19-55 and :163-197) ---------------------------------
 * col is [C*kh*kw, OH*OW]; row (c, ky, kx) holds the input sampled at (oy*sh - ph + ky*dh, ox*sw - pw + kx*dw),
 * zero outside the image. */
static int conv_out(int in, int k, int p, int s, int d) { return (in + 2 * p - (d * (k - 1) + 1)) / s + 1; }

void oracle_im2col(const float* im, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw, int dh,
                   int dw, float* col) {
  const int OH = conv_out(H, kh, ph, sh, dh), OW = conv_out(W, kw, pw, sw, dw);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) {
        float* dst = col + (((size_t)c * kh + ky) * kw + kx) * OH * OW;
        for (int oy = 0; oy < OH; ++oy) {
          const int iy = oy * sh - ph + ky * dh;
          for (int ox = 0; ox < OW; ++ox) {
            const int ix = ox * sw - pw + kx * dw;
            dst[oy * OW + ox] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? im[((size_t)c * H + iy) * W + ix] : 0.f;
          }
        }
      }
}

/* im is the (larger) image of the deconvolution output: [C,H,W]; col is [C*kh*kw, OH*OW] where (OH,OW)
 * is the conv-output size of that image = the deconvolution INPUT size.  Overlap-add in the reference's
 * order: channel, kernel row, kernel col, then output rows/cols (im2col.cpp:176-195). */
void oracle_col2im(const float* col, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw, int dh,
                   int dw, float* im) {
  const int OH = conv_out(H, kh, ph, sh, dh), OW = conv_out(W, kw, pw, sw, dw);
  memset(im, 0, (size_t)C * H * W * sizeof(float));
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) {
        const float* src = col + (((size_t)c * kh + ky) * kw + kx) * OH * OW;
        for (int oy = 0; oy < OH; ++oy) {
          const int iy = oy * sh - ph + ky * dh;
          if (iy < 0 || iy >= H) continue;
          for (int ox = 0; ox < OW; ++ox) {
            const int ix = ox * sw - pw + kx * dw;
            if (ix >= 0 && ix < W) im[((size_t)c * H + iy) * W + ix] += src[oy * OW + ox];
          }
        }
      }
}

/* ---- Convolution forward (conv_layer.cpp:8-40; base_conv_layer.cpp:257-280) --------------------------
 * per image: col = im2col(x) unless 1x1/stride1/pad0 (is_1x1_, base_conv_layer.cpp:111-116);
 * y = W[Cout, Cin*kh*kw] * col; y += bias * ones (rank-1 GEMM, :274-280).  group == 1 on this path. */
void oracle_conv_forward(const float* x, int N, int C, int H, int W, const float* w, const float* bias, int Cout,
                         int kh, int kw, int ph, int pw, int sh, int sw, int dh, int dw, float* y) {
  const int OH = conv_out(H, kh, ph, sh, dh), OW = conv_out(W, kw, pw, sw, dw);
  const int Kd = C * kh * kw, S = OH * OW;
  const int is1x1 = kh == 1 && kw == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0;
  float* col = is1x1 ? NULL : (float*)malloc((size_t)Kd * S * sizeof(float));
  for (int n = 0; n < N; ++n) {
    const float* xin = x + (size_t)n * C * H * W;
    float* yo = y + (size_t)n * Cout * S;
    const float* cb = xin;
    if (!is1x1) {
      oracle_im2col(xin, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, col);
      cb = col;
    }
    oracle_sgemm(0, Cout, S, Kd, w, cb, 0.f, yo);
    if (bias)
      for (int co = 0; co < Cout; ++co) {
        float* r = yo + (size_t)co * S;
        const float b = bias[co];
        for (int i = 0; i < S; ++i) r[i] += b * 1.f;
      }
  }
  free(col);
}

/* ---- Deconvolution forward (deconv_layer.cpp:8-40; base_conv_layer.cpp:283-298) -----------------------
 * weights [Cin, Cout, kh, kw]; per image: col[Cout*kh*kw, H*W] = W^T * x (CblasTrans), y = col2im(col),
 * then bias.  Output size s*(in-1) + d*(k-1)+1 - 2p. */
void oracle_deconv_forward(const float* x, int N, int C, int H, int W, const float* w, const float* bias, int Cout,
                           int kh, int kw, int ph, int pw, int sh, int sw, int dh, int dw, float* y) {
  const int OH = sh * (H - 1) + dh * (kh - 1) + 1 - 2 * ph, OW = sw * (W - 1) + dw * (kw - 1) + 1 - 2 * pw;
  const int Kd = Cout * kh * kw, S = H * W;
  float* col = (float*)malloc((size_t)Kd * S * sizeof(float));
  for (int n = 0; n < N; ++n) {
    const float* xin = x + (size_t)n * C * S;
    float* yo = y + (size_t)n * Cout * OH * OW;
    oracle_sgemm(1, Kd, S, C, w, xin, 0.f, col); /* A = w viewed [C, Kd], transposed */
    oracle_col2im(col, Cout, OH, OW, kh, kw, ph, pw, sh, sw, dh, dw, yo);
    if (bias)
      for (int co = 0; co < Cout; ++co) {
        float* r = yo + (size_t)co * OH * OW;
        const float b = bias[co];
        for (int i = 0; i < OH * OW; ++i) r[i] += b * 1.f;
      }
  }
  free(col);
}

/* ---- BatchNorm, use_global_stats (batch_norm_layer.cpp:86-93,105-111,138-149) -------------------------
 * scale_factor = blob2[0]==0 ? 0 : 1/blob2[0]; mean = blob0*sf; var = blob1*sf;
 * y = (x - mean) / pow(var + eps, 0.5)   — float arithmetic, division (not multiply by reciprocal). */
void oracle_batchnorm_forward(float* x, int N, int C, int S, const float* b0, const float* b1, const float* b2,
                              float eps) {
  const float sf = b2[0] == 0.f ? 0.f : 1.f / b2[0];
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < C; ++c) {
    const float mean = b0[c] * sf;
    float var = b1[c] * sf;
    var = var + eps;
    const float sd = powf(var, 0.5f);
    for (int n = 0; n < N; ++n) {
      float* p = x + ((size_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) p[i] = (p[i] - mean) / sd;
    }
  }
}

/* ---- Scale with bias along the channel axis (scale_layer.cpp:109-134; bias_layer.cpp:72-87) ------------ */
void oracle_scale_forward(float* x, int N, int C, int S, const float* gamma, const float* beta) {
  for (int n = 0; n < N; ++n)
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int c = 0; c < C; ++c) {
      float* p = x + ((size_t)n * C + c) * S;
      const float g = gamma[c];
      for (int i = 0; i < S; ++i) p[i] = p[i] * g;
      if (beta) {
        const float b = beta[c];
        for (int i = 0; i < S; ++i) p[i] += b;
      }
    }
}

/* ---- ReLU (relu_layer.cpp:9-19): max(x,0) + slope*min(x,0) --------------------------------------------- */
void oracle_relu_forward(float* x, size_t n, float slope) {
  for (size_t i = 0; i < n; ++i) {
    const float v = x[i];
    x[i] = (v > 0.f ? v : 0.f) + slope * (v < 0.f ? v : 0.f);
  }
}

/* ---- Sigmoid (sigmoid_layer.cpp:9-22): 1. / (1. + exp(-x)) ---------------------------------------------- */
void oracle_sigmoid_forward(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = (float)(1. / (1. + expf(-x[i])));
}

/* ---- Eltwise SUM, coefficients 1 (eltwise_layer.cpp:59-65): top = 0; top += a; top += b ---------------- */
void oracle_eltwise_sum(const float* a, const float* b, float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    float t = 0.f;
    t += 1.f * a[i];
    t += 1.f * b[i];
    y[i] = t;
  }
}

/* ---- MAX pooling (pooling_layer.cpp:79-123 shapes, :140-187 loop) ------------------------------------ */
int oracle_pool_out(int in, int k, int p, int s) {
  int o = (int)ceilf((float)(in + 2 * p - k) / (float)s) + 1;
  if (p && (o - 1) * s >= in + p) --o;
  return o;
}
/* kernel_h/kernel_w, stride_h/stride_w, pad_h/pad_w as pooling_layer.cpp:22-77 accepts them (the path uses the square k3 s2 p0
 * form only; the rectangular form exists so that the reference's own literals test_pooling_layer.cpp:121-372 can pin this loop) */
void oracle_maxpool_forward_rect(const float* x, int N, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                                 float* y) {
  const int OH = oracle_pool_out(H, kh, ph, sh), OW = oracle_pool_out(W, kw, pw, sw);
  for (int nc = 0; nc < N * C; ++nc) {
    const float* xi = x + (size_t)nc * H * W;
    float* yo = y + (size_t)nc * OH * OW;
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox) {
        int hs = oy * sh - ph, ws = ox * sw - pw;
        int he = hs + kh < H ? hs + kh : H, we = ws + kw < W ? ws + kw : W;
        if (hs < 0) hs = 0;
        if (ws < 0) ws = 0;
        float m = -FLT_MAX;
        for (int iy = hs; iy < he; ++iy)
          for (int ix = ws; ix < we; ++ix)
            if (xi[iy * W + ix] > m) m = xi[iy * W + ix];
        yo[oy * OW + ox] = m;
      }
  }
}
void oracle_maxpool_forward(const float* x, int N, int C, int H, int W, int k, int s, int p, float* y) {
  oracle_maxpool_forward_rect(x, N, C, H, W, k, k, s, s, p, p, y);
}

/* ---- Crop (fork-specific, crop_layer.cpp:25-50): top = bottom0[:, :, oh:oh+H1, ow:ow+W1] ------------------ */
void oracle_crop_forward(const float* x, int N, int C, int H, int W, int oh, int ow, int H1, int W1, float* y) {
  for (int nc = 0; nc < N * C; ++nc)
    for (int h = 0; h < H1; ++h)
      memcpy(y + ((size_t)nc * H1 + h) * W1, x + ((size_t)nc * H + oh + h) * W + ow, (size_t)W1 * sizeof(float));
}
