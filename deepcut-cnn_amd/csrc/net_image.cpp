// net_image.cpp — see net_internal.h: image entry, pose decode, multi-person consumers, plan / profile / debug text.
#include "net_internal.h"

namespace dc {

// ---- image entry: the demo's pre-processing on the device ------------------------------------------------------------
// python/pose/estimate_pose.py:83-103: replicate the last row/column 64 px, scipy.misc.imresize(.., scale, 'bilinear')
// (= Pillow's 8-bit two-pass resample to (int(W*s), int(H*s)); identity when the size does not change), subtract the
// BGR mean, paste on a zero canvas whose sides are rounded up to the stride.  The resample is integer arithmetic with
// 22-bit fixed-point weights; the weights are computed here on the host in double precision exactly as
// Pillow's precompute_coeffs / normalize_coeffs_8bpc do, so the device result is bit-identical to the reference's.
#pragma clang fp contract(off)
ResampleTable::~ResampleTable() {
  dev_free(dev_bounds);
  dev_free(dev_coeffs);
}

void resample_coeffs(int in_size, int out_size, int& ksize, std::vector<int>& bounds, std::vector<int>& coeffs) {
  const int kPrecisionBits = 32 - 8 - 2;
  double filterscale, scale;
  filterscale = scale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;  // bilinear: support 1
  ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  coeffs.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      const double w = v < 1.0 ? 1.0 - v : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      coeffs[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
    }
    bounds[(size_t)xx * 2] = xmin;
    bounds[(size_t)xx * 2 + 1] = xmax;
  }
}

std::shared_ptr<ResampleTable> Net::resample_table(int in_size, int out_size) {
  auto key = std::make_pair(in_size, out_size);
  auto it = resample_.find(key);
  if (it != resample_.end()) return it->second;
  auto t = std::make_shared<ResampleTable>();
  std::vector<int> b, c;
  resample_coeffs(in_size, out_size, t->ksize, b, c);
  t->bounds = b;
  dev_alloc((void**)&t->dev_bounds, b.size() * sizeof(int));
  dev_alloc((void**)&t->dev_coeffs, c.size() * sizeof(int));
  dev_upload(t->dev_bounds, b.data(), b.size() * sizeof(int), stream);
  dev_upload(t->dev_coeffs, c.data(), c.size() * sizeof(int), stream);
  if (resample_.size() > 64) resample_.clear();  // a pyramid uses a handful; bound the cache anyway
  resample_[key] = t;
  return t;
}

void image_canvas_size(int h, int w, double scale, int& out_h, int& out_w, int& new_h, int& new_w) {
  const int kStride = 8, kPad = 64;
  out_w = (int)(std::ceil((double)w * scale / kStride) * kStride);  // estimate_pose.py:85-88
  out_h = (int)(std::ceil((double)h * scale / kStride) * kStride);
  new_w = (int)((double)(w + kPad) * scale);  // scipy.misc.imresize: (array(im.size) * scale).astype(int)
  new_h = (int)((double)(h + kPad) * scale);
}

void Net::forward_images(const unsigned char* bgr, int n, int h, int w, double scale, bool is_device, float* prob, float* loc,
                         float* next, double* pose, void* user_stream) {
  if (Context::get().mode != DC_MODE_GPU)
    throw DcError(DC_ENOCPU, "forward_images() in CPU mode: libdeepcut_hip provides the MI355X path only");
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  // (the net's own stream exists only after ensure_device(): prep_images resolves a null `s` to it)
  prep_images(bgr, n, h, w, scale, is_device, user_stream);
  void* s = user_stream ? user_stream : stream;
  enqueue_plan(s);
  emit_maps(prob, loc, next, is_device, s);
  if (pose) {
    decode_pose(scale, pose, is_device, (user_stream || own_async) ? s : nullptr);
  }
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void Net::prep_images(const unsigned char* bgr, int n, int h, int w, double scale, bool is_device, void* s) {
  if (n <= 0 || h <= 0 || w <= 0 || !(scale > 0)) throw DcError(DC_EINVAL, "forward_images: n, height, width and scale must be positive");
  int out_h, out_w, new_h, new_w;
  image_canvas_size(h, w, scale, out_h, out_w, new_h, new_w);
  if (new_h < 1 || new_w < 1 || out_h < 8 || out_w < 8)
    throw DcError(DC_ESHAPE, "forward_images: scale " + std::to_string(scale) + " leaves no pixels of a " + std::to_string(h) + "x" +
                                 std::to_string(w) + " image");
  Storage& in = begin_batch(n, out_h, out_w);
  if (in.dim(1) != 3) throw DcError(DC_ESHAPE, "forward_images needs a 3-channel input blob");
  if (!s) s = stream;
  const int kPad = 64;
  const int ph = h + kPad, pw = w + kPad;              // the replicate-padded image (never materialised)
  const int use_h = std::min(out_h, new_h), use_w = std::min(out_w, new_w);  // part of the resized image on the canvas
  const unsigned char* src = bgr;
  const size_t bytes = (size_t)n * h * w * 3;
  if (!is_device) {
    if (bytes > img_cap_) {
      dev_free(img_dev_);
      img_dev_ = nullptr;
      dev_alloc((void**)&img_dev_, bytes);
      img_cap_ = bytes;
    }
    HIPCHECK(hipMemcpyAsync(img_dev_, bgr, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    src = img_dev_;
  }
  const bool need_x = new_w != pw, need_y = new_h != ph;
  ImagePrepParams q{};
  q.src = src;
  q.n = n, q.h = h, q.w = w;
  q.out_h = out_h, q.out_w = out_w, q.use_h = use_h, q.use_w = use_w;
  q.dst = in.dev, q.dst_esize = in.esize, q.dst_cp = in.cp();
  q.mean[0] = 104.f, q.mean[1] = 117.f, q.mean[2] = 123.f;  // _MEAN, estimate_pose.py:26
  std::shared_ptr<ResampleTable> hold_y, hold_x;  // the tables outlive a cache flush until the launches are enqueued
  if (need_y) {
    hold_y = resample_table(ph, new_h);
    const ResampleTable& ty = *hold_y;
    q.y_bounds = ty.dev_bounds, q.y_coeffs = ty.dev_coeffs, q.y_ksize = ty.ksize;
    // rows of the (padded, horizontally resampled) image the kept output rows read
    q.row0 = ty.bounds[0];
    q.rows = ty.bounds[(size_t)(use_h - 1) * 2] + ty.bounds[(size_t)(use_h - 1) * 2 + 1] - q.row0;
  } else {
    q.row0 = 0, q.rows = use_h;
  }
  if (need_x) {
    hold_x = resample_table(pw, new_w);
    const ResampleTable& tx = *hold_x;
    q.x_bounds = tx.dev_bounds, q.x_coeffs = tx.dev_coeffs, q.x_ksize = tx.ksize;
    const size_t tb = (size_t)n * q.rows * use_w * 4;
    if (tb > tmp_cap_) {
      dev_free(tmp_dev_);
      tmp_dev_ = nullptr;
      dev_alloc((void**)&tmp_dev_, tb);
      tmp_cap_ = tb;
    }
    q.tmp = tmp_dev_;
  }
  KCHECK(launch_image_prep(q, s));
  in.head = HEAD_AT_GPU;
}

// _pose_from_mats (python/pose/estimate_pose.py:131-143) on the device: reads the `prob` and `loc_pred`
// images of the last forward where they live (channel views of the merged head tensor included) and
// returns 5 x J doubles per image — the 10 MB of maps need not cross PCIe for single-person decoding.
void Net::decode_pose(double scale, double* out, bool is_device, void* user_stream) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "decode_pose() in CPU mode");
  auto ip = blob_index.find("prob"), il = blob_index.find("loc_pred");
  if (ip == blob_index.end() || il == blob_index.end()) throw DcError(DC_EINVAL, "net has no 'prob' / 'loc_pred' blobs");
  Storage& P = *blobs[ip->second]->st;
  Storage& L = *blobs[il->second]->st;
  if (P.head == UNINITIALIZED || L.head == UNINITIALIZED) throw DcError(DC_EINVAL, "decode_pose: run forward() first");
  if (L.dim(1) != 2 * P.dim(1) || L.dim(2) != P.dim(2) || L.dim(3) != P.dim(3) || L.dim(0) != P.dim(0))
    throw DcError(DC_ESHAPE, "decode_pose: loc_pred must have 2 channels per joint and the score map's size");
  ensure_device();
  auto img = [&](Storage& s, const void*& ptr, int& cp, int& c0) {
    if (s.view_of >= 0) {
      ptr = storages[s.view_of]->dev;
      cp = storages[s.view_of]->cp();
      c0 = s.view_c0;
    } else {
      if (s.head == HEAD_AT_CPU || s.head == UNINITIALIZED) sync_to_device(s);
      ptr = s.dev;
      cp = s.cp();
      c0 = 0;
    }
  };
  const void *pp, *lp;
  int pcp, pc0, lcp, lc0;
  const int pes = P.view_of >= 0 ? storages[P.view_of]->esize : P.esize;
  img(P, pp, pcp, pc0);
  img(L, lp, lcp, lc0);
  const int NB = P.dim(0), J = P.dim(1), H = P.dim(2), W = P.dim(3);
  void* s = user_stream ? user_stream : stream;
  const size_t cnt = (size_t)NB * 5 * J;
  if (is_device) {
    KCHECK(launch_pose_decode(pp, pcp, pc0, lp, lcp, lc0, pes, NB, H, W, J, scale, out, s));
    if (!user_stream) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
    return;
  }
  if (cnt > pose_cap) {
    dev_free(pose_dev);
    pose_dev = nullptr;
    dev_alloc((void**)&pose_dev, cnt * sizeof(double));
    pose_cap = cnt;
  }
  KCHECK(launch_pose_decode(pp, pcp, pc0, lp, lcp, lc0, pes, NB, H, W, J, scale, pose_dev, s));
  HIPCHECK(hipMemcpyAsync(out, pose_dev, cnt * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)s));
  HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

Net::MapRef Net::map_ref(const char* blob_name) {
  auto it = blob_index.find(blob_name);
  if (it == blob_index.end()) throw DcError(DC_EINVAL, std::string("net has no '") + blob_name + "' blob");
  Storage& s = *blobs[it->second]->st;
  if (s.elided && s.view_of < 0)
    throw DcError(DC_EUNSUP, std::string("'") + blob_name + "' is not computed in the current plan (DC_OPT_OUTPUTS leaves it out, or it is folded into a fused kernel)");
  if (s.head == UNINITIALIZED) throw DcError(DC_EINVAL, std::string("'") + blob_name + "': run forward() first");
  if (s.shape.size() != 4) throw DcError(DC_ESHAPE, std::string("'") + blob_name + "' is not a 4-D map");
  MapRef r{};
  if (s.view_of >= 0) {
    Storage& b = *storages[s.view_of];
    r.ptr = b.dev, r.cp = b.cp(), r.c0 = s.view_c0, r.es = b.esize;
  } else {
    if (s.head == HEAD_AT_CPU) sync_to_device(s);
    r.ptr = s.dev, r.cp = s.cp(), r.c0 = 0, r.es = s.esize;
  }
  r.NB = s.dim(0), r.C = s.dim(1), r.H = s.dim(2), r.W = s.dim(3);
  return r;
}

void* Net::scratch(size_t bytes) {
  if (bytes > scratch_cap_) {
    dev_free(scratch_dev_);
    scratch_dev_ = nullptr;
    dev_alloc((void**)&scratch_dev_, bytes);
    scratch_cap_ = bytes;
  }
  return scratch_dev_;
}

// Part candidates: non-maximum suppression of every score map + location refinement, on the device.
void Net::detect_parts(double scale, float thr, int radius, int max_det, int* counts, double* dets) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "detect_parts() in CPU mode");
  if (!(scale > 0) || !(thr >= 0.f) || radius < 0 || radius > 64 || max_det < 1 || max_det > 4096)
    throw DcError(DC_EINVAL, "detect_parts: scale > 0, threshold >= 0, 0 <= radius <= 64, 1 <= max_det <= 4096");
  ensure_device();
  const MapRef P = map_ref("prob"), L = map_ref("loc_pred");
  if (L.C != 2 * P.C || L.H != P.H || L.W != P.W || L.NB != P.NB || L.es != P.es)
    throw DcError(DC_ESHAPE, "detect_parts: loc_pred must have 2 channels per joint and the score map's size");
  const int lists = P.NB * P.C;
  const size_t cnt_b = ((size_t)lists * sizeof(int) + 255) / 256 * 256;
  const size_t spill_b = (size_t)lists * P.H * P.W * sizeof(unsigned long long);  // every cell may be a local maximum
  const size_t out_b = (size_t)lists * max_det * 5 * sizeof(double);
  unsigned char* base = (unsigned char*)scratch(cnt_b + spill_b + out_b);
  int* cnt = (int*)base;
  unsigned long long* spill = (unsigned long long*)(base + cnt_b);
  double* out = (double*)(base + cnt_b + spill_b);
  KCHECK(launch_part_select(P.ptr, P.cp, P.c0, L.ptr, L.cp, L.c0, P.es, P.NB, P.H, P.W, P.C, thr, radius, scale, max_det, spill, cnt, out,
                            stream));
  HIPCHECK(hipMemcpyAsync(counts, cnt, (size_t)lists * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHECK(hipMemcpyAsync(dets, out, out_b, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
}

// Pairwise regression of the next joint from a set of detections (cells), on the device.
void Net::decode_pairwise(double scale, int ndet, const int* det, const double* mean, const double* stdev, double* out) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "decode_pairwise() in CPU mode");
  if (!(scale > 0) || ndet < 0) throw DcError(DC_EINVAL, "decode_pairwise: scale > 0, ndet >= 0");
  if (ndet == 0) return;
  ensure_device();
  const MapRef N = map_ref("next_pred");
  if (N.C % 2) throw DcError(DC_ESHAPE, "decode_pairwise: next_pred must have 2 channels per regression edge");
  const int E = N.C / 2;
  const size_t det_b = ((size_t)ndet * 3 * sizeof(int) + 255) / 256 * 256, st_b = (size_t)E * 2 * sizeof(double);
  const size_t out_b = (size_t)ndet * E * 2 * sizeof(double);
  unsigned char* base = (unsigned char*)scratch(det_b + 2 * st_b + out_b);
  int* ddet = (int*)base;
  double* dmean = (double*)(base + det_b);
  double* dstd = dmean + (size_t)E * 2;
  double* dout = (double*)(base + det_b + 2 * st_b);
  HIPCHECK(hipMemcpyAsync(ddet, det, (size_t)ndet * 3 * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
  if (mean) HIPCHECK(hipMemcpyAsync(dmean, mean, st_b, hipMemcpyHostToDevice, (hipStream_t)stream));
  if (stdev) HIPCHECK(hipMemcpyAsync(dstd, stdev, st_b, hipMemcpyHostToDevice, (hipStream_t)stream));
  KCHECK(launch_pairwise_decode(N.ptr, N.cp, N.c0, N.es, N.NB, N.H, N.W, E, scale, ndet, ddet, mean ? dmean : nullptr,
                                stdev ? dstd : nullptr, dout, stream));
  HIPCHECK(hipMemcpyAsync(out, dout, out_b, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
}

std::string Net::plan_text() {
  ensure_plan();
  std::ostringstream os;
  os << "# plan for input";
  for (int d : plan_input_shape) os << " " << d;
  os << ": " << plan.size() << " launches, " << plan_flops / 1e9 << " GFLOP algorithmic, fuse=" << fuse
     << (dtype == 1 ? ", dtype=f16" : ", dtype=f32") << "\n";
  for (size_t i = 0; i < plan.size(); ++i) {
    const Launch& l = plan[i];
    os << i << "\t" << l.kernel << "\t";
    if (l.kind == Launch::CONV)
      os << "M=" << l.cg.M << " N=" << l.cg.Cout << " K=" << l.cg.Ktot << " taps=" << l.cg.nty * l.cg.ntx
         << (l.cg.ncls > 1 ? " classes=" + std::to_string(l.cg.ncls) : std::string()) << " grid=" << l.grid
         << (l.in2 >= 0 ? " +resid" : "") << (l.relu ? " +relu" : "") << (l.cg.sigmoid_ch ? " +sigmoid" : "");
    os << "\t" << l.label << "\n";
  }
  return os.str();
}

std::string Net::profile_text(int iters) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "profile in CPU mode");
  if (!plan_valid) throw DcError(DC_EINVAL, "profile_text: run forward() first");
  ensure_device();
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  std::ostringstream os;
  os << "idx\tkernel\tus\tGFLOP\tTFLOP/s\tgrid\tlabel\n";
  double total_us = 0;
  for (size_t i = 0; i < plan.size(); ++i) {
    const Launch& l = plan[i];
    run_launch(l, stream);  // warm
    HIPCHECK(hipEventRecord(e0, (hipStream_t)stream));
    for (int k = 0; k < iters; ++k) run_launch(l, stream);
    HIPCHECK(hipEventRecord(e1, (hipStream_t)stream));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1000.0 / iters;
    total_us += us;
    char buf[512];
    std::snprintf(buf, sizeof buf, "%zu\t%s\t%.2f\t%.3f\t%.2f\t%ld\t%s\n", i, l.kernel.c_str(), us, l.flops / 1e9,
                  us > 0 ? l.flops / us / 1e6 : 0.0, l.grid, l.label.c_str());
    os << buf;
  }
  os << "# sum of per-launch times: " << total_us << " us\n";
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return os.str();
}


// Net::ForwardDebugInfo / InputDebugInfo (net.cpp:648-681 of the reference, `debug_info: true`): the mean absolute value of
// every top blob and every parameter blob, in the reference's own line format, from the blobs of the LAST forward — what
// somebody bisecting a mismatch against a Caffe debug_info log needs.  Differences that follow from the lowering: an
// in-place layer chain (conv -> BatchNorm -> Scale -> ReLU on one blob) runs as one kernel, so the blob is only ever
// seen after the LAST layer of the chain: the line is printed for that layer, the earlier in-place layers of the chain get
// a `(folded into ...)` note; a blob swallowed by residual / head fusion (DC_OPT_FUSE >= 1) is reported as elided — run with
// DC_OPT_FUSE 0 to materialise all 220 Caffe-visible blobs.
std::string Net::debug_info_text() {
  if (!plan_valid) throw DcError(DC_EINVAL, "debug_info: run forward() first");
  std::ostringstream os;
  char buf[384];
  auto mean_abs = [&](Storage& st) -> double {
    sync_to_host(st);
    const float* h = st.host_ptr();
    const size_t n = st.count();
    double a = 0;
    for (size_t i = 0; i < n; ++i) a += std::fabs((double)h[i]);
    return n ? a / (double)n : 0.0;
  };
  for (int b : inputs) {
    std::snprintf(buf, sizeof buf, "    [Forward] Input %s data: %g\n", blobs[b]->name.c_str(), mean_abs(*blobs[b]->st));
    os << buf;
  }
  // the last layer (in file order) that writes each blob: the only moment its value exists here
  std::vector<int> last_writer(blobs.size(), -1);
  for (size_t i = 0; i < layers.size(); ++i)
    for (int t : layers[i].tops) last_writer[t] = (int)i;
  for (size_t i = 0; i < layers.size(); ++i) {
    const LayerRec& L = layers[i];
    for (int t : L.tops) {
      Storage& st = *blobs[t]->st;
      if (last_writer[t] != (int)i) {
        std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, top blob %s data: (in place: folded into layer %s)\n", L.name.c_str(),
                      blobs[t]->name.c_str(), layers[last_writer[t]].name.c_str());
      } else if (st.elided) {
        std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, top blob %s data: (elided by fusion; DC_OPT_FUSE 0 materialises it)\n",
                      L.name.c_str(), blobs[t]->name.c_str());
      } else {
        std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, top blob %s data: %g\n", L.name.c_str(), blobs[t]->name.c_str(), mean_abs(st));
      }
      os << buf;
    }
    for (size_t k = 0; k < L.params.size(); ++k) {
      Storage& st = *L.params[k]->st;
      const float* h = st.host_ptr();
      const size_t n = st.count();
      double a = 0;
      for (size_t q = 0; q < n; ++q) a += std::fabs((double)h[q]);
      // Net::AppendParam (net.cpp:469-482): the ParamSpec's name when it has one, else the index
      const auto specs = L.def.subs("param");
      const std::string pname = k < specs.size() && !specs[k]->str("name").empty() ? specs[k]->str("name") : std::to_string(k);
      std::snprintf(buf, sizeof buf, "    [Forward] Layer %s, param blob %s data: %g\n", L.name.c_str(), pname.c_str(), n ? a / (double)n : 0.0);
      os << buf;
    }
  }
  return os.str();
}

}  // namespace dc
