// net_run.cpp — see net_internal.h: execution of a lowered plan and the entries built on it.
#include "net_internal.h"

namespace dc {

// ---- execution ----------------------------------------------------------------------------------------
void Net::ensure_device() {
  if (device_count() <= 0)
    throw DcError(DC_EDEVICE, "no HIP device visible: libdeepcut_hip has no CPU compute path (the CPU restatement of the "
                              "reference is test-only, under oracle/)");
  Context& c = Context::get();
  if (device < 0) device = c.device;
  HIPCHECK(hipSetDevice(device));
  if (!stream) {
    hipStream_t s;
    HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    stream = s;
  }
}

void Net::upload_vecs() {
  std::lock_guard<std::mutex> lk(shared->mu);  // an image may be shared with a clone that uploads at the same moment
  std::vector<DevVec*> todo;
  for (auto& l : plan)
    for (const std::shared_ptr<DevVec>* vp : {&l.w, &l.scale, &l.shift, &l.wino_w, &l.wino_scale})
      if (*vp && !(*vp)->dev && !(*vp)->host.empty()) todo.push_back(vp->get());
  for (DevVec* vq : todo) {
    DevVec& v = *vq;
    if (!v.dev && !v.host.empty()) {
      if (v.as_half) {  // filter image of an fp16 net: upload as float, convert on the device, keep the half copy
        float* tmp = nullptr;
        dev_alloc((void**)&tmp, v.host.size() * sizeof(float));
        struct TmpGuard {
          float* p;
          ~TmpGuard() { dev_free(p); }
        } tmp_guard{tmp};
        dev_upload(tmp, v.host.data(), v.host.size() * sizeof(float), stream);
        dev_alloc((void**)&v.dev, v.host.size() * 2);
        KCHECK(launch_f32_to_f16(tmp, v.dev, (long)v.host.size(), stream));
        HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
      } else {
        dev_alloc((void**)&v.dev, v.host.size() * sizeof(float));
        dev_upload(v.dev, v.host.data(), v.host.size() * sizeof(float), stream);
      }
      v.uploaded = v.host.size();
      std::vector<float>().swap(v.host);  // the packed image lives in HBM only
    }
  }
}

// Per-shape tile selection by measurement ("benchmark mode"): every distinct GEMM signature of the plan
// is timed once per process with each eligible tile variant on the real buffers (all variants compute
// the same values up to fp32 summation order) and the fastest is kept.  DC_AUTOTUNE=0 keeps the cost
// model's choice; DC_CONV_VARIANT forces one variant.
// DC_TUNE_CACHE=<file>: "signature tile-name" per line.  Caller holds shared.mu.
void Net::release_graph() {
  if (graph_exec) {
    (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    graph_exec = nullptr;
  }
}

// SyncedMemory::to_gpu (syncedmem.cpp:49-77): UNINITIALIZED -> a zeroed device image that is authoritative
// (HEAD_AT_GPU); HEAD_AT_CPU -> upload, SYNCED.  The device image of a 4-D blob is channels-last.
Net* Net::create_for_layer(const std::string& layer_text, int phase, const std::vector<std::vector<int>>& bottom_shapes) {
  TextMsg m = parse_text_proto(layer_text);
  const TextMsg* L = &m;
  if (m.sub("layer") && !m.has("type")) L = m.sub("layer");  // given with the enclosing `layer { }`
  std::vector<std::string> bottoms = L->strs("bottom");
  if (bottoms.size() != bottom_shapes.size())
    throw DcError(DC_EINVAL, "layer '" + L->str("name") + "' declares " + std::to_string(bottoms.size()) + " bottom(s), " +
                                 std::to_string(bottom_shapes.size()) + " given");
  std::string text = "name: \"" + L->str("name") + "\"\n";
  std::set<std::string> seen;
  for (size_t i = 0; i < bottoms.size(); ++i) {
    if (!seen.insert(bottoms[i]).second) throw DcError(DC_EUNSUP, "layer '" + L->str("name") + "': the same bottom twice");
    text += "input: \"" + bottoms[i] + "\"\ninput_shape {";
    for (int d : bottom_shapes[i]) text += " dim: " + std::to_string(d);
    text += " }\n";
  }
  text += L == &m ? "layer {\n" + layer_text + "\n}\n" : layer_text + "\n";
  std::unique_ptr<Net> n(Net::create(text, phase));
  n->fuse = 0;
  return n.release();
}

void Net::run_launch(const Launch& l, void* s) {
  Storage& X = *storages[l.in];
  Storage& Y = *storages[l.out];
  switch (l.kind) {
    case Launch::CONV: {
      ConvGemmParams g = l.cg;
      g.dbg = nullptr;
      g.x = X.dev;
      g.y = Y.dev_at(l.y_off);
      g.resid = l.in2 >= 0 ? storages[l.in2]->dev_at(l.y_off) : nullptr;
      g.w = reinterpret_cast<const unsigned char*>(l.w->dev) + (size_t)l.w_off * (size_t)g.esize;
      g.scale = l.scale ? l.scale->dev + l.c_off : nullptr;
      g.shift = l.shift ? l.shift->dev + l.c_off : nullptr;
      const bool wino = is_wino_variant(l.variant);  // Winograd F(2x2,3x3) form of a stride-1 3x3 layer (8 or 16 waves per workgroup)
      if (wino) {
        if (!l.wino_w) throw DcError(DC_EINVAL, "launch '" + l.label + "' has no Winograd filter image");
        g.w = l.wino_w->dev;
        if (l.variant == kWinoHalf) {  // float16: the form's own epilogue scale (row scale of ITS image, the 1/4 of the staged pixels)
          if (!l.wino_scale || l.in2 >= 0) throw DcError(DC_EINVAL, "launch '" + l.label + "' cannot run as the float16 Winograd form");
          g.scale = l.wino_scale->dev + l.c_off;
        }
      }
      static const int dbg_idx = env_int("DC_DEBUG_TIMING", -1);
      // index of this launch in the plan (autotuning passes copies, which have none)
      const bool in_plan = !plan.empty() && std::greater_equal<const Launch*>()(&l, plan.data()) &&
                           std::less<const Launch*>()(&l, plan.data() + plan.size());
      const int my_idx = in_plan ? (int)(&l - plan.data()) : -1;
      if (dbg_idx >= 0 && my_idx == dbg_idx) {
        // device-side phase timestamps of ONE launch (diagnostics only): per wave the shader cycle counter at up to 8 phase
        // boundaries (slots 0..7) and the chip-wide 100 MHz clock at start / end (slots 8, 9)
        const int nwv = wino ? (l.variant == kWinoVariant16 ? 16 : l.variant == kStreamHalf || l.variant == kStemHalf || l.variant == kStreamFloat || l.variant == kStemFloat ? 4 : 8) : conv_variant(l.variant).WR * conv_variant(l.variant).WC * conv_variant(l.variant).WK;
        const long n = (l.grid * 2 + 64) * nwv * 12;  // the XCD-aware maps pad the grid (at most 8 x the longest XCD list)
        long long* d = nullptr;
        dev_alloc((void**)&d, n * sizeof(long long));
        dev_zero(d, n * sizeof(long long), s);
        // DC_DEBUG_TIMING_INSITU=1: ONE launch, in stream order right behind the launch before it (cold filters, the other kernel's tail)
        // instead of three launches on an idle chip (the last one warm)
        static const bool insitu = env_int("DC_DEBUG_TIMING_INSITU", 0) != 0;
        for (int rep = 0; rep < (insitu ? 1 : 3); ++rep) {
          g.dbg = d;
          if (!insitu) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
          if (wino) KCHECK(launch_wino_conv(g, s, l.variant));
          else KCHECK(launch_conv_gemm(g, l.variant, s));
          HIPCHECK(hipStreamSynchronize((hipStream_t)s));
        }
        std::vector<long long> h(n);
        HIPCHECK(hipMemcpyAsync(h.data(), d, n * sizeof(long long), hipMemcpyDeviceToHost, (hipStream_t)s));
        HIPCHECK(hipStreamSynchronize((hipStream_t)s));
        dev_free(d);
        double dsum[8] = {0}, karg = 0;
        long cnt = 0;
        long long t0min = 0, t0max = 0, t7max = 0;
        for (long i = 0; i < (l.grid * 2 + 64) * nwv; ++i) {
          const long long* w = &h[i * 12];
          if (w[7] == 0 || w[0] == 0) continue;  // workgroup of the padded XCD grid that exited at once
          for (int k = 1; k < 8; ++k) dsum[k] += (double)(w[k] - w[k - 1]);
          karg += (double)(w[10] - w[8]);
          if (!cnt || w[8] < t0min) t0min = w[8];
          if (!cnt || w[8] > t0max) t0max = w[8];
          if (!cnt || w[9] > t7max) t7max = w[9];
          ++cnt;
        }
        std::fprintf(stderr, "[dc timing] first wave start -> last wave start %.2f us | first start -> last end %.2f us | wave entry -> kernel "
                     "arguments there %.2f us | waves %ld\n",
                     (t0max - t0min) / 100.0, (t7max - t0min) / 100.0, karg / std::max(cnt, 1L) / 100.0, cnt);
        // conv_gemm slots: 0 start, 1 filter loads + epilogue constants issued, 2 rows decoded + activation loads issued,
        // 3 output offsets in LDS, 4 first tile staged (K-loop entry), 5 K-loop exit, 6 split-K exchange done, 7 stores issued
        static const char* kGemm[7] = {"filter-load issue", "row decode + activation-load issue", "output offsets to LDS", "wait+stage+barrier",
                                       "K loop", "split-K exchange", "epilogue math+stores"};
        static const char* kWino[7] = {"index setup", "first loads issued", "two stages in LDS", "K loop", "partials to LDS + barrier",
                                       "inverse transform + epilogue constants", "shortcut + stores"};
        // ws1x1 (stream1x1.hip) slots: 0 start, 1 every prologue request issued, 2 first stage + filters + constants landed, 3 the peeled
        // first D steps done, 4 the other steps (and the late half's last epilogue) done, 5 requests drained
        static const char* kStream[7] = {"prologue requests issued", "first stage + filters landed", "the first D steps", "the other steps",
                                         "drain", "-", "exit"};
        std::string line;
        for (int k = 1; k < 8; ++k) {
          char buf[96];
          std::snprintf(buf, sizeof buf, "%s%s %.0f", k > 1 ? " | " : "", (l.variant == kStreamHalf || l.variant == kStreamFloat || l.variant == kStemFloat ? kStream : wino ? kWino : kGemm)[k - 1], dsum[k] / std::max(cnt, 1L));
          line += buf;
        }
        std::fprintf(stderr, "[dc timing] launch %d %s %s\n  mean cycles per wave: %s\n", my_idx, l.kernel.c_str(), l.label.c_str(), line.c_str());
        g.dbg = nullptr;
      }
      if (wino) {
        KCHECK(launch_wino_conv(g, s, l.variant));
        break;
      }
      {
        const int rc = launch_conv_gemm(g, l.variant, s);
        if (rc == (int)hipErrorInvalidValue)
          throw DcError(DC_EUNSUP, "launch '" + l.label + "': unsupported geometry (a tensor of 2 GiB or more per launch — "
                                   "split the batch — or a tap / K layout this variant cannot take)");
        KCHECK(rc);
      }
      break;
    }
    case Launch::POOL:
      KCHECK(launch_maxpool(X.dev, Y.dev, X.esize, X.dim(0), X.dim(2), X.dim(3), X.cp(), Y.dim(2), Y.dim(3), l.pk, l.ps, l.pp, s));
      break;
    case Launch::ELT:
      KCHECK(launch_eltwise(X.dev, l.in2 >= 0 ? storages[l.in2]->dev : nullptr, l.scale ? l.scale->dev : nullptr,
                            l.shift ? l.shift->dev : nullptr, Y.dev, Y.esize, (long)Y.dev_count(), Y.cp(), l.relu,
                            l.sigmoid, s));
      break;
    case Launch::CROP:
      KCHECK(launch_crop(X.dev, Y.dev, X.esize, X.dim(0), X.dim(2), X.dim(3), X.cp(), l.oh, l.ow, Y.dim(2), Y.dim(3), s));
      break;
  }
}

void Net::run_plan(int start, int end, void* s) {
  for (auto& l : plan) {
    if (l.last_layer < start || l.first_layer > end) continue;
    if (l.first_layer < start || l.last_layer > end)
      throw DcError(DC_EINVAL, "forward range [" + std::to_string(start) + "," + std::to_string(end) +
                                   "] cuts through the fused group '" + l.label + "'; use DC_OPT_FUSE 0 for partial ranges");
    run_launch(l, s);
  }
}

static void prepare_buffers(Net& n, bool& grew) {
  grew = false;
  auto prep = [&](int sidx) {
    Storage& s = *n.storages[sidx];
    size_t need = s.dev_count();
    if (!s.dev || s.dev_cap < std::max<size_t>(need, 8) * (size_t)s.esize) {
      s.ensure_dev(need);
      grew = true;
    }
  };
  for (auto& l : n.plan) {
    prep(l.in);
    prep(l.out);
    if (l.in2 >= 0) prep(l.in2);
  }
}

// Everything a launch sequence of the CURRENT input shape needs, without running it and without touching the inputs: the plan of
// that shape active (re-derived if a blob was reshaped since the last forward), images uploaded, every buffer of the plan allocated at
// its present size, tiles chosen.  (plan_text() / dc_net_flops only lower: plan_valid alone does not mean buffers exist.)
void Net::prepare_to_run() {
  ensure_plan();
  ensure_device();
  upload_vecs();
  bool grew;
  prepare_buffers(*this, grew);
  if (!tuned) autotune();
}

void Net::forward(int start, int end) {
  if (Context::get().mode != DC_MODE_GPU)
    throw DcError(DC_ENOCPU, "forward() in CPU mode: libdeepcut_hip provides the MI355X path only — call set_mode_gpu() "
                             "(the CPU restatement of the reference is test infrastructure under oracle/)");
  ensure_plan();
  ensure_device();
  upload_vecs();
  bool grew;
  prepare_buffers(*this, grew);
  const bool whole = start <= 0 && end >= (int)layers.size() - 1;
  // The timing passes replay launches of the WHOLE plan (in-place and Eltwise ones included) before the inputs are synced:
  // harmless while every output they overwrite is scratch, i.e. for a full forward.  A partial range may start from
  // intermediate blobs the caller placed on the device (mutable_gpu_data): those must not be clobbered, so a partial
  // forward runs with the tiles the plan has (cost model / tune cache) and leaves the tuning to the next full forward.
  if (!tuned && whole) autotune();
  // inputs of the executed range whose host copy is authoritative go up first (SyncedMemory::to_gpu)
  for (auto& l : plan) {
    if (l.last_layer < start || l.first_layer > end) continue;
    for (int sidx : {l.in, l.in2})
      if (sidx >= 0) {
        Storage& s = *storages[sidx];
        if (s.head == HEAD_AT_CPU || s.head == UNINITIALIZED) {
          bool produced_earlier = false;
          for (auto& m : plan) {
            if (&m == &l) break;
            if (m.last_layer < start || m.first_layer > end) continue;
            if (m.out == sidx) produced_earlier = true;
          }
          if (!produced_earlier) {
            ensure_device();
            // pinned host copy (non-parameter blobs): the copy engine reads it behind our back until the stream is drained —
            // which forward() does before it returns; pageable memory is staged by the runtime at enqueue time
            storage_to_device_impl(s, stream, false);
          }
        }
      }
  }
  if (use_graph && whole) {
    if (graph_exec && graph_buf_gen != buf_gen_) release_graph();  // a buffer it addresses was reallocated since
    if (!graph_exec) {
      graph_exec = capture_graph(stream, [&](void* cs) { run_plan(start, end, cs); });
      graph_buf_gen = buf_gen_;
      ++stats.graph_instantiations;
    }
    HIPCHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  } else {
    run_plan(start, end, stream);
  }
  for (auto& l : plan) {
    if (l.last_layer < start || l.first_layer > end) continue;
    storages[l.out]->head = HEAD_AT_GPU;
  }
  for (int v : plan_views_) storages[v]->head = HEAD_AT_GPU;
  // the outputs the caller has been reading through host pointers travel now, behind the last launch (Storage::host_wanted)
  std::vector<Storage*> delivered;
  if (whole)
    for (int bi : outputs) {
      Storage& st = *blobs[bi]->st;
      if (!st.host_wanted || st.head != HEAD_AT_GPU || st.elided) continue;
      if (!st.host_touched) {  // not read since the last delivery: stop sending it
        st.host_wanted = false;
        continue;
      }
      st.host_touched = false;
      storage_download_enqueue(st, stream, st.view_of >= 0 ? storages[st.view_of].get() : nullptr);
      delivered.push_back(&st);
    }
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  for (Storage* st : delivered) st->head = SYNCED;
}

// Common front half of the batched entries: shape the input blob, (re)build the plan, make the device state ready.
Storage& Net::begin_batch(int n, int h, int w) {
  if (Context::get().mode != DC_MODE_GPU)
    throw DcError(DC_ENOCPU, "forward_batch() in CPU mode: libdeepcut_hip provides the MI355X path only");
  if (inputs.size() != 1) throw DcError(DC_EINVAL, "forward_batch needs a single-input net");
  Storage& in = *blobs[inputs[0]]->st;
  int C = in.dim(1);
  in.reshape({n, C, h, w});
  ensure_plan();
  ensure_device();
  upload_vecs();
  bool grew;
  prepare_buffers(*this, grew);
  if (!tuned) autotune();
  return in;
}

// Enqueue every launch of the plan on stream s (the input image is already in HBM).
void Net::enqueue_plan(void* s) {
  const int last = (int)layers.size() - 1;
  if (use_graph) {
    // the launch sequence is captured once on the net's own stream and replayed on whichever stream
    // the caller works on (a graph is not tied to its capture stream)
    if (graph_exec && graph_buf_gen != buf_gen_) release_graph();  // a buffer it addresses was reallocated since
    if (!graph_exec) {
      graph_exec = capture_graph(stream, [&](void* cs) { run_plan(0, last, cs); });
      graph_buf_gen = buf_gen_;
      ++stats.graph_instantiations;
    }
    HIPCHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)s));
  } else {
    run_plan(0, last, s);
  }
  for (auto& l : plan) storages[l.out]->head = HEAD_AT_GPU;
  for (int v : plan_views_) storages[v]->head = HEAD_AT_GPU;
}

// Copy the three output maps out as NCHW (host or device destination), enqueued on s: float32, or — dst_esize 2, fp16
// nets only — the half values as they are in HBM (half the gather payload).
void Net::emit_maps(void* prob, void* loc, void* next, bool is_device, void* s, int dst_esize) {
  struct Out {
    const char* name;
    void* dst;
  } outs[3] = {{"prob", prob}, {"loc_pred", loc}, {"next_pred", next}};
  if (dst_esize != 4 && !(dst_esize == 2 && dtype == 1))
    throw DcError(DC_EINVAL, "maps are emitted as float32, or as float16 from a float16 net (DC_OPT_DTYPE 1)");
  for (auto& o : outs) {
    if (!o.dst) continue;
    auto it = blob_index.find(o.name);
    if (it == blob_index.end()) throw DcError(DC_EINVAL, std::string("net has no blob '") + o.name + "'");
    Storage& st = *blobs[it->second]->st;
    size_t m = st.count();
    const void* src = st.view_of >= 0 ? storages[st.view_of]->dev : st.dev;
    const int ses = st.view_of >= 0 ? storages[st.view_of]->esize : st.esize;
    const int scp = st.view_of >= 0 ? storages[st.view_of]->cp() : st.cp();
    const int sc0 = st.view_of >= 0 ? st.view_c0 : 0;
    if (st.head == UNINITIALIZED) throw DcError(DC_EINVAL, std::string("'") + o.name + "': run a forward first");
    if (is_device) {
      KCHECK(launch_nhwc_to_nchw(src, o.dst, ses, st.dim(0), st.dim(1), st.dim(2), st.dim(3), scp, sc0, s, dst_esize));
    } else {
      st.ensure_stage(m);  // sized in floats: large enough for either element type
      KCHECK(launch_nhwc_to_nchw(src, st.stage, ses, st.dim(0), st.dim(1), st.dim(2), st.dim(3), scp, sc0, s, dst_esize));
      HIPCHECK(hipMemcpyAsync(o.dst, st.stage, m * (size_t)dst_esize, hipMemcpyDeviceToHost, (hipStream_t)s));
    }
  }
}

void Net::emit_last_maps(void* prob, void* loc, void* next, int elem, bool is_device, void* user_stream) {
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "emit_maps() in CPU mode");
  if (elem != 0 && elem != 1) throw DcError(DC_EINVAL, "element type must be 0 (float32) or 1 (float16)");
  ensure_device();
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  void* s = user_stream ? user_stream : stream;
  emit_maps(prob, loc, next, is_device, s, elem == 1 ? 2 : 4);
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void Net::forward_batch(const float* input, int n, int h, int w, bool is_device, float* prob, float* loc, float* next,
                        void* user_stream, bool host_async) {
  if (host_async && (is_device || user_stream)) throw DcError(DC_EINVAL, "forward_host_async takes host buffers and runs on the net's own stream");
  Storage& in = begin_batch(n, h, w);
  const int C = in.dim(1);
  const bool own_async = user_stream == (void*)-1;  // DC_STREAM_OWN: the net's stream, no final sync
  if (own_async) user_stream = nullptr;
  void* s = user_stream ? user_stream : stream;
  size_t cnt = in.count();
  if (is_device) {
    KCHECK(launch_nchw_to_nhwc(input, in.dev, in.esize, n, C, h, w, in.cp(), s));
  } else {
    in.ensure_stage(cnt);
    HIPCHECK(hipMemcpyAsync(in.stage, input, cnt * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)s));
    KCHECK(launch_nchw_to_nhwc(in.stage, in.dev, in.esize, n, C, h, w, in.cp(), s));
  }
  in.head = HEAD_AT_GPU;
  enqueue_plan(s);
  emit_maps(prob, loc, next, is_device, s);
  if (!(is_device && (user_stream || own_async)) && !host_async) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void Net::forward_host_images(const float* const* inputs, int n, int h, int w) {
  Storage& in = begin_batch(n, h, w);
  const int C = in.dim(1);
  const size_t img = (size_t)C * h * w;
  in.ensure_stage(img * n);
  for (int i = 0; i < n; ++i)
    HIPCHECK(hipMemcpyAsync(reinterpret_cast<float*>(in.stage) + i * img, inputs[i], img * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
  KCHECK(launch_nchw_to_nhwc(reinterpret_cast<const float*>(in.stage), in.dev, in.esize, n, C, h, w, in.cp(), stream));
  in.head = HEAD_AT_GPU;
  enqueue_plan(stream);
}

// n independent requests of one image each -> one batch-n launch plan: at batch 1 a res4 layer is 196 workgroups on 256 CUs
// and a third of its time is fixed cost; the same layers at batch 2-4 fill the chip and pay the fixed cost once.  The
// per-request NCHW device buffers are gathered into / scattered from the batch image by the layout kernels themselves.
void Net::forward_requests(int n, const float* const* inputs, int h, int w, float* const* prob, float* const* loc, float* const* next,
                           void* user_stream) {
  Storage& in = begin_batch(n, h, w);
  const int C = in.dim(1);
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  void* s = user_stream ? user_stream : stream;
  const long img = (long)h * w * in.cp();
  for (int i = 0; i < n; ++i) KCHECK(launch_nchw_to_nhwc(inputs[i], in.dev_at(i * img), in.esize, 1, C, h, w, in.cp(), s));
  in.head = HEAD_AT_GPU;
  enqueue_plan(s);
  struct Out {
    const char* name;
    float* const* dst;
  } outs[3] = {{"prob", prob}, {"loc_pred", loc}, {"next_pred", next}};
  for (auto& o : outs) {
    if (!o.dst) continue;
    const MapRef m = map_ref(o.name);
    const long per = (long)m.H * m.W * m.cp;
    for (int i = 0; i < n; ++i)
      if (o.dst[i])
        KCHECK(launch_nhwc_to_nchw((const unsigned char*)m.ptr + (size_t)i * per * m.es, o.dst[i], m.es, 1, m.C, m.H, m.W, m.cp, m.c0, s));
  }
  if (!(user_stream || own_async)) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

}  // namespace dc
