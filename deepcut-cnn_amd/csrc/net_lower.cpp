// net_lower.cpp — see net_internal.h: lowering of the layer list to a launch plan; the per-shape plan cache.
#include "net_internal.h"

namespace dc {

// ---- lowering ---------------------------------------------------------------------------------------
namespace {
struct LOp {
  enum Kind { CONV, DECONV, POOL, ELT, CROP } kind = CONV;
  std::vector<int> lids;
  int in = -1, in2 = -1, out = -1;
  int wl = -1;
  std::vector<int> wls;      // weight layers when several sibling layers are concatenated along Cout
  std::vector<double> a, b;  // folded per-channel affine (empty = identity)
  int sigmoid_ch = -1;       // >= 0: logistic on the first sigmoid_ch output channels only
  bool relu = false, sigmoid = false;
  int oh = 0, ow = 0;
  bool fused_crop = false;
  bool dead = false;
};

// cost model used to pick the tile variant (cycles; see DESIGN.md "Tile selection")
double variant_cost(const ConvGemmParams& p, int v) {
  const ConvVariant& cv = conv_variant(v);
  int bk = conv_variant_bk(v);
  int FM = cv.BM / cv.WR / 32, FN = cv.BN / cv.WC / 32;
  const double wps = cv.WR * cv.WC * cv.WK / 4.0;  // waves per SIMD of one workgroup
  double wgs = (double)conv_grid(p, v);
  // matrix-pipe cycles per k of one 32x32 fragment: 64/2 (v_mfma_f32_32x32x2_f32) or 32/16 (..._32x32x16_f16)
  const double cyc_per_k = p.esize == 2 ? 2.0 : 32.0;
  double mfma = (double)FM * FN * p.Ktot * cyc_per_k / cv.WK;
  double tiles = (double)p.Ktot / bk;
  // the matrix pipe serialises the MFMAs of co-resident waves; a second wave hides most per-tile overhead
  double per_wg = mfma * wps + tiles * (wps > 1 ? 60.0 : 220.0) + 2500.0;
  double rounds = std::ceil(wgs / 256.0);
  // the matrix pipe is shared by co-resident waves, so rounds serialise; partial last round costs a full one
  double t_mfma = rounds * per_wg;
  double bytes = wgs * (double)p.Ktot * (cv.BM + cv.BN) * (double)p.esize;
  double t_l2 = bytes / 4500.0;  // ~11 TB/s aggregate L2->LDS at 2.4 GHz
  return std::max(t_mfma, t_l2);
}
}  // namespace

// 64-bit content hash of a parameter blob (four independent multiply-xor lanes so that it runs at memory speed)
uint64_t content_hash(const float* p, size_t n) {
  uint64_t h[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
  const uint32_t* u = reinterpret_cast<const uint32_t*>(p);
  size_t i = 0;
  for (; i + 4 <= n; i += 4)
    for (int k = 0; k < 4; ++k) h[k] = (h[k] ^ u[i + k]) * 0x100000001b3ull + (h[k] >> 29);
  for (; i < n; ++i) h[0] = (h[0] ^ u[i]) * 0x100000001b3ull + (h[0] >> 29);
  return (h[0] * 31 + h[1]) * 31 + (h[2] * 31 + h[3]) + n;
}


void Net::build_plan() {
  const int nL = (int)layers.size();
  auto sid = [&](int bi) { return blobs[bi]->st->id; };
  std::vector<LOp> ops;
  std::vector<char> absorbed(nL, 0);

  auto inplace_on = [&](int j, int storage) {
    const LayerRec& L = layers[j];
    return !L.is_split && L.bottoms.size() == 1 && L.tops.size() == 1 && L.bottoms[0] == L.tops[0] &&
           sid(L.tops[0]) == storage;
  };
  auto ensure_affine = [&](LOp& op, int C) {
    if (op.a.empty()) {
      op.a.assign(C, 1.0);
      op.b.assign(C, 0.0);
    }
  };
  auto fold_bn = [&](LOp& op, const LayerRec& L) {  // batch_norm_layer.cpp:86-93,138-149
    int C = L.params[0]->st->dim(0);
    ensure_affine(op, C);
    const float* mean = L.params[0]->st->host_ptr();
    const float* var = L.params[1]->st->host_ptr();
    float sfv = L.params[2]->st->host_ptr()[0];
    double sf = sfv == 0.f ? 0.0 : 1.0 / (double)sfv;
    for (int c = 0; c < C; ++c) {
      double s = 1.0 / std::sqrt((double)var[c] * sf + (double)L.bn_eps);
      op.a[c] = op.a[c] * s;
      op.b[c] = (op.b[c] - (double)mean[c] * sf) * s;
    }
  };
  auto fold_scale = [&](LOp& op, const LayerRec& L) {  // scale_layer.cpp:109-134, bias_layer.cpp:72-87
    int C = L.params[0]->st->dim(0);
    ensure_affine(op, C);
    const float* g = L.params[0]->st->host_ptr();
    const float* be = L.scale_bias ? L.params[1]->st->host_ptr() : nullptr;
    for (int c = 0; c < C; ++c) {
      op.a[c] = op.a[c] * (double)g[c];
      op.b[c] = op.b[c] * (double)g[c] + (be ? (double)be[c] : 0.0);
    }
  };
  // absorb the in-place BatchNorm / Scale / ReLU / Sigmoid layers that directly follow layer i on `op.out`
  auto absorb_chain = [&](LOp& op, int i, bool allow_affine) {
    int j = i + 1;
    while (j < nL && !op.relu && !op.sigmoid && inplace_on(j, op.out)) {
      const LayerRec& L = layers[j];
      if (L.type == "BatchNorm" && allow_affine) fold_bn(op, L);
      else if (L.type == "Scale" && allow_affine) fold_scale(op, L);
      else if (L.type == "ReLU") op.relu = true;
      else if (L.type == "Sigmoid") op.sigmoid = true;
      else break;
      absorbed[j] = 1;
      op.lids.push_back(j);
      ++j;
    }
  };

  // pass 1: one op per layer group
  for (int i = 0; i < nL; ++i) {
    if (absorbed[i] || layers[i].is_split) continue;
    const LayerRec& L = layers[i];
    LOp op;
    op.lids.push_back(i);
    op.in = sid(L.bottoms[0]);
    op.out = sid(L.tops[0]);
    if (L.type == "Convolution" || L.type == "Deconvolution") {
      op.kind = L.type == "Convolution" ? LOp::CONV : LOp::DECONV;
      op.wl = i;
      if (L.conv.bias) {
        const float* bias = L.params[1]->st->host_ptr();
        op.a.assign(L.conv.num_output, 1.0);
        op.b.assign(bias, bias + L.conv.num_output);
      }
      if (op.in != op.out) absorb_chain(op, i, true);
    } else if (L.type == "Pooling") {
      op.kind = LOp::POOL;
    } else if (L.type == "Eltwise") {
      op.kind = LOp::ELT;
      op.in2 = sid(L.bottoms[1]);
      absorb_chain(op, i, false);
    } else if (L.type == "Crop") {
      op.kind = LOp::CROP;
      op.oh = L.crop_oh;
      op.ow = L.crop_ow;
    } else {  // stand-alone BatchNorm / Scale / ReLU / Sigmoid
      op.kind = LOp::ELT;
      if (L.type == "BatchNorm") fold_bn(op, L);
      else if (L.type == "Scale") fold_scale(op, L);
      else if (L.type == "ReLU") op.relu = true;
      else if (L.type == "Sigmoid") op.sigmoid = true;
      if (!op.relu && !op.sigmoid) absorb_chain(op, i, true);
    }
    ops.push_back(std::move(op));
  }

  // pass 1b (DC_OPT_OUTPUTS): ops that only feed unwanted net outputs are dropped — a backward walk over the op list: an op is live
  // if the tensor it writes is a wanted output or is read by a live op further down (in-place ops keep their tensor needed)
  if (outputs_mask != -1) {
    std::vector<char> needed(storages.size(), 0);
    for (size_t i = 0; i < outputs.size(); ++i)
      if (i >= 31 || ((outputs_mask >> i) & 1)) needed[sid(outputs[i])] = 1;
    for (int k = (int)ops.size() - 1; k >= 0; --k) {
      LOp& op = ops[k];
      if (!needed[op.out]) {
        op.dead = true;
        continue;
      }
      if (op.in != op.out && op.in2 != op.out) needed[op.out] = 0;  // produced here: nothing above needs to
      needed[op.in] = 1;
      if (op.in2 >= 0) needed[op.in2] = 1;
    }
  }

  // pass 2: residual-add and deconvolution-head fusion
  if (fuse >= 1) {
    const int nS = (int)storages.size();
    auto analyse = [&](std::vector<int>& prod, std::vector<std::vector<int>>& cons) {
      prod.assign(nS, -1);
      cons.assign(nS, {});
      for (int k = 0; k < (int)ops.size(); ++k) {
        if (ops[k].dead) continue;
        cons[ops[k].in].push_back(k);
        if (ops[k].in2 >= 0) cons[ops[k].in2].push_back(k);
        prod[ops[k].out] = k;
      }
    };
    std::vector<int> prod;
    std::vector<std::vector<int>> cons;
    for (int e = 0; e < (int)ops.size(); ++e) {
      LOp& E = ops[e];
      if (E.dead || E.kind != LOp::ELT || E.in2 < 0 || !E.a.empty() || E.in == E.out || E.in2 == E.out) continue;
      analyse(prod, cons);
      int cand[2][2] = {{E.in2, E.in}, {E.in, E.in2}};
      if (prod[E.in] > prod[E.in2]) std::swap(cand[0], cand[1]);
      for (auto& c : cand) {
        int X = c[0], other = c[1];
        int pk = prod[X];
        if (pk < 0 || cons[X].size() != 1) continue;
        LOp& P = ops[pk];
        if (P.kind == LOp::CONV && !P.relu && !P.sigmoid && P.in2 < 0 && prod[other] < pk && P.in != P.out) {
          P.in2 = other;
          P.out = E.out;
          P.relu = E.relu;
          P.sigmoid = E.sigmoid;
          P.lids.insert(P.lids.end(), E.lids.begin(), E.lids.end());
          E.dead = true;
          break;
        }
        if (P.kind == LOp::CROP) {
          int dk = prod[P.in];
          if (dk < 0 || cons[P.in].size() != 1) continue;
          LOp& D = ops[dk];
          if (D.kind != LOp::DECONV || D.relu || D.sigmoid || D.in2 >= 0) continue;
          LOp F = D;
          F.in2 = other;
          F.out = E.out;
          F.oh = P.oh;
          F.ow = P.ow;
          F.fused_crop = true;
          F.relu = E.relu;
          F.sigmoid = E.sigmoid;
          F.lids.insert(F.lids.end(), P.lids.begin(), P.lids.end());
          F.lids.insert(F.lids.end(), E.lids.begin(), E.lids.end());
          D.dead = true;
          P.dead = true;
          ops[e] = F;  // executes at the Eltwise's position: both operands are ready there
          break;
        }
      }
    }
  }


  // pass 3 (DC_OPT_FUSE >= 2): sibling heads.  The three DeeperCut heads (part scores, location refinement,
  // pairwise regression) are the same Deconvolution on res5c + the same 1x1 skip convolution on res3's
  // last block, differing only in Cout (14 / 28 / 364): run them as ONE 406-channel skip GEMM and ONE
  // 406-channel deconvolution (res5c's 2048-deep rows are read once instead of three times, and the
  // 14/28-channel GEMMs no longer pad to 32-wide MFMA tiles).  The named output blobs become channel
  // views of the concatenated tensor; the Sigmoid of the score head moves into the epilogue.
  for (auto& st : storages) {
    st->view_of = -1;
    st->view_c0 = 0;
    st->view_cp = 0;
  }
  plan_views_.clear();
  if (fuse >= 2) {
    const int nS0 = (int)storages.size();
    std::vector<int> prod(nS0, -1);
    std::vector<std::vector<int>> cons(nS0);
    for (int k = 0; k < (int)ops.size(); ++k) {
      if (ops[k].dead) continue;
      cons[ops[k].in].push_back(k);
      if (ops[k].in2 >= 0) cons[ops[k].in2].push_back(k);
      prod[ops[k].out] = k;
    }
    auto same_geom = [&](int la, int lb) {
      const ConvSpec &x = layers[la].conv, &y = layers[lb].conv;
      return x.kh == y.kh && x.kw == y.kw && x.sh == y.sh && x.sw == y.sw && x.ph == y.ph && x.pw == y.pw && x.dh == y.dh &&
             x.dw == y.dw;
    };
    std::vector<char> used(ops.size(), 0);
    for (int f0 = 0; f0 < (int)ops.size(); ++f0) {
      if (used[f0] || ops[f0].dead || ops[f0].kind != LOp::DECONV || !ops[f0].fused_crop || ops[f0].in2 < 0 || ops[f0].relu ||
          ops[f0].sigmoid || !ops[f0].wls.empty())
        continue;
      std::vector<int> grp;  // deconv ops
      for (int f = f0; f < (int)ops.size(); ++f) {
        const LOp& F = ops[f];
        if (used[f] || F.dead || F.kind != LOp::DECONV || !F.fused_crop || F.in2 < 0 || F.relu || F.sigmoid || !F.wls.empty()) continue;
        if (F.in != ops[f0].in || F.oh != ops[f0].oh || F.ow != ops[f0].ow || !same_geom(F.wl, ops[f0].wl)) continue;
        const int ck = prod[F.in2];
        if (ck < 0 || cons[F.in2].size() != 1) continue;
        const LOp& Cq = ops[ck];
        const int c0k = prod[ops[f0].in2];
        if (Cq.kind != LOp::CONV || Cq.relu || Cq.sigmoid || Cq.in2 >= 0 || !Cq.wls.empty() || c0k < 0 || Cq.in != ops[c0k].in ||
            !same_geom(Cq.wl, ops[c0k].wl))
          continue;
        if (storages[F.out]->shape[2] != storages[ops[f0].out]->shape[2] || storages[F.out]->shape[3] != storages[ops[f0].out]->shape[3])
          continue;
        grp.push_back(f);
      }
      if (grp.size() < 2) continue;
      // sigmoid folding: a head whose only consumer is an out-of-place Sigmoid
      struct Member {
        int f, c, sig_elt, final_out;
      };
      std::vector<Member> mem;
      for (int f : grp) {
        Member m{f, prod[ops[f].in2], -1, ops[f].out};
        const auto& cs = cons[ops[f].out];
        if (cs.size() == 1) {
          const LOp& E = ops[cs[0]];
          if (E.kind == LOp::ELT && E.in2 < 0 && E.a.empty() && E.sigmoid && !E.relu && E.in != E.out) {
            m.sig_elt = cs[0];
            m.final_out = E.out;
          }
        }
        if (m.sig_elt < 0 && !cs.empty()) {  // some other kernel reads this head: it cannot become a strided view
          m.f = -1;
        }
        mem.push_back(m);
      }
      mem.erase(std::remove_if(mem.begin(), mem.end(), [](const Member& m) { return m.f < 0; }), mem.end());
      if (mem.size() < 2) continue;
      std::stable_sort(mem.begin(), mem.end(), [](const Member& x, const Member& y) { return (x.sig_elt >= 0) > (y.sig_elt >= 0); });
      int ctot = 0, sig_ch = 0;
      for (auto& m : mem) {
        const int c = layers[ops[m.f].wl].conv.num_output;
        if (m.sig_elt >= 0) sig_ch += c;
        ctot += c;
      }
      auto aux = [&](const std::string& key, std::vector<int> shape) {
        auto it = aux_index_.find(key);
        int id;
        if (it == aux_index_.end()) {
          auto st = std::make_shared<Storage>();
          st->id = (int)storages.size();
          st->owner = this;
          st->esize = dtype == 1 ? 2 : 4;
          storages.push_back(st);
          id = st->id;
          aux_index_[key] = id;
        } else {
          id = it->second;
        }
        storages[id]->reshape(shape);
        return id;
      };
      const Storage& o0 = *storages[ops[mem[0].f].out];
      const std::string gkey = std::to_string(ops[mem[0].f].lids.front());
      const int T1 = aux("heads_skip:" + gkey, {o0.dim(0), ctot, o0.dim(2), o0.dim(3)});
      const int T2 = aux("heads_out:" + gkey, {o0.dim(0), ctot, o0.dim(2), o0.dim(3)});
      LOp MC = ops[mem[0].c], MF = ops[mem[0].f];
      MC.wls.clear();
      MF.wls.clear();
      MC.lids.clear();
      MF.lids.clear();
      MC.a.clear();
      MC.b.clear();
      MF.a.clear();
      MF.b.clear();
      int c0 = 0;
      for (auto& m : mem) {
        const LOp &Cm = ops[m.c], &Fm = ops[m.f];
        const int c = layers[Fm.wl].conv.num_output;
        MC.wls.push_back(Cm.wl);
        MF.wls.push_back(Fm.wl);
        MC.lids.insert(MC.lids.end(), Cm.lids.begin(), Cm.lids.end());
        MF.lids.insert(MF.lids.end(), Fm.lids.begin(), Fm.lids.end());
        if (m.sig_elt >= 0) MF.lids.insert(MF.lids.end(), ops[m.sig_elt].lids.begin(), ops[m.sig_elt].lids.end());
        for (int k = 0; k < c; ++k) {
          MC.a.push_back(Cm.a.empty() ? 1.0 : Cm.a[k]);
          MC.b.push_back(Cm.b.empty() ? 0.0 : Cm.b[k]);
          MF.a.push_back(Fm.a.empty() ? 1.0 : Fm.a[k]);
          MF.b.push_back(Fm.b.empty() ? 0.0 : Fm.b[k]);
        }
        Storage& v = *storages[m.final_out];
        v.view_of = T2;
        v.view_c0 = c0;
        plan_views_.push_back(m.final_out);
        c0 += c;
      }
      MC.out = T1;
      MF.in2 = T1;
      MF.out = T2;
      MF.sigmoid_ch = sig_ch;
      // the merged ops execute where the LAST member deconvolution stood (all operands are ready there)
      int last_f = 0;
      for (auto& m : mem) {
        last_f = std::max(last_f, m.f);
        ops[m.c].dead = true;
        ops[m.f].dead = true;
        used[m.f] = 1;
        if (m.sig_elt >= 0) ops[m.sig_elt].dead = true;
      }
      for (auto& m : mem)
        if (m.sig_elt >= 0 && m.sig_elt < last_f) { /* sigmoid stood before the last head: fine, it is folded */ }
      ops[last_f] = MF;
      ops[last_f].dead = false;
      used[last_f] = 1;
      ops.insert(ops.begin() + last_f, MC);  // skip GEMM right before it
      used.insert(used.begin() + last_f, 1);
      break;  // one head group per net is all the path has; indices moved, stop scanning
    }
  }

  // tensors combined element-wise / pooled / cropped must agree on channel pitch: propagate before any
  // launch parameters are derived from cp()
  for (bool changed = true; changed;) {
    changed = false;
    for (auto& op : ops) {
      if (op.dead || op.kind == LOp::CONV || op.kind == LOp::DECONV) continue;
      bool p4 = storages[op.in]->pad4 || storages[op.out]->pad4 || (op.in2 >= 0 && storages[op.in2]->pad4);
      if (!p4) continue;
      for (int sx : {op.in, op.in2, op.out})
        if (sx >= 0 && !storages[sx]->pad4) storages[sx]->pad4 = true, changed = true;
    }
  }
  {
    std::vector<char> live(storages.size(), 0);
    for (int bi : inputs) live[blobs[bi]->st->id] = 1;
    for (auto& op : ops)
      if (!op.dead) live[op.out] = 1;
    for (auto& st : storages) st->elided = !live[st->id] && st->view_of < 0;
  }

  // finalize: launches
  plan.clear();
  plan_flops = 0;
  ++stats.lowerings;
  // the packed-image cache is shared with the clones: the first executor to lower after a parameter change empties it
  // (images still referenced by another executor's plans stay alive until that executor re-lowers too)
  std::lock_guard<std::mutex> pack_lock(shared->mu);
  if (shared->packed_gen != shared->weights_gen) {
    shared->vec_by_key.clear();
    shared->packed_gen = shared->weights_gen;
    for (auto& L : layers)
      for (auto& pb : L.params) pb->st->packed_hash = content_hash(pb->st->host_ptr(), pb->st->count());
    ++stats.repacks;
  }
  auto get_vec = [&](const std::string& key, const std::function<void(std::vector<float>&)>& fill) {
    auto it = shared->vec_by_key.find(key);
    if (it != shared->vec_by_key.end()) return it->second;
    auto v = std::make_shared<DevVec>();
    fill(v->host);
    shared->vec_by_key[key] = v;
    return v;
  };
  auto label_of = [&](const LOp& op) {
    std::string s;
    for (size_t k = 0; k < op.lids.size(); ++k) {
      if (k) s += "+";
      s += layers[op.lids[k]].name;
    }
    return s;
  };
  const int force_variant = env_int("DC_CONV_VARIANT", -1);
  const int es = dtype == 1 ? 2 : 4;          // bytes per activation / filter element
  const int kmin = dtype == 1 ? 64 : 32;      // smallest K tile of the dtype's variants (one 128-byte line)
  const std::string dkey = dtype == 1 ? "h:" : "";

  auto affine_vecs = [&](const LOp& op, Launch& l, int C) {
    if (op.a.empty()) return;
    std::string key = std::to_string(op.lids.front()) + ":" + std::to_string(op.lids.size()) + ":" + std::to_string(op.wls.size());
    l.scale = get_vec("a:" + key, [&](std::vector<float>& h) {
      h.resize(C);
      for (int c = 0; c < C; ++c) h[c] = (float)op.a[c];
    });
    l.shift = get_vec("b:" + key, [&](std::vector<float>& h) {
      h.resize(C);
      for (int c = 0; c < C; ++c) h[c] = (float)op.b[c];
    });
  };
  // float16 filter images: per-output-channel power-of-two pre-scaling (exact in fp32; undone by the epilogue's fp32 scale), see
  // DevVec::row_scale.  Called right after the image of a launch is made / found; replaces the launch's scale vector by
  // a[c] * 2^-k(c) (keyed by the image, since k depends on the image's rows).
  static const bool half_rowscale = env_int("DC_HALF_ROWSCALE", 1) != 0;
  auto half_row_scale = [&](Launch& l, const LOp& op, int OC) {
    if (dtype != 1 || !half_rowscale || !l.w) return;
    DevVec& Wv = *l.w;
    if (Wv.row_scale.empty()) {
      if (Wv.host.empty()) return;  // an image uploaded before this feature existed in the process: leave it
      struct Seg { size_t off; int K; };
      std::vector<Seg> segs;
      if (l.cg.ncls > 1)
        for (int q = 0; q < l.cg.ncls; ++q) segs.push_back({(size_t)l.cg.cls[q].w_off, l.cg.cls[q].Ktot});
      else
        segs.push_back({0, l.cg.Ktot});
      for (const Seg& sg : segs)
        if (sg.off + (size_t)OC * sg.K > Wv.host.size())
          throw DcError(DC_EINVAL, "launch '" + l.label + "': filter image of " + std::to_string(Wv.host.size()) + " elements is smaller than " +
                                       std::to_string(OC) + " rows of " + std::to_string(sg.K));
      Wv.row_scale.assign(OC, 1.f);
      for (int c = 0; c < OC; ++c) {
        float mx = 0.f;
        for (const Seg& sg : segs) {
          const float* r = Wv.host.data() + sg.off + (size_t)c * sg.K;
          for (int k = 0; k < sg.K; ++k) mx = std::max(mx, std::fabs(r[k]));
        }
        if (!(mx > 0.f) || !std::isfinite(mx)) continue;
        int k = 13 - std::ilogb(mx);
        k = std::max(-60, std::min(60, k));
        if (k == 0) continue;
        const float f = std::ldexp(1.f, k);
        for (const Seg& sg : segs) {
          float* r = Wv.host.data() + sg.off + (size_t)c * sg.K;
          for (int q = 0; q < sg.K; ++q) r[q] *= f;
        }
        Wv.row_scale[c] = std::ldexp(1.f, -k);
      }
    }
    std::shared_ptr<DevVec> rs = l.w;  // keeps row_scale alive inside the fill
    char wkey[40];  // the image's identity: its address (images and these vectors live and die together in vec_by_key)
    std::snprintf(wkey, sizeof wkey, "%p", (void*)l.w.get());
    l.scale = get_vec(std::string("ha:") + wkey + ":" + std::to_string(op.lids.front()) + ":" + std::to_string(op.lids.size()), [&](std::vector<float>& h) {
      h.resize(OC);
      for (int c = 0; c < OC; ++c) h[c] = (float)((op.a.empty() ? 1.0 : op.a[c]) * (double)rs->row_scale[c]);
    });
  };
  // -1: where measured faster (autotune); 0: never; 1: wherever eligible, 8 waves per workgroup; 2: wherever eligible, the 16-wave form
  const int wino_mode = env_int("DC_WINOGRAD", -1);
  const int stem_mode = env_int("DC_STEM", -1);          // the float16 stem kernel: -1 where measured faster, 0 never, 1 forced
  const int stream_mode = env_int("DC_STREAM1X1", -1);  // the streaming form of the float16 dense 1x1 layers: -1 where measured faster, 0 never, 1 wherever eligible
  auto choose_variant = [&](Launch& l, int kgcd) {
    int best = -1;
    double bc = 0;
    const bool mc = l.cg.ncls > 1;  // multi-class launches need a tile with a multi-class instantiation
    for (int v = 0; v < conv_num_variants(); ++v) {
      if (kgcd % conv_variant_bk(v) != 0 || conv_variant_esize(v) != es || (mc && !conv_variant_multiclass(v))) continue;
      if (force_variant >= 0 && v != force_variant) continue;
      double c = variant_cost(l.cg, v);
      if (best < 0 || c < bc) best = v, bc = c;
    }
    if (best < 0)
      for (int v = 0; v < conv_num_variants(); ++v) {
        if (kgcd % conv_variant_bk(v) != 0 || conv_variant_esize(v) != es || (mc && !conv_variant_multiclass(v))) continue;
        double c = variant_cost(l.cg, v);
        if (best < 0 || c < bc) best = v, bc = c;
      }
    if (best < 0) throw DcError(DC_EUNSUP, "launch '" + l.label + "': no tile variant takes K segments of " + std::to_string(kgcd) + " elements");
    l.variant = best;
    l.kernel = std::string("conv_gemm<") + conv_variant(best).name + ">";
    l.grid = conv_grid(l.cg, best);
  };
  auto use_wino = [&](Launch& l, int wv) {
    l.variant = wv;
    l.kernel = wino_kernel_label(wv);
    l.grid = wino_grid(l.cg);
  };

  // Channel split of a wide-but-ragged GEMM (the merged heads: N = 406 = 3 x 128 + 22).  On 128-wide tiles a quarter of the
  // fourth column block is padding (26 % of the launch's MFMA work and filter fetches for nothing); as two launches — the
  // first floor(N / 128) * 128 channels, then the tail on a narrow tile — the padding is 22 -> 32/64 channels.  Host-side only:
  // the second launch is the same kernel on offset filter rows / epilogue constants / output channels.
  // DC_HEAD_SPLIT=1 switches it on.  OFF by default — measured (round 4, EXPERIMENTS.md): the tail launch re-reads all of res5c's
  // 2048-deep rows for 22 channels (float16 grouped pyramid: 900 us -> 720 + 177 us; float32 batch 1: 457 -> 459 images/s in
  // flight, 337 -> 333 alone): the padding it removes is paid back as operand traffic.  Kept as a switch with its parity test.
  const int head_split = env_int("DC_HEAD_SPLIT", 0);
  auto push_split = [&](Launch&& l, int kgcd) {
    const int OC = l.cg.Cout;
    const bool want = head_split == 1;
    const int c0 = OC / 128 * 128;
    if (!want || OC < 256 || c0 == OC || OC - c0 > 64 || force_variant >= 0) {
      plan.push_back(std::move(l));
      return;
    }
    Launch a = l, b = l;
    const double fa = (double)c0 / OC;
    a.cg.Cout = c0;
    a.flops = l.flops * fa;
    a.label += " [ch 0-" + std::to_string(c0 - 1) + "]";
    a.cg.sigmoid_ch = std::min(l.cg.sigmoid_ch, c0);
    b.cg.Cout = OC - c0;
    b.flops = l.flops * (1.0 - fa);
    b.label += " [ch " + std::to_string(c0) + "-" + std::to_string(OC - 1) + "]";
    b.cg.sigmoid_ch = std::max(0, l.cg.sigmoid_ch - c0);
    b.y_off = l.y_off + c0;
    b.c_off = l.c_off + c0;
    if (l.cg.ncls > 1) {
      for (int q = 0; q < l.cg.ncls; ++q) b.cg.cls[q].w_off = l.cg.cls[q].w_off + (long)c0 * l.cg.cls[q].Ktot;
    } else {
      b.w_off = l.w_off + (long)c0 * l.cg.Ktot;
    }
    choose_variant(a, kgcd);
    choose_variant(b, kgcd);
    plan.push_back(std::move(a));
    plan.push_back(std::move(b));
  };

  for (auto& op : ops) {
    if (op.dead) continue;
    Launch base;
    base.label = label_of(op);
    base.first_layer = *std::min_element(op.lids.begin(), op.lids.end());
    base.last_layer = *std::max_element(op.lids.begin(), op.lids.end());
    base.in = op.in;
    base.in2 = op.in2;
    base.out = op.out;
    base.relu = op.relu;
    base.sigmoid = op.sigmoid;
    Storage& X = *storages[op.in];
    Storage& Y = *storages[op.out];
    const int N = X.dim(0), C = X.dim(1), H = X.dim(2), W = X.dim(3);
    const int CP = X.cp();
    const int OC = Y.dim(1), OHt = Y.dim(2), OWt = Y.dim(3), OCP = Y.cp();
    if (op.kind == LOp::CONV) {
      const LayerRec& L = layers[op.wl];
      const ConvSpec& c = L.conv;
      Launch l = base;
      l.kind = Launch::CONV;
      ConvGemmParams& g = l.cg;
      g.esize = es;
      g.x_img_stride = (long)H * W * CP;
      g.x_row_stride = W * CP;
      g.x_rows = H;
      g.x_rowlen = W * CP;
      g.sy = c.sh;
      g.sx = c.sw * CP;
      int kgcd;
      const bool rowtap = (CP % kmin) != 0;
      if (rowtap) {
        // small-channel input (the 3->4 channel stem): one tap per kernel ROW, the kw adjacent pixels of
        // that row being contiguous in NHWC; K per tap = kw*CP rounded up to 32 with zero weights
        if (c.dw != 1 || CP % (16 / es) != 0)
          throw DcError(DC_EUNSUP, "layer '" + L.name + "': convolution over " + std::to_string(C) +
                                       " channels needs dilation_w 1 (row-tap path) or a multiple of " + std::to_string(kmin) + " channels");
        int klen = (c.kw * CP + kmin - 1) / kmin * kmin;
        if (c.kh > kMaxTaps) throw DcError(DC_EUNSUP, "layer '" + L.name + "': kernel too tall");
        g.nty = c.kh;
        g.ntx = 1;
        g.dy0 = -c.ph;
        g.ddy = c.dh;
        g.x0 = -c.pw * CP;
        g.ddx = 0;
        g.klen = klen;
        g.Ktot = c.kh * klen;
        kgcd = klen;
        // sibling layers merged into one launch (the skip convolutions of the heads) are concatenated along Cout here too:
        // round 2 packed only the first member in this path — a float16 net whose skip level has fewer than 64 channels
        // ran its second and third head on rows beyond the image (found by the row scaling's bounds check in round 3)
        const std::vector<int> members = op.wls.empty() ? std::vector<int>{op.wl} : op.wls;
        l.w = get_vec(dkey + "w:" + std::to_string(members.front()) + "x" + std::to_string(members.size()) + "r", [&](std::vector<float>& h) {
          h.assign((size_t)OC * g.Ktot, 0.f);
          int cbase = 0;
          for (int ml : members) {
            const float* w = layers[ml].params[0]->st->host_ptr();  // [Cout][Cin][kh][kw]
            const int cm = layers[ml].conv.num_output;
            for (int co = 0; co < cm; ++co)
              for (int ci = 0; ci < C; ++ci)
                for (int ky = 0; ky < c.kh; ++ky)
                  for (int kx = 0; kx < c.kw; ++kx)
                    h[(size_t)(cbase + co) * g.Ktot + ky * klen + kx * CP + ci] = w[(((size_t)co * C + ci) * c.kh + ky) * c.kw + kx];
            cbase += cm;
          }
        });
      } else {
        if (c.kh * c.kw > kMaxTaps)
          throw DcError(DC_EUNSUP, "layer '" + L.name + "': more than " + std::to_string(kMaxTaps) + " kernel taps");
        g.nty = c.kh;
        g.ntx = c.kw;
        g.dy0 = -c.ph;
        g.ddy = c.dh;
        g.x0 = -c.pw * CP;
        g.ddx = c.dw * CP;
        g.klen = CP;
        g.Ktot = c.kh * c.kw * CP;
        kgcd = CP;
        const std::vector<int> members = op.wls.empty() ? std::vector<int>{op.wl} : op.wls;
        l.w = get_vec(dkey + "w:" + std::to_string(members.front()) + "x" + std::to_string(members.size()), [&](std::vector<float>& h) {
          h.assign((size_t)OC * g.Ktot, 0.f);
          const int taps = c.kh * c.kw;
          int cbase = 0;
          for (int ml : members) {  // sibling layers concatenated along Cout
            const float* w = layers[ml].params[0]->st->host_ptr();
            const int cm = layers[ml].conv.num_output;
            for (int co = 0; co < cm; ++co)
              for (int ci = 0; ci < C; ++ci) {
                const float* src = w + ((size_t)co * C + ci) * taps;
                float* dst = h.data() + (size_t)(cbase + co) * g.Ktot + ci;
                for (int tp = 0; tp < taps; ++tp) dst[(size_t)tp * CP] = src[tp];
              }
            cbase += cm;
          }
        });
      }
      g.NB = N;
      g.OH = OHt;
      g.OW = OWt;
      g.M = N * OHt * OWt;
      g.Cout = OC;
      g.y_img_stride = (long)OHt * OWt * OCP;
      g.y_row_stride = OWt * OCP;
      g.y_pix_stride = OCP;
      g.relu = op.relu;
      g.sigmoid_ch = op.sigmoid ? OC : 0;
      affine_vecs(op, l, OC);
      l.flops = 2.0 * g.M * (double)OC * C * c.kh * c.kw;
      plan_flops += l.flops;
      l.w->as_half = dtype == 1;
          half_row_scale(l, op, OC);
      choose_variant(l, kgcd);
      // stride-1 3x3 layers can also run as Winograd F(2x2,3x3): keep the transformed filters next to the direct ones
      // and let the per-shape timing decide (kernels.hip, wino_f23_kernel)
      if (!rowtap && wino_mode != 0 && op.wls.empty() && wino_eligible(g) && dtype == 0) {
        l.wino_w = get_vec(dkey + "wino:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          h.assign(wino_packed_floats(c.num_output, C), 0.f);
          wino_pack_filters(L.params[0]->st->host_ptr(), c.num_output, C, h.data());
        });
        if (wino_mode >= 1 && (force_variant < 0 || is_wino_variant(force_variant))) use_wino(l, wino_mode == 2 ? kWinoVariant16 : kWinoVariant);
      } else if (!rowtap && wino_mode != 0 && op.wls.empty() && dtype == 1 && op.in2 < 0 && wino_eligible(g)) {
        // float16 (wino_f16.hip): the image is packed in MFMA fragment order with its own per-channel power-of-two row scale
        // (DevVec::row_scale: G g G^T has other maxima than g); the form's epilogue scale = folded affine x row scale x 4 (the
        // kernel stages the pixels pre-multiplied by 1/4 so that B^T d B cannot overflow float16)
        std::vector<float> rs_made;  // filled only if the image is packed now (else the DevVec found in the cache carries its row scale)
        l.wino_w = get_vec(dkey + "wino:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          h.assign(wino_half_packed_elems(c.num_output, C), 0.f);
          rs_made.assign(c.num_output, 1.f);
          wino_half_pack_filters(L.params[0]->st->host_ptr(), c.num_output, C, half_rowscale, h.data(), rs_made.data());
        });
        if (!rs_made.empty()) l.wino_w->row_scale = std::move(rs_made);
        l.wino_w->as_half = true;
        std::shared_ptr<DevVec> ws = l.wino_w;
        char wkey[40];
        std::snprintf(wkey, sizeof wkey, "%p", (void*)ws.get());
        l.wino_scale = get_vec(std::string("hwa:") + wkey + ":" + std::to_string(op.lids.front()) + ":" + std::to_string(op.lids.size()), [&](std::vector<float>& h) {
          h.resize(OC);
          for (int q = 0; q < OC; ++q) h[q] = (float)((op.a.empty() ? 1.0 : op.a[q]) * 4.0 * (double)(ws->row_scale.empty() ? 1.f : ws->row_scale[q]));
        });
        if (wino_mode >= 1 && (force_variant < 0 || is_wino_variant(force_variant))) use_wino(l, kWinoHalf);
      } else if (!rowtap && stream_mode != 0 && op.wls.empty() && dtype == 1 && g.klen == C && g.Ktot == C && stream1x1_eligible(g)) {
        // float16 dense 1x1 layers (stream1x1.hip): the same row-scaled filters as the direct image, in MFMA fragment order;
        // scale / shift / shortcut are the launch's own.  The per-shape timing decides (DC_STREAM1X1=1: wherever eligible, 0: never)
        std::shared_ptr<DevVec> direct = l.w;
        l.wino_w = get_vec(dkey + "ws:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          const float* src = L.params[0]->st->host_ptr();  // [OC][C][1][1]
          std::vector<float> scaled((size_t)OC * C);
          for (int co = 0; co < OC; ++co) {
            const float f = direct->row_scale.empty() ? 1.f : 1.f / direct->row_scale[co];  // an exact power of two
            for (int k = 0; k < C; ++k) scaled[(size_t)co * C + k] = src[(size_t)co * C + k] * f;
          }
          h.assign(stream1x1_packed_elems(OC, C), 0.f);
          stream1x1_pack_filters(scaled.data(), OC, C, h.data());
        });
        l.wino_w->as_half = true;
        if (stream_mode >= 1 && (force_variant < 0 || is_wino_variant(force_variant))) use_wino(l, kStreamHalf);
      }
      if (!rowtap && !l.wino_w && stream_mode != 0 && op.wls.empty() && dtype == 0 && g.klen == C && g.Ktot == C && stream1x1f_eligible(g)) {
        // float32 dense 1x1 layers with 64 / 128 / 256 / 512 input channels (stream1x1_f32.hip): the filters in the order of its 16x16x4 matrix steps
        l.wino_w = get_vec(dkey + "wsf:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          h.assign(stream1x1f_packed_elems(OC, C), 0.f);
          stream1x1f_pack_filters(L.params[0]->st->host_ptr(), OC, C, h.data());
        });
        if (stream_mode >= 1 && (force_variant < 0 || is_wino_variant(force_variant))) use_wino(l, kStreamFloat);
      }
      if (rowtap && stem_mode != 0 && op.wls.empty() && dtype == 1 && C <= 4 && stem7x7_eligible(g)) {
        // float16 stem (stem_f16.hip): the same row-scaled filters as the row-tap image, the 28 real elements of every kernel row in MFMA
        // operand order; scale / shift are the launch's own.  The per-shape timing decides (DC_STEM=1: forced, 0: never)
        std::shared_ptr<DevVec> direct = l.w;
        l.wino_w = get_vec(dkey + "stem:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          const float* src = L.params[0]->st->host_ptr();  // [64][C][7][7]
          std::vector<float> scaled((size_t)64 * C * 49);
          for (int co = 0; co < 64; ++co) {
            const float f = direct->row_scale.empty() ? 1.f : 1.f / direct->row_scale[co];  // an exact power of two
            for (int q = 0; q < C * 49; ++q) scaled[(size_t)co * C * 49 + q] = src[(size_t)co * C * 49 + q] * f;
          }
          h.assign(stem7x7_packed_elems(), 0.f);
          stem7x7_pack_filters(scaled.data(), C, h.data());
        });
        l.wino_w->as_half = true;
        if (stem_mode >= 1 && (force_variant < 0 || is_wino_variant(force_variant))) use_wino(l, kStemHalf);
      }
      if (rowtap && stem_mode != 0 && op.wls.empty() && dtype == 0 && C <= 4 && stem_ws_eligible(g)) {
        // float32 stem on the streaming skeleton (stream1x1_f32.hip, "ws7x7f"): the row-tap image's 224 columns + 32 of zeros in the order of
        // its 16x16x4 matrix steps; scale / shift are the launch's own.  The per-shape timing decides (DC_STEM=1: forced, 0: never)
        std::shared_ptr<DevVec> direct = l.w;
        l.wino_w = get_vec(dkey + "stemws:" + std::to_string(op.wl), [&](std::vector<float>& h) {
          const float* w = L.params[0]->st->host_ptr();  // [64][C][7][7]
          std::vector<float> rt((size_t)64 * 224, 0.f);  // the row-tap order: k = ky 32 + kx 4 + ci
          for (int co = 0; co < 64; ++co)
            for (int ci = 0; ci < C; ++ci)
              for (int ky = 0; ky < 7; ++ky)
                for (int kx = 0; kx < 7; ++kx) rt[(size_t)co * 224 + ky * 32 + kx * 4 + ci] = w[(((size_t)co * C + ci) * 7 + ky) * 7 + kx];
          h.assign(stem_ws_packed_elems(), 0.f);
          stem_ws_pack_filters(rt.data(), h.data());
        });
        if (stem_mode >= 1 && (force_variant < 0 || is_wino_variant(force_variant))) use_wino(l, kStemFloat);
      }
      if (l.wino_w || rowtap) plan.push_back(std::move(l));
      else push_split(std::move(l), kgcd);
    } else if (op.kind == LOp::DECONV) {
      // stride-s transposed convolution = s*s ordinary gather-GEMMs, one per output residue class
      // (Y mod s, X mod s): output pixel (s*i + r) receives tap k iff (r + p - k*d) % s == 0, from input
      // row i + (r + p - k*d)/s  (col2im_cpu, im2col.cpp:163-197, inverted: output-stationary).
      const LayerRec& L = layers[op.wl];
      const ConvSpec& c = L.conv;
      if (CP % kmin != 0)
        throw DcError(DC_EUNSUP, "layer '" + L.name + "': deconvolution input channels must be a multiple of " + std::to_string(kmin));
      const int DH = c.sh * (H - 1) + c.dh * (c.kh - 1) + 1 - 2 * c.ph;  // full deconv output
      const int DW = c.sw * (W - 1) + c.dw * (c.kw - 1) + 1 - 2 * c.pw;
      const int oh = op.fused_crop ? op.oh : 0, ow = op.fused_crop ? op.ow : 0;
      plan_flops += 2.0 * (double)C * H * W * N * OC * c.kh * c.kw;  // SURVEY §8(d) definition (OC = all member heads)
      // one record per residue class; they become ONE multi-class launch (kernels.h ConvClass) when a multi-class
      // instantiation of the chosen tile exists, else one launch each
      struct ClassRec {
        ConvGemmParams g;
        long y_off;
        int ry, rx;
        std::vector<std::pair<int, int>> tky, tkx;
      };
      std::vector<ClassRec> recs;
      for (int ry = 0; ry < c.sh; ++ry)
        for (int rx = 0; rx < c.sw; ++rx) {
          // rows of this class inside the (cropped) output window: Y = s*i + ry, y = Y - oh in [0, OHt)
          auto range = [](int r, int s, int off, int outn, int full, int& i0, int& cnt) {
            int lo = off - r;  // s*i >= lo
            i0 = lo <= 0 ? 0 : (lo + s - 1) / s;
            int hiY = std::min(full, off + outn) - 1;  // last Y
            int i1 = (hiY - r) >= 0 ? (hiY - r) / s : -1;
            cnt = i1 - i0 + 1;
          };
          int i0, nh, j0, nw;
          range(ry, c.sh, oh, OHt, DH, i0, nh);
          range(rx, c.sw, ow, OWt, DW, j0, nw);
          if (nh <= 0 || nw <= 0) continue;
          ClassRec rec;
          rec.ry = ry, rec.rx = rx;
          auto &tky = rec.tky, &tkx = rec.tkx;  // (k, source offset)
          for (int k = 0; k < c.kh; ++k)
            if ((ry + c.ph - k * c.dh) % c.sh == 0) tky.push_back({k, (ry + c.ph - k * c.dh) / c.sh});
          for (int k = 0; k < c.kw; ++k)
            if ((rx + c.pw - k * c.dw) % c.sw == 0) tkx.push_back({k, (rx + c.pw - k * c.dw) / c.sw});
          ConvGemmParams& g = rec.g;
          g = ConvGemmParams{};
          g.esize = es;
          g.x_img_stride = (long)H * W * CP;
          g.x_row_stride = W * CP;
          g.x_rows = H;
          g.x_rowlen = W * CP;
          g.sy = 1;
          g.sx = CP;
          const int ntaps = (int)(tky.size() * tkx.size());
          if (ntaps > kMaxTaps) throw DcError(DC_EUNSUP, "layer '" + L.name + "': too many taps");
          if (ntaps == 0) throw DcError(DC_EUNSUP, "layer '" + L.name + "': deconvolution with kernel smaller than stride");
          // the taps of a residue class form an arithmetic grid (k advances by s/gcd(s,d))
          g.nty = (int)tky.size();
          g.ntx = (int)tkx.size();
          g.dy0 = tky[0].second + i0;
          g.ddy = tky.size() > 1 ? tky[1].second - tky[0].second : 0;
          g.x0 = (tkx[0].second + j0) * CP;
          g.ddx = tkx.size() > 1 ? (tkx[1].second - tkx[0].second) * CP : 0;
          for (size_t q = 1; q < tky.size(); ++q)
            if (tky[q].second - tky[q - 1].second != g.ddy) throw DcError(DC_EUNSUP, "layer '" + L.name + "': irregular tap grid");
          for (size_t q = 1; q < tkx.size(); ++q)
            if ((tkx[q].second - tkx[q - 1].second) * CP != g.ddx) throw DcError(DC_EUNSUP, "layer '" + L.name + "': irregular tap grid");
          g.klen = CP;
          g.Ktot = ntaps * CP;
          g.NB = N;
          g.OH = nh;
          g.OW = nw;
          g.M = N * nh * nw;
          g.Cout = OC;
          g.y_img_stride = (long)OHt * OWt * OCP;
          g.y_row_stride = c.sh * OWt * OCP;
          g.y_pix_stride = c.sw * OCP;
          rec.y_off = ((long)(c.sh * i0 + ry - oh) * OWt + (c.sw * j0 + rx - ow)) * OCP;
          g.relu = op.relu;
          g.sigmoid_ch = op.sigmoid_ch >= 0 ? op.sigmoid_ch : (op.sigmoid ? OC : 0);
          recs.push_back(std::move(rec));
        }
      if (recs.empty()) throw DcError(DC_ESHAPE, "layer '" + L.name + "': empty deconvolution output");
      const std::vector<int> members = op.wls.empty() ? std::vector<int>{op.wl} : op.wls;
      // filter image of one class: [OC][taps of the class][CP], sibling layers concatenated along Cout
      auto fill_class = [&](const ClassRec& rec, float* h) {
        int cbase = 0;
        for (int ml : members) {
          const float* w = layers[ml].params[0]->st->host_ptr();  // [Cin][Cout][kh][kw]
          const int cm = layers[ml].conv.num_output;
          int t2 = 0;
          for (auto& a : rec.tky)
            for (auto& b : rec.tkx) {
              for (int co = 0; co < cm; ++co)
                for (int ci = 0; ci < C; ++ci)
                  h[(size_t)(cbase + co) * rec.g.Ktot + (size_t)t2 * CP + ci] = w[(((size_t)ci * cm + co) * c.kh + a.first) * c.kw + b.first];
              ++t2;
            }
          cbase += cm;
        }
      };
      const std::string wkey = dkey + "w:" + std::to_string(members.front()) + "x" + std::to_string(members.size());
      bool merged = false;
      if (recs.size() > 1 && (int)recs.size() <= kMaxClasses && env_int("DC_DECONV_MERGE", 1) != 0) {
        // heaviest class first: its workgroups are dispatched first, the light classes fill the tail
        std::stable_sort(recs.begin(), recs.end(), [](const ClassRec& a, const ClassRec& b) { return a.g.Ktot > b.g.Ktot; });
        Launch l = base;
        l.kind = Launch::CONV;
        l.label += " [" + std::to_string(recs.size()) + " classes]";
        l.cg = recs[0].g;
        l.cg.ncls = (int)recs.size();
        long woff = 0;
        for (size_t q = 0; q < recs.size(); ++q) {
          const ConvGemmParams& g = recs[q].g;
          ConvClass& k = l.cg.cls[q];
          k.nty = g.nty, k.ntx = g.ntx, k.dy0 = g.dy0, k.ddy = g.ddy, k.x0 = g.x0, k.ddx = g.ddx, k.Ktot = g.Ktot;
          k.OH = g.OH, k.OW = g.OW, k.M = g.M;
          k.w_off = woff;
          k.y_off = recs[q].y_off;
          woff += (long)OC * g.Ktot;
          l.flops += 2.0 * g.M * (double)OC * C * g.nty * g.ntx;
        }
        affine_vecs(op, l, OC);
        // a multi-class tile must exist among the candidates of this K granularity (or be the forced one)
        bool have_mc = false;
        for (int v = 0; v < conv_num_variants(); ++v)
          if (CP % conv_variant_bk(v) == 0 && conv_variant_esize(v) == es && conv_variant_multiclass(v) &&
              (force_variant < 0 || force_variant == v))
            have_mc = true;
        if (have_mc) {
          std::string key = wkey + ":mc";
          for (auto& r : recs) key += ":" + std::to_string(r.ry) + "," + std::to_string(r.rx);
          l.w = get_vec(key, [&](std::vector<float>& h) {
            h.assign((size_t)woff, 0.f);
            for (size_t q = 0; q < recs.size(); ++q) fill_class(recs[q], h.data() + l.cg.cls[q].w_off);
          });
          l.w->as_half = dtype == 1;
          half_row_scale(l, op, OC);
          choose_variant(l, CP);
          push_split(std::move(l), CP);
          merged = true;
        }
      }
      if (!merged)
        for (auto& rec : recs) {
          Launch l = base;
          l.kind = Launch::CONV;
          l.label += " [class " + std::to_string(rec.ry) + "," + std::to_string(rec.rx) + "]";
          l.cg = rec.g;
          l.y_off = rec.y_off;
          affine_vecs(op, l, OC);
          l.w = get_vec(wkey + ":" + std::to_string(rec.ry) + "," + std::to_string(rec.rx), [&](std::vector<float>& h) {
            h.assign((size_t)OC * rec.g.Ktot, 0.f);
            fill_class(rec, h.data());
          });
          l.flops = 2.0 * rec.g.M * (double)OC * C * rec.g.nty * rec.g.ntx;
          l.w->as_half = dtype == 1;
          half_row_scale(l, op, OC);
          choose_variant(l, CP);
          plan.push_back(std::move(l));
        }
    } else if (op.kind == LOp::POOL) {
      const LayerRec& L = layers[op.lids[0]];
      Launch l = base;
      l.kind = Launch::POOL;
      l.kernel = "maxpool";
      l.pk = L.pool_k;
      l.ps = L.pool_s;
      l.pp = L.pool_p;
      plan.push_back(std::move(l));
    } else if (op.kind == LOp::ELT) {
      Launch l = base;
      l.kind = Launch::ELT;
      l.kernel = "eltwise";
      affine_vecs(op, l, C);
      plan.push_back(std::move(l));
    } else {
      Launch l = base;
      l.kind = Launch::CROP;
      l.kernel = "crop";
      l.oh = op.oh;
      l.ow = op.ow;
      plan.push_back(std::move(l));
    }
  }
  plan_valid = true;
  tuned = false;
  plan_input_shape.clear();
  for (int bi : inputs)
    for (int d : blobs[bi]->st->shape) plan_input_shape.push_back(d);
  cur_last_use_ = ++use_clock_;
  release_graph();
}

// ---- per-shape plan cache -----------------------------------------------------------------------------
static std::vector<int> input_signature(const Net& n) {
  std::vector<int> sig;
  for (int bi : n.inputs)
    for (int d : n.blobs[bi]->st->shape) sig.push_back(d);
  return sig;
}

void Net::mark_weights_changed() {
  std::lock_guard<std::mutex> lk(shared->mu);
  ++shared->weights_gen;
}

// Parameters are handed out writable on every access (pycaffe's Blob.data is mutable_cpu_data, _caffe.cpp:273), so an
// access alone says nothing: the blobs touched since the last run are re-hashed here and only a CONTENT change moves
// the shared generation.  Every executor of the model compares that generation with the one its plans came from.
void Net::check_weights() {
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lk(shared->mu);
    if (!shared->touched.empty()) {
      bool changed = shared->packed_gen != shared->weights_gen;  // nothing packed yet: hashes are not meaningful
      for (auto& w : shared->touched)
        if (auto st = w.lock()) {
          st->touch_listed = false;
          if (!changed && content_hash(st->host_ptr(), st->count()) != st->packed_hash) changed = true;
        }
      shared->touched.clear();
      if (changed && shared->packed_gen == shared->weights_gen) ++shared->weights_gen;
    }
    gen = shared->weights_gen;
  }
  if (gen != seen_weights_gen) {
    invalidate_plans();
    seen_weights_gen = gen;
  }
}

void Net::invalidate_plans() {
  if (stream && (plan_valid || !parked_.empty())) (void)hipStreamSynchronize((hipStream_t)stream);  // nothing in flight reads them
  release_graph();
  for (auto& ps : parked_)
    if (ps->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)ps->graph_exec);
  parked_.clear();
  plan.clear();
  plan_valid = false;
  tuned = false;
}

void Net::park_current() {
  if (!plan_valid) return;
  std::unique_ptr<PlanState> ps(new PlanState());
  ps->input_shape = plan_input_shape;
  ps->plan.swap(plan);
  ps->flops = plan_flops;
  ps->views.swap(plan_views_);
  for (auto& st : storages) ps->sstate.push_back({st->id, st->view_of, st->view_c0, st->elided});
  for (auto& kv : aux_index_) ps->aux_shapes.push_back({kv.second, storages[kv.second]->shape});
  ps->graph_exec = graph_exec;
  ps->graph_buf_gen = graph_buf_gen;
  ps->tuned = tuned;
  ps->last_use = cur_last_use_;
  graph_exec = nullptr;
  plan_valid = false;
  parked_.push_back(std::move(ps));
  static const int cap = std::max(1, env_int("DC_PLAN_CACHE", 16));
  while ((int)parked_.size() > cap) {  // least recently used shape goes
    size_t lru = 0;
    for (size_t i = 1; i < parked_.size(); ++i)
      if (parked_[i]->last_use < parked_[lru]->last_use) lru = i;
    if (parked_[lru]->graph_exec) {
      if (stream) (void)hipStreamSynchronize((hipStream_t)stream);
      (void)hipGraphExecDestroy((hipGraphExec_t)parked_[lru]->graph_exec);
    }
    parked_.erase(parked_.begin() + lru);
  }
}

// Make the plan of the CURRENT input shape the active one.  Shapes of every blob are re-derived first (Layer::Forward
// calls Reshape on every forward, layer.hpp:451-456); a shape met before costs that walk and a swap, nothing else.
void Net::ensure_plan() {
  check_weights();
  reshape();
  const std::vector<int> sig = input_signature(*this);
  if (plan_valid && sig == plan_input_shape) {
    cur_last_use_ = ++use_clock_;
    return;
  }
  for (size_t i = 0; i < parked_.size(); ++i) {
    if (parked_[i]->input_shape != sig) continue;
    std::unique_ptr<PlanState> ps = std::move(parked_[i]);
    parked_.erase(parked_.begin() + i);
    park_current();
    plan_input_shape = ps->input_shape;
    plan.swap(ps->plan);
    plan_flops = ps->flops;
    plan_views_.swap(ps->views);
    for (auto& ss : ps->sstate) {
      Storage& st = *storages[ss.id];
      st.view_of = ss.view_of, st.view_c0 = ss.view_c0, st.elided = ss.elided;
    }
    for (auto& as : ps->aux_shapes) storages[as.first]->reshape(as.second);
    graph_exec = ps->graph_exec;
    graph_buf_gen = ps->graph_buf_gen;
    tuned = ps->tuned;
    plan_valid = true;
    cur_last_use_ = ++use_clock_;
    ++stats.plan_hits;
    return;
  }
  park_current();
  build_plan();
}

void Net::reserve(int n, int h, int w) { begin_batch(n, h, w); }

}  // namespace dc
