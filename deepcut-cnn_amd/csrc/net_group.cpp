// net_group.cpp — see net_internal.h / net.h: NetGroup, the same model over several tensors as one launch sequence.
#include "net_internal.h"

namespace dc {

// ---- NetGroup: the same model over several tensors as ONE launch sequence (net.h) ---------------------------------------
// Candidate streams for the lanes of groups (per device, process-wide, never destroyed: a destroyed stream would hand its
// hardware-queue slot to the next one created).  WHICH of them a group's lanes run on is decided by measurement
// (NetGroup::choose_lane_streams): a HIP process has a handful of hardware queues, the runtime binds a stream to one of them at
// creation, and whether two streams really run side by side cannot be asked.
static std::mutex g_lane_mu;
static std::map<void*, int> g_lane_users;  // candidate stream -> groups whose lanes run on it (two groups in flight must not share one)
// the candidates are the process-wide executor pool of streams.cpp, handed out as a COPY: the pool grows under its own lock, and
// a reference into it dangled when another thread's group asked for more lanes (round-4 advice)
static std::vector<void*> lane_stream_candidates(int device, size_t want) { return executor_stream_pool(device, want); }
static void lane_streams_release(const std::vector<void*>& side) {
  std::lock_guard<std::mutex> lk(g_lane_mu);
  for (void* st : side)
    if (st && g_lane_users[st] > 0) --g_lane_users[st];
}

NetGroup* NetGroup::create(const std::vector<Net*>& members) {
  if (members.empty()) throw DcError(DC_EINVAL, "a group needs at least one net");
  for (Net* n : members) {
    if (!n) throw DcError(DC_EINVAL, "null net in group");
    if (n->shared != members[0]->shared)
      throw DcError(DC_EINVAL, "the members of a group must be executors of ONE model: a net and its clones (dc_net_clone)");
    if (n->dtype != members[0]->dtype || n->fuse != members[0]->fuse)
      throw DcError(DC_EINVAL, "the members of a group must agree on DC_OPT_DTYPE and DC_OPT_FUSE");
    if (n->inputs.size() != 1) throw DcError(DC_EINVAL, "group members must be single-input nets");
  }
  for (size_t i = 0; i < members.size(); ++i)
    for (size_t j = i + 1; j < members.size(); ++j)
      if (members[i] == members[j]) throw DcError(DC_EINVAL, "the same net twice in a group (every member needs its own activations: clone it)");
  std::unique_ptr<NetGroup> g(new NetGroup());
  g->nets = members;
  return g.release();
}

void* NetGroup::stream() { return nets[0]->stream; }

void NetGroup::drop_plan(GroupPlan& gp) {
  // nothing enqueued may still replay the graphs.  The device-wide wait covers the members' streams, the lanes' and the caller's
  // without touching a member: a group may be destroyed AFTER its nets (a garbage collector finalises a cycle in any order)
  {
    RuntimeLock rl;  // (no capture of ours is open while the device is waited for)
    (void)hipDeviceSynchronize();
  }
  gp.drop_graphs();
}

NetGroup::~NetGroup() {
  for (auto& gp : plans_) drop_plan(*gp);
  for (void* e : lane_events_) (void)hipEventDestroy((hipEvent_t)e);
  if (fork_event_) (void)hipEventDestroy((hipEvent_t)fork_event_);
  for (auto& kv : lane_choice_) lane_streams_release(kv.second);  // (the streams themselves belong to the process-wide list)
}

void GroupPlan::drop_graphs() {
  for (void*& g : lane_graphs)
    if (g) (void)hipGraphExecDestroy((hipGraphExec_t)g), g = nullptr;
}

void NetGroup::set_lanes(int n) {
  if (n < 0) throw DcError(DC_EINVAL, "lanes must be 0 (automatic) or positive");
  if (n == lanes_opt_) return;
  lanes_opt_ = n;
  for (auto& kv : lane_choice_) lane_streams_release(kv.second);
  lane_choice_.clear();
  for (auto& gp : plans_) drop_plan(*gp);  // every merged plan was cut for the old lane count
  plans_.clear();
  cur_ = nullptr;
}

// The merged plan of the members' CURRENT shapes (every member has been through begin_batch: its plan is active, its
// buffers allocated, its filter images uploaded, its own tiles chosen).
GroupPlan& NetGroup::ensure_plan() {
  std::vector<std::vector<int>> shapes;
  for (Net* n : nets) shapes.push_back(n->plan_input_shape);
  GroupPlan* hit = nullptr;
  for (auto& gp : plans_)
    if (gp->shapes == shapes) hit = gp.get();
  if (hit) {
    if (plan_current(*hit)) {
      hit->last_use = ++use_clock_;
      ++stats.plan_hits;
      return *hit;
    }
    drop_plan(*hit);  // a member re-lowered (weights, options), reallocated a buffer or changed a tile: merge again (choices are cached)
    hit->launches.clear();
    hit->tuned = false;
    try {
      merge(*hit);
    } catch (...) {  // a half-merged plan must not be found again
      forget_plan(hit);
      throw;
    }
    hit->last_use = ++use_clock_;
    return *hit;
  }
  static const size_t cap = (size_t)std::max(1, env_int("DC_GROUP_PLAN_CACHE", 8));
  while (plans_.size() >= cap) {
    size_t lru = 0;
    for (size_t i = 1; i < plans_.size(); ++i)
      if (plans_[i]->last_use < plans_[lru]->last_use) lru = i;
    drop_plan(*plans_[lru]);
    forget_plan(plans_[lru].get());
  }
  plans_.emplace_back(new GroupPlan());
  GroupPlan& gp = *plans_.back();
  gp.shapes = shapes;
  try {
    merge(gp);
  } catch (...) {
    forget_plan(&gp);
    throw;
  }
  gp.last_use = ++use_clock_;
  return gp;
}

// does the merged plan still describe its members (their lowering, buffers, filter images, tiles)?
bool NetGroup::plan_current(const GroupPlan& gp) const {
  for (size_t c = 0; c < nets.size(); ++c)
    if (gp.shapes[c] != nets[c]->plan_input_shape || gp.lowerings[c] != (uint64_t)nets[c]->stats.lowerings || gp.buf_gens[c] != nets[c]->buf_gen_ ||
        gp.weight_gens[c] != nets[c]->seen_weights_gen || gp.tile_gens[c] != nets[c]->tile_gen_)
      return false;
  return true;
}

// the plan of the last forward, for the calls that launch from it outside a forward: a member that has been reshaped, re-lowered or
// re-tiled on its own since then has moved the buffers the prepared launches point at
GroupPlan& NetGroup::current_plan() {
  if (!cur_) throw DcError(DC_EINVAL, "group: run a forward first");
  if (!plan_current(*cur_)) throw DcError(DC_EINVAL, "group: a member changed (shape, weights, tiles) since the group's last forward: run a forward first");
  return *cur_;
}

void NetGroup::forget_plan(GroupPlan* gp) {
  if (cur_ == gp) cur_ = nullptr;
  for (size_t i = 0; i < plans_.size(); ++i)
    if (plans_[i].get() == gp) {
      plans_.erase(plans_.begin() + (long)i);
      return;
    }
}

// the kernel arguments of a merged launch for a tile (common block + problem table); returns the grid, <= 0 if the tile cannot take it
static long group_args(const GroupLaunch& gl, int variant, ConvMultiArgs& a) {
  a.p = gl.p;
  a.t = gl.table;
  if (variant == kStreamHalf) {
    if (!gl.ws_w) return -1;
    a.p.w = gl.ws_w;
  }
  return prepare_conv_multi(a.p, a.t, gl.nprob, variant);
}
// the tiles a merged launch can be timed on: every multi-problem tile of its K granularity and type, and the streaming form
static std::vector<int> group_candidates(const GroupLaunch& gl) {
  std::vector<int> c;
  for (int v = 0; v < conv_num_variants(); ++v)
    if (conv_variant_multiproblem(v) && gl.p.klen % conv_variant_bk(v) == 0 && conv_variant_esize(v) == gl.p.esize) c.push_back(v);
  if (gl.ws_w && env_int("DC_STREAM1X1", -1) != 0) c.push_back(kStreamHalf);
  return c;
}

void NetGroup::merge(GroupPlan& gp) {
  const size_t NM = nets.size();
  gp.lowerings.resize(NM), gp.buf_gens.resize(NM), gp.weight_gens.resize(NM), gp.tile_gens.resize(NM);
  for (size_t c = 0; c < NM; ++c) {
    gp.tile_gens[c] = nets[c]->tile_gen_;
    gp.lowerings[c] = (uint64_t)nets[c]->stats.lowerings;
    gp.buf_gens[c] = nets[c]->buf_gen_;
    gp.weight_gens[c] = nets[c]->seen_weights_gen;
  }
  ++stats.merges;
  gp.flops = 0;
  for (Net* n : nets) gp.flops += n->plan_flops;
  const size_t NL = nets[0]->plan.size();
  for (Net* n : nets)
    if (n->plan.size() != NL) throw DcError(DC_EINVAL, "group: the members' plans differ in length (different fusion options or graphs?)");
  const bool grouping = env_int("DC_GROUP", 1) != 0;  // 0: every launch member by member (A/B of the merge itself)
  // LANES.  The members are dealt to `nlanes` lanes — snake order over their sizes: largest with smallest — and every lane is
  // merged on its own and runs on a stream of its own, concurrently with the others: the launches of one lane fill the
  // dispatch ramps and the tails of the other's (a grouped 4-scale float16 pyramid batch: 12.06 ms as one lane, 10.61 ms as two
  // lanes of two scales — what two independent groups in flight reach, for ONE request), at the price of fetching a layer's
  // filters once per lane.  Default (measured on float16 batch-8 members, tools/group_profile.py --scales): TWO members run as two
  // lanes — nothing merged, plain concurrency: 8.26 against 9.27 ms merged (544x736 + 680x920), 5.44 against 6.10 (408x552 + 544x736) —,
  // THREE as one lane (one merged launch per layer: 11.15 against 12.06 ms for the lop-sided {A, C} | {B}), FOUR or more as two lanes
  // of merged members.  dc_group_set_lanes / DC_GROUP_LANES override.
  int nl = lanes_opt_ > 0 ? lanes_opt_ : env_int("DC_GROUP_LANES", 0);
  if (nl <= 0) nl = NM == 3 ? 1 : 2;
  nl = std::max(1, std::min<int>(nl, (int)NM));
  gp.nlanes = nl;
  gp.lane_members.assign(nl, {});
  {
    std::vector<size_t> order(NM);
    for (size_t c = 0; c < NM; ++c) order[c] = c;
    auto rows = [&](size_t c) {
      long r = 1;
      for (int d : nets[c]->plan_input_shape) r *= d;
      return r;
    };
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return rows(x) > rows(y); });
    for (size_t j = 0; j < NM; ++j) {
      const size_t r = j / nl, c = j % nl;
      gp.lane_members[r % 2 == 0 ? c : nl - 1 - c].push_back((int)order[j]);
    }
    for (auto& lm : gp.lane_members) std::sort(lm.begin(), lm.end());
  }
  for (int lane = 0; lane < nl; ++lane) {
  const std::vector<int>& mem = gp.lane_members[lane];
  const size_t NMl = mem.size();
  for (size_t i = 0; i < NL; ++i) {
    const Launch& l0 = nets[mem[0]]->plan[i];
    // (a lane with a single member runs that member's own launches: a one-problem multi-problem launch is the same work behind a
    //  longer prologue — measured 6 % slower at float16 batch 8)
    bool mergeable = grouping && l0.kind == Launch::CONV && NMl >= 2;
    for (size_t cc = 0; cc < NMl && mergeable; ++cc) {
      const size_t c = (size_t)mem[cc];
      const Launch& l = nets[c]->plan[i];
      const ConvGemmParams &g = l.cg, &g0 = l0.cg;
      // (a member on the float16 Winograd form merges as the direct layer it also is: the form is a one-round kernel that wins alone on
      //  a 240-workgroup grid — round 6: with the 544x736 member's conv4_x 3x3 launches kept out of the merge, the four scales of that
      //  layer ran as 17.8 + 31.2 + 17.0 + 14.3 us where the merged direct launch takes 60.5.  The float32 forms stay member by member.)
      //  (The float32 streaming forms, round 6 — ws1x1f and the stem on its skeleton —, merge the same way: they win by a few per cent
      //   on a member's own grid only.)
      const bool wino_apart = is_wino_variant(l.variant) && l.variant != kWinoHalf && l.variant != kStreamHalf && l.variant != kStreamFloat && l.variant != kStemFloat;
      if (l.kind != Launch::CONV || wino_apart || l.w != l0.w || l.scale != l0.scale || l.shift != l0.shift || l.c_off != l0.c_off ||
          l.w_off != l0.w_off ||
          (l.in2 >= 0) != (l0.in2 >= 0) || g.esize != g0.esize || g.klen != g0.klen || g.sy != g0.sy || g.sx != g0.sx || g.Cout != g0.Cout ||
          g.relu != g0.relu || g.sigmoid_ch != g0.sigmoid_ch)
        mergeable = false;
    }
    if (mergeable) {
      // a multi-problem tile must exist for this K granularity
      bool have = false;
      for (int v = 0; v < conv_num_variants(); ++v)
        if (conv_variant_multiproblem(v) && l0.cg.klen % conv_variant_bk(v) == 0 && conv_variant_esize(v) == l0.cg.esize) have = true;
      mergeable = have;
    }
    if (!mergeable) {
      for (size_t cc = 0; cc < NMl; ++cc) {
        const size_t c = (size_t)mem[cc];
        if (nets[c]->plan[i].kind != l0.kind) throw DcError(DC_EINVAL, "group: the members' plans differ at launch " + std::to_string(i));
        GroupLaunch gl;
        gl.lane = lane;
        gl.multi = false;
        gl.index = (int)i;
        gl.member = (int)c;
        gl.label = nets[c]->plan[i].label;
        gp.launches.push_back(std::move(gl));
      }
      continue;
    }
    // the problems: per member, its single problem or its deconvolution classes; heaviest K first, then the most pixels
    struct Rec {
      ConvProblem q;
      int member;
      std::string key;
    };
    std::vector<Rec> recs;
    std::string keys;
    for (size_t cc = 0; cc < NMl; ++cc) {
      const size_t c = (size_t)mem[cc];
      Net& n = *nets[c];
      const Launch& l = n.plan[i];
      const ConvGemmParams& g = l.cg;
      Storage& X = *n.storages[l.in];
      Storage& Y = *n.storages[l.out];
      const int nc = g.ncls > 1 ? g.ncls : 1;
      for (int k = 0; k < nc; ++k) {
        ConvProblem q{};
        const long yo = g.ncls > 1 ? l.y_off + g.cls[k].y_off : l.y_off;
        q.x = X.dev;
        q.y = Y.dev_at(yo);
        q.resid = l.in2 >= 0 ? n.storages[l.in2]->dev_at(yo) : nullptr;
        q.x_img_stride = g.x_img_stride, q.y_img_stride = g.y_img_stride;
        q.x_row_stride = g.x_row_stride, q.x_rows = g.x_rows, q.x_rowlen = g.x_rowlen;
        q.y_row_stride = g.y_row_stride, q.y_pix_stride = g.y_pix_stride;
        q.NB = g.NB;
        if (g.ncls > 1) {
          const ConvClass& cl = g.cls[k];
          q.w_off = cl.w_off;
          q.nty = cl.nty, q.ntx = cl.ntx, q.dy0 = cl.dy0, q.ddy = cl.ddy, q.x0 = cl.x0, q.ddx = cl.ddx, q.Ktot = cl.Ktot;
          q.OH = cl.OH, q.OW = cl.OW, q.M = cl.M;
        } else {
          q.w_off = l.w_off;
          q.nty = g.nty, q.ntx = g.ntx, q.dy0 = g.dy0, q.ddy = g.ddy, q.x0 = g.x0, q.ddx = g.ddx, q.Ktot = g.Ktot;
          q.OH = g.OH, q.OW = g.OW, q.M = g.M;
        }
        recs.push_back({q, (int)c, std::string()});
      }
      keys += (cc ? "|" : "") + n.tune_key(l);
    }
    // order of the problems = order in which every XCD walks them.  Default: tensor after tensor (the residue classes of ONE
    // member's deconvolution next to each other: they read the same 2048-deep input rows through different taps, which the
    // memory-side cache then still holds), biggest tensor first, inside a tensor the heaviest class first.
    // DC_GROUP_ORDER=1: heaviest K first across all members (class-major).
    if (env_int("DC_GROUP_ORDER", 0) == 1)
      std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.q.Ktot != b.q.Ktot ? a.q.Ktot > b.q.Ktot : a.q.M > b.q.M; });
    else {
      std::vector<long> msum(NM, 0);
      for (auto& r : recs) msum[r.member] += r.q.M;
      std::stable_sort(recs.begin(), recs.end(), [&](const Rec& a, const Rec& b) {
        if (a.member != b.member) return msum[a.member] != msum[b.member] ? msum[a.member] > msum[b.member] : a.member < b.member;
        return a.q.Ktot > b.q.Ktot;
      });
    }
    for (size_t r0 = 0, part = 0; r0 < recs.size(); r0 += kMaxProblems, ++part) {
      GroupLaunch gl;
      gl.lane = lane;
      gl.multi = true;
      gl.index = (int)i;
      gl.nprob = (int)std::min<size_t>(kMaxProblems, recs.size() - r0);
      gl.p = l0.cg;  // the layer's common fields: esize, klen, sy, sx, Cout, relu, sigmoid_ch
      gl.p.ncls = 0;
      gl.p.dbg = nullptr;
      gl.p.x = nullptr, gl.p.y = nullptr, gl.p.resid = nullptr;
      gl.p.w = l0.w->dev;
      {  // the streaming form of a dense float16 1x1 layer (stream1x1.hip) is a candidate if every member carries the (shared) image
        bool all = l0.takes_wino(kStreamHalf) && l0.wino_w->dev;
        for (int c : mem) all = all && nets[c]->plan[i].wino_w == l0.wino_w && nets[c]->plan[i].takes_wino(kStreamHalf);
        gl.ws_w = all ? l0.wino_w->dev : nullptr;
      }
      gl.p.scale = l0.scale ? l0.scale->dev + l0.c_off : nullptr;
      gl.p.shift = l0.shift ? l0.shift->dev + l0.c_off : nullptr;
      for (int k = 0; k < gl.nprob; ++k) {
        gl.table.prob[k] = recs[r0 + k].q;
        gl.prob_member.push_back(recs[r0 + k].member);
      }
      {
        double fl = 0;
        for (int c : mem) fl += nets[c]->plan[i].flops;
        gl.flops = fl * gl.nprob / (double)recs.size();
      }
      gl.key = "G" + std::to_string(gl.nprob) + (recs.size() > (size_t)kMaxProblems ? "p" + std::to_string(part) : "") + ":" + keys;
      gl.label = l0.label + " x" + std::to_string(NMl) + (gl.nprob != (int)NMl ? " [" + std::to_string(gl.nprob) + " problems]" : "");
      gp.launches.push_back(std::move(gl));
    }
  }
  }  // lane
  // tile of every merged launch: the shared choice table, else (until the group is timed) the widest member's own tile
  {
    std::lock_guard<std::mutex> lk(nets[0]->shared->mu);
    for (auto& gl : gp.launches) {
      if (!gl.multi) continue;
      auto it = nets[0]->shared->tune_cache.find(gl.key);
      int v = it != nets[0]->shared->tune_cache.end() ? it->second : -1;
      const int forced = env_int("DC_CONV_VARIANT", -1);
      if (forced >= 0 && forced < conv_num_variants() && conv_variant_multiproblem(forced) && gl.p.klen % conv_variant_bk(forced) == 0 &&
          conv_variant_esize(forced) == gl.p.esize)
        v = forced;
      if (forced < 0 && gl.ws_w && env_int("DC_STREAM1X1", -1) >= 1) v = kStreamHalf;  // (forced on: wherever eligible, as in Net's lowering)
      gl.variant = v;
    }
  }
  for (auto& gl : gp.launches) {
    if (!gl.multi) continue;
    int v = gl.variant;
    auto usable = [&](int cand) {
      if (cand != kStreamHalf && (cand < 0 || cand >= conv_num_variants() || is_wino_variant(cand))) return false;
      ConvMultiArgs a;
      return group_args(gl, cand, a) > 0;
    };
    if (!usable(v)) {
      v = -1;
      size_t big = (size_t)gp.lane_members[gl.lane][0];  // the lane's member with the most pixels
      for (int c : gp.lane_members[gl.lane])
        if (nets[c]->plan[gl.index].cg.M > nets[big]->plan[gl.index].cg.M) big = (size_t)c;
      if (usable(nets[big]->plan[gl.index].variant)) v = nets[big]->plan[gl.index].variant;
      for (int cand = 0; v < 0 && cand < conv_num_variants(); ++cand)
        if (usable(cand)) v = cand;
      if (v < 0) throw DcError(DC_EUNSUP, "group launch '" + gl.label + "': no multi-problem tile takes it");
    }
    apply_variant(gp, gl, v);
  }
}

// prepare the launch (common block + problem table = its kernel arguments) for a tile
void NetGroup::apply_variant(GroupPlan&, GroupLaunch& gl, int variant) {
  ConvMultiArgs a;
  const long grid = group_args(gl, variant, a);
  if (grid <= 0) throw DcError(DC_EUNSUP, "group launch '" + gl.label + "': tile " + conv_variant(variant).name + " cannot take it");
  gl.args = a;
  gl.variant = variant;
  gl.grid = grid;
}

// Tile of every merged launch by measurement, once per distinct signature (shared with every group of the model through
// the model's choice table, persisted with DC_TUNE_CACHE like the single-problem choices).
void NetGroup::autotune(GroupPlan& gp) {
  gp.tuned = true;
  if (env_int("DC_AUTOTUNE", 1) == 0 || env_int("DC_CONV_VARIANT", -1) >= 0) return;
  Net& n0 = *nets[0];
  std::lock_guard<std::mutex> lk(n0.shared->mu);
  std::map<std::string, int>& cache = n0.shared->tune_cache;
  bool timed_any = false;
  std::set<std::string> fresh;  // the signatures timed by THIS call: a choice that is in the table stays (other plans run it)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  void* s = stream();
  for (auto& gl : gp.launches) {
    if (!gl.multi || cache.count(gl.key)) continue;
    timed_any = true;
    fresh.insert(gl.key);
    std::vector<std::pair<float, int>> c;
    for (int v : group_candidates(gl)) {
      ConvMultiArgs p;
      const long grid = group_args(gl, v, p);
      if (grid <= 0) continue;
      KCHECK(launch_conv_multi(p, v, grid, s));  // warm
      float ms = 1e30f;
      for (int t2 = 0; t2 < 2; ++t2) {
        HIPCHECK(hipEventRecord(e0, (hipStream_t)s));
        for (int r = 0; r < 3; ++r) KCHECK(launch_conv_multi(p, v, grid, s));
        HIPCHECK(hipEventRecord(e1, (hipStream_t)s));
        HIPCHECK(hipEventSynchronize(e1));
        float m2 = 0;
        HIPCHECK(hipEventElapsedTime(&m2, e0, e1));
        ms = std::min(ms, m2);
      }
      c.push_back({ms, v});
    }
    std::sort(c.begin(), c.end());
    if (!c.empty()) {
      cache[gl.key] = c.front().second;
      for (auto& tm : c) tm.first *= 5.f / 3.f;  // the report prints "ms of a 5-launch burst"
      n0.shared->tune_timings[gl.key] = c;
    }
  }
  // (2) in situ, as Net::autotune does: the candidates within 15 % of a signature's best (at most 4) once more inside whole
  // passes over the GROUP plan (events around every launch of the signature, best of 3 passes per candidate).  Timed alone a
  // launch re-reads warm filters and meets an idle chip; in the sequence it follows another kernel's tail — on the two-pyramid
  // group the isolated pass took a 128x128 tile for the 256->1024+shortcut layers that is 15 % slower there than the 64x128 one.
  if (timed_any && env_int("DC_TUNE_INSITU", 1) != 0) {
    std::map<std::string, std::vector<int>> shortlist;
    size_t rounds = 0;
    for (auto& gl : gp.launches) {
      if (!gl.multi || shortlist.count(gl.key) || !fresh.count(gl.key)) continue;
      auto t = n0.shared->tune_timings.find(gl.key);
      if (t == n0.shared->tune_timings.end()) continue;
      std::vector<int> sl;
      for (auto& c : t->second)
        if (sl.size() < 4 && c.first <= t->second.front().first * 1.15f) sl.push_back(c.second);
      if (sl.size() >= 2) rounds = std::max(rounds, sl.size()), shortlist[gl.key] = sl;
    }
    if (rounds) {
      std::vector<size_t> idx;
      for (size_t i = 0; i < gp.launches.size(); ++i)
        if (gp.launches[i].multi && shortlist.count(gp.launches[i].key)) idx.push_back(i);
      std::vector<hipEvent_t> ev(2 * idx.size(), nullptr);
      struct EvList {
        std::vector<hipEvent_t>& ev;
        ~EvList() {
          for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        }
      } ev_list{ev};
      for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
      std::map<std::string, std::vector<float>> best;
      for (auto& kv : shortlist) best[kv.first].assign(kv.second.size(), 1e30f);
      for (size_t r = 0; r < rounds; ++r) {
        for (size_t i : idx) {
          const std::vector<int>& sl = shortlist[gp.launches[i].key];
          const int v = sl[std::min(r, sl.size() - 1)];
          if (gp.launches[i].variant != v) apply_variant(gp, gp.launches[i], v);
        }
        for (int pass = 0; pass < 3; ++pass) {
          size_t j = 0;
          for (size_t i = 0; i < gp.launches.size(); ++i) {
            const GroupLaunch& gl = gp.launches[i];
            const bool watched = j < idx.size() && idx[j] == i;
            if (watched) HIPCHECK(hipEventRecord(ev[2 * j], (hipStream_t)s));
            if (gl.multi) KCHECK(launch_conv_multi(gl.args, gl.variant, gl.grid, s));
            else nets[gl.member]->run_launch(nets[gl.member]->plan[gl.index], s);
            if (watched) {
              HIPCHECK(hipEventRecord(ev[2 * j + 1], (hipStream_t)s));
              ++j;
            }
          }
          HIPCHECK(hipStreamSynchronize((hipStream_t)s));
          std::map<std::string, float> sum;
          for (size_t q = 0; q < idx.size(); ++q) {
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ev[2 * q], ev[2 * q + 1]));
            sum[gp.launches[idx[q]].key] += ms;
          }
          for (auto& kv : sum) {
            const size_t e = std::min(r, shortlist[kv.first].size() - 1);
            best[kv.first][e] = std::min(best[kv.first][e], kv.second);
          }
        }
      }
      for (auto& kv : best) {
        size_t arg = 0;
        for (size_t e = 1; e < kv.second.size(); ++e)
          if (kv.second[e] < kv.second[arg]) arg = e;
        cache[kv.first] = shortlist[kv.first][arg];
      }
    }
  }
  for (auto& gl : gp.launches) {
    if (!gl.multi) continue;
    auto it = cache.find(gl.key);
    if (it != cache.end() && it->second != gl.variant) apply_variant(gp, gl, it->second);
  }
  if (timed_any) {
    ++stats.autotune_runs;
    write_tune_cache_locked(*n0.shared);
    // the other cached plans of this group follow the table as well (same signatures at other shape sets)
    for (auto& other : plans_) {
      if (other.get() == &gp) continue;
      bool touched = false;
      for (auto& gl : other->launches) {
        if (!gl.multi) continue;
        auto it = cache.find(gl.key);
        if (it != cache.end() && it->second != gl.variant) {
          apply_variant(*other, gl, it->second);
          touched = true;
        }
      }
      if (touched) other->drop_graphs();
    }
  }
  gp.drop_graphs();
}

void NetGroup::run(GroupPlan& gp, int lane, void* s) {
  for (auto& gl : gp.launches) {
    if (lane >= 0 && gl.lane != lane) continue;
    if (gl.multi) {
      const int rc = launch_conv_multi(gl.args, gl.variant, gl.grid, s);
      if (rc != 0) throw DcError(DC_EDEVICE, "group launch '" + gl.label + "' failed: " + hipGetErrorString((hipError_t)rc));
    } else {
      Net& n = *nets[gl.member];
      n.run_launch(n.plan[gl.index], s);
    }
  }
}

void NetGroup::enqueue(void* s) {
  GroupPlan& gp = ensure_plan();
  cur_ = &gp;
  if (!gp.tuned) {
    HIPCHECK(hipStreamSynchronize((hipStream_t)s));  // the inputs are in place; the timing launches run on the group's own stream
    autotune(gp);
  }
  bool use_graph = true;
  for (Net* n : nets) use_graph = use_graph && n->use_graph;
  const int nl = gp.nlanes;
  if (use_graph && (int)gp.lane_graphs.size() != nl) gp.lane_graphs.assign(nl, nullptr);
  // With more than one lane, lane 0 runs on the caller's stream and every other lane on a stream of the device's candidate list,
  // forked from and joined back into the caller's stream by events — WHICH candidate is measured (choose_lane_streams).
  if (nl > 1) {
    while ((int)lane_events_.size() < nl) {
      hipEvent_t ev;
      HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      lane_events_.push_back(ev);
    }
    if (!fork_event_) {
      hipEvent_t ev;
      HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      fork_event_ = ev;
    }
  }
  // (graphs first: the measurement below replays them)
  if (use_graph)
    for (int lane = 0; lane < nl; ++lane) {
      if (gp.lane_graphs[lane]) continue;
      gp.lane_graphs[lane] = capture_graph(stream(), [&](void* cs) { run(gp, lane, cs); });
      ++stats.graph_instantiations;
    }
  if (nl > 1) {
    auto it = lane_choice_.find(s);
    if (it == lane_choice_.end() || (int)it->second.size() != nl) {
      if (lane_choice_.size() >= 8) {  // a caller that keeps changing streams: start over rather than grow
        for (auto& kv : lane_choice_) lane_streams_release(kv.second);
        lane_choice_.clear();
      }
      choose_lane_streams(gp, s, use_graph);
      it = lane_choice_.find(s);
    }
    launch_lanes(gp, s, it->second, use_graph);
  } else {
    launch_lanes(gp, s, {}, use_graph);
  }
  for (Net* n : nets) {
    for (auto& l : n->plan) n->storages[l.out]->head = HEAD_AT_GPU;
    for (int v : n->plan_views_) n->storages[v]->head = HEAD_AT_GPU;
  }
}

// one grouped forward: lane 0 on s, lane k on side[k] (side[0] unused), fork / join by events
void NetGroup::launch_lanes(GroupPlan& gp, void* s, const std::vector<void*>& side, bool use_graph) {
  const int nl = gp.nlanes;
  if (nl > 1) HIPCHECK(hipEventRecord((hipEvent_t)fork_event_, (hipStream_t)s));
  for (int lane = 0; lane < nl; ++lane) {
    void* ls = lane == 0 ? s : side[lane];
    if (lane > 0) HIPCHECK(hipStreamWaitEvent((hipStream_t)ls, (hipEvent_t)fork_event_, 0));
    if (use_graph) HIPCHECK(hipGraphLaunch((hipGraphExec_t)gp.lane_graphs[lane], (hipStream_t)ls));
    else run(gp, lane, ls);
    if (lane > 0) {
      HIPCHECK(hipEventRecord((hipEvent_t)lane_events_[lane], (hipStream_t)ls));
      HIPCHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)lane_events_[lane], 0));
    }
  }
}

// Which streams do the lanes beyond the first run on?  Measured, per caller stream: the forward itself is timed (warm run + one
// timed run, events on s) with the side lanes on successive candidates, and the assignment with the shortest forward stays.  A
// side stream that shares a hardware queue with the caller's stream — or with another side lane's — runs its lane AFTER the other
// one (a grouped float16 pyramid batch: 12.0 instead of 10.7 ms); which candidate does depends on everything the process created
// before, so nothing but a measurement on the real launch sequence tells.  Costs a dozen forwards, once per (group, caller stream).
void NetGroup::choose_lane_streams(GroupPlan& gp, void* s, bool use_graph) {
  const int nl = gp.nlanes;
  const size_t ncand = 6;
  const std::vector<void*> cand = lane_stream_candidates(nets[0]->device, ncand + (size_t)nl);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  HIPCHECK(hipStreamSynchronize((hipStream_t)s));
  float best = 1e30f, best_free = 1e30f;
  std::vector<void*> best_set, best_free_set;
  for (size_t first = 0; first + (size_t)(nl - 1) <= cand.size(); ++first) {
    std::vector<void*> side(nl, nullptr);
    for (int k = 1; k < nl; ++k) side[k] = cand[first + (size_t)k - 1];
    launch_lanes(gp, s, side, use_graph);  // warm
    HIPCHECK(hipEventRecord(e0, (hipStream_t)s));
    launch_lanes(gp, s, side, use_graph);
    HIPCHECK(hipEventRecord(e1, (hipStream_t)s));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms, best_set = side;
    bool free_ = true;
    {
      std::lock_guard<std::mutex> lk(g_lane_mu);
      for (int k = 1; k < nl; ++k) free_ = free_ && g_lane_users[side[k]] == 0;
    }
    for (int k = 1; k < nl; ++k) free_ = free_ && pool_stream_users(side[k]) == 0;  // ... nor an executor's own (streams.cpp)
    if (free_ && ms < best_free) best_free = ms, best_free_set = side;
  }
  // a stream no other group's lanes run on, if one is (nearly) as good: two groups in flight whose side lanes share ONE stream
  // run those lanes one after the other (measured alone, both would pick the same winner)
  if (!best_free_set.empty() && best_free <= best * 1.04f) best_set = best_free_set;
  if (best_set.empty()) throw DcError(DC_EDEVICE, "group: no stream could be created for the lanes beyond the first (dc_group_set_lanes(g, 1) runs one lane)");
  {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    for (int k = 1; k < nl; ++k) ++g_lane_users[best_set[k]];
  }
  auto old = lane_choice_.find(s);
  if (old != lane_choice_.end()) lane_streams_release(old->second);
  lane_choice_[s] = best_set;
}

void NetGroup::forward_batch(const float* const* inputs, const int* n, const int* h, const int* w, bool is_device, float* const* prob,
                             float* const* loc, float* const* next, void* user_stream) {
  if (!inputs || !n || !h || !w) throw DcError(DC_EINVAL, "group forward: null argument");
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  std::vector<Storage*> ins;
  for (size_t c = 0; c < nets.size(); ++c) ins.push_back(&nets[c]->begin_batch(n[c], h[c], w[c]));
  void* s = user_stream ? user_stream : stream();
  for (size_t c = 0; c < nets.size(); ++c) {
    Storage& in = *ins[c];
    const int C = in.dim(1);
    if (is_device) {
      KCHECK(launch_nchw_to_nhwc(inputs[c], in.dev, in.esize, n[c], C, h[c], w[c], in.cp(), s));
    } else {
      in.ensure_stage(in.count());
      HIPCHECK(hipMemcpyAsync(in.stage, inputs[c], in.count() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)s));
      KCHECK(launch_nchw_to_nhwc(in.stage, in.dev, in.esize, n[c], C, h[c], w[c], in.cp(), s));
    }
    in.head = HEAD_AT_GPU;
  }
  enqueue(s);
  for (size_t c = 0; c < nets.size(); ++c)
    nets[c]->emit_maps(prob ? prob[c] : nullptr, loc ? loc[c] : nullptr, next ? next[c] : nullptr, is_device, s);
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

void NetGroup::forward_images(const unsigned char* const* bgr, const int* n, const int* h, const int* w, const double* scale, bool is_device,
                              float* const* prob, float* const* loc, float* const* next, double* const* pose, void* user_stream) {
  if (!bgr || !n || !h || !w || !scale) throw DcError(DC_EINVAL, "group forward_images: null argument");
  if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "forward_images() in CPU mode: libdeepcut_hip provides the MI355X path only");
  const bool own_async = user_stream == (void*)-1;
  if (own_async) user_stream = nullptr;
  nets[0]->ensure_device();
  void* s = user_stream ? user_stream : stream();
  for (size_t c = 0; c < nets.size(); ++c) nets[c]->prep_images(bgr[c], n[c], h[c], w[c], scale[c], is_device, s);
  enqueue(s);
  for (size_t c = 0; c < nets.size(); ++c) {
    nets[c]->emit_maps(prob ? prob[c] : nullptr, loc ? loc[c] : nullptr, next ? next[c] : nullptr, is_device, s);
    // decode on the group's stream; a host destination synchronises inside (the maps are complete there: same stream)
    if (pose && pose[c]) nets[c]->decode_pose(scale[c], pose[c], is_device, s);
  }
  if (!(is_device && (user_stream || own_async))) HIPCHECK(hipStreamSynchronize((hipStream_t)s));
}

int NetGroup::num_launches() { return cur_ ? (int)cur_->launches.size() : 0; }
int NetGroup::num_multi_launches() {
  int m = 0;
  if (cur_)
    for (auto& gl : cur_->launches) m += gl.multi ? 1 : 0;
  return m;
}
double NetGroup::flops() { return cur_ ? cur_->flops : 0.0; }

// same line formats as Net::plan_text / Net::profile_text (tools/breakdown.py aggregates both)
std::string NetGroup::plan_text() {
  if (!cur_) throw DcError(DC_EINVAL, "group: run a forward first");
  std::ostringstream os;
  os << "# group of " << nets.size() << " executors in " << cur_->nlanes << " lane" << (cur_->nlanes > 1 ? "s" : "") << ": " << cur_->launches.size() << " launches (" << num_multi_launches() << " multi-problem), "
     << cur_->flops / 1e9 << " GFLOP algorithmic" << (nets[0]->dtype == 1 ? ", dtype=f16" : ", dtype=f32") << "\n";
  for (size_t i = 0; i < cur_->launches.size(); ++i) {
    const GroupLaunch& gl = cur_->launches[i];
    os << i << "\t";
    if (gl.multi) {
      long M = 0;
      int kmax = 0;
      for (int k = 0; k < gl.nprob; ++k) M += gl.table.prob[k].M, kmax = std::max(kmax, gl.table.prob[k].Ktot);
      os << "conv_gemm_mp<" << conv_variant(gl.variant).name << ">\tM=" << M << " N=" << gl.p.Cout << " K=" << kmax << " problems=" << gl.nprob
         << " grid=" << gl.grid << (cur_->nlanes > 1 ? " lane=" + std::to_string(gl.lane) : std::string()) << (gl.table.prob[0].resid ? " +resid" : "")
         << (gl.p.relu ? " +relu" : "") << (gl.p.sigmoid_ch ? " +sigmoid" : "");
    } else {
      os << nets[gl.member]->plan[gl.index].kernel << "\tmember " << gl.member;
    }
    os << "\t" << gl.label << "\n";
  }
  return os.str();
}

std::string NetGroup::tune_report_text() {
  if (!cur_) throw DcError(DC_EINVAL, "group: run a forward first");
  std::lock_guard<std::mutex> lk(nets[0]->shared->mu);
  std::vector<std::string> order;
  std::map<std::string, std::pair<int, int>> seen;
  for (auto& gl : cur_->launches) {
    if (!gl.multi) continue;
    auto it = seen.find(gl.key);
    if (it == seen.end()) order.push_back(gl.key), seen[gl.key] = {gl.variant, 1};
    else ++it->second.second;
  }
  std::string out;
  for (auto& k : order) {
    out += k + "\t" + conv_variant(seen[k].first).name + "\t" + std::to_string(seen[k].second) + "\t";
    auto t = nets[0]->shared->tune_timings.find(k);
    if (t != nets[0]->shared->tune_timings.end())
      for (size_t i = 0; i < t->second.size(); ++i) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "%s%s:%.2f", i ? " " : "", conv_variant(t->second[i].second).name, t->second[i].first * 1000.f / 5.f);
        out += buf;
      }
    out += "\n";
  }
  return out;
}

void NetGroup::set_tile(const std::string& key, const std::string& tile) {
  current_plan();
  int v = -1;
  for (int i = 0; v < 0 && i < conv_num_variants(); ++i)
    if (tile == conv_variant(i).name) v = i;
  if (tile == conv_variant(kStreamHalf).name) v = kStreamHalf;
  if (v < 0) throw DcError(DC_EINVAL, "no tile variant named '" + tile + "'");
  bool any = false;
  for (auto& gl : cur_->launches) {
    if (!gl.multi || gl.key != key) continue;
    ConvMultiArgs a;
    if (!conv_variant_multiproblem(v) || group_args(gl, v, a) <= 0)
      throw DcError(DC_EUNSUP, "tile '" + tile + "' cannot take group launch '" + gl.label + "'");
    any = true;
  }
  if (!any) throw DcError(DC_EINVAL, "the group's current plan has no launch with signature '" + key + "'");
  // nothing of this plan may be in flight while its launches change (the forward may have run on a caller's stream and on lane
  // streams: the device-wide wait covers them all)
  {
    RuntimeLock rl;
    (void)hipDeviceSynchronize();
  }
  // EVERY cached plan that carries the signature (other shape sets of this group share it through the model's table and the
  // DC_TUNE_CACHE file: re-tiling only the current plan left them on the old tile — round-4 advice)
  for (auto& gp : plans_) {
    bool touched = false;
    for (auto& gl : gp->launches)
      if (gl.multi && gl.key == key && gl.variant != v) {
        apply_variant(*gp, gl, v);
        touched = true;
      }
    if (touched) gp->drop_graphs();
  }
  {
    std::lock_guard<std::mutex> lk(nets[0]->shared->mu);
    auto it = nets[0]->shared->tune_cache.find(key);
    if (it == nets[0]->shared->tune_cache.end() || it->second != v) {
      nets[0]->shared->tune_cache[key] = v;
      write_tune_cache_locked(*nets[0]->shared);
    }
  }
  cur_->drop_graphs();
}

std::string NetGroup::profile_text(int iters) {
  GroupPlan& gp = current_plan();
  void* s = stream();
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } guard{e0, e1};
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  std::ostringstream os;
  os << "idx\tkernel\tus\tGFLOP\tTFLOP/s\tgrid\tlabel\n";
  double total_us = 0;
  auto one = [&](const GroupLaunch& gl) {
    if (gl.multi) KCHECK(launch_conv_multi(gl.args, gl.variant, gl.grid, s));
    else nets[gl.member]->run_launch(nets[gl.member]->plan[gl.index], s);
  };
  for (size_t i = 0; i < gp.launches.size(); ++i) {
    const GroupLaunch& gl = gp.launches[i];
    one(gl);
    HIPCHECK(hipEventRecord(e0, (hipStream_t)s));
    for (int k = 0; k < iters; ++k) one(gl);
    HIPCHECK(hipEventRecord(e1, (hipStream_t)s));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    total_us += us;
    const double fl = gl.multi ? gl.flops : nets[gl.member]->plan[gl.index].flops;
    const std::string kn = gl.multi ? std::string("conv_gemm_mp<") + conv_variant(gl.variant).name + ">" : nets[gl.member]->plan[gl.index].kernel;
    char buf[640];
    std::snprintf(buf, sizeof buf, "%zu\t%s\t%.2f\t%.3f\t%.2f\t%ld\t%s\n", i, kn.c_str(), us, fl / 1e9, us > 0 ? fl / us / 1e6 : 0.0,
                  gl.multi ? gl.grid : nets[gl.member]->plan[gl.index].grid, gl.label.c_str());
    os << buf;
  }
  os << "# sum of per-launch times: " << total_us << " us\n";
  return os.str();
}

}  // namespace dc
