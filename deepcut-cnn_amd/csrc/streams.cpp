// streams.cpp — which HIP streams a set of executors runs on, chosen by measurement (round 5).
//
// A HIP process has four hardware queues; the runtime binds every stream to one of them when it is created, several streams share
// a queue, and work on streams that share a queue runs one after the other.  Nothing in the API tells which queue a stream got,
// it depends on every stream the process created before, and a do-nothing-kernel probe of stream pairs does not predict what real
// forwards do (EXPERIMENTS.md E).  Measured with four batch-1 executors on the 70 four-subsets of eight streams: 12 subsets at
// 478-492 images/s, 53 at 419-434, 5 at 376-390 (profiles/r04_stream_subsets.txt).  Round 4 let bench.py search the subsets; a
// caller of the library got luck.  Here the choice is the library's: a process-wide pool of candidate streams per device, the
// executors' REAL forwards timed on candidate assignments (greedy: executor k takes the free candidate on which executors 0..k
// together run fastest; ~20 bursts of a few forwards instead of 70 subsets), and every executor adopts its stream as its own
// (DC_STREAM_OWN, dc_net_stream).  The reference has no counterpart: its Net runs on the one legacy stream of its thread
// (src/caffe/common.cpp:99-158).
#include <chrono>

#include "net_internal.h"

namespace dc {

static std::mutex g_pool_mu;
static std::map<int, std::vector<void*>>& pool_of_devices() {
  static std::map<int, std::vector<void*>>* p = new std::map<int, std::vector<void*>>();  // never destroyed: streams live to exit
  return *p;
}
static std::map<void*, int>& pool_users() {
  static std::map<void*, int>* p = new std::map<void*, int>();
  return *p;
}

// a COPY of the first `want` candidates of the device's pool (created on demand, never destroyed): the pool may grow under
// another thread's feet, a reference into it would dangle (round-4 advice)
std::vector<void*> executor_stream_pool(int device, size_t want) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  std::vector<void*>& p = pool_of_devices()[device];
  while (p.size() < want) {
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    p.push_back(st);
  }
  return std::vector<void*>(p.begin(), p.begin() + (long)std::min(want, p.size()));
}
bool is_pool_stream(void* s) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& kv : pool_of_devices())
    if (std::find(kv.second.begin(), kv.second.end(), s) != kv.second.end()) return true;
  return false;
}
void pool_stream_acquire(void* s) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  ++pool_users()[s];
}
void pool_stream_release(void* s) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = pool_users().find(s);
  if (it != pool_users().end() && it->second > 0) --it->second;
}
int pool_stream_users(void* s) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = pool_users().find(s);
  return it == pool_users().end() ? 0 : it->second;
}

void* Net::own_stream() {
  ensure_device();
  return stream;
}

void Net::adopt_stream(void* s) {
  if (s == stream) return;
  if (stream) {
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    if (stream_borrowed_) pool_stream_release(stream);
    else (void)hipStreamDestroy((hipStream_t)stream);
  }
  stream = s;
  stream_borrowed_ = true;
  pool_stream_acquire(s);
}

// see the header of this file.  Every executor must have an active plan with its buffers (a forward or dc_net_reserve at the
// shape it will serve); inputs are whatever the input blobs hold (timing only).  rates: forwards per second of `reps` forwards per
// executor with the chosen streams / with the first n candidates (what a process that creates its streams first would get).
void Net::choose_streams(const std::vector<Net*>& nets, int ncand, int reps, double* rate_chosen, double* rate_first) {
  const int n = (int)nets.size();
  if (n < 1) throw DcError(DC_EINVAL, "choose_streams: no executors");
  for (Net* e : nets) {
    if (!e) throw DcError(DC_EINVAL, "choose_streams: null net");
    if (Context::get().mode != DC_MODE_GPU) throw DcError(DC_ENOCPU, "choose_streams() in CPU mode: libdeepcut_hip provides the MI355X path only");
    if (!e->plan_valid) throw DcError(DC_EINVAL, "choose_streams: every executor needs a lowered shape first (a forward, or dc_net_reserve)");
    // plan_valid alone proves nothing about buffers (plan_text() / dc_net_flops lower without allocating; a blob may have been
    // reshaped since the last forward): the timed forwards below must find every buffer of the current shape in place
    e->prepare_to_run();
    if (e->device != nets[0]->device) throw DcError(DC_EINVAL, "choose_streams: the executors must sit on one device");
  }
  if (ncand <= 0) ncand = 8;
  ncand = std::max(ncand, n);
  reps = std::max(1, reps);
  HIPCHECK(hipSetDevice(nets[0]->device));
  std::vector<void*> cand = executor_stream_pool(nets[0]->device, (size_t)ncand);
  if ((int)cand.size() < n) throw DcError(DC_EDEVICE, "choose_streams: could not create enough streams");
  for (Net* e : nets) e->synchronize();
  auto burst = [&](const std::vector<void*>& assign) {
    const int m = (int)assign.size();
    auto run = [&]() {
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < m; ++k) nets[k]->enqueue_plan(assign[k]);
      for (int k = 0; k < m; ++k) HIPCHECK(hipStreamSynchronize((hipStream_t)assign[k]));
    };
    run();  // warm (graphs captured, L2s in their steady state)
    double best = 1e30;
    for (int t = 0; t < 2; ++t) {
      const auto t0 = std::chrono::steady_clock::now();
      run();
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
  };
  // streams that executors OUTSIDE this set already run on are taken only if nothing else is left
  std::vector<void*> own_now;
  for (Net* e : nets)
    if (e->stream_borrowed_) own_now.push_back(e->stream);
  auto foreign_users = [&](void* s) {
    int u = pool_stream_users(s);
    for (void* o : own_now)
      if (o == s) --u;
    return std::max(0, u);
  };
  std::vector<void*> chosen;
  std::vector<char> used(cand.size(), 0);
  for (int k = 0; k < n; ++k) {
    double best = 1e30;
    int best_i = -1;
    for (int pass = 0; pass < 2 && best_i < 0; ++pass)  // pass 0: streams nobody else runs on; pass 1: any
      for (size_t i = 0; i < cand.size(); ++i) {
        if (used[i] || (pass == 0 && foreign_users(cand[i]) > 0)) continue;
        if (k == 0) {  // a lone executor runs alike on every stream: the first free candidate
          best_i = (int)i;
          break;
        }
        std::vector<void*> a = chosen;
        a.push_back(cand[i]);
        const double t = burst(a);
        if (t < best) best = t, best_i = (int)i;
      }
    if (best_i < 0) throw DcError(DC_EDEVICE, "choose_streams: no candidate stream left");
    used[(size_t)best_i] = 1;
    chosen.push_back(cand[(size_t)best_i]);
  }
  if (rate_chosen) *rate_chosen = (double)reps * n / burst(chosen);
  if (rate_first) *rate_first = (double)reps * n / burst(std::vector<void*>(cand.begin(), cand.begin() + n));
  for (int k = 0; k < n; ++k) nets[k]->adopt_stream(chosen[(size_t)k]);
}

}  // namespace dc
