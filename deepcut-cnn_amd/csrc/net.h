// net.h — graph runtime of the DeeperCut forward path: the MI355X-native counterpart of
// caffe::Net / caffe::Layer / caffe::Blob / caffe::SyncedMemory / caffe::Caffe for the TEST-phase
// forward of models/deepercut/*.prototxt (reference: src/caffe/net.cpp, include/caffe/layer.hpp,
// src/caffe/blob.cpp, src/caffe/syncedmem.cpp, src/caffe/common.cpp).
//
// It is not a layer-by-layer interpreter.  Net::Init semantics (InsertSplits, in-place tops, output
// discovery, name-matched weight loading) are reproduced so that the pycaffe-visible surface is the
// reference's, but execution goes through a *lowered plan*: BatchNorm+Scale(+bias)+ReLU are folded
// into the producing convolution's epilogue, residual Eltwise adds and the Deconvolution+Crop+Eltwise
// heads are fused, activations live channels-last (NHWC) in HBM, and every convolution /
// deconvolution is one launch (4 for a stride-2 deconvolution: one per output parity class) of the
// gather-GEMM MFMA kernel in kernels.hip.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "formats.h"
#include "kernels.h"

namespace dc {

// ---- Caffe context (common.cpp:13-20: thread-local mode + device) ---------------------------------
struct Context {
  int mode = 0;     // DC_MODE_CPU (reference default, common.cpp:55,100)
  int device = 0;
  static Context& get();
};
int device_count();  // 0 without a GPU; never throws

// ---- SyncedMemory + Blob ----------------------------------------------------------------------------
enum Head { UNINITIALIZED = 0, HEAD_AT_CPU = 1, HEAD_AT_GPU = 2, SYNCED = 3 };

struct Net;
struct ModelShared;
struct Storage {
  std::vector<int> shape;  // logical Caffe shape (N,C,H,W for activations)
  float* host = nullptr;
  size_t host_cap = 0;
  bool host_pinned = false;
  unsigned char* dev = nullptr;  // NHWC image (float or _Float16 elements, `esize` bytes each), channel pitch cp()
  size_t dev_cap = 0;            // capacity in bytes
  int esize = 4;                 // bytes per device element (2 in fp16 nets)
  float* stage = nullptr;  // device NCHW staging for up/download
  size_t stage_cap = 0;
  int head = UNINITIALIZED;
  bool pad4 = false;   // channel pitch rounded up to one 16-byte vector (tensors read by the gather-GEMM)
  bool is_param = false;
  bool elided = false;  // absorbed by fusion in the current plan: never materialised
  // net outputs the caller reads through host pointers after forward() (pycaffe: blobs['prob'].data): once an output was
  // downloaded on demand, the following forwards send it to the (pinned) host copy right behind the head launch, inside the
  // forward's own synchronisation — `.data` then finds it there (SYNCED) instead of paying a kernel, a copy and a stream
  // round trip per map.  An output that was NOT touched between two forwards stops being sent (next_pred is 9.1 of the
  // 10.2 MB and the demo never reads it, estimate_pose.py:104-112).
  bool host_wanted = false, host_touched = false;
  int view_of = -1;     // >= 0: this blob is channels [view_c0, view_c0+C) of storage `view_of` (merged heads)
  int view_c0 = 0, view_cp = 0;
  Net* owner = nullptr;  // activations: the net whose stream moves this blob between host and device (null: stand-alone blob)
  std::shared_ptr<ModelShared> shared;  // params: the model state a net and its clones own jointly (outlives any one of them)
  uint64_t packed_hash = 0;  // params: content hash when the filter images were last packed (see ModelShared::touched)
  bool touch_listed = false; // params: already on ModelShared::touched (an access loop must not grow the list)
  int id = -1;

  ~Storage();
  size_t count() const;
  int dim(int i) const { return i < (int)shape.size() ? shape[i] : 1; }
  int cp() const;                 // channel pitch of the device image
  size_t dev_count() const;       // elements of the device image
  void reshape(const std::vector<int>& s);  // Blob::Reshape: capacity only grows (blob.cpp:23-43)
  float* host_ptr();              // allocates + zero-fills on first touch (syncedmem.cpp:25-31)
  void ensure_dev(size_t n);  // n elements
  void* dev_at(long elem) const { return dev + elem * esize; }
  void ensure_stage(size_t n);
};

struct NetBlob {
  std::string name;
  std::shared_ptr<Storage> st;
  bool standalone = false;  // dc_blob_create: owned by the caller, moved on the default stream of the thread's device
};

// SyncedMemory::to_gpu / to_cpu (syncedmem.cpp:25-101) for a blob that may or may not belong to a net: `stream` is the
// owner's stream (null: the default stream), `base` the concatenated tensor a channel view lives in (else null)
void storage_to_device(Storage& s, void* stream);
void storage_to_host(Storage& s, void* stream, Storage* base);
void storage_download_enqueue(Storage& s, void* stream, Storage* base);  // the device -> host part of it, no wait, head untouched
void storage_mutable_device(Storage& s, void* stream);  // SyncedMemory::mutable_gpu_data: device image authoritative
// Blob::CopyFrom (blob.cpp:435-474): wherever the source is authoritative; device copies re-pitch through an NCHW stage
void storage_copy(Storage& dst, Storage& src, Storage* src_base, void* stream);

struct ConvSpec {
  int num_output = 0, kh = 1, kw = 1, sh = 1, sw = 1, ph = 0, pw = 0, dh = 1, dw = 1, group = 1;
  bool bias = true;
};

struct LayerRec {
  std::string name, type;
  std::vector<int> bottoms, tops;  // indices into Net::blobs
  TextMsg def;                     // the layer { } message
  std::vector<std::shared_ptr<NetBlob>> params;
  bool is_split = false;
  // parsed parameters
  ConvSpec conv;
  int pool_k = 1, pool_s = 1, pool_p = 0;
  float bn_eps = 1e-5f;
  bool scale_bias = false;
  float relu_slope = 0.f;
  int crop_oh = 0, crop_ow = 0;
};

// ---- lowered plan -----------------------------------------------------------------------------------
struct DevVec {  // a packed filter / affine vector; shared between a Net and its clones
  std::vector<float> host;
  bool as_half = false;  // filter image of an fp16 net: converted to _Float16 on upload
  // fp16 filter images: output channel c of the image was multiplied by an exact power of two 2^k(c) that brings its largest
  // weight into [2^13, 2^14) — filters of 1e-5 (a head on a trunk with large activations) would otherwise sit in float16's
  // subnormal range with a handful of significant bits —; row_scale[c] = 2^-k(c) goes into the launch's fp32 epilogue scale
  std::vector<float> row_scale;
  float* dev = nullptr;
  size_t uploaded = 0;
  DevVec() = default;
  DevVec(const DevVec&) = delete;
  DevVec& operator=(const DevVec&) = delete;
  ~DevVec();
};

// What a Net and its clones own JOINTLY (shared_ptr): the packed filter / affine images in HBM, the tile choices measured
// on the device, and the generation counter of the parameters.  Parameter blobs point here (not at a Net), so a blob
// obtained through one executor stays safe to touch after that executor is gone, and a parameter write reaches every
// executor: each compares `weights_gen` with the generation its plans were lowered from before it runs.
struct ModelShared {
  std::mutex mu;
  std::map<std::string, std::shared_ptr<DevVec>> vec_by_key;  // packed-weight cache (plans hold the images they use alive)
  uint64_t weights_gen = 1;   // bumped when parameter CONTENT changed (copy_from, a write through params that changed bytes)
  uint64_t packed_gen = 0;    // generation vec_by_key was packed from
  std::vector<std::weak_ptr<Storage>> touched;  // params handed out writable since the last check (Blob.data is always mutable
                                                // in pycaffe, _caffe.cpp:273: a read must not cost a 263 MB re-pack)
  std::map<std::string, int> tune_cache;        // GEMM signature -> fastest variant, one timing per process and model
  bool tune_file_loaded = false;
  std::map<std::string, std::vector<std::pair<float, int>>> tune_timings;  // signature -> (ms of a 5-launch burst, variant), sorted: pass 1 of autotune
};

struct ResampleTable {  // Pillow-style 8-bit bilinear resample of one axis: taps and 22-bit weights, on the device
  int ksize = 0;
  std::vector<int> bounds;  // host copy of [out][2] (first tap, tap count)
  int* dev_bounds = nullptr;
  int* dev_coeffs = nullptr;
  ResampleTable() = default;
  ResampleTable(const ResampleTable&) = delete;
  ResampleTable& operator=(const ResampleTable&) = delete;
  ~ResampleTable();
};
// host side of Pillow's precompute_coeffs + normalize_coeffs_8bpc (bilinear, whole-image box)
void resample_coeffs(int in_size, int out_size, int& ksize, std::vector<int>& bounds, std::vector<int>& coeffs);
// estimate_pose.py:85-88,96: canvas (stride-8) and resized-image sizes for an h x w image at `scale`
void image_canvas_size(int h, int w, double scale, int& out_h, int& out_w, int& new_h, int& new_w);

struct Launch {
  enum Kind { CONV, POOL, ELT, CROP } kind = CONV;
  std::string label;    // e.g. "res4b3_branch2b+bn+scale+relu"
  std::string kernel;   // variant / kernel name
  int first_layer = 0, last_layer = 0;
  int in = -1, in2 = -1, out = -1;  // storage ids (in2: residual / second operand)
  // CONV
  ConvGemmParams cg{};  // pointers filled at launch time
  int variant = 0;
  std::shared_ptr<DevVec> w, scale, shift;  // packed filters / folded affine (kept alive by the plan)
  std::shared_ptr<DevVec> wino_w;           // Winograd-transformed filters (eligible 3x3 layers) / the fragment-order image of the streaming
                                            // form (eligible float16 1x1 layers, stream1x1.hip), else null
  std::shared_ptr<DevVec> wino_scale;       // float16 nets: the epilogue scale of the Winograd form (folded affine x the image's row scale x 4)
  // a Winograd form this launch can run as: the image exists and the form serves the net's element type
  bool takes_wino(int v) const {
    return is_wino_variant(v) && (bool)wino_w && wino_variant_esize(v) == cg.esize && cg.ncls <= 1 && (v == kStreamHalf || v == kStreamFloat) == (cg.nty == 1 && cg.ntx == 1) &&
           (v == kStemHalf || v == kStemFloat) == (cg.nty == 7 && cg.ntx == 1);
  }
  long y_off = 0;                      // element offset of this launch's first output (deconvolution classes, channel splits)
  long w_off = 0;                      // element offset of this launch's first filter row inside `w` (channel splits)
  int c_off = 0;                       // first output channel of this launch inside scale / shift (channel splits)
  double flops = 0;                    // algorithmic 2*MAC
  long grid = 0;
  // POOL
  int pk = 0, ps = 0, pp = 0;
  // ELT
  int relu = 0, sigmoid = 0;
  // CROP
  int oh = 0, ow = 0;
};

// Everything that depends on the INPUT SHAPE: the lowered launches, which storages are views / elided, the captured
// hipGraph.  A Net keeps one per shape it has met (LRU): Layer::Forward re-derives shapes on every call
// (layer.hpp:451-456) and the demo's scale loop changes the shape on every iteration (estimate_pose.py:81-128), so a
// 4-scale pyramid must not re-lower 734 layers and re-instantiate a 160-node graph four times per image.
struct PlanState {
  std::vector<int> input_shape;
  std::vector<Launch> plan;
  double flops = 0;
  std::vector<int> views;  // storages that are channel views in this plan
  struct StorageState {
    int id, view_of, view_c0;
    bool elided;
  };
  std::vector<StorageState> sstate;                       // per-storage fusion state to restore on activation
  std::vector<std::pair<int, std::vector<int>>> aux_shapes;  // tensors created by the lowering (merged heads)
  void* graph_exec = nullptr;
  uint64_t graph_buf_gen = 0;  // Net::buf_gen_ when the graph was captured (a reallocated buffer makes it stale)
  bool tuned = false;
  uint64_t last_use = 0;
};

struct NetStats {  // dc_net_stats
  long long lowerings = 0, graph_instantiations = 0, plan_hits = 0, autotune_runs = 0, buffer_growths = 0, repacks = 0;
};

struct Net {
  std::string name;
  int phase = 1;
  std::vector<LayerRec> layers;  // after InsertSplits
  std::vector<std::shared_ptr<NetBlob>> blobs;
  std::map<std::string, int> blob_index;
  std::vector<std::shared_ptr<Storage>> storages;
  std::vector<int> inputs, outputs;  // blob indices

  int fuse = 2;
  int dtype = 0;  // 0: float activations/filters; 1: _Float16 activations/filters, fp32 accumulate + epilogue
  int use_graph = 0;
  int outputs_mask = -1;  // DC_OPT_OUTPUTS: bit i = output i (order of `outputs`) is wanted
  std::shared_ptr<ModelShared> shared;   // joint with every clone
  uint64_t seen_weights_gen = 0;         // generation the cached plans were lowered from
  // the ACTIVE plan (the fields below are swapped with a parked PlanState when the input shape changes)
  bool plan_valid = false;
  std::vector<int> plan_input_shape;
  std::vector<Launch> plan;
  std::vector<int> plan_views_;          // storages that are channel views in the current plan
  double plan_flops = 0;
  void* graph_exec = nullptr;
  uint64_t graph_buf_gen = 0;
  bool tuned = false;                    // tile variants of the current plan were timed on the device
  std::vector<std::unique_ptr<PlanState>> parked_;  // plans of the other shapes met so far (LRU, DC_PLAN_CACHE entries)
  uint64_t use_clock_ = 0, cur_last_use_ = 0;
  uint64_t buf_gen_ = 1;                 // bumped whenever a device buffer of this net is reallocated
  uint64_t tile_gen_ = 1;                // bumped whenever a launch of the active plan changes its tile (autotune, set_tile): a
                                         // NetGroup that captured this member's launches in ITS graph re-merges
  NetStats stats;
  std::map<std::string, int> aux_index_; // concatenated-head tensors created by the lowering
  void* stream = nullptr;
  bool stream_borrowed_ = false;  // `stream` belongs to the process-wide executor pool (streams.cpp): never destroyed with the net
  int device = -1;
  double* pose_dev = nullptr;
  size_t pose_cap = 0;
  std::string text_buf;
  std::string proto_text;  // kept for clone()

  ~Net();
  static Net* create(const std::string& prototxt_text, int phase, const Net* clone_of = nullptr);
  Net* clone();           // same graph and input shape, SHARED parameters and packed device weights
  void synchronize();     // wait for everything enqueued on the net's own stream
  // streams.cpp: the executors' own streams chosen by timing their forwards on candidate assignments (hardware-queue sharing decides
  // what "in flight" is worth and the API does not say which queue a stream got)
  static void choose_streams(const std::vector<Net*>& nets, int ncand, int reps, double* rate_chosen, double* rate_first);
  void adopt_stream(void* s);  // a pool stream becomes the net's own
  void* own_stream();          // the net's own stream, created on first use
  void set_dtype(int d);  // 0 float32 / 1 float16 device images (DC_OPT_DTYPE)
  void set_outputs_mask(int mask);  // DC_OPT_OUTPUTS
  void copy_from(const std::string& path);
  void save(const std::string& path);
  void reshape();         // propagate input shapes through every layer (Net::Reshape)
  void build_plan();      // lower to launches for the current shapes (host only; no device needed)
  void ensure_plan();     // make the plan of the current input shape the active one: cache hit, or build_plan()
  void invalidate_plans();  // drop every cached plan and graph (option / weight change)
  void mark_weights_changed();  // parameter content changed: every executor re-packs and re-lowers before its next run
  void reserve(int n, int h, int w);  // lower + allocate + tune for a shape without running it
  void prepare_to_run();              // the same for the current input shape (no reshape)
  // one reference layer stand-alone (Layer<Dtype>::SetUp on given bottoms): a net whose inputs are the layer's bottoms
  static Net* create_for_layer(const std::string& layer_text, int phase, const std::vector<std::vector<int>>& bottom_shapes);
  void forward(int start, int end);
  // host_async (host buffers only): nothing is waited for — the upload, the forward and the downloads are enqueued on the net's own
  // stream and the caller collects with synchronize(); truly asynchronous for pinned buffers (dc_host_alloc), staged by the runtime otherwise
  void forward_batch(const float* input, int n, int h, int w, bool is_device, float* prob, float* loc,
                     float* next, void* user_stream, bool host_async = false);
  // the same from n separate HOST images (one pointer each; pinned memory: the DMA engines read them in place), asynchronous on the
  // net's own stream — what dc_forward_batch uses when the caller's arrays are pinned: no staging copy on the host
  void forward_host_images(const float* const* inputs, int n, int h, int w);
  // cross-request batching: n independent single-image requests (device buffers, one pointer set per request) as ONE batch-n forward
  void forward_requests(int n, const float* const* inputs, int h, int w, float* const* prob, float* const* loc, float* const* next,
                        void* user_stream);
  // image entry: pre-processing (estimate_pose.py:83-103) + forward + optional decode, all on the device
  void forward_images(const unsigned char* bgr, int n, int h, int w, double scale, bool is_device, float* prob, float* loc,
                      float* next, double* pose, void* user_stream);
  void sync_to_host(Storage& s);       // SyncedMemory::to_cpu
  void sync_to_device(Storage& s);     // SyncedMemory::to_gpu
  void decode_pose(double scale, double* out, bool is_device, void* user_stream);  // after a forward
  // the maps of the last forward as NCHW float32 (elem 0) or float16 (elem 1, fp16 nets), host or device destination
  void emit_last_maps(void* prob, void* loc, void* next, int elem, bool is_device, void* user_stream);
  // multi-person consumers of the maps of the last forward (encoding: pose_data_layer.cpp:686-802)
  void detect_parts(double scale, float thr, int radius, int max_det, int* counts, double* dets);
  void decode_pairwise(double scale, int ndet, const int* det, const double* mean, const double* stdev, double* out);
  // front half of forward_images: the uint8 pixels -> the network's NHWC input image, enqueued on s (the plan of the canvas
  // shape is active afterwards); returns the canvas height / width
  void prep_images(const unsigned char* bgr, int n, int h, int w, double scale, bool is_device, void* s);
  std::string plan_text();
  std::string profile_text(int iters);
  std::string tune_report_text();  // per GEMM signature of the current plan: tile in use, launches, the isolated timings of autotune
  void set_tile(const std::string& key, const std::string& tile);  // override the tile of one signature (tuning under the caller's own load)
  std::string debug_info_text();  // Net::ForwardDebugInfo (net.cpp:648-681): mean |x| of every top / parameter blob
  int layer_index(const std::string& name) const;

 private:
  friend struct NetGroup;
  void init_from(const TextMsg& root);
  void setup_layer(LayerRec& L);
  void reshape_layer(LayerRec& L);
  void ensure_device();
  void upload_vecs();
  void run_launch(const Launch& l, void* s);
  void run_plan(int start, int end, void* s);
  const Net* clone_src_ = nullptr;  // during create() of a clone only
  void release_graph();
  void check_weights();   // compare the shared weight generation / touched parameters with what the plans were built from
  void park_current();
  void autotune();
  std::string tune_key(const Launch& l) const;
  Storage& begin_batch(int n, int h, int w);
  void enqueue_plan(void* s);
  void emit_maps(void* prob, void* loc, void* next, bool is_device, void* s, int dst_esize = 4);
  std::shared_ptr<ResampleTable> resample_table(int in_size, int out_size);
  std::map<std::pair<int, int>, std::shared_ptr<ResampleTable>> resample_;
  struct MapRef {
    const void* ptr;
    int cp, c0, es, NB, C, H, W;
  };
  MapRef map_ref(const char* blob_name);  // device image of an output map (channel views of the merged heads included)
  unsigned char* scratch_dev_ = nullptr;  // candidates / detections / pairwise scratch
  size_t scratch_cap_ = 0;
  void* scratch(size_t bytes);
  unsigned char* img_dev_ = nullptr;  // uint8 source images uploaded from the host
  size_t img_cap_ = 0;
  unsigned char* tmp_dev_ = nullptr;  // horizontally resampled rows
  size_t tmp_cap_ = 0;
};

// ---- pyramid-grouped execution (round 4) ---------------------------------------------------------------------------------
// The demo runs the scales of an image pyramid one forward after the other (python/pose/estimate_pose.py:81-128), the
// reference one SGEMM per image and layer (base_conv_layer.cpp:326-341): four launches per layer over the SAME filters, each
// starting on L2s that hold neither its filters nor its input, each with its own dispatch ramp and tail.  A NetGroup runs
// several executors of ONE model (a net and its clones, each at its own input shape) as ONE launch sequence: launch i of
// the group is launch i of every member's plan merged into a multi-problem gather-GEMM (kernels.h ConvProblem) — the
// residue classes of the deconvolution heads become problems too —, so a 4-scale pyramid is 161 launches (one lane) or 318 (the default two
// concurrent lanes of two scales, GroupPlan::lane_members) instead of 632.
// Members keep their blobs, plans and tile choices: a member can still be run alone.  Launches that cannot merge (Winograd
// form, max-pool, stand-alone element-wise layers) run member by member inside the same sequence.
struct GroupLaunch {
  bool multi = false;
  int lane = 0;               // which lane of the plan runs it (GroupPlan::lane_members)
  int index = 0;              // index into every member's plan
  int member = -1;            // !multi: the member whose launch `index` this is
  ConvGemmParams p{};         // multi: the layer's common block (not yet prepared for a variant)
  const void* ws_w = nullptr; // ... the layer's stream1x1 filter image if every member has one (variant kStreamHalf reads it instead of p.w)
  ConvMultiTable table{};     // ... and the problems (pointers filled, not yet prepared)
  ConvMultiArgs args{};       // both, prepared for `variant`: the kernel arguments
  int nprob = 0;
  int variant = -1;
  long grid = 0;
  std::string key, label;
  double flops = 0;
  std::vector<int> prob_member;  // per problem: the member it belongs to (diagnostics)
};
struct GroupPlan {
  std::vector<std::vector<int>> shapes;   // per member: its input shape
  std::vector<uint64_t> lowerings, buf_gens, weight_gens, tile_gens;  // per member, when the plan was merged
  std::vector<GroupLaunch> launches;
  int nlanes = 1;
  std::vector<std::vector<int>> lane_members;  // per lane: member indices (ascending)
  std::vector<void*> lane_graphs;              // per lane: the captured hipGraphExec, or null
  bool tuned = false;
  uint64_t last_use = 0;
  double flops = 0;
  void drop_graphs();
};
struct GroupStats {
  long long merges = 0, graph_instantiations = 0, autotune_runs = 0, plan_hits = 0;
};
struct NetGroup {
  std::vector<Net*> nets;  // borrowed: executors of one model (a net and its clones), alive as long as the group
  GroupStats stats;
  std::string text_buf;
  ~NetGroup();
  static NetGroup* create(const std::vector<Net*>& members);
  // one batch per member (member c gets inputs[c] of n[c] x 3 x h[c] x w[c]); pointers as Net::forward_batch
  void forward_batch(const float* const* inputs, const int* n, const int* h, const int* w, bool is_device, float* const* prob,
                     float* const* loc, float* const* next, void* user_stream);
  // image entry: member c pre-processes n[c] images of h[c] x w[c] at scale[c] (Net::forward_images), then ONE grouped
  // forward, then per member the maps / the decoded pose
  void forward_images(const unsigned char* const* bgr, const int* n, const int* h, const int* w, const double* scale, bool is_device,
                      float* const* prob, float* const* loc, float* const* next, double* const* pose, void* user_stream);
  // lanes: 0 = automatic (2 members: two lanes; 3: one; 4 and more: two), else that many (at most one per member); every merged plan is dropped
  void set_lanes(int n);
  int lanes() const { return cur_ ? cur_->nlanes : lanes_opt_; }
  std::string plan_text();
  std::string profile_text(int iters);
  // as Net::tune_report_text / Net::set_tile, for the merged launches of the last forward's plan (signature = "G<problems>:" + the
  // members' signatures): what deepcut_tools.tune_in_flight walks when the load is groups in flight
  std::string tune_report_text();
  void set_tile(const std::string& key, const std::string& tile);
  int num_launches();
  int num_multi_launches();
  double flops();

 private:
  std::vector<std::unique_ptr<GroupPlan>> plans_;
  GroupPlan* cur_ = nullptr;
  uint64_t use_clock_ = 0;
  int lanes_opt_ = 0;
  std::vector<void*> lane_events_;  // per lane: its join event
  std::map<void*, std::vector<void*>> lane_choice_;  // caller's stream -> per lane the side stream measured best (index 0 unused; borrowed)
  void* fork_event_ = nullptr;
  GroupPlan& ensure_plan();   // after every member's begin_batch: the merged plan of the members' current shapes
  void merge(GroupPlan& gp);
  bool plan_current(const GroupPlan& gp) const;
  GroupPlan& current_plan();
  void forget_plan(GroupPlan* gp);  // out of the cache (a merge that threw, an eviction); cur_ never dangles
  void autotune(GroupPlan& gp);
  void apply_variant(GroupPlan& gp, GroupLaunch& gl, int variant);
  void run(GroupPlan& gp, int lane, void* s);  // lane < 0: every launch
  void enqueue(void* s);
  void launch_lanes(GroupPlan& gp, void* s, const std::vector<void*>& side, bool use_graph);
  void choose_lane_streams(GroupPlan& gp, void* s, bool use_graph);
  void drop_plan(GroupPlan& gp);
  void* stream();
};

// ---- runtime.cpp: pinned host memory (dc_host_alloc / dc_host_free) ---------------------------------------------------------
void* host_alloc_pinned(size_t bytes);
void host_free_pinned(void* p);

// ---- multi_gpu.cpp: in-process multi-GPU forward (dc_comm_*, dc_forward_batch) ----------------------------------------------
struct Comm;
Comm* comm_create(int nexec, const int* devices, int transport);
void comm_destroy(Comm* c);
int comm_transport(const Comm* c);
int comm_nexec(const Comm* c);
void comm_forward(Comm* c, Net* const* nets, int nexec, const float* const* inputs, const int (*hw)[2], int n, float* const* prob, float* const* loc,
                  float* const* next);
int comm_item_executor(const Comm* c, int i);
void comm_root_maps(const Comm* c, int i, const void** prob, const void** loc, const void** next, int dims[5]);
std::vector<std::vector<int>> lpt_schedule(const std::vector<double>& cost, int nexec);

}  // namespace dc
