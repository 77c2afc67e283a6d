/*
 * deepcut_hip.h — C ABI of libdeepcut_hip.so, the MI355X (gfx950) forward path of the
 * DeeperCut part detector.
 *
 * This is the drop-in boundary for the ONE hot path of eldar/deepcut-cnn: the TEST-phase
 * forward of models/deepercut/ResNet-152.prototxt behind caffe::Net::ForwardFromTo /
 * pycaffe net.forward().  The reference has no C ABI (it binds C++ to Python through
 * Boost.Python, python/caffe/_caffe.cpp); every entry point below names the reference
 * interface it stands in for (file:line relative to the reference tree).  A maintainer
 * binds these with ctypes / pybind / Boost.Python exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DC_E* code on failure; the message is
 *     available from dc_last_error() (thread local).  Nothing aborts the process (the
 *     reference LOG(FATAL)s; we never continue silently either).
 *   - handles are opaque; a dc_blob* is owned by its net (or, for dc_blob_create, by the
 *     caller) and stays valid until that owner is destroyed.
 *   - host tensors are float32, C-contiguous NCHW exactly as caffe::Blob (blob.hpp:153-164).
 *     Device tensors are channels-last (NHWC) float32; see DESIGN.md "Data layout in HBM".
 *   - mode/device are per thread, like caffe::Caffe (common.cpp:13-20).
 *   - there is NO CPU compute path in this library: dc_net_forward in CPU mode fails with
 *     DC_ENOCPU.  The CPU restatement of the reference lives in oracle/ and is test-only.
 */
#ifndef DEEPCUT_HIP_H_
#define DEEPCUT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK 0
#define DC_EINVAL (-1)   /* bad argument / malformed model                         */
#define DC_EIO (-2)      /* "Could not open file ..." (_caffe.cpp:45-52)            */
#define DC_ESHAPE (-3)   /* shape / blob-count mismatch (net.cpp:822-834)           */
#define DC_EUNSUP (-4)   /* layer type or parameter outside the supported path      */
#define DC_EDEVICE (-5)  /* HIP runtime error / no gfx950 device                    */
#define DC_ENOCPU (-6)   /* forward requested in CPU mode: not provided             */

#define DC_MODE_CPU 0 /* caffe::Caffe::CPU (common.hpp:107) */
#define DC_MODE_GPU 1 /* caffe::Caffe::GPU                  */
#define DC_PHASE_TRAIN 0 /* caffe.proto:253-256 */
#define DC_PHASE_TEST 1

/* dc_net_set_option keys */
#define DC_OPT_FUSE 1       /* 0: every named blob materialised (Caffe-visible semantics);
                               1: + residual-add and Deconvolution+Crop+Eltwise head fusion;
                               2 (default): + the sibling heads run as one concatenated GEMM     */
#define DC_OPT_HIPGRAPH 2   /* 1: replay the per-shape launch sequence as a hipGraph        */
#define DC_OPT_DTYPE 3      /* 0 (default): float32 activations and filters in HBM, v_mfma_f32_32x32x2_f32;
                               1: float16 activations and filters, v_mfma_f32_32x32x16_f16 with float32
                               accumulation and epilogue (BASELINE configs[2]); host blobs stay float32 */

#define DC_OPT_OUTPUTS 4    /* bit i set = the i-th output blob of the net (dc_net_num_outputs order: alphabetical, net.cpp:268-273) is wanted;
                               default -1 = all.  The lowering drops every launch that only feeds unwanted outputs (the demo reads `prob`
                               and `loc_pred` only, python/pose/estimate_pose.py:231-241, and the 364-channel `next_pred` head is 23.3 of the
                               241 GFLOP of a 544x736 forward); an unwanted output blob is elided: reading it is DC_EUNSUP.  The wanted
                               maps equal the full forward's up to the summation order of the tile chosen for the narrower head GEMM
                               (bit for bit under the same tile, tests/test_gpu_outputs.py) */

typedef struct dc_net dc_net;
typedef struct dc_blob dc_blob;

/* ---- error / context ---------------------------------------------------------------- */
const char* dc_last_error(void);
const char* dc_version(void);

/* Caffe::set_mode / Caffe::mode  (common.hpp:148-150; _caffe.cpp:38-39)                  */
int dc_set_mode(int mode);
int dc_get_mode(void);
/* Caffe::SetDevice (common.cpp:140-158; _caffe.cpp:221)                                  */
int dc_set_device(int device_id);
int dc_get_device(void);
/* number of visible HIP devices (0 when there is no GPU; never fails)                    */
int dc_device_count(void);

/* ---- Net ------------------------------------------------------------------------------ */
/* Net<float>::Net(param_file, phase) + CopyTrainedLayersFrom(weights)
 * (net.cpp:31-37,843-858; _caffe.cpp:76-96).  caffemodel may be NULL.                    */
int dc_net_create(const char* prototxt_path, const char* caffemodel_path, int phase, dc_net** out);
/* same, from an in-memory prototxt string                                                */
int dc_net_create_from_text(const char* prototxt_text, const char* caffemodel_path, int phase,
                            dc_net** out);
int dc_net_destroy(dc_net* net);
/* A second executor of the same model on the same device: own activations / stream / hipGraph, SHARED
 * parameter blobs and packed filter images (no reference counterpart; Net::ShareTrainedLayersWith,
 * net.cpp:751-769, is the nearest idea).  Used to keep several independent forwards in flight.           */
int dc_net_clone(dc_net* net, dc_net** out);
/* wait for everything enqueued on the net's own stream (see DC_STREAM_OWN)                                */
int dc_net_synchronize(dc_net* net);
/* non-blocking: *busy = 1 while work enqueued on the net's own stream (DC_STREAM_OWN) has not finished, else 0 — what a
 * dispatcher polls to hand queued requests to whichever executor is free (deepcut_tools.Pipeline); no reference counterpart  */
int dc_net_busy(dc_net* net, int* busy);
int dc_net_set_option(dc_net* net, int key, int value);
int dc_net_get_option(dc_net* net, int key, int* value);
/* Net::CopyTrainedLayersFrom(file) (net.cpp:805-858): match by layer name, check blob
 * count and shape, ignore unmatched source layers.  Formats: binary NetParameter in the current
 * `layer` form or the deprecated V1 / V0 `layers` form (upgraded as upgrade_proto.cpp:19-78 does),
 * and — for names ending in ".h5", the reference's rule (net.cpp:843-850) — HDF5 weights
 * /data/<layer>/<param index> (CopyTrainedLayersFromHDF5, net.cpp:861-909).                     */
int dc_net_copy_from(dc_net* net, const char* caffemodel_path);
/* Net::ToProto + WriteProtoToBinaryFile (net.cpp:910-925; _caffe.cpp:98-102)             */
int dc_net_save(dc_net* net, const char* caffemodel_path);
const char* dc_net_name(dc_net* net);

/* Net::layer_names / layers()[i]->type() (net.hpp:126-133), AFTER InsertSplits           */
int dc_net_num_layers(dc_net* net);
const char* dc_net_layer_name(dc_net* net, int i);
const char* dc_net_layer_type(dc_net* net, int i);
/* Net::blob_names / blobs (net.hpp:122-125,135), creation order, split blobs included    */
int dc_net_num_blobs(dc_net* net);
const char* dc_net_blob_name(dc_net* net, int i);
/* Net::blob_by_name (net.cpp:947-957); unknown name -> DC_EINVAL                         */
int dc_net_blob(dc_net* net, const char* name, dc_blob** out);
/* Net::input_blob_indices / output_blob_indices (net.hpp:182-189): outputs are the
 * unconsumed blobs in alphabetical order (net.cpp:268-273)                               */
int dc_net_num_inputs(dc_net* net);
const char* dc_net_input_name(dc_net* net, int i);
int dc_net_num_outputs(dc_net* net);
const char* dc_net_output_name(dc_net* net, int i);
/* Net::layers()[i]->blobs()[j]  (pycaffe net.params, pycaffe.py:40-51)                   */
int dc_net_layer_num_params(dc_net* net, const char* layer_name);
int dc_net_param(dc_net* net, const char* layer_name, int idx, dc_blob** out);

/* Net::Reshape (net.cpp:744-749): propagate the current input shapes                     */
int dc_net_reshape(dc_net* net);
/* Net::ForwardFromTo(start, end) (net.cpp:565-581; _caffe.cpp:231 "_forward").  Layer
 * indices are those of dc_net_layer_name.  Re-derives every shape from the current input
 * shape (Layer::Forward calls Reshape, layer.hpp:451-456).  Synchronous: on return all
 * outputs are computed.  *loss (may be NULL) receives 0 (no loss layers on this path).   */
int dc_net_forward(dc_net* net, int start, int end, float* loss);
/* Net::ForwardPrefilled convenience: whole net                                           */
int dc_net_forward_all(dc_net* net);

/* ---- Blob (caffe::Blob<float> + SyncedMemory) ----------------------------------------- */
/* Blob::shape (blob.hpp:52-71).  dims must hold 4 ints (params may have 1 axis)          */
int dc_blob_num_axes(dc_blob* b);
int dc_blob_shape(dc_blob* b, int* ndim, int* dims /*[8]*/);
int dc_blob_count(dc_blob* b);
/* Blob::Reshape (blob.cpp:23-43; _caffe.cpp:181-193): capacity only grows                */
int dc_blob_reshape(dc_blob* b, int ndim, const int* dims);
/* Blob::cpu_data / mutable_cpu_data (blob.cpp:82-86,105-109 -> syncedmem.cpp:25-77,
 * 103-128).  The pointer is host memory owned by the blob, NCHW, valid until a reshape
 * grows the blob or the owner is destroyed.  mutable_: host becomes authoritative
 * (HEAD_AT_CPU) so the next forward re-uploads; a pending device result is downloaded first. */
int dc_blob_cpu_data(dc_blob* b, const float** out);
int dc_blob_mutable_cpu_data(dc_blob* b, float** out);
/* SyncedMemory::head() (syncedmem.hpp:59): 0 UNINITIALIZED 1 HEAD_AT_CPU 2 HEAD_AT_GPU 3 SYNCED */
int dc_blob_head(dc_blob* b);
/* Blob::gpu_data (blob.cpp:88-92): device pointer of the channels-last (NHWC) image of the
 * blob, plus its channel pitch (>= channels; the 3-channel input is stored with pitch 4). */
int dc_blob_gpu_data(dc_blob* b, const void** dev_ptr, int* channel_pitch);

/* Blob<float>() / Blob<float>(shape) (blob.hpp:26-33): a blob of its own, owned by the caller — the bottoms and tops a
 * stand-alone Layer is driven with, or the backing store of a caffe::SyncedMemory.  It moves between host and device on
 * the default stream of the calling thread's device (Caffe::SetDevice).  dc_blob_destroy refuses a net's blob.       */
int dc_blob_create(int ndim, const int* dims, dc_blob** out);
int dc_blob_destroy(dc_blob* b);
/* Blob::mutable_gpu_data (blob.cpp:111-115 -> SyncedMemory::mutable_gpu_data, syncedmem.cpp:130-139): the device image
 * (channels-last for 4-D blobs, plain otherwise) becomes authoritative (HEAD_AT_GPU); an UNINITIALIZED blob gets a
 * zeroed image, a host-side one is uploaded first.  Refused for blobs a fused plan never materialises.            */
int dc_blob_mutable_gpu_data(dc_blob* b, void** dev_ptr, int* channel_pitch);
/* Blob::CopyFrom(source, copy_diff=false, reshape) (blob.cpp:435-474): copies wherever the source is authoritative —
 * host to host, or device image to device image (re-pitched) leaving dst HEAD_AT_GPU.  Shapes must agree unless
 * `reshape`.                                                                                                      */
int dc_blob_copy_from(dc_blob* dst, dc_blob* src, int reshape);

/* ---- one reference layer stand-alone: Layer<Dtype>::SetUp(bottom, top) (layer.hpp:67-74) ---------------------------
 * layer_param_text: the text-format LayerParameter (the body of a `layer { }` message, or the message itself).  The
 * result is a net whose inputs are the layer's bottoms (named as in the text, shaped like `bottoms`), with DC_OPT_FUSE 0:
 * Layer::Reshape = dc_blob_reshape on its inputs + dc_net_reshape, Layer::Forward_gpu = dc_net_forward_all, the layer's
 * blobs() = dc_net_param(net, <layer name>, i).  include/caffe_facade.hpp wraps this as caffe::Layer<float>.        */
int dc_net_create_for_layer(const char* layer_param_text, int phase, int nbottom, dc_blob* const* bottoms, dc_net** out);

/* ---- batched / sharded extension (no reference counterpart: the reference forwards one
 * image at a time, conv_layer.cpp:31).  Runs `n` same-shape images as one batch:
 * inputs  : host or device NCHW float32 [n,3,H,W] (is_device selects)
 * outputs : prob [n,14,h,w], loc_pred [n,28,h,w], next_pred [n,364,h,w] NCHW float32, host or
 *           device like the input; any of them may be NULL to skip the copy-out.
 * stream  : hipStream_t to enqueue on.  NULL = the net's own stream, synchronous.  DC_STREAM_OWN = the
 *           net's own stream, asynchronous for device buffers (pair with dc_net_synchronize).  With a
 *           caller stream and device buffers the call is asynchronous on that stream.              */
#define DC_STREAM_OWN ((void*)-1)
int dc_net_forward_batch(dc_net* net, const float* input, int n, int h, int w, int is_device,
                         float* prob, float* loc_pred, float* next_pred, void* stream);

/* The same with HOST buffers and NO wait: the upload of the batch, the forward and the downloads of the maps are enqueued on the
 * net's own stream; dc_net_synchronize (or dc_net_busy) tells when the outputs are there, and input and outputs must stay valid
 * until then.  With buffers from dc_host_alloc (pinned) the copies run on the DMA engines beside other executors' kernels, which is
 * what lets several executors keep host-in / host-out requests in flight (deepcut_tools.Pipeline.submit_host); pageable buffers
 * work too, staged by the runtime.  Replaces the blocking copies of SyncedMemory::to_gpu / to_cpu (src/caffe/syncedmem.cpp:25-77) for
 * callers that do not need the reference's synchronous contract.                                                          */
int dc_net_forward_host_async(dc_net* net, const float* input, int n, int h, int w, float* prob, float* loc_pred, float* next_pred);
/* pinned (page-locked) host memory for the entry above; the reference pins its blobs the same way in GPU mode
 * (CaffeMallocHost, include/caffe/syncedmem.hpp:15-44)                                                                 */
int dc_host_alloc(size_t bytes, void** out);
int dc_host_free(void* p);

/* ---- executor streams chosen by measurement -----------------------------------------------------------------------------
 * A HIP process has four hardware queues; a stream is bound to one of them at creation and streams that share a queue run one
 * after the other — with four executors "in flight" the throughput is 380 to 490 images/s depending on which streams they got
 * (profiles/r04_stream_subsets.txt), and the API does not say.  dc_nets_choose_streams times the executors' REAL forwards (each must
 * have run or reserved its shape) on assignments of a process-wide pool of `candidates` (0 = 8) streams, `reps` (0 = 3) forwards
 * per executor and burst, and makes the best assignment the executors' own streams (DC_STREAM_OWN, dc_net_stream).  rate_chosen /
 * rate_first (may be NULL): forwards per second with the chosen streams / with the first n streams of the pool.  No reference
 * counterpart (one legacy stream per thread, src/caffe/common.cpp:99-158).                                                */
int dc_nets_choose_streams(dc_net* const* nets, int n, int candidates, int reps, double* rate_chosen, double* rate_first);
/* the net's own stream (a hipStream_t), created on first use: what DC_STREAM_OWN enqueues on                              */
int dc_net_stream(dc_net* net, void** out);

/* Cross-request batching: `n` independent single-image requests — one DEVICE input pointer ([3,H,W] float32) and one
 * set of DEVICE output pointers per request (the arrays, or single entries, may be NULL) — run as ONE batch-n forward;
 * request i's maps land in its own buffers.  What a server does with concurrent batch-1 requests (the reference forwards
 * one image at a time, conv_layer.cpp:31).  stream as dc_net_forward_batch.                                       */
int dc_net_forward_requests(dc_net* net, int n, const float* const* inputs, int h, int w, float* const* prob,
                            float* const* loc_pred, float* const* next_pred, void* stream);

/* The maps of the LAST forward copied out as NCHW, host or device destination, any pointer NULL to skip: elem 0 =
 * float32; elem 1 = float16, offered by fp16 nets (DC_OPT_DTYPE 1) only — the values as they are in HBM, i.e. half the
 * bytes for the gather of the maps to rank 0 (no reference counterpart; Blob::cpu_data of the three outputs).
 * stream as dc_net_forward_batch.                                                                             */
int dc_net_emit_maps(dc_net* net, void* prob, void* loc_pred, void* next_pred, int elem, int is_device, void* stream);

/* ---- pose decoding on the device (python/pose/estimate_pose.py:131-143 `_pose_from_mats`): after a
 * forward, writes pose[n][5][J] doubles (x, y, confidence, and the refinement vector in the reference's
 * (row, column) order, all divided by `scale`) to a host buffer (is_device=0) or a device buffer.        */
int dc_net_decode_pose(dc_net* net, double scale, double* pose, int is_device, void* stream);

/* ---- image entry: the demo's pre-processing on the device + forward + optional decode -------------------
 * python/pose/estimate_pose.py:83-128 for `n` same-size images at one scale, without the float canvas ever
 * existing on the host: replicate the last row / column 64 px (:89-95), scipy.misc.imresize(img, scale,
 * 'bilinear') = Pillow's 8-bit two-pass bilinear resample to (int((W+64)*scale), int((H+64)*scale)) (:96;
 * bit-exact: 22-bit fixed-point weights computed as Pillow does; the identity at scale 1), subtract the BGR
 * mean [104,117,123] (:97), paste on a zero canvas of ceil(H*scale/8)*8 x ceil(W*scale/8)*8 (:85-88,
 * :99-103), written straight into the `data` blob's image in HBM; then the forward; then, if `pose` is not
 * NULL, `_pose_from_mats` (:131-143) as dc_net_decode_pose does.
 * images  : n * height * width * 3 bytes, BGR, HWC, packed; host (is_device=0) or device memory.
 * outputs : as dc_net_forward_batch (any may be NULL); pose = n*5*J doubles or NULL, host/device like the rest.
 * stream  : as dc_net_forward_batch.  The net input is reshaped to the canvas size.                       */
int dc_net_forward_images(dc_net* net, const unsigned char* images, int n, int height, int width, double scale,
                          int is_device, float* prob, float* loc_pred, float* next_pred, double* pose, void* stream);
/* the canvas (= network input) height and width dc_net_forward_images uses for an image at `scale` */
int dc_image_canvas_size(int height, int width, double scale, int* canvas_h, int* canvas_w);

/* ---- multi-person consumers of the maps (no reference code: the reference repository stops at the maps) --------
 * What they invert is the label encoding of the reference's training layer (src/caffe/layers/pose_data_layer.cpp:
 * 686-802): a map cell (row, col) stands for the image point pt = (col*8+4, row*8+4)/scale; loc_pred holds
 * (joint - pt)*scale/sqrt(53); next_pred channel pair l holds ((next joint - pt)*scale - mean[l]) / std[l] for
 * regression edge l (edges, means and stds come from the model's `joint_pairs_stats` file, caffe.proto:1184).
 *
 * dc_net_detect_parts: non-maximum suppression of every score map of the last forward, on the device: a cell is a
 * candidate if prob >= threshold and it is the maximum of its (2*radius+1)^2 window (ties: the lower row-major cell
 * index).  Per image and joint the candidates are ordered by (score descending, cell ascending) and the first
 * max_det are returned: counts[n*J + j] and dets[((n*J + j)*max_det + k)*5 + {0..4}] = x, y (refined with loc_pred,
 * divided by scale), score, cell row, cell column.  Host buffers; synchronous.
 * dc_net_decode_pairwise: for ndet detections given as (image, cell row, cell column) triples and every edge l,
 * out[(d*E + l)*2 + {0,1}] = pt + (next_pred[2l + k] at the cell * std[l][k] + mean[l][k]) / scale, E = channels/2;
 * mean / std may be NULL (0 / 1).  Host buffers; synchronous.                                               */
int dc_net_detect_parts(dc_net* net, double scale, float threshold, int radius, int max_det, int* counts, double* dets);
int dc_net_decode_pairwise(dc_net* net, double scale, int ndet, const int* detections, const double* mean,
                           const double* stdev, double* out);


/* ---- introspection used by bench.py / DESIGN.md ----------------------------------------- */
/* algorithmic FLOPs (2*MAC of conv+deconv) of the current shape               */
int dc_net_flops(dc_net* net, double* flops);
/* number of kernel launches in the current plan                                           */
int dc_net_num_launches(dc_net* net);
/* counters of the per-shape plan cache: Layer::Forward re-derives every shape on every call (layer.hpp:451-456) and the
 * demo changes the input shape once per scale (estimate_pose.py:81-128); a shape met before must cost neither a
 * re-lowering nor a graph instantiation.  out[i] for i < n:                                                     */
#define DC_STAT_LOWERINGS 0        /* times the layer graph was lowered to a launch plan                  */
#define DC_STAT_GRAPH_INSTANTIATIONS 1 /* hipGraph captures + instantiations                              */
#define DC_STAT_PLAN_HITS 2        /* shape changes served from the cache                                 */
#define DC_STAT_AUTOTUNE_RUNS 3    /* plans for which at least one GEMM signature had to be timed         */
#define DC_STAT_BUFFER_GROWTHS 4   /* device buffers (re)allocated                                        */
#define DC_STAT_REPACKS 5          /* times the filter images were re-packed from the parameter blobs     */
#define DC_STAT_CACHED_PLANS 6     /* shapes currently cached (LRU of DC_PLAN_CACHE, default 16)          */
#define DC_NUM_STATS 7
int dc_net_stats(dc_net* net, long long* out, int n);
/* lower, allocate and tune the plan of an [n,3,h,w] input without running it: reserving the LARGEST shape of a
 * pyramid first means no buffer grows (and no captured graph goes stale) while the smaller ones are met        */
int dc_net_reserve(dc_net* net, int n, int h, int w);
/* the HIP device this net executes on (-1 until its first device use: then Caffe::SetDevice's value, common.cpp:140) */
int dc_net_device(dc_net* net);
/* human-readable launch plan of the current shape (kernel variant, tile, grid per op);
 * pointer valid until the next call on this net                                           */
const char* dc_net_plan_text(dc_net* net);
/* time each op of the current plan with hipEvents on the net's stream (iters runs each);
 * returns a text table (op, kernel, us, GFLOP, TFLOP/s); pointer valid until next call    */
const char* dc_net_profile_text(dc_net* net, int iters);
/* Net::ForwardDebugInfo / InputDebugInfo (src/caffe/net.cpp:648-681, `debug_info: true` in the NetParameter,
 * caffe.proto:88): one line per input, top blob and parameter blob with its mean absolute value after the LAST
 * forward, in the reference's log format ("    [Forward] Layer conv1, top blob conv1 data: 0.0645").  In-place chains
 * run as one kernel here: the value is reported at the chain's last layer; with DC_OPT_FUSE 0 every Caffe-visible
 * blob is materialised (fused blobs are reported as elided otherwise).  NULL + dc_last_error() on failure;
 * pointer valid until the next call on this net.                                                                   */
const char* dc_net_debug_info(dc_net* net);
/* Tile choices of the current shape, for a tuner that works under the caller's own load (deepcut_tools.tune_in_flight): one line
 * per GEMM signature of the plan, "<signature>\t<tile in use>\t<launches>\t<tile>:<us timed alone> ..." (fastest first; the
 * signature is the key DC_TUNE_CACHE files use).  dc_net_set_tile overrides the tile of one signature in this executor's
 * current plan and in the choice table it shares with its clones; the captured graph is dropped (re-captured by the next
 * forward): call it while the executor is idle (nothing of it in flight on any stream).  DC_EUNSUP if the tile cannot take a
 * launch of the signature.  No reference counterpart: the reference has one
 * SGEMM per layer (math_functions.cu:13-27).                                                                            */
const char* dc_net_tune_report(dc_net* net);
int dc_net_set_tile(dc_net* net, const char* signature, const char* tile);
/* the gather-GEMM's tile-variant table (csrc/kernels.hip): number of entries, name and element size (4 float / 2 half) of
 * entry i — what the environment switch DC_CONV_VARIANT=<i> forces and the names dc_net_plan_text / DC_TUNE_CACHE use.
 * No reference counterpart: the reference has one SGEMM (math_functions.cu:13-27); diagnostics only.                  */
int dc_conv_variant_count(void);
const char* dc_conv_variant_name(int i);
int dc_conv_variant_esize(int i);
/* the float16 Winograd form's filter image (csrc/wino_f16.hip, tile `wino_h23`), made on the host exactly as the lowering makes it:
 * g = [cout][cin][3][3] (Caffe order, convolution_param of a stride-1 3x3 layer; cin % 16 == 0, cout % 32 == 0) -> out[16 * cout * cin]
 * = U = G g G^T per (co, ci) in double, channel co multiplied by row_scale[co]^-1 — an exact power of two bringing its largest |U|
 * into [2^13, 2^14) when `rowscale` is non-zero, else 1 —, in MFMA fragment order [cout/32][4 i][cin/16][4 j][64 lanes][8]:
 * lane = 32 * ((ci % 16) / 8) + co % 32, element = ci % 8.  Diagnostics / tests (the host half of the kernel's parity:
 * tests/test_wino_half_pack.py); the reference has no counterpart (its 3x3 layers are im2col + SGEMM, base_conv_layer.cpp:257-280). */
int dc_wino_half_pack(const float* g, int cout, int cin, int rowscale, float* out, float* row_scale);

/* the filter images of the two float16 kernels added in round 6, made on the host exactly as the lowering makes them (diagnostics / tests:
 * tests/test_stream_pack.py emulates the matrix instruction's operand layout on them; the reference has no counterpart — its 1x1 layers
 * are one SGEMM per image, base_conv_layer.cpp:326-341, its stem im2col + SGEMM, base_conv_layer.cpp:257-280):
 *  dc_stream1x1_pack: g = [cout][k] (a 1x1 filter bank; cout % 32 == 0, k % 16 == 0) -> out[cout * k] in the ROW-operand order of
 *    v_mfma_f32_32x32x16_f16, [cout/32][k/16][64 lanes][8]: lane = 32 * ((kk % 16) / 8) + co % 32, element = kk % 8 (csrc/stream1x1.hip);
 *  dc_stem7x7_pack:   g = [64][c][7][7] (c <= 4) -> out[14336] = [fragment 2][kernel row 7][K step 2][64 lanes][8], element e = kx * 4 + ci
 *    of a kernel row at lane 32 * ((e % 16) / 8) + co % 32, position e % 8 of K step e / 16, zeros elsewhere (csrc/stem_f16.hip).        */
int dc_stream1x1_pack(const float* g, int cout, int k, float* out);
int dc_stem7x7_pack(const float* g, int c, float* out);
/*  dc_stream1x1f_pack: the float32 form of the streaming 1x1 kernel (csrc/stream1x1_f32.hip, tile `ws1x1f`): g = [cout][k] (cout % 16 == 0,
 *    k % 16 == 0) -> out[cout * k] in the ROW-operand order of v_mfma_f32_16x16x4_f32 with the K range cut into four runs,
 *    [cout/16][k/16 vectors][64 lanes][4]: lane = 16 q + co % 16 holds run q, element e of vector j = g[co][q k/4 + 4 j + e] — matrix step
 *    4 j + e of a 16-pixel step multiplies column (q k/4 + 4 j + e) of the filters with the same element of the pixel rows (one 16-byte LDS
 *    read of a pixel's row feeds four matrix steps).  tests/test_stream_pack.py.                                                        */
int dc_stream1x1f_pack(const float* g, int cout, int k, float* out);

/* ---- pyramid-grouped execution: several executors of ONE model, each at its own input shape, as ONE launch sequence ----
 * Replaces the scale loop of the demo (python/pose/estimate_pose.py:81-128: one net.forward() per scale, every shape change a
 * full Reshape) and, per layer, the reference's one-SGEMM-per-image loop (src/caffe/layers/base_conv_layer.cpp:326-341,
 * conv_layer.cpp:31): launch i of the group is launch i of EVERY member merged into one multi-problem gather-GEMM, so a layer's
 * filters are pulled through the L2s once for all scales and the chip sees one dispatch ramp and one tail per layer
 * (a 4-scale pyramid: 161 launches as one lane, 318 as the default two concurrent lanes, instead of 632).  Members are a net and its clones (dc_net_clone: shared parameters, own
 * activations); they stay usable on their own, and their blobs hold the results of a grouped forward exactly as after their own
 * (dc_net_blob / dc_net_decode_pose / dc_net_emit_maps / dc_net_detect_parts on a member see them).  Results equal the members'
 * own forwards up to the fp32 summation order of the tile chosen (bit-identical for the same tile).  The group borrows the nets: it
 * must not RUN after one of them is gone (destroying it afterwards is harmless).  Arrays below have one entry per member, in the order given at creation.                          */
typedef struct dc_group dc_group;
int dc_group_create(dc_net* const* nets, int n, dc_group** out);
int dc_group_destroy(dc_group* group);
int dc_group_size(dc_group* group);
/* LANES: the members are dealt to `lanes` lanes (largest with smallest), every lane is merged on its own and runs on a stream of
 * its own, concurrently with the others — the launches of one lane fill the dispatch ramps and tails of the other's (one grouped
 * 4-scale float16 pyramid batch: 12.1 ms as one lane, 10.6 ms as two), at the price of one filter fetch per lane and layer.
 * 0 (default) = automatic: two members run as two lanes (plain concurrency: faster than merging two tensors), three as one lane,
 * four or more as two lanes of merged members.  Drops the merged plans.                                                      */
int dc_group_set_lanes(dc_group* group, int lanes);
/* dc_net_forward_batch for every member at once: member c forwards inputs[c] = n[c] x 3 x h[c] x w[c]; output pointer arrays
 * (or single entries) may be NULL.  stream as dc_net_forward_batch (NULL = the first member's own stream, synchronous).      */
int dc_group_forward_batch(dc_group* group, const float* const* inputs, const int* n, const int* h, const int* w, int is_device,
                           float* const* prob, float* const* loc_pred, float* const* next_pred, void* stream);
/* dc_net_forward_images for every member at once (member c: n[c] images of height[c] x width[c] at scale[c]; the usual case is
 * the SAME images at the scales of a pyramid): pre-processing per member, ONE grouped forward, then per member the maps and —
 * pose[c] not NULL — the decoded pose.                                                                                       */
int dc_group_forward_images(dc_group* group, const unsigned char* const* images, const int* n, const int* height, const int* width,
                            const double* scale, int is_device, float* const* prob, float* const* loc_pred, float* const* next_pred,
                            double* const* pose, void* stream);
/* the merged plan of the last forward: one line per launch ("conv_gemm_mp<tile> problems=.. grid=.." or "member c: <kernel>");
 * NULL + dc_last_error() before the first forward; pointer valid until the next call on this group                            */
const char* dc_group_plan_text(dc_group* group);
/* every launch of that plan timed with hipEvents (iters runs each), same table as dc_net_profile_text                        */
const char* dc_group_profile_text(dc_group* group, int iters);
/* dc_net_tune_report / dc_net_set_tile for the merged launches of the last forward's plan: signature = "G<problems>:" + the members'
 * signatures joined by '|' (the key DC_TUNE_CACHE files carry for them); set_tile wants the group idle (it synchronises the members'
 * own streams; work enqueued on a caller's stream is the caller's to wait for)                                              */
const char* dc_group_tune_report(dc_group* group);
int dc_group_set_tile(dc_group* group, const char* signature, const char* tile);
#define DC_GSTAT_MERGES 0               /* times the members' plans were merged into a group plan                  */
#define DC_GSTAT_GRAPH_INSTANTIATIONS 1 /* hipGraph captures + instantiations of group plans                       */
#define DC_GSTAT_AUTOTUNE_RUNS 2        /* group plans for which at least one merged signature had to be timed     */
#define DC_GSTAT_PLAN_HITS 3            /* forwards served by a cached group plan                                  */
#define DC_GSTAT_LAUNCHES 4             /* launches of the last forward's plan                                     */
#define DC_GSTAT_MULTI_LAUNCHES 5       /* ... of which multi-problem                                              */
#define DC_GSTAT_LANES 6                /* lanes of the last forward's plan                                        */
#define DC_NUM_GSTATS 7
int dc_group_stats(dc_group* group, long long* out, int n);
/* algorithmic FLOPs of the last grouped forward (the members' dc_net_flops summed)                                          */
int dc_group_flops(dc_group* group, double* out);

/* ---- in-process multi-GPU forward (SURVEY 8(b)'s dc_forward_batch) -------------------------------------------------------
 * A communicator over `nexec` executors: executor k runs on devices[k] (NULL: k modulo the visible devices) on a host thread of its
 * own (mode and device are per thread, src/caffe/common.cpp:13-20).  transport: how the maps travel to the root executor's device —
 * DC_COMM_RCCL: one grouped ncclRecv x (n-1) / ncclSend exchange (librccl.so is opened with dlopen at the first use; needs one
 * executor per device); DC_COMM_PEER: hipMemcpyPeerAsync (device-to-device copies when executors share a device: the loop-back
 * transport of the 1-GPU tests); DC_COMM_AUTO: RCCL when it loads, the devices are distinct AND the communicators it makes move a
 * byte from every peer to the root at creation (dc_comm_create probes them), else PEER; an explicit DC_COMM_RCCL reports the failure
 * instead.  ENVIRONMENT: on hosts whose driver offers dmabuf IPC only (the MI355X boxes this was built on), RCCL's peer buffers need
 * HSA_ENABLE_IPC_MODE_LEGACY=0 in the process environment BEFORE the HIP runtime is loaded — without it ncclCommInitAll / the first
 * exchange fail with `hipIpcGetMemHandle: invalid argument`.  The library does not change the environment of its host process.
 * The calling thread's current HIP device is the same after dc_comm_create / dc_forward_batch as before.
 * dc_forward_batch: `n` host images (inputs[i]: 3 x hw[i][0] x hw[i][1] float32 NCHW, shapes may differ) are dealt to the executors
 * longest-processing-time-first over H*W, every executor forwards the same-shape images of its share on nets[k], a group of 8 (float16:
 * 16) or more images as two sub-batches, so that staging, forward and the way back of consecutive (sub-)batches overlap (round 6) —
 * nets[k] must live on devices[k]: replicas created under dc_set_device(k), or clones where executors share a device —, the maps are
 * gathered on the root's device (dc_comm_root_maps: NCHW float32 device pointers of image i, dims = {prob, loc_pred, next_pred
 * channels, map height, map width}, valid until the next call) and copied to prob[i] / loc_pred[i] / next_pred[i] (host; arrays or
 * entries may be NULL).  Synchronous.  The reference has no inference-time multi-GPU path (its P2PSync, src/caffe/parallel.cpp:287-322,
 * sums gradients along a tree); the consumer this serves is a tools/caffe.cpp-style C++ program (tools/caffe.cpp:302-388).   */
#define DC_COMM_AUTO 0
#define DC_COMM_RCCL 1
#define DC_COMM_PEER 2
typedef struct dc_comm dc_comm;
int dc_comm_create(int nexec, const int* devices, int transport, dc_comm** out);
int dc_comm_destroy(dc_comm* comm);
int dc_comm_transport(dc_comm* comm); /* DC_COMM_RCCL or DC_COMM_PEER (negative: error) */
int dc_forward_batch(dc_comm* comm, dc_net* const* nets, int nexec, const float* const* inputs, const int (*hw)[2], int n,
                     float* const* prob, float* const* loc_pred, float* const* next_pred);
int dc_comm_item_executor(dc_comm* comm, int i); /* which executor forwarded image i of the last call (negative: error) */
int dc_comm_root_maps(dc_comm* comm, int i, const void** prob, const void** loc_pred, const void** next_pred, int dims[5]);
/* the schedule alone (host only): exec_of_item[i] = executor of item i for `n` items of the given costs on `nexec` executors */
int dc_lpt_schedule(const double* cost, int n, int nexec, int* exec_of_item);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCUT_HIP_H_ */
