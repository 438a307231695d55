/*
 * rva.h -- C ABI of librva.so: the MI355X-native detect/track hot path.
 *
 * Boundary: everything the reference's Detector / Tracker plugin API computes between "a decoded
 * frame is available" and "tracks are visible to the host" (SURVEY.md section 8).  The reference is
 * pure Python and has no FFI of its own; each entry point below names the reference function(s)
 * it replaces (paths relative to /root/reference/src/realtime_analytics/).  Plain pointers and
 * sizes only: no torch / Python types cross this boundary.
 *
 * Conventions
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  All device work is
 *     enqueued on it and nothing synchronises unless the function says so ("host-synchronous").
 *   - "device" pointers are HIP device addresses (e.g. torch.Tensor.data_ptr()).
 *   - Return value: RVA_OK or an error code; rva_last_error(ctx) gives the text.
 *   - No function allocates device memory once rva_reserve()/…_create() have sized the context, so a
 *     caller may capture a whole tick into a hipGraph.
 *   - There is NO CPU fallback: if no HIP device is usable, rva_create() fails.
 */
#ifndef RVA_H
#define RVA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVA_ABI_VERSION 1

enum rva_status {
    RVA_OK = 0,
    RVA_ERR_ARG = 1,         /* bad argument / unsupported shape */
    RVA_ERR_HIP = 2,         /* a HIP runtime call failed */
    RVA_ERR_CAPACITY = 3,    /* a capacity given at create/reserve time was exceeded */
    RVA_ERR_UNAVAILABLE = 4  /* optional component (rocDecode) not present on this machine */
};

enum rva_dtype { RVA_F16 = 0, RVA_F32 = 1, RVA_F64 = 2 /* only where an entry point says so */ };

typedef struct rva_ctx rva_ctx;
typedef struct rva_tracker rva_tracker;
typedef void *rva_stream_t;

int rva_abi_version(void);
int rva_create(int device, rva_ctx **out);
void rva_destroy(rva_ctx *ctx);
const char *rva_last_error(const rva_ctx *ctx);

/* Pre-size every scratch buffer for `batch` images x `anchors` head rows (host-synchronous; call
 * before graph capture).  Optional: buffers otherwise grow on first use. */
int rva_reserve(rva_ctx *ctx, int batch, int anchors);

/* ----------------------------------------------------------------------------------------------
 * Letterbox geometry -- detector.py:209-230 and the `meta` dict of :259-263.
 * -------------------------------------------------------------------------------------------- */
typedef struct rva_letterbox {
    int32_t src_w, src_h;   /* meta["orig_shape"] = (src_h, src_w) */
    int32_t dst_w, dst_h;   /* detector input size (input_hw) */
    int32_t new_w, new_h;   /* int(w*scale), int(h*scale) */
    int32_t pad_left, pad_top; /* meta["pad"] */
    double scale;           /* meta["scale"] (Python float) */
} rva_letterbox;

int rva_letterbox_meta(int src_w, int src_h, int dst_w, int dst_h, rva_letterbox *out);

/* ----------------------------------------------------------------------------------------------
 * K1 pre-process -- replaces _TensorRTBaseDetector._preprocess (detector.py:198-264) for a whole
 * tick of frames in ONE launch: [colour convert] -> cv2.resize(INTER_LINEAR) letterbox -> pad 114
 * -> BGR2RGB -> astype(dtype) * (1/255) -> CHW, written straight into the batch tensor
 * out[n, 3, dst_h, dst_w] (device).  n <= RVA_MAX_BATCH per call.
 *
 * nv12: y_ptrs[i] / uv_ptrs[i] are device pointers of pitch-linear NV12 surfaces (what a hardware
 *       decoder hands over instead of video_stream.py:173's BGR ndarray); BT.601 limited range,
 *       nearest chroma, converted at full resolution before the resize, as FFmpeg+OpenCV would.
 * bgr:  frames[i] is a device copy of the uint8 [src_h, src_w, 3] BGR frame of FramePacket.frame.
 * meta_out (host, may be NULL) receives the geometry shared by all n frames.
 * -------------------------------------------------------------------------------------------- */
#define RVA_MAX_BATCH 64

int rva_preprocess_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                              const int32_t *pitches, int n, int src_w, int src_h, void *out,
                              int out_dtype, int dst_w, int dst_h, rva_letterbox *meta_out,
                              rva_stream_t stream);

/* Steady-state form of rva_preprocess_nv12_batch: writes only the CONTENT region of out[n, 3, dst_h, dst_w] -- the
 * letterbox border is a constant (114/255, detector.py:233-241) and the caller guarantees it is already there (an
 * earlier rva_preprocess_nv12_batch into the same tensor with the same geometry put it there).  Same values as the full
 * call; 1080p -> 640: 2,764,800 instead of 3,840,000 bytes of traffic per frame.  Geometries without an integer
 * source/content ratio fall back to writing the whole tensor. */
int rva_preprocess_nv12_content_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                      const int32_t *pitches, int n, int src_w, int src_h, void *out,
                                      int out_dtype, int dst_w, int dst_h, rva_letterbox *meta_out,
                                      rva_stream_t stream);

int rva_preprocess_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes, int n,
                             int src_w, int src_h, void *out, int out_dtype, int dst_w, int dst_h,
                             rva_letterbox *meta_out, rva_stream_t stream);

/* Measurement aid (bench.py roofline leg): the NEXT integer-ratio K1 launch of this context is issued with
 * hipExtLaunchKernelGGL so that the two HIP events (hipEvent_t handles created with timing enabled) are written by
 * the kernel's own dispatch -- they bracket exactly that kernel.  One-shot; pass NULL, NULL to cancel. */
int rva_profile_next_preprocess(rva_ctx *ctx, void *start_event, void *stop_event);

/* Clip-frame pre-process -- replaces the per-frame body of CNNLSTMDetector._preprocess_sequence
 * (temporal_detector.py:340-359): stretch-resize to (dst_w, dst_h), BGR2RGB, /255.0, (x-mean)/std
 * (ImageNet constants), CHW; out[n, 3, dst_h, dst_w] in out_dtype (float32 math, cast last). */
int rva_preprocess_clip_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                   const int32_t *pitches, int n, int src_w, int src_h, void *out,
                                   int out_dtype, int dst_w, int dst_h, rva_stream_t stream);

int rva_preprocess_clip_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes,
                                  int n, int src_w, int src_h, void *out, int out_dtype, int dst_w,
                                  int dst_h, rva_stream_t stream);

/* Frame pre-process of the remaining classification / temporal heads: stretch-resize to
 * (dst_w, dst_h) with cv2.resize defaults, BGR2RGB, image.astype(float32) / 255.0, (image - mean) / std, CHW.
 *   norm RVA_NORM_IMAGENET_F32: float32 ImageNet constants -- CNNLSTMDetector (temporal_detector.py:350-354) and the
 *        ResNet classifiers' _preprocess (detector.py:980-1001);
 *   norm RVA_NORM_VIDEO_F32:    float32 mean 0.45 / std 0.225 on every channel -- CNN3DDetector (:570-573);
 *   norm RVA_NORM_IMAGENET_F64: ImageNet constants held in float64 arrays, so the float32 image is promoted and the
 *        subtraction / division run in float64 -- ConvGRUDetector (:741-743).  The reference then hands over float64
 *        (out_dtype RVA_F64) or casts float64 -> float16 in one rounding (RVA_F16); RVA_F32 is that value rounded once.
 *   layout RVA_LAYOUT_NCHW: out[n][3][H][W] (frames stacked on axis 0: CNN-LSTM / ConvGRU clips [T,C,H,W], ResNet);
 *   layout RVA_LAYOUT_CNHW: out[3][n][H][W] (3D-CNN clips [C,T,H,W], temporal_detector.py:583-590).
 * RVA_F64 is accepted with RVA_NORM_IMAGENET_F64 only.  rva_preprocess_clip_* = (IMAGENET_F32, NCHW). */
enum rva_frame_norm { RVA_NORM_IMAGENET_F32 = 0, RVA_NORM_VIDEO_F32 = 1, RVA_NORM_IMAGENET_F64 = 2 };
enum rva_frame_layout { RVA_LAYOUT_NCHW = 0, RVA_LAYOUT_CNHW = 1 };
int rva_preprocess_frames_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                     const int32_t *pitches, int n, int src_w, int src_h, void *out,
                                     int out_dtype, int dst_w, int dst_h, int norm, int layout,
                                     rva_stream_t stream);
int rva_preprocess_frames_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes,
                                    int n, int src_w, int src_h, void *out, int out_dtype, int dst_w,
                                    int dst_h, int norm, int layout, rva_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K2+K3 post-process -- replaces _TensorRTBaseDetector._postprocess with _xywh2xyxy, _scale_boxes,
 * _nms and module _iou (detector.py:266-375, 469-481) for a batch of head tensors.
 *
 * raw: device, [batch, d1, d2] contiguous, float16 or float32 (float16 is widened exactly).  Each
 *      image is oriented like detector.py:282-283: d1 < d2 means [channels, anchors], otherwise
 *      [anchors, channels].  channels < 5 -> every count is 0 (detector.py:285-287).
 * metas: host array of `batch` geometries (or 1 entry broadcast when n_metas == 1).
 * classes: host array (config.classes) or NULL.
 * Outputs (device), per image b in NMS order (descending score; ties: ascending anchor):
 *   out_boxes[b][i][4] xyxy frame pixels, out_scores[b][i], out_cls[b][i],
 *   out_anchor[b][i]  anchor row of the detection, out_cand[b][i] its index among the thresholded
 *   candidates (the value the reference's `keep` list holds), out_counts[b], out_ncand[b].
 *   Arrays are [batch, max_det]; out_anchor/out_cand/out_ncand may be NULL.
 * The reference has no detection cap: pass max_det = anchors for exact behaviour; if an image keeps
 * more than max_det boxes the extra ones are dropped and bit 0 of rva_post_status() is set.
 * -------------------------------------------------------------------------------------------- */
int rva_postprocess_batch(rva_ctx *ctx, const void *raw, int raw_dtype, int batch, int d1, int d2,
                          double conf_thr, double iou_thr, const int32_t *classes, int n_classes,
                          const rva_letterbox *metas, int n_metas, int max_det, float *out_boxes,
                          float *out_scores, int32_t *out_cls, int32_t *out_anchor, int32_t *out_cand,
                          int32_t *out_counts, int32_t *out_ncand, rva_stream_t stream);

/* Host-synchronous: bit 0 = max_det overflow, bit 1 = candidate capacity overflow since the last call. */
int rva_post_status(rva_ctx *ctx, rva_stream_t stream, int *flags);

/* Host-synchronous diagnostic: the number of images since the last call whose NMS scanned the kept list through the centre-bin
 * filter (K3: proper boxes -- x1 <= x2, y1 <= y2 for every candidate of the image -- and 0.15 <= iou_thr <= 0.999; other images
 * take the unfiltered scan, same result).  Nothing in the reference corresponds to it; the tests use it to know which path ran. */
int rva_post_filter_stats(rva_ctx *ctx, rva_stream_t stream, int *binned_images);

/* ----------------------------------------------------------------------------------------------
 * K4 tracker -- replaces IouTracker (tracker.py:45-147) for all streams of this process.
 *
 * One table per stream lives in HBM (structure-of-arrays, `capacity` rows, insertion order ==
 * the reference's dict order).  An update tick processes any subset of streams concurrently (one
 * wavefront per stream; the per-detection greedy loop of tracker.py:55-92 stays sequential inside
 * it), then rva_tracker_assign_ids() hands out the reference's GLOBAL ids (tracker.py:47) in the
 * canonical order "stream-minor within the tick".
 * Memory: 64 B per table row (n_streams x capacity rows) + eight pinned, device-mapped snapshot slots of the same size + the
 * float64 IoU matrix of the busy-scene form, n_streams x min(capacity, 512) x (capacity + 512) doubles capped at 512 MiB
 * (201 MB at 32 streams x capacity 1024); if that matrix cannot be allocated the tracker runs the in-loop form (same results).
 * -------------------------------------------------------------------------------------------- */
int rva_tracker_create(rva_ctx *ctx, int n_streams, int capacity, int max_age, double max_iou_distance,
                       int min_hits, rva_tracker **out);
void rva_tracker_destroy(rva_tracker *trk);

/* Detections straight from rva_postprocess_batch (device, float32, [batch, max_det] layout).
 * slot_of_stream: host int32[n_streams]; entry s = batch row holding stream s's detections this
 * tick, or -1 if stream s has no frame this tick (its table is untouched), or -2 for a skipped
 * frame (update(name, []) of pipeline.py:215: ages every track), or -3 when ANOTHER update launch of
 * the same tick owns stream s (a tick with several detectors / frame geometries issues one launch per
 * group; -3 leaves the stream's table and its new-track count alone).
 * filter_thr: filter_detections' threshold (pipeline.py:182), applied on the widened score. */
int rva_tracker_update_f32(rva_tracker *trk, const int32_t *slot_of_stream, const float *boxes,
                           const float *scores, const int32_t *cls, const int32_t *counts, int max_det,
                           double filter_thr, rva_stream_t stream);

/* The same update with the pre-detector gates of StreamWorker._process_packet decided ON THE DEVICE, so a gated
 * tick needs no host round trip (the detector has then run on every delivered frame; a gated-out frame becomes a
 * skipped frame here):
 *   motion gate (pipeline.py:156-163, utils/frame_filter.py:26-40): motion_row[s] (host int32[n_streams]) = row of
 *     the device array motion_counts (written by rva_motion_*_batch earlier on this stream) holding stream s's
 *     changed-pixel count, or -1 when the stream has no motion gate; a count of -1 (first frame) always passes, else
 *     the frame is processed iff count >= the stream's motion_min_count (rva_tracker_set_gates);
 *   adaptive-fps gate (pipeline.py:107-116, 165-170) and _adjust_adaptive_state (:242-262): frame index, idle
 *     frames and process_every live in HBM per stream and are advanced by this kernel with len(filtered) and
 *     len(tracks) of the update.
 * Streams whose slot is -2 are skipped by the caller's decision; their gate state advances like any other frame. */
int rva_tracker_update_gated_f32(rva_tracker *trk, const int32_t *slot_of_stream, const float *boxes,
                                 const float *scores, const int32_t *cls, const int32_t *counts, int max_det,
                                 double filter_thr, const int32_t *motion_counts, const int32_t *motion_row,
                                 rva_stream_t stream);

/* Gate parameters per stream (host int32[n_streams] each; host-synchronous): adaptive_enabled = StreamConfig.adaptive_fps,
 * max_process_every = max(1, int(round(target_fps / max(min_target_fps, 1)))), idle_tolerance =
 * max(int(idle_frame_tolerance), 1) (pipeline.py:107-116); motion_min_count = smallest changed-pixel count c with
 * float(c) / float(w * h) >= motion_threshold (0 when the stream has no motion gate).  reset_state != 0 also rewinds
 * every stream's frame index / idle counter / process_every to their initial values. */
int rva_tracker_set_gates(rva_tracker *trk, const int32_t *adaptive_enabled, const int32_t *max_process_every,
                          const int32_t *idle_tolerance, const int32_t *motion_min_count, int reset_state);

/* Detections supplied by the host API (IouTracker.update(stream_name, detections)): device arrays
 * double boxes[total][4], double conf[total], int64 cls[total]; stream s owns rows
 * [offsets[s], offsets[s+1]) when active[s] == 1 (host arrays, n_streams(+1) entries); active[s] == 0: no
 * update for the stream this tick; active[s] == 2: another update launch of this tick owns it (left alone). */
int rva_tracker_update_f64(rva_tracker *trk, const int32_t *active, const int32_t *offsets,
                           const double *boxes, const double *conf, const int64_t *cls,
                           rva_stream_t stream);

/* Per-stream multiplier applied (in float64) to the widened box of every detection fed through
 * rva_tracker_update_f32: _rescale_detections of pipeline.py:224-240 (1 / max(downsample_ratio, 1e-6); 1.0 = off).
 * Host-synchronous; scales: host double[n_streams]. */
int rva_tracker_set_box_scale(rva_tracker *trk, const double *scales);

/* Device address of int32 new_counts[n_streams] written by the last update (for the multi-GPU
 * all-gather of SURVEY.md 8e). */
int32_t *rva_tracker_new_counts(rva_tracker *trk);

/* Give final ids to the tracks created by the last update.  counts_all: device int32[n_global] =
 * new-track counts of ALL streams of the job in canonical order (NULL: this process owns every
 * stream; uses its own new_counts).  global_index: host int32[n_streams] position of each local
 * stream in that order (NULL: identity).  Advances the id counter by sum(counts_all). */
int rva_tracker_assign_ids(rva_tracker *trk, const int32_t *counts_all, int n_global,
                           const int32_t *global_index, rva_stream_t stream);

/* Host-synchronous read-back of one stream's table in insertion order (Track fields of
 * tracker.py:18-27).  Returns the row count in *n (rows beyond `cap` are not copied).
 * last_det[i] = index (within the stream's detection list of the latest update) of the last
 * detection that created/overwrote row i, or -1: lets the host carry the optional temporal
 * fields of tracker.py:58-67,88-90 without shipping strings to the device.  May be NULL. */
int rva_tracker_read(rva_tracker *trk, int stream_id, int cap, int64_t *ids, int32_t *cls, int32_t *age,
                     int32_t *hits, double *conf, double *boxes, int32_t *last_det, int32_t *n,
                     rva_stream_t stream);

/* Host-synchronous read-back of every stream at once: arrays are [n_streams, capacity(,4)],
 * counts[n_streams]; any pointer may be NULL.  Only rows [0, counts[s]) of stream s are written (one device
 * kernel gathers the live rows into a pinned slot); the rest of the caller's arrays is left untouched. */
int rva_tracker_read_all(rva_tracker *trk, int64_t *ids, int32_t *cls, int32_t *age, int32_t *hits,
                         double *conf, double *boxes, int32_t *last_det, int32_t *counts,
                         rva_stream_t stream);

#define RVA_SNAPSHOT_SLOTS 8   /* ticks whose tables may be on their way to the host at once */
/* Pipelined read-back: snapshot_async enqueues device-to-host copies of every table into pinned
 * staging slot 0 .. RVA_SNAPSHOT_SLOTS - 1 on `stream` (no host wait); snapshot_fetch (wait != 0) waits for that slot's
 * copies only and hands the arrays out ([n_streams, capacity(,4)] like read_all).  Lets tick k+1 be
 * enqueued before tick k's tracks are consumed.  When snapshot_async is captured into a hipGraph no
 * event is recorded: synchronise on the graph launch yourself and call snapshot_fetch with wait = 0. */
int rva_tracker_snapshot_async(rva_tracker *trk, int slot, rva_stream_t stream);
int rva_tracker_snapshot_fetch(rva_tracker *trk, int slot, int wait, int64_t *ids, int32_t *cls, int32_t *age,
                               int32_t *hits, double *conf, double *boxes, int32_t *last_det,
                               int32_t *counts);

/* What rides along in a snapshot slot besides the tables (no wait: call after snapshot_fetch / the graph launch has
 * completed): emitted[s] = len(filtered) of the stream's last update (pipeline.py:187), processed[s] = 1 processed /
 * 0 skipped frame / -1 no frame, *flags = tracker flags (bit 0: a table overflowed `capacity`, detections dropped)
 * | post-process flags << 8 (bit 8: more survivors than max_det, bit 9: more thresholded anchors than the sort holds).
 * The reference is unbounded in all three places, so a caller must treat a non-zero value as an error. */
int rva_tracker_snapshot_status(rva_tracker *trk, int slot, int32_t *emitted, int32_t *processed, int32_t *flags);

/* Host-synchronous: next id the counter will hand out; flags bit 0 = a table overflowed `capacity`. */
int rva_tracker_state(rva_tracker *trk, int64_t *next_id, int *flags, rva_stream_t stream);
int rva_tracker_set_next_id(rva_tracker *trk, int64_t next_id, rva_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Fused detector primitives (NHWC float16) -- the arithmetic ONNX Runtime performs inside
 * `session.run` (detector.py:608) for a YOLOv8 graph, one pass per layer instead of MIOpen's
 * conv + bias + SiLU + concat chains.  Activations are [batch*H*W, channels] with a row stride
 * (ld, in elements) so layers read / write channel slices of shared concat buffers.
 *
 * rva_conv2d_nhwc_f16: out = [SiLU](conv(in, weights) + bias) [+ residual]; ksize 1 or 3 (pad ksize/2),
 *   stride 1 or 2, Cin % 8 == 0, Cout % 8 == 0.  weights: float16 [rva_conv_cout_pad(Cout)][ksize*ksize][CinPad]
 *   with CinPad = Cin rounded up to 32 (padding rows / channels zero), bias: float32 [rva_conv_cout_pad(Cout)].
 *   MFMA implicit GEMM, fp32 accumulate.
 * rva_stem_conv_f16: first layer, 3x3 stride 2 on the PLANAR tensor K1 writes ([batch,3,H,W]);
 *   weights float16 [64][32]: row = output channel (zero beyond Cout); column order with j = c*3 + ky:
 *   k = 2*j + kx for kx in {0,1}, k = 18 + j for kx = 2, zero for k >= 27; bias float32 [64]; Cout <= 64; NHWC out.
 * rva_maxpool5_nhwc_f16 / rva_upsample2x_nhwc_f16: SPPF pooling, FPN nearest upsample, on channel slices.
 * rva_yolo_head_f16: DFL expectation + dist2bbox + sigmoid of one pyramid level into
 *   out[batch, 4+nc, anchors_total] at anchor_offset -- the `[B,84,8400]` tensor rva_postprocess_batch reads.
 * -------------------------------------------------------------------------------------------- */
int rva_conv_cout_pad(int Cout);
/* number of explicit kernel variants rva_conv2d_nhwc_f16_v accepts (1..N); the plan's autotuner iterates over them */
int rva_conv_num_variants(void);
int rva_conv2d_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias,
                        void *out, int ldo, const void *residual, int ldr, int batch, int H, int W, int Cin,
                        int Cout, int ksize, int stride, int act, rva_stream_t stream);
/* Same with an explicit kernel variant, so that a plan can time the applicable variants per layer once and keep the
 * fastest (0 = heuristic; 1..rva_conv_num_variants() = one kernel family and tile each, listed beside the dispatch in
 * csrc/rva_conv.hip: register-staged gather / resident-chunk / row-reuse kernels, and the LDS-DMA large-tile kernels
 * -- row reuse for 3x3 stride 1, gather with 64- or 32-channel K-steps for 1x1 and strided 3x3).  RVA_ERR_ARG if a
 * variant does not apply to the shape. */
int rva_conv2d_nhwc_f16_v(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias,
                          void *out, int ldo, const void *residual, int ldr, int batch, int H, int W, int Cin,
                          int Cout, int ksize, int stride, int act, int variant, rva_stream_t stream);
/* 1x1 convolution whose input is torch.cat([nearest-2x-upsample(low), skip], channel) -- the FPN pattern -- without
 * materialising either the upsampled tensor or the concatenation: low is [batch, H/2, W/2, c_low], skip is
 * [batch, H, W, c_skip] (row strides ld_low / ld_skip), weights [rva_conv_cout_pad(Cout)][1][c_low + c_skip].
 * c_low % 64 == c_skip % 64 == 0, H and W even.  variant: 0 or one of the LDS-DMA gather variants 33..39. */
int rva_conv1x1_upcat_f16(rva_ctx *ctx, const void *low, int ld_low, int c_low, const void *skip, int ld_skip,
                          int c_skip, const void *weights, const float *bias, void *out, int ldo, int batch,
                          int H, int W, int Cout, int act, int variant, rva_stream_t stream);
/* Last 1x1 convolution of a detect-head branch fused with the head decode: mode 1 = box branch (Cout = 64 DFL logits ->
 * expectation + dist2bbox, rows 0..3 of out), mode 2 = class branch (Cout = nc logits -> sigmoid, rows 4..4+nc), written
 * straight into out[batch, 4+nc, anchors_total] at anchor_offset -- the logits never reach HBM.  Same arithmetic and
 * rounding points as rva_conv2d_nhwc_f16 followed by rva_yolo_head_f16 (bit-identical).  Cin % 64 == 0; variant 0 or 33..39. */
int rva_conv1x1_head_f16(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias, int batch,
                         int H, int W, int Cin, int Cout, int mode, void *out, int nc, int anchors_total,
                         int anchor_offset, float stride_px, int variant, rva_stream_t stream);
int rva_stem_conv_f16(rva_ctx *ctx, const void *in_planar, const void *weights, const float *bias,
                      void *out, int ldo, int batch, int H, int W, int Cout, rva_stream_t stream);

/* rva_c2f_pair32_f16: one C2f bottleneck with 32 channels and a shortcut -- y = x + SiLU(conv3x3(SiLU(conv3x3(x)))) -- in one launch;
 * replaces two session-internal Conv nodes + Add of the reference's ONNX graph (`session.run`, detector.py:597-609) and two
 * rva_conv2d_nhwc_f16 launches of this library, bit-identical to them (same operations, same order); the 32-channel intermediate
 * stays in LDS.  in / out: NHWC fp16 slices with row strides ldi / ldo (halfs, multiples of 8), 32 channels each, not overlapping;
 * w1, w2: [64][9][32] fp16 as rva_conv2d_nhwc_f16 takes them (rows >= 32 unused), b1, b2: [64] fp32. */
int rva_c2f_pair32_f16(rva_ctx *ctx, const void *in, int ldi, const void *w1, const float *b1, const void *w2, const float *b2, void *out,
                       int ldo, int batch, int H, int W, rva_stream_t stream);

/* rva_stem2_f16: the stem and the first downsampling convolution of YOLOv8s (3 -> 32 -> 64 channels, both 3x3 stride 2
 * pad 1 with bias + SiLU; ultralytics model.0 and model.1, reference call site detector.py:597-609 via the exported graph)
 * in ONE launch: the half-resolution 32-channel tensor stays in LDS.  in_planar, w1 ([64][32] fp16, column order of
 * rva_stem_conv_f16) and b1 ([64] fp32) as for rva_stem_conv_f16 with Cout = 32; w2 / b2 in the layout of
 * rva_conv2d_nhwc_f16 for Cin = 32, Cout = 64 ([64][9][32] fp16, [64] fp32); out: NHWC fp16 [batch, Ho, Wo, 64] with row
 * stride ldo (>= 64, multiple of 8), Ho = ((H-1)/2)/2 + 1.  W % 8 == 0.  Results equal stem -> conv up to fp32 summation
 * order inside the stem's 27-tap dot product. */
int rva_stem2_f16(rva_ctx *ctx, const void *in_planar, const void *w1, const float *b1, const void *w2, const float *b2,
                  void *out, int ldo, int batch, int H, int W, rva_stream_t stream);

/* SPPF's three chained 5x5/1 max pools in one launch: out1 = pool5(in), out2 = pool5(out1) = pool9(in),
 * out3 = pool5(out2) = pool13(in) (stride 1, -inf padding), all three with row stride ldo.  H*W*64 bytes must fit LDS
 * (H*W <= 2400); larger maps use rva_maxpool5_nhwc_f16 three times. */
int rva_sppf_pool3_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, void *out1, void *out2, void *out3, int ldo,
                            int batch, int H, int W, int C, rva_stream_t stream);
int rva_maxpool5_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, void *out, int ldo, int batch, int H,
                          int W, int C, rva_stream_t stream);
int rva_upsample2x_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, void *out, int ldo, int batch, int H,
                            int W, int C, rva_stream_t stream);
int rva_yolo_head_f16(rva_ctx *ctx, const void *box_logits, int ldb, const void *cls_logits, int ldc,
                      void *out, int batch, int h, int w, int nc, int anchors_total, int anchor_offset,
                      float stride, rva_stream_t stream);
/* The three pyramid levels (strides 8 / 16 / 32) in one launch; arrays of 3, anchor offsets follow from h*w. */
int rva_yolo_head3_f16(rva_ctx *ctx, const void *const *box_logits, const int32_t *ldb, const void *const *cls_logits,
                       const int32_t *ldc, void *out, int batch, const int32_t *h, const int32_t *w, int nc,
                       int anchors_total, const float *strides, rva_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * The fused YOLOv8 detector as ONE object (round 4) -- replaces `self.session.run` of the reference's ONNX Runtime
 * backend (/root/reference/src/realtime_analytics/detector.py:597-609; the network it runs is the exported ultralytics graph,
 * detector.py:575-586) for `half: true`.
 *
 * rva_yolov8_plan_create: `convs` = the network's n_convs convolutions with BatchNorm folded, in MODULE ORDER, each in the
 *   checkpoint's own layout (fp32 host arrays, weight [cout][cin][k][k], bias [cout] or NULL).  Module order: b0, b1, C2f(b2), b3,
 *   C2f(b4), b5, C2f(b6), b7, C2f(b8), SPPF(b9) = cv1 cv2, C2f(h12), C2f(h15), h16, C2f(h18), h19, C2f(h21), detect.box[0..2][0..2],
 *   detect.cls[0..2][0..2]; C2f(x) = cv1, cv2, then cv1 cv2 of every bottleneck (ultralytics model.0 .. model.22).  The call
 *   checks every shape against the descriptor (RVA_ERR_ARG names the first convolution that does not fit), packs the weights
 *   for the kernels above, allocates all activation buffers in HBM and lays down the static list of launches.
 * rva_yolov8_plan_run: input = fp16 planar [batch,3,height,width] (what rva_preprocess_* writes), output = fp16
 *   [batch, 4+nc, anchors] (what rva_postprocess_batch reads; anchors = H/8*W/8 + H/16*W/16 + H/32*W/32); every launch of the
 *   forward pass goes to `stream`, no host synchronisation, no allocation: capturable.  _run_lanes additionally forks the
 *   stride-8 / stride-16 detect branches onto two side streams (events inside the plan) and joins them at the end: +6 % for one
 *   pass at a time; with several passes in flight use _run.  _run_range(first, last) replays steps [first, last) (a caller that
 *   wants an event at `quiet_step`, where the pass enters its 20x20 layers).
 * Kernel selection: every convolution step ("tunable") carries a kernel variant of rva_conv2d_nhwc_f16_v (0 = heuristic).
 *   A tuner times rva_yolov8_plan_launch_tunable(index, variant) on the plan's own buffers (RVA_ERR_ARG = variant does not
 *   apply to that layer) and fixes its choice with _set_variant; _tunable_desc gives "Cin->Cout kKsS HxW" (the key of a
 *   persisted selection).  Results do not depend on the variant beyond fp32 summation order.
 * -------------------------------------------------------------------------------------------- */
#define RVA_PLAN_NO_STEM2 1   /* rva_yolov8_desc.flags: stem and first downsampling convolution as two launches (A/B switch) */
#define RVA_PLAN_NO_CIN_PAD 2 /* ... convolutions with Cin % 32 != 0 keep their Cin (default: declared rounded up to 32, zero weights) */
#define RVA_PLAN_NO_PAIR32 4  /* ... the 32-channel C2f bottlenecks (YOLOv8s at 160 x 160) as two convolution launches instead of rva_c2f_pair32_f16 */
typedef struct rva_yolov8_plan rva_yolov8_plan;
typedef struct rva_yolov8_desc {
    int32_t batch, height, width;     /* input tensor; height and width multiples of 32 */
    int32_t widths[5];                /* c1..c5 (YOLOv8s: 32 64 128 256 512) */
    int32_t depth_backbone[4];        /* bottlenecks of the C2f blocks b2, b4, b6, b8 (YOLOv8s: 1 2 2 1) */
    int32_t depth_head;               /* bottlenecks of h12, h15, h18, h21 (YOLOv8s: 1) */
    int32_t nc, reg_max;              /* classes (multiple of 8), DFL bins (16) */
    int32_t n_convs;                  /* length of `convs` */
    int32_t flags;
} rva_yolov8_desc;
typedef struct rva_conv_weights {
    const float *weight;              /* [cout][cin][k][k] */
    const float *bias;                /* [cout] or NULL */
    int32_t cout, cin, k, stride;
} rva_conv_weights;
int rva_yolov8_plan_create(rva_ctx *ctx, const rva_yolov8_desc *desc, const rva_conv_weights *convs, rva_yolov8_plan **out);
void rva_yolov8_plan_destroy(rva_yolov8_plan *plan);
int rva_yolov8_plan_info(const rva_yolov8_plan *plan, int32_t *anchors, int32_t *out_rows, int32_t *n_steps, int32_t *n_tunable,
                         int32_t *quiet_step);
int rva_yolov8_plan_run(rva_yolov8_plan *plan, const void *input, void *output, rva_stream_t stream);
int rva_yolov8_plan_run_lanes(rva_yolov8_plan *plan, const void *input, void *output, rva_stream_t stream, rva_stream_t side1,
                              rva_stream_t side2);
int rva_yolov8_plan_run_range(rva_yolov8_plan *plan, const void *input, void *output, int first, int last, rva_stream_t stream);
int rva_yolov8_plan_tunable_desc(const rva_yolov8_plan *plan, int index, char *buf, int len);
int rva_yolov8_plan_launch_tunable(rva_yolov8_plan *plan, int index, int variant, void *output, rva_stream_t stream);
int rva_yolov8_plan_set_variant(rva_yolov8_plan *plan, int index, int variant);
int rva_yolov8_plan_get_variant(const rva_yolov8_plan *plan, int index);

/* ----------------------------------------------------------------------------------------------
 * K5 motion gate -- replaces MotionFilter.should_process (utils/frame_filter.py:26-40)
 * for a tick of NV12 surfaces: gray -> 5x5 Gaussian -> |diff| against prev_blur[i] -> counts[i] = number of
 * pixels with diff > 25 (device int32[n]; -1 where prev_blur[i] is NULL = first frame of that stream).
 * blur_out[i] (device uint8 [h][w]) receives the new blurred frame; pass it as prev_blur next tick.
 * The caller compares counts[i] / (w*h) >= motion_threshold (frame_filter.py:38-40).
 * -------------------------------------------------------------------------------------------- */
int rva_motion_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                          const int32_t *pitches, const void *const *prev_blur, void *const *blur_out, int n,
                          int w, int h, int32_t *counts, rva_stream_t stream);
/* Same with per-stream ROI masks (uint8 [h][w], 0 = outside; entries may be NULL) applied first, and for uint8 BGR
 * device images (the downsampled frames): the gate sees what the reference's frame_for_detection holds. */
int rva_motion_nv12_masked_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                 const int32_t *pitches, const void *const *masks, const void *const *prev_blur,
                                 void *const *blur_out, int n, int w, int h, int32_t *counts, rva_stream_t stream);
int rva_motion_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes,
                         const void *const *prev_blur, void *const *blur_out, int n, int w, int h, int32_t *counts,
                         rva_stream_t stream);

/* ROI + downsample in front of the detector (utils/frame_filter.py:43-57, pipeline.py:149-154):
 * rva_preprocess_nv12_masked_batch = rva_preprocess_nv12_batch with apply_roi (pixels whose mask byte is 0
 * become BGR 0,0,0 before the resize); rva_resize_nv12_to_bgr_batch = apply_roi + downsample: cv2.resize
 * INTER_LINEAR to (dst_w, dst_h) as uint8 BGR images out_bgr[n][dst_h][dst_w][3] (feed rva_preprocess_bgr_batch). */
int rva_preprocess_nv12_masked_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                     const int32_t *pitches, const void *const *masks, int n, int src_w, int src_h,
                                     void *out, int out_dtype, int dst_w, int dst_h, rva_letterbox *meta_out,
                                     rva_stream_t stream);
int rva_resize_nv12_to_bgr_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                 const int32_t *pitches, const void *const *masks, int n, int src_w, int src_h,
                                 void *out_bgr, int dst_w, int dst_h, rva_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K6 annotated preview -- replaces the pixel work of KafkaSink._render_frame (sinks/kafka_sink.py:200-294) and of
 * StreamWorker._maybe_save_snapshot (pipeline.py:264-290): uint8 BGR image out_bgr[dst_h][dst_w][3] (device) =
 * the NV12 surface converted (BT.601 as in K1), box-averaged down by the integer `ratio` (1 = same size; 0 = out_bgr
 * already holds the base image, draw only), then `n_rects` filled rectangles (device int32[n][4] x0,y0,x1,y1
 * inclusive, device uint8[n][4] b,g,r,-) in painter's order and `n_glyphs` characters of a built-in 5x7 font (device
 * int32[m][3] x, y, code; digits, 'I', 'D', space) in white at `glyph_scale`.  WHAT is drawn is decided by the host
 * (preview.py, pinned against a call-level recording of the reference); how OpenCV rasterises lines and glyphs is unpinned.
 * -------------------------------------------------------------------------------------------- */
int rva_preview_nv12(rva_ctx *ctx, const void *y, const void *uv, int pitch, int src_w, int src_h, int ratio,
                     void *out_bgr, int dst_w, int dst_h, const int32_t *rects, const uint8_t *colors, int n_rects,
                     const int32_t *glyphs, int n_glyphs, int glyph_scale, rva_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * D1 decode -- stands where VideoStream.open()/frames() sit on cv2.VideoCapture(url, CAP_FFMPEG) (video_stream.py:76,
 * 173): an H.264 / H.265 Annex-B elementary stream goes to the VCN decoder through rocDecode and comes back as NV12
 * surfaces in HBM (device pointers + pitch, the input form of rva_preprocess_nv12_batch); no frame visits the host.
 * librocdecode is looked up with dlopen at run time: rva_decode_available() / rva_decoder_create() report
 * RVA_ERR_UNAVAILABLE on a machine that does not have it (the demuxer and the retry / back-off / reconnect policy of
 * video_stream.py:155-243 live on the host side: mp4.py, video_stream.py of this package).
 * -------------------------------------------------------------------------------------------- */
typedef struct rva_decoder rva_decoder;
enum rva_codec { RVA_CODEC_H264 = 0, RVA_CODEC_HEVC = 1 };

int rva_decode_available(char *detail, int detail_len);

/* One decode session (parser + decoder) on ctx's device.  num_surfaces: decode surfaces to ask for (the stream's own
 * minimum wins when larger; <= 0: 8).  8-bit 4:2:0 streams only (what maps onto NV12). */
int rva_decoder_create(rva_ctx *ctx, int codec, int num_surfaces, rva_decoder **out);
void rva_decoder_destroy(rva_decoder *dec);

/* Feed ONE access unit (host memory, Annex-B start codes, parameter sets in band in front of sync samples); pts in
 * 10 MHz units.  end_of_stream != 0 flushes the decoder (data may be NULL / size 0).  Decoding is submitted from inside
 * this call (parser callbacks); decoded pictures queue up for rva_decoder_next_frame. */
int rva_decoder_feed(rva_decoder *dec, const uint8_t *data, int size, int64_t pts, int end_of_stream);

/* Next picture in display order: *pic_index >= 0 and device pointers *y / *uv (interleaved) with a common *pitch, the
 * display size *width x *height (cropped to whole chroma pairs), or *pic_index == -1 when nothing is displayable yet
 * (feed more data).  Waits for that picture's decode to finish.  *pic_index is a ticket (decoder generation << 10 | the
 * library's picture index): the surface stays valid until rva_decoder_release(dec, that ticket) hands it back -- also
 * across a change of picture size mid-stream, where the previous decoder lives on until its last picture is released.
 * The library probed first is the one RVA_ROCDECODE_LIB names (then librocdecode.so on the default search path). */
int rva_decoder_next_frame(rva_decoder *dec, void **y, void **uv, int32_t *pitch, int32_t *width, int32_t *height,
                           int64_t *pts, int32_t *pic_index);
int rva_decoder_release(rva_decoder *dec, int pic_index);

/* ----------------------------------------------------------------------------------------------
 * K7 device JPEG encoder (round 4) -- replaces cv2.imencode('.jpg', frame, [IMWRITE_JPEG_QUALITY, q, ...]) of
 * KafkaSink._render_frame (/root/reference/src/realtime_analytics/sinks/kafka_sink.py:260-284) and of
 * StreamWorker._maybe_save_snapshot (pipeline.py:264-290): a uint8 BGR image in HBM ([height] rows of `pitch` bytes, 3 bytes per
 * pixel: what rva_preview_nv12 writes) -> a baseline JFIF stream in `out` (device), its length in *out_size (device or mapped
 * host int32), both written asynchronously on `stream`.  libjpeg's arithmetic step by step (colour conversion, 4:2:0 downsampling,
 * edge rules, islow DCT, quantisation tables of `quality`), Annex-K Huffman tables, one restart interval per MCU row: a decoder
 * reconstructs exactly the picture it reconstructs from libjpeg's own file of the same quality (the reference asks for the
 * progressive, Huffman-optimised form of the same coefficients).  rva_jpeg_max_bytes: a capacity that fits photographic content
 * (128 B per 8x8 block); rva_jpeg_status (host-synchronous): bit 0 = the stream did not fit since the last call.
 * -------------------------------------------------------------------------------------------- */
int rva_jpeg_max_bytes(int width, int height);
int rva_jpeg_encode_bgr(rva_ctx *ctx, const void *bgr, int pitch, int width, int height, int quality, void *out,
                        int out_capacity, int32_t *out_size, rva_stream_t stream);
int rva_jpeg_status(rva_ctx *ctx, rva_stream_t stream, int *flags);

#ifdef __cplusplus
}
#endif
#endif /* RVA_H */
