/*
 * vortex_hip.h — C ABI of libvortex_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for Vortex's detection pre/post-process path.  Plain
 * pointers and sizes only; every pointer named `d_*` is DEVICE memory, every
 * other pointer is HOST memory read before the call returns.  `stream` is a
 * hipStream_t passed as void*; kernels are launched on it with the CURRENT HIP
 * device, which must be the device the pointers live on.  No entry point
 * allocates, frees or synchronises: the caller owns outputs and workspace (sizes
 * from the *_workspace_bytes queries) and reads counts / status words back itself
 * (one exception, a set-up probe outside the data path: vx_streams_overlap waits
 * for the two streams it is given).
 * Process-global state the library keeps (all per device, set up on first use,
 * thread-safe): the "large dynamic LDS" attribute of four kernels, one hipEvent
 * for vx_yolo_fused_postprocess_overlapped, the address of the static scratch of
 * vx_batched_nms_noalloc, and the opt-in profiling hooks.
 *
 * All reference citations are relative to
 *   /root/reference/src/development/vortex/development/   (unless they start with src/runtime)
 *
 * Return value: VX_OK or a negative VX_ERR_* code (argument / limit checks are
 * done on the host before any launch).  Data-dependent conditions that the
 * reference raises on (NaN in predictions) are reported through the device
 * `d_status` word, see VX_STATUS_*.
 */
#ifndef VORTEX_HIP_H
#define VORTEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK                 0
#define VX_ERR_INVALID_ARG   (-1)   /* null pointer, non-positive dimension, bad enum   */
#define VX_ERR_UNSUPPORTED   (-2)   /* a documented limit exceeded (see each function)  */
#define VX_ERR_WORKSPACE     (-3)   /* workspace smaller than *_workspace_bytes()       */
#define VX_ERR_LAUNCH        (-4)   /* hipGetLastError() != hipSuccess after a launch    */
#define VX_ERR_ALIGNMENT     (-5)   /* a device pointer is not aligned as documented     */
#define VX_ERR_SCRATCH_LIMIT (-6)   /* vx_batched_nms_noalloc: the call needs more than the static scratch holds */

/* bits of the device status word (int32, caller zeroes it, kernels OR into it) */
#define VX_STATUS_NAN         1     /* predictions contain NaN: postprocess/yolov3.py:35-36 -> ValueError */
#define VX_STATUS_BAD_CLASS   2     /* class index outside [0, n_classes) (n_classes <= 0: [0, VX_MAX_CLASSES)) */
#define VX_STATUS_RANGE       4     /* NMS, more than one class: some box coordinate lies outside (-1024, 1024) */
#define VX_STATUS_RANGE2      16    /* NMS, more than one class: some box coordinate lies outside [0, 2048]
                                       (VX_MAX_COORDINATE).  The per-class sweep equals the reference's single pass over
                                       2049*class-shifted boxes while EVERY coordinate of an image batch stays in ONE of
                                       the two ranges (normalised boxes / pixel boxes up to max_coordinate): a caller
                                       refuses the result only when BOTH bits are set */
#define VX_STATUS_DEFERRED    8     /* vx_batched_nms_small: some image had more candidates than the single-launch
                                       kernel sorts (16384); repeat the call with vx_batched_nms */

#define VX_MAX_CLASSES        4096  /* sort key holds the class in 12 bits               */
#define VX_MAX_CANDIDATES     1048575 /* per image; sort key holds the index in 20 bits  */
#define VX_MAX_COORDINATE     2048  /* postprocess/utils/nms.py:94 */

#define VX_MAX_FORMAT_COLS    64    /* columns of one formatted result row                */

/* use_threshold argument of the decoders: bit 0 = apply the score threshold.  vx_yolo_fused_decode also takes
 * bit 1 = "threshold first": read the objectness plane first and the box/class planes only where a row
 * survives.  Same outputs bit for bit; HBM traffic falls from (5+C) planes to 1 plane + the sectors of the
 * survivors.  Deviation (why it is opt-in): the reference's NaN check (postprocess/yolov3.py:35-36) then
 * covers every objectness value but the box/class values of surviving lanes only. */
#define VX_DECODE_THRESHOLD       1
#define VX_DECODE_THRESHOLD_FIRST 2
/* vx_yolo_fused_postprocess only: run the NMS through vx_batched_nms_small (see there: VX_STATUS_DEFERRED) */
#define VX_POSTPROCESS_NMS_OPTIMISTIC 4

#define VX_RESIZE_STRETCH     0     /* src/runtime/vortex/runtime/basic_runtime.py:66-68 */
#define VX_RESIZE_LETTERBOX   1     /* build-side extension (basic_runtime.py:70-73 is a TODO) */

int         vx_version(void);
const char *vx_error_string(int code);

/* ---------------------------------------------------------------- K1 pre-process
 * Replaces Normalize.forward / FlipNormalize.forward
 *   (networks/modules/preprocess/normalizer.py:74-77, :91-95; to_tensor :13-31,
 *    normalize :34-60, flip :5-11)
 * and, when (Hs,Ws) != (Hd,Wd), BaseRuntime.resize_stretch in front of it
 *   (src/runtime/vortex/runtime/basic_runtime.py:66-68, :75-90).
 * d_src u8[B,Hs,Ws,3] (NHWC) -> d_dst f32[B,3,Hd,Wd] (NCHW, contiguous):
 *   dst[b,c,y,x] = ((float(px) / scaler) - mean[c]) / std[c], px = src[b,y,x, flip ? 2-c : c]
 * bit-exact with the reference for same-size input.  Resize: u8 bilinear, half-pixel
 * centres, 11-bit fixed-point weights (OpenCV INTER_LINEAR scheme; parity unpinned).
 * d_dst must be 16-byte aligned, d_src 4-byte aligned. */
int vx_preprocess_u8hwc_f32nchw(const uint8_t *d_src, float *d_dst,
                                int B, int Hs, int Ws, int Hd, int Wd,
                                const float mean[3], const float std[3], int scaler,
                                int flip, int resize_mode, int pad_value, void *stream);

/* Resize only: BaseRuntime.resize_stretch / resize_batch (src/runtime/vortex/runtime/basic_runtime.py:66-68,
 * :75-90; cv2.resize default INTER_LINEAR).  d_src u8[B,Hs,Ws,3] -> d_dst u8[B,Hd,Wd,3], same fixed-point
 * scheme as above (parity unpinned: cv2 is not available to pin it). */
int vx_resize_u8hwc(const uint8_t *d_src, uint8_t *d_dst, int B, int Hs, int Ws, int Hd, int Wd,
                    int resize_mode, int pad_value, void *stream);

/* ---------------------------------------------------------------- K2 YOLOv3 grid/anchor decode
 * Replaces YoloV3Layer.forward, eval branch (networks/modules/heads/detection/yolov3.py:263-304;
 * buffers :202-247) for ONE scale, writing straight into its slice of the concatenated
 * tensor of YoloV3.forward (networks/models/detection/yolov3.py:72-76).
 * d_in f32[B, 3*(5+C), G, G]  ->  d_out rows [row_offset, row_offset + 3*G*G) of f32[B, R_total, 5+C]
 *   anchor_wh[a] = anchors[mask[a]] / stride (grid units, as the reference's anchor_wh buffer)
 * Transcendentals are tolerance-class (1e-5 relative). Limit: 5+C <= 255. d_out 16-byte aligned. */
int vx_yolo_head_decode(const float *d_in, float *d_out, int B, int G, int C,
                        const float anchor_wh[6], float stride,
                        int64_t out_batch_stride /* floats between images in d_out */,
                        int64_t row_offset, void *stream);
/* The three scales in ONE launch: YoloV3Layer x3 (heads/detection/yolov3.py:263-304) + the concat of YoloV3.forward
 * (networks/models/detection/yolov3.py:72-76).  d_in0/1/2 = raw heads coarse -> fine, f32[B, 3*(5+C), G, G] with
 * G = img_size/32, /16, /8; d_out f32[B, R, 5+C], R = 3*sum(G*G), rows in the reference's order (coarse scale first).
 * anchors_px = the 9 (w,h) anchors in pixels; scale k uses masks (6,7,8), (3,4,5), (0,1,2) (:319-343) and
 * anchors / stride in fp32 (:244).  Same arithmetic, same bits as three vx_yolo_head_decode calls; the small scales
 * fill the tail of the large one instead of paying a launch ramp each.  d_out 16-byte aligned, 5+C <= 255. */
int vx_yolo_head_decode3(const float *d_in0, const float *d_in1, const float *d_in2, float *d_out, int B,
                         int img_size, int C, const float anchors_px[18], void *stream);

/* ---------------------------------------------------------------- K3 YOLOv3 decoder
 * Replaces YoloV3Decoder.forward (networks/modules/postprocess/yolov3.py:29-68, yolo2xywh :9-17)
 * for all B images in one call (the reference loops per image, base_postprocess.py:131-138).
 * d_pred f32[B,R,5+C] (16-byte aligned).  Per image i, K_i rows with obj > score_threshold
 * (all rows if use_threshold == 0), ascending row order:
 *   d_boxes f32[B,R,4] xyxy/img_size, d_scores f32[B,R] = obj*class_conf,
 *   d_cls i64[B,R] (first max wins), d_det f32[B,R,6] = box|class_conf|float(cls),
 *   d_counts i32[B] = K_i.  Rows beyond K_i are left untouched.
 * NaN anywhere in d_pred sets VX_STATUS_NAN in *d_status.  Bit-exact (no transcendentals). */
size_t vx_yolo_decode_workspace_bytes(int B, int R);
int vx_yolo_decode(const float *d_pred, int B, int R, int C, float img_size,
                   float score_threshold, int use_threshold,
                   float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                   int32_t *d_counts, int32_t *d_status,
                   void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- K2+K3 fused
 * YoloV3Layer (3 scales) + concat + YoloV3Decoder without materialising f32[B,R,5+C].
 * d_heads[s] f32[B,3*(5+C),G_s,G_s] for G_s = S/32, S/16, S/8 (coarse -> fine, the h1,h2,h3
 * order of heads/detection/yolov3.py:369-376); anchors_px[9][2] = the 9 anchors in pixels
 * (models/detection/yolov3.py:26-29), masks (6,7,8),(3,4,5),(0,1,2) applied inside.
 * Outputs as vx_yolo_decode with R = 3*sum(G_s^2).  Results are bit-identical to
 * vx_yolo_head_decode x3 followed by vx_yolo_decode. */
size_t vx_yolo_fused_workspace_bytes(int B, int S);
int vx_yolo_fused_decode(const float *d_head0, const float *d_head1, const float *d_head2,
                         int B, int S, int C, const float anchors_px[18],
                         float score_threshold, int use_threshold,
                         float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                         int32_t *d_counts, int32_t *d_status,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- K4 prior-box decode (+landmarks)
 * FPN-SSD / RetinaFace decoder.  The reference's source for it is absent from the snapshot
 * (evidence: scripts/export_check/README.md:54-63,:107,:113); build-authored spec, see
 * oracle/vortex_oracle.py:prior_box_decode.  d_in f32[B,P,4+2L+NC] = loc|landmarks|logits
 * (logit 0 = background), d_priors f32[P,4] normalised cx,cy,w,h.
 * Outputs as vx_yolo_decode with row width D = 6+2L: d_det f32[B,P,D].
 * One kernel (round 3): the survivors of a group of rows are listed on chip and written, in row order, straight into
 * the four output tensors once the groups in front have published their counts; the workspace only holds one ticket
 * counter per image and one hand-off word per group (a few KB; zeroed by the call itself on `stream`).
 * Limits: 4+2L+NC <= 63, NC >= 2, P <= VX_MAX_CANDIDATES. */
size_t vx_prior_decode_workspace_bytes(int B, int P, int n_landmarks);
int vx_prior_decode(const float *d_in, const float *d_priors, int B, int P,
                    int n_landmarks, int n_logits, float variance0, float variance1,
                    float score_threshold, int use_threshold,
                    float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                    int32_t *d_counts, int32_t *d_status,
                    void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- K5-K7 class-aware batched NMS
 * Replaces batched_nms + BatchedNMS.forward (networks/modules/postprocess/utils/nms.py:57-98,
 * :105-123) and the per-image loop of BatchedNMSPostProcess.forward (base_postprocess.py:131-138).
 * Per image i the first n_i = d_counts[i] (or `capacity` when d_counts == NULL) entries of
 *   d_boxes f32[B,capacity,4], d_scores f32[B,capacity], d_cls i64[B,capacity], d_det f32[B,capacity,D]
 * are candidates.  Greedy NMS in descending score order (ties: ascending index) on
 * fl32(box + fl32(2049*cls)) exactly as the reference's offset trick; suppress when
 * inter/(area_i+area_j-inter) > iou_threshold.  Outputs, per image:
 *   d_keep_idx i64[B,out_capacity]   indices into the candidate list, score-descending
 *   d_out_det  f32[B,out_capacity,D] = d_det rows gathered by keep (index_select, nms.py:122)
 *   d_keep_counts i32[B]
 * top_k > 0 truncates each keep list to its first top_k entries (build-side extension);
 * out_capacity >= min(capacity, top_k>0 ? top_k : capacity).
 * Limits: capacity <= VX_MAX_CANDIDATES, 0 <= cls < n_classes <= VX_MAX_CLASSES (else VX_STATUS_BAD_CLASS),
 * every coordinate in (-1024, 1024) or every coordinate in [0, 2048] so that classes cannot overlap after the
 * offset (checked on the device -> VX_STATUS_RANGE / VX_STATUS_RANGE2, see there), iou_threshold >= 0 (taken as
 * fp32: the reference hands torchvision a Python float, i.e. the fp32 IoU is compared with a double there; the two
 * agree for thresholds that are fp32 values, e.g. fp32 tensors).
 * n_classes == 1 declares class-agnostic NMS (every cls must be 0): no shift, any coordinate range.
 * n_classes is the caller's bound on the class indices (<= 0: VX_MAX_CLASSES); with 4..256 declared classes
 * images beyond the single-launch kernel are partitioned by class before sorting (see vx_nms.hip).
 * capacity > 16384 (round 3, same results): a class segment of 257..16384 boxes is resolved over the whole chip
 * (vx_nms_bm.h: spatial tiles -> pair kernel -> edge-list scan; the sweeps remain as fall-back for untame boxes and
 * overflowing edge lists); top_k > 0 first cuts every image to a score prefix (greedy NMS of a score prefix is the
 * prefix of the full NMS) and redoes an image whose prefix keeps fewer than top_k boxes on the full path in the same
 * call -- class ids are then range-checked for the prefix's candidates only. */
size_t vx_batched_nms_workspace_bytes(int B, int capacity);
int vx_batched_nms(const float *d_boxes, const float *d_scores, const int64_t *d_cls,
                   const float *d_det, const int32_t *d_counts,
                   int B, int capacity, int D, int n_classes /* <=0: VX_MAX_CLASSES */,
                   float iou_threshold, int top_k,
                   int64_t *d_keep_idx, float *d_out_det, int32_t *d_keep_counts,
                   int out_capacity, int32_t *d_status,
                   void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- DETR post-process
 * Replaces DETRPostProcess.forward / _process_single_batch (networks/models/detection/detr.py:539-559)
 * and cxcywh_to_xyxy (detr.py:932-936) for all images in one launch.
 *   d_in f32[B,Q,4+n_logits]: per query (cx,cy,w,h) then n_logits logits, the last one = "no object".
 * Per query: p = softmax(logits); conf,label = max / first argmax of p[0 .. n_logits-2];
 * kept iff conf > score_threshold (strict); row = (cx-w/2, cy-h/2, cx+w/2, cy+h/2, conf, float(label)).
 *   d_out f32[B,Q,6]: the kept rows of image b, in query order, at d_out[b][0 .. d_counts[b]).
 *   d_counts i32[B].
 * Tolerance class (softmax/exp): 1e-5 relative on conf; boxes and labels exact. */
int vx_detr_decode(const float *d_in, int B, int Q, int n_logits, float score_threshold,
                   float *d_out, int32_t *d_counts, void *stream);

/* ---------------------------------------------------------------- classification post-process
 * Replaces SoftmaxPostProcess.forward (networks/modules/postprocess/base_postprocess.py:140-150; caller
 * networks/models/classification.py:98, output_format :45-49) for dim=1:
 *   d_in f32[B,n_classes] logits -> d_out f32[B,2] = (float(first arg-max of softmax), max probability).
 * Tolerance class (softmax/exp): 1e-5 relative on the probability; labels exact. */
int vx_softmax_max(const float *d_in, int B, int n_classes, float *d_out, void *stream);

/* Optimistic variant of vx_batched_nms (same arguments, same results) for capacity > 16384: only the cheap paths are
 * launched.  Without a class count in 4..256: ONE launch; images with more than 16384 candidates are NOT processed.
 * With 4 <= n_classes <= 256: that launch plus the class-partitioned path (histogram, scatter, per-class sort + sweep,
 * run merge / top-k select); images with ONE CLASS of more than 16384 candidates are not processed; with top_k > 0
 * neither is an image whose score prefix keeps fewer than top_k boxes.  An unprocessed
 * image has keep count 0 and VX_STATUS_DEFERRED is set in *d_status -- the caller checks the bit after synchronising
 * and, if set, repeats the call with vx_batched_nms.  No reference counterpart: it spares the ~8 launches of the
 * general path (global sort, per-class sweep), which find nothing to do in the common case. */
int vx_batched_nms_small(const float *d_boxes, const float *d_scores, const int64_t *d_cls,
                         const float *d_det, const int32_t *d_counts,
                         int B, int capacity, int D, int n_classes, float iou_threshold, int top_k,
                         int64_t *d_keep_idx, float *d_out_det, int32_t *d_keep_counts,
                         int out_capacity, int32_t *d_status,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* Allocation-free variant of vx_batched_nms (same arguments minus the workspace, same results).  It has NO
 * counterpart in the reference; it exists for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The workspace is a static device array of
 * vx_batched_nms_noalloc_max_bytes() = 64 MiB per GPU (held until the process ends); a call whose
 * vx_batched_nms_workspace_bytes(B, capacity) exceeds it returns VX_ERR_SCRATCH_LIMIT (e.g. B <= 106 at
 * capacity 10647, B <= 14 at 22743, B <= 7 at 50400, B <= 3 at 100000: capacities above 16384 carry the edge
 * lists and spatial records of the bitmask path).  The call itself performs no hipMalloc / hipFree; the first call
 * on a device looks the array's address up (make it outside graph capture).  All calls on one GPU share the
 * array: calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
size_t vx_batched_nms_noalloc_max_bytes(void);
int vx_batched_nms_noalloc(const float *d_boxes, const float *d_scores, const int64_t *d_cls,
                           const float *d_det, const int32_t *d_counts,
                           int B, int capacity, int D, int n_classes, float iou_threshold, int top_k,
                           int64_t *d_keep_idx, float *d_out_det, int32_t *d_keep_counts,
                           int out_capacity, int32_t *d_status, void *stream);

/* ---------------------------------------------------------------- tuple packing
 * The `tuple(results)` of BatchedNMSPostProcess.forward (base_postprocess.py:131-138) as one launch:
 * rows [0, min(d_counts[b],capacity)) of d_src f32[B,capacity,D] are copied back to back into d_dst
 * (image order), *d_total (optional) = number of rows written.  d_dst holds >= B*capacity*D floats.
 * meta_dst (optional): n_meta int32 words are copied from d_meta_src (the counts|keep_counts|status words of
 * the call chain) to meta_dst, which may be PINNED HOST memory mapped into the device (plain stores, visible
 * to the host once an event recorded after this call has completed) -- the path then needs no D2H memcpy. */
int vx_pack_rows(const float *d_src, const int32_t *d_counts, int B, int capacity, int D,
                 float *d_dst, int32_t *d_total,
                 const int32_t *d_meta_src, int32_t *meta_dst, int n_meta, void *stream);

/* ---------------------------------------------------------------- whole YOLOv3 post-process, one host call
 * vx_yolo_fused_decode -> vx_batched_nms -> vx_pack_rows chained on `stream` (no allocation, no synchronisation):
 * the batch-level equivalent of YoloV3PostProcess.forward (postprocess/yolov3.py:71-79 over
 * base_postprocess.py:110-138) from the three raw head tensors.
 *   d_packed f32[B*out_cap*6]: kept rows (x1,y1,x2,y2,class_conf,class) of image 0, then image 1, ...
 *                              out_cap = min(top_k, R) when top_k > 0, else R = 3*sum(G_s^2)
 *   d_meta   i32[2B+1]       : counts[B] | keep_counts[B] | status  (every word written here; status bits as above)
 *   meta_mirror (optional)   : the same 2B+1 words copied to this device-accessible address (pinned host memory)
 * use_threshold takes the VX_DECODE_* bits and VX_POSTPROCESS_NMS_OPTIMISTIC.  The workspace holds every
 * intermediate.  Up to 16384 rows per image (S <= 512) the chain is two launches behind one fill: the decode, then
 * the single-launch NMS kernel, which also packs the decode's per-tile records and -- with B <= the device's
 * compute units -- the kept rows (no vx_k_cand_finish / vx_k_pack_rows launch; same outputs). */
size_t vx_yolo_fused_postprocess_workspace_bytes(int B, int S, int top_k);
int vx_yolo_fused_postprocess(const float *d_head0, const float *d_head1, const float *d_head2,
                              int B, int S, int C, const float anchors_px[18],
                              float score_threshold, int use_threshold, float iou_threshold, int top_k,
                              float *d_packed, int32_t *d_meta, int32_t *meta_mirror,
                              void *d_workspace, size_t workspace_bytes, void *stream);

/* The same call with the latency-bound tail (NMS, packing, the mirror) on `tail_stream`, ordered behind the decode by
 * an event the library records on `stream` (one hipEvent per device, created on first use): `stream` then carries
 * only the HBM-bound decode, so the caller's next kernel on it (the next batch's pre-process) overlaps this batch's
 * tail.  Outputs are complete once `tail_stream` has drained; the caller keeps d_packed / d_meta / d_workspace alive
 * until then.  No reference counterpart (the reference is synchronous Python); same results as the call above. */
int vx_yolo_fused_postprocess_overlapped(const float *d_head0, const float *d_head1, const float *d_head2,
                                         int B, int S, int C, const float anchors_px[18],
                                         float score_threshold, int use_threshold, float iou_threshold, int top_k,
                                         float *d_packed, int32_t *d_meta, int32_t *meta_mirror,
                                         void *d_workspace, size_t workspace_bytes, void *stream, void *tail_stream);

/* The per-batch step of a YOLOv3 predictor around the backbone in ONE host call: vx_preprocess_u8hwc_f32nchw
 * (d_img u8[B,Hs,Ws,3] -> d_norm f32[B,3,S,S]: Normalize.forward, normalizer.py:74-77, with the stretch / letterbox
 * resize of basic_runtime.py:66-90 when (Hs,Ws) != (S,S)) followed by vx_yolo_fused_postprocess on the three head
 * tensors handed in (BasePredictor.forward, predictor/base_module.py:98-102, minus the network between the two).
 * Arguments, outputs and return codes are those of the two calls; nothing is allocated or synchronised. */
int vx_yolo_pipeline_step(const uint8_t *d_img, float *d_norm, int Hs, int Ws, const float mean[3],
                          const float std[3], int scaler, int flip, int resize_mode, int pad_value,
                          const float *d_head0, const float *d_head1, const float *d_head2, int B, int S, int C,
                          const float anchors_px[18], float score_threshold, int use_threshold,
                          float iou_threshold, int top_k, float *d_packed, int32_t *d_meta, int32_t *meta_mirror,
                          void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- result formatting
 * vx_pack_rows + the output_format column slicing of get_prediction_results (predictor/utils.py:11-33) /
 * BaseRuntime.__call__ (src/runtime/vortex/runtime/basic_runtime.py:103-111) + the relative->absolute
 * scaling of InferenceHelper.adjust_coordinates (src/runtime/vortex/runtime/helper.py:292-307), one launch:
 *   d_dst f32[sum_b n_b, n_cols], row-major, images back to back; column j = d_src column h_cols[j],
 *   multiplied (fp32) by d_image_wh[b][0] when h_kinds[j]==1 (x), by d_image_wh[b][1] when ==2 (y).
 * h_cols i32[n_cols] / h_kinds u8[n_cols] are HOST arrays (read during the call); h_kinds or d_image_wh
 * NULL = no scaling.  n_cols <= VX_MAX_FORMAT_COLS, D <= 256. */
int vx_format_results(const float *d_src, const int32_t *d_counts, int B, int capacity, int D,
                      const int32_t *h_cols, const uint8_t *h_kinds, int n_cols,
                      const float *d_image_wh, float *d_dst, int32_t *d_total, void *stream);

/* ---------------------------------------------------------------- measurement hooks (no reference
 * counterpart; bench.py's roofline leg).  When enabled, every kernel family launched by this library is
 * bracketed by HIP events recorded on the launch stream (events are created here, once, outside the
 * data path); vx_profile_get synchronises on them and returns the summed / min / max kernel time of
 * the launches recorded since the last vx_profile_enable.  Process-global, not thread-safe.
 * Each event pair costs a few microseconds of stream time; vx_profile_select restricts the bracketing to
 * the slots whose bit is set in slot_mask (default: all), e.g. only the dominant kernel inside a timed run. */
int vx_profile_enable(int max_launches_per_kernel);
int vx_profile_select(unsigned slot_mask);
int vx_profile_disable(void);
int vx_profile_slots(void);
int vx_profile_get(int slot, const char **name, int *launches, float *total_ms, float *min_ms, float *max_ms);
/* NMS throughput leg of the measurement: while d_counter (a caller-owned, zeroed, 8-byte aligned device word) is
 * set, the NMS sweeps add the number of IoU tests they execute on live lanes to it (one atomic add per wave and
 * class segment); NULL switches the counting off (the default).  Process-global, like the hooks above. */
int vx_profile_set_iou_counter(uint64_t *d_counter);
/* No counterpart in the reference.  Do kernels on stream_a and stream_b of this process run side by side?  HIP spreads a
 * process's streams over a few hardware queues (GPU_MAX_HW_QUEUES), and two streams that share a queue run strictly one
 * after the other.  The probe runs a one-wave ~0.1 ms dependent chain alone on stream_a, then one on each stream, and sets
 * *overlap = 1 when the pair takes < 1.5x the single time (times in ms to ms_alone / ms_both, either may be NULL).
 * Synchronises both streams; a set-up call, not a data-path one (vortex_amd.streams.StreamRing picks its pair with it). */
int vx_streams_overlap(void *stream_a, void *stream_b, int *overlap, float *ms_alone, float *ms_both);
/* Diagnostics of the single-launch NMS kernel: while profiling is enabled, workgroup b writes 8 shader-clock
 * stamps (keys, sort, gather, wave sweeps, block sweeps, second sort, output) as the first 8 words of u64[B][136] at
 * this byte offset of the workspace handed to vx_batched_nms (0 if the arguments are out of range); words 8.. hold,
 * per wave of the image's last column-owner sweep, cycles spent waiting / applying / on the chain / building rows,
 * total cycles and IoU tests. */
size_t vx_batched_nms_stamps_offset(int B, int capacity);
/* Diagnostics of the bitmask NMS path (capacity > 16384): byte offset, inside the NMS workspace, of its int[64] counter
 * block ([0] large segments registered, [1] work items; [8..13] only in a -DVX_BM_DEBUG build); 0 = no such path. */
size_t vx_batched_nms_bm_counters_offset(int B, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* VORTEX_HIP_H */
