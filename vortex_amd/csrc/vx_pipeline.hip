// One host call for the whole YOLOv3 post-process: fused head decode -> class-aware NMS -> packed rows
// (+ the counts/status words mirrored for the host).  Replaces, in one call,
//   YoloV3Layer x3 + concat   (heads/detection/yolov3.py:263-304, models/detection/yolov3.py:72-76)
//   YoloV3Decoder             (postprocess/yolov3.py:29-68)
//   BatchedNMS per image      (postprocess/utils/nms.py:57-123)
//   the per-image loop + tuple of BatchedNMSPostProcess.forward (base_postprocess.py:131-138)
// It only chains the library's own entry points on the caller's stream: no allocation, no synchronisation;
// the caller supplies one workspace for every intermediate.  Saves the host ~4 marshalled calls and ~8 tensor
// allocations per batch (the synchronous Python API is host-bound at 256 images per call).
#include "vx_common.h"
#include <mutex>

static inline size_t vx_al(size_t x) { return (x + 255) & ~(size_t)255; }

struct vx_pipe_ws {
    char *fused; size_t fused_bytes;
    float *boxes, *scores, *det, *out_det;
    int64_t *cls, *keep_idx;
    char *nms; size_t nms_bytes;
    int *sync; size_t sync_bytes;        // vx_pack_info::sync (tickets, finished count, status, one word per image)
    size_t total;
};

static vx_pipe_ws vx_pipe_carve(void *ws, int B, int S, int out_cap) {
    vx_pipe_ws w = {};
    int R = 0;
    const int div[3] = {32, 16, 8};
    for (int i = 0; i < 3; ++i) { int G = S / div[i]; R += 3 * G * G; }
    char *p = (char *)ws;
    size_t off = 0, n = (size_t)B * R, m = (size_t)B * out_cap;
    w.fused_bytes = vx_yolo_fused_workspace_bytes(B, S);
    w.nms_bytes = vx_batched_nms_workspace_bytes(B, R);
    w.fused = p + off; off += vx_al(w.fused_bytes);
    w.boxes = (float *)(p + off); off += vx_al(n * 4 * sizeof(float));
    w.scores = (float *)(p + off); off += vx_al(n * sizeof(float));
    w.cls = (int64_t *)(p + off); off += vx_al(n * sizeof(int64_t));
    w.det = (float *)(p + off); off += vx_al(n * 6 * sizeof(float));
    w.keep_idx = (int64_t *)(p + off); off += vx_al(m * sizeof(int64_t));
    w.out_det = (float *)(p + off); off += vx_al(m * 6 * sizeof(float));
    w.nms = p + off; off += vx_al(w.nms_bytes);
    w.sync_bytes = vx_al(VX_PACK_SYNC_WORDS(B) * sizeof(int));
    w.sync = (int *)(p + off); off += w.sync_bytes;
    w.total = off;
    return w;
}

static int vx_pipe_rows(int S) {
    int R = 0;
    const int div[3] = {32, 16, 8};
    for (int i = 0; i < 3; ++i) { int G = S / div[i]; R += 3 * G * G; }
    return R;
}

extern "C" size_t vx_yolo_fused_postprocess_workspace_bytes(int B, int S, int top_k) {
    if (B <= 0 || S < 32) return 0;
    int R = vx_pipe_rows(S);
    if (R > VX_MAX_CANDIDATES) return 0;
    int out_cap = top_k > 0 && top_k < R ? top_k : R;
    return vx_pipe_carve(nullptr, B, S, out_cap).total;
}

// One event per device orders the tail stream behind the decode (vx_yolo_fused_postprocess_overlapped); created on
// first use, kept until the process ends.  Re-recording it while an earlier wait is still queued is well defined:
// a wait refers to the record that preceded it.
static hipEvent_t vx_decoded_event() {
    static hipEvent_t ev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
    return ev[dev];
}

// compute units of the current device (cached per device)
static int vx_pipe_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
        cus[dev] = n;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

static int vx_pipe_impl(const float *d_head0, const float *d_head1, const float *d_head2, int B, int S,
                        int C, const float anchors_px[18], float score_threshold, int use_threshold,
                        float iou_threshold, int top_k, float *d_packed, int32_t *d_meta,
                        int32_t *meta_mirror, void *d_workspace, size_t workspace_bytes, void *stream, void *tail_stream) {
    if (!d_packed || !d_meta || !d_workspace) return VX_ERR_INVALID_ARG;
    if (B <= 0 || S < 32 || C <= 0) return VX_ERR_INVALID_ARG;
    if (((uintptr_t)d_workspace & 255) || ((uintptr_t)d_packed & 3) || ((uintptr_t)d_meta & 3)) return VX_ERR_ALIGNMENT;
    const int R = vx_pipe_rows(S);
    if (R > VX_MAX_CANDIDATES) return VX_ERR_UNSUPPORTED;
    const int out_cap = top_k > 0 && top_k < R ? top_k : R;
    vx_pipe_ws w = vx_pipe_carve(d_workspace, B, S, out_cap);
    if (workspace_bytes < w.total) return VX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int32_t *counts = d_meta, *keep_counts = d_meta + B, *status = d_meta + 2 * B;
    // R <= 16384 (every YOLOv3 size up to 512): the single-launch NMS kernel packs the decode's per-tile records itself
    // (one launch less per batch) ...
    const bool from_stage = vx_next_pow2(R) <= 16384;
    // ... and packs the kept rows too when its B workgroups are all resident at once
    const bool pack_in_kernel = from_stage && B <= vx_pipe_cus();
    // The words the chain needs zeroed -- its status word (accumulated by atomicOr) and, when the NMS kernel packs, that
    // kernel's hand-off words, the status word among them -- are cleared by the decode kernel itself when it stages its
    // records for the NMS kernel (vx_stage_info.zero_ptr; the decode then reports a NaN through its tile counts, not
    // through the status word): no fill launch in front of the chain (4 us of GPU time + a launch gap per batch).
    // Without staging only the status word is cleared, by a fill (counts and keep_counts are written for every image;
    // a memset of all 2B+1 words is split by the runtime into two fill kernels).
    if (pack_in_kernel) status = w.sync + 2;
    if (!from_stage && hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) return VX_ERR_LAUNCH;
    vx_stage_info info = {};
    info.zero_ptr = pack_in_kernel ? w.sync : status;
    info.zero_words = pack_in_kernel ? (int)(w.sync_bytes / sizeof(int)) : 1;
    int rc = from_stage
                 ? vx_internal_yolo_fused_stage(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold,
                                                status, w.fused, w.fused_bytes, stream, &info)
                 : vx_yolo_fused_decode(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold, w.boxes,
                                        w.scores, w.cls, w.det, counts, status, w.fused, w.fused_bytes, stream);
    if (rc != VX_OK) return rc;
    if (tail_stream != stream) {                       // NMS + packing on their own stream, behind the decode
        static std::mutex mu;                          // record + wait as one step: the event is shared per device
        std::lock_guard<std::mutex> lock(mu);
        hipEvent_t ev = vx_decoded_event();
        if (!ev || hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent((hipStream_t)tail_stream, ev, 0) != hipSuccess)
            return VX_ERR_LAUNCH;
        stream = tail_stream;
    }
    if (from_stage && info.tiles <= 256) {
        vx_pack_info pk;
        pk.dst = d_packed; pk.sync = w.sync; pk.meta = d_meta; pk.mirror = meta_mirror;
        rc = vx_internal_nms_from_stage(info, w.boxes, w.scores, w.cls, w.det, counts, B, R, 6, C, iou_threshold, top_k,
                                        w.keep_idx, w.out_det, keep_counts, out_cap, status, w.nms, w.nms_bytes, stream,
                                        pack_in_kernel ? &pk : nullptr);
        if (rc != VX_OK) return rc;
        if (pack_in_kernel) return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
    } else if (from_stage)
        return VX_ERR_UNSUPPORTED;                     // (cannot happen: R <= 16384 rows are at most 3 * 22 tiles)
    else
        rc = (use_threshold & VX_POSTPROCESS_NMS_OPTIMISTIC)
             ? vx_batched_nms_small(w.boxes, w.scores, w.cls, w.det, counts, B, R, 6, C, iou_threshold, top_k, w.keep_idx,
                                    w.out_det, keep_counts, out_cap, status, w.nms, w.nms_bytes, stream)
             : vx_batched_nms(w.boxes, w.scores, w.cls, w.det, counts, B, R, 6, C, iou_threshold, top_k, w.keep_idx,
                              w.out_det, keep_counts, out_cap, status, w.nms, w.nms_bytes, stream);
    if (rc != VX_OK) return rc;
    return vx_pack_rows(w.out_det, keep_counts, B, out_cap, 6, d_packed, nullptr, d_meta, meta_mirror, 2 * B + 1, stream);
}

extern "C" int vx_yolo_fused_postprocess(const float *d_head0, const float *d_head1, const float *d_head2, int B, int S,
                                         int C, const float anchors_px[18], float score_threshold, int use_threshold,
                                         float iou_threshold, int top_k, float *d_packed, int32_t *d_meta,
                                         int32_t *meta_mirror, void *d_workspace, size_t workspace_bytes, void *stream) {
    return vx_pipe_impl(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold, iou_threshold,
                        top_k, d_packed, d_meta, meta_mirror, d_workspace, workspace_bytes, stream, stream);
}

// Same call with the latency-bound tail (NMS, packing, the mirror of the counts) on `tail_stream`, ordered behind the
// decode by an event: the caller's `stream` holds only the HBM-bound decode, so its next kernel (the next batch's
// pre-process) starts while this batch's tail runs.  The outputs are complete when `tail_stream` has drained.
extern "C" int vx_yolo_fused_postprocess_overlapped(const float *d_head0, const float *d_head1, const float *d_head2,
                                                    int B, int S, int C, const float anchors_px[18],
                                                    float score_threshold, int use_threshold, float iou_threshold,
                                                    int top_k, float *d_packed, int32_t *d_meta, int32_t *meta_mirror,
                                                    void *d_workspace, size_t workspace_bytes, void *stream,
                                                    void *tail_stream) {
    return vx_pipe_impl(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold, iou_threshold,
                        top_k, d_packed, d_meta, meta_mirror, d_workspace, workspace_bytes, stream, tail_stream);
}

// The whole per-batch step of a YOLOv3 predictor around the backbone, one host call: pre-process of THIS batch's
// frames (d_img -> d_norm, the backbone's input) and post-process of the head tensors the caller hands in (normally
// the backbone's output for the previous d_norm), chained on `stream`.  A synchronous batch-1 caller pays the host
// side of a library call once instead of twice (marshalling + argument checks: ~5 us of a ~70 us step).
// Same results as vx_preprocess_u8hwc_f32nchw followed by vx_yolo_fused_postprocess.
extern "C" int vx_yolo_pipeline_step(const uint8_t *d_img, float *d_norm, int Hs, int Ws, const float mean[3],
                                     const float std[3], int scaler, int flip, int resize_mode, int pad_value,
                                     const float *d_head0, const float *d_head1, const float *d_head2, int B, int S, int C,
                                     const float anchors_px[18], float score_threshold, int use_threshold,
                                     float iou_threshold, int top_k, float *d_packed, int32_t *d_meta, int32_t *meta_mirror,
                                     void *d_workspace, size_t workspace_bytes, void *stream) {
    int rc = vx_preprocess_u8hwc_f32nchw(d_img, d_norm, B, Hs, Ws, S, S, mean, std, scaler, flip, resize_mode, pad_value, stream);
    if (rc != VX_OK) return rc;
    return vx_pipe_impl(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold, iou_threshold,
                        top_k, d_packed, d_meta, meta_mirror, d_workspace, workspace_bytes, stream, stream);
}
