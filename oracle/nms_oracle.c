/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. Nothing under vortex_amd/ may load this.
 *
 * CPU restatement of the greedy NMS that the reference obtains from the
 * third-party call `torchvision.ops.nms`
 * (reference call sites: src/development/vortex/development/networks/modules/
 *  postprocess/utils/nms.py:52 and :97; dependency pinned in
 *  src/development/requirements.txt:3-4 as torchvision<=0.10.1, torch<=1.9.1).
 *
 * torchvision is absent from /root/reference and from this image, so this file
 * restates its published CPU algorithm (torchvision/csrc/ops/cpu/nms_kernel.cpp
 * of the 0.10 line, from memory of the upstream source):
 *   areas = (x2 - x1) * (y2 - y1)
 *   order = argsort(scores, descending)
 *   for i in order: if suppressed -> skip; keep i;
 *       for every later j not suppressed:
 *           w = max(0, min(x2i,x2j) - max(x1i,x1j)); h likewise; inter = w*h
 *           ovr = inter / (area_i + area_j - inter);  if (ovr > thr) suppress j
 * all in fp32, one rounding per operation (build with -ffp-contract=off).
 *
 * PARITY UNPINNED for the NMS core: the reference holds no golden vector at
 * this boundary. What IS pinned: the IoU arithmetic against the
 * reference's known answers in tests/utils/test_prediction.py:29-56, and every-
 * thing around the call (offset trick, gather, tuple packing) by running the
 * reference's own wrapper with this function injected as torchvision.ops.nms
 * (tests/golden/gen_golden.py).
 *
 * Tie rule (torch's sort of that era was not stable, so the reference leaves
 * ties unspecified): equal scores keep ascending original index, i.e. a stable
 * descending sort. The HIP path uses the same rule.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int64_t i; } vx_key_t;

/* stable descending merge sort on (score desc, index asc) */
static void merge_sort(vx_key_t *a, vx_key_t *tmp, int64_t n) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n;
            int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t p = lo, q = mid, o = lo;
            while (p < mid && q < hi) {
                /* take right only if strictly greater -> stable */
                if (a[q].s > a[p].s) tmp[o++] = a[q++]; else tmp[o++] = a[p++];
            }
            while (p < mid) tmp[o++] = a[p++];
            while (q < hi) tmp[o++] = a[q++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(vx_key_t));
    }
}

/* boxes: [n,4] xyxy fp32; returns number kept; keep[] gets original indices in
 * descending-score order. */
int64_t vx_oracle_nms(const float *boxes, const float *scores, int64_t n,
                      float iou_threshold, int64_t *keep) {
    if (n <= 0) return 0;
    vx_key_t *ord = (vx_key_t *)malloc((size_t)n * sizeof(vx_key_t));
    vx_key_t *tmp = (vx_key_t *)malloc((size_t)n * sizeof(vx_key_t));
    float *area = (float *)malloc((size_t)n * sizeof(float));
    uint8_t *sup = (uint8_t *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        ord[i].s = scores[i]; ord[i].i = i;
        area[i] = (boxes[4*i+2] - boxes[4*i+0]) * (boxes[4*i+3] - boxes[4*i+1]);
    }
    merge_sort(ord, tmp, n);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = ord[_i].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4*i], iy1 = boxes[4*i+1], ix2 = boxes[4*i+2], iy2 = boxes[4*i+3];
        float ia = area[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = ord[_j].i;
            if (sup[j]) continue;
            /* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a */
            float xx1 = ix1 < boxes[4*j]   ? boxes[4*j]   : ix1;
            float yy1 = iy1 < boxes[4*j+1] ? boxes[4*j+1] : iy1;
            float xx2 = boxes[4*j+2] < ix2 ? boxes[4*j+2] : ix2;
            float yy2 = boxes[4*j+3] < iy2 ? boxes[4*j+3] : iy2;
            float w = xx2 - xx1; w = 0.0f < w ? w : 0.0f;
            float h = yy2 - yy1; h = 0.0f < h ? h : 0.0f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr > iou_threshold) sup[j] = 1;
        }
    }
    free(ord); free(tmp); free(area); free(sup);
    return nk;
}

/* IoU pieces exposed so tests can pin them to the reference's known answers
 * (tests/utils/test_prediction.py:29-56: inter 9/0/4/9, union 9/18/14/49). */
float vx_oracle_intersection(const float *a, const float *b) {
    float xx1 = a[0] < b[0] ? b[0] : a[0];
    float yy1 = a[1] < b[1] ? b[1] : a[1];
    float xx2 = b[2] < a[2] ? b[2] : a[2];
    float yy2 = b[3] < a[3] ? b[3] : a[3];
    float w = xx2 - xx1; w = 0.0f < w ? w : 0.0f;
    float h = yy2 - yy1; h = 0.0f < h ? h : 0.0f;
    return w * h;
}

float vx_oracle_union(const float *a, const float *b) {
    float aa = (a[2] - a[0]) * (a[3] - a[1]);
    float ab = (b[2] - b[0]) * (b[3] - b[1]);
    return aa + ab - vx_oracle_intersection(a, b);
}
