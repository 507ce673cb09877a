// DETR post-process: softmax over (classes + no-object), best foreground class, strict threshold on
// its probability, cxcywh -> xyxy, survivors kept in query order.
// Follows DETRPostProcess._process_single_batch (networks/models/detection/detr.py:550-559) and
// cxcywh_to_xyxy (detr.py:932-936).
//
// Layout: in f32[B, Q, 4 + NC1] row-major (box | logits, the LAST logit is "no object").
// One workgroup (4 waves) per image; a wave owns one query row at a time: lanes stride the logits
// (coalesced), butterfly reductions give the row max, the exp sum and the (max prob, first index)
// pair, so every lane ends with the same bits whatever the row length.  Lane i of a wave keeps the
// result of the wave's i-th row, then one ballot per 256-row chunk compacts the survivors in query
// order.  HBM-bound by the single read of the input (Q*(4+NC1)*4 B per image).
#include "vx_common.h"
#include "vx_profile.h"

#define VX_DETR_WAVES 4

__device__ __forceinline__ float vx_wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, VX_WAVE));
    return v;
}
__device__ __forceinline__ float vx_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, VX_WAVE);   // same tree on every lane
    return v;
}

// softmax over lg[0..n_all) followed by (max probability, first index) over lg[0..n_best); executed by a whole
// wave, every lane returns the same bits.  Probabilities are exp(l - max) / sum as torch's softmax.
__device__ __forceinline__ void vx_wave_softmax_best(const float *__restrict__ lg, int n_all, int n_best, int lane,
                                                     float &p, int &pi) {
    float m = -INFINITY;
    for (int c = lane; c < n_all; c += VX_WAVE) m = fmaxf(m, lg[c]);
    m = vx_wave_max(m);
    float s = 0.f;
    for (int c = lane; c < n_all; c += VX_WAVE) s += vx_exp(lg[c] - m);
    s = vx_wave_sum(s);
    p = -1.f;                                                 // probabilities are >= 0
    pi = 0x7fffffff;
    for (int c = lane; c < n_best; c += VX_WAVE) {
        float q = vx_exp(lg[c] - m) / s;
        if (q > p) { p = q; pi = c; }                         // first max within the lane (c ascending)
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {                       // (max p, lowest index) across lanes
        float op = __shfl_xor(p, k, VX_WAVE);
        int oi = __shfl_xor(pi, k, VX_WAVE);
        if (op > p || (op == p && oi < pi)) { p = op; pi = oi; }
    }
}

__global__ __launch_bounds__(VX_DETR_WAVES *VX_WAVE) void vx_k_detr_decode(
    const float *__restrict__ in, int Q, int NC1, float thr, float *__restrict__ out, int *__restrict__ counts) {
    __shared__ int s_wcnt[VX_DETR_WAVES];
    const int b = blockIdx.x;
    const int lane = vx_lane();
    const int wave = threadIdx.x >> 6;
    const int ncol = 4 + NC1;
    const int NC = NC1 - 1;                                   // foreground classes
    const float *img = in + (size_t)b * Q * ncol;
    float *dst = out + (size_t)b * Q * 6;
    int base = 0;                                             // survivors of earlier chunks (uniform)

    for (int chunk = 0; chunk < Q; chunk += VX_DETR_WAVES * VX_WAVE) {
        const int row0 = chunk + wave * VX_WAVE;
        float bx = 0.f, by = 0.f, bw = 0.f, bh = 0.f, conf = 0.f;
        int label = 0;
        bool keep = false;
        const int nrows = min(VX_WAVE, Q - row0);             // may be <= 0 for trailing waves
        for (int i = 0; i < nrows; ++i) {
            const float *row = img + (size_t)(row0 + i) * ncol;
            const float *lg = row + 4;
            float p;
            int pi;
            vx_wave_softmax_best(lg, NC1, NC, lane, p, pi);
            float v = lane < 4 ? row[lane] : 0.f;
            float cx = __shfl(v, 0, VX_WAVE), cy = __shfl(v, 1, VX_WAVE);
            float w = __shfl(v, 2, VX_WAVE), h = __shfl(v, 3, VX_WAVE);
            if (lane == i) {
                bx = cx; by = cy; bw = w; bh = h; conf = p; label = pi;
                keep = p > thr;                               // strict, detr.py:556
            }
        }
        u64 mask = __ballot(keep);
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < VX_DETR_WAVES; ++w) {
            int c = s_wcnt[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (keep) {
            float *o = dst + (size_t)(base + before + vx_mbcnt(mask)) * 6;
            o[0] = bx - 0.5f * bw;                            // detr.py:934-935
            o[1] = by - 0.5f * bh;
            o[2] = bx + 0.5f * bw;
            o[3] = by + 0.5f * bh;
            o[4] = conf;
            o[5] = (float)label;
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[b] = base;
}

extern "C" int vx_detr_decode(const float *d_in, int B, int Q, int n_logits, float score_threshold,
                              float *d_out, int32_t *d_counts, void *stream) {
    if (!d_in || !d_out || !d_counts || B <= 0 || Q <= 0 || n_logits < 2) return VX_ERR_INVALID_ARG;
    if ((long long)Q * (4 + n_logits) > 2147483647LL) return VX_ERR_UNSUPPORTED;
    if (((uintptr_t)d_in | (uintptr_t)d_out | (uintptr_t)d_counts) & 3) return VX_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_DETR, st);
    hipLaunchKernelGGL(vx_k_detr_decode, dim3(B), dim3(VX_DETR_WAVES * VX_WAVE), 0, st, d_in, Q, n_logits,
                       score_threshold, d_out, d_counts);
    vx_prof_end(VX_PROF_DETR, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

// Classification post-process: softmax over the classes of a row, (first arg-max as float, max probability).
// Follows SoftmaxPostProcess.forward (networks/modules/postprocess/base_postprocess.py:147-150) for dim=1:
// out[b] = (float(label), conf).  One wave per row.
__global__ __launch_bounds__(VX_DETR_WAVES *VX_WAVE) void vx_k_softmax_max(const float *__restrict__ in, int B, int NC,
                                                                          float *__restrict__ out) {
    const int row = blockIdx.x * VX_DETR_WAVES + (threadIdx.x >> 6);
    if (row >= B) return;
    const int lane = vx_lane();
    float p;
    int pi;
    vx_wave_softmax_best(in + (size_t)row * NC, NC, NC, lane, p, pi);
    if (lane == 0) {
        out[2 * (size_t)row] = (float)pi;
        out[2 * (size_t)row + 1] = p;
    }
}

extern "C" int vx_softmax_max(const float *d_in, int B, int n_classes, float *d_out, void *stream) {
    if (!d_in || !d_out || B <= 0 || n_classes <= 0) return VX_ERR_INVALID_ARG;
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 3) return VX_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_SOFTMAX, st);
    hipLaunchKernelGGL(vx_k_softmax_max, dim3((unsigned)((B + VX_DETR_WAVES - 1) / VX_DETR_WAVES)),
                       dim3(VX_DETR_WAVES * VX_WAVE), 0, st, d_in, B, n_classes, d_out);
    vx_prof_end(VX_PROF_SOFTMAX, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}
