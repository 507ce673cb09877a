// On-device result formatting: the output_format column slicing of
// get_prediction_results (predictor/utils.py:11-33) / BaseRuntime.__call__ (basic_runtime.py:103-111)
// and the relative -> absolute scaling of InferenceHelper.adjust_coordinates (helper.py:292-307),
// applied while the kept rows of all images are packed back to back, so the host receives one dense
// f32[total, n_cols] array (one D2H) instead of B device tensors and B*fields numpy passes.
//
// Output column j = input column col[j]; kind[j] = 0 copy, 1 multiply by the image width,
// 2 multiply by the image height (fp32 multiply by float(im_w / im_h): numpy's in-place `*= int`).
#include "vx_common.h"
#include "vx_profile.h"

struct vx_fmt_cols {
    int n;
    unsigned char col[VX_MAX_FORMAT_COLS];
    unsigned char kind[VX_MAX_FORMAT_COLS];
};

__global__ __launch_bounds__(256) void vx_k_format_results(const float *__restrict__ src, const int *__restrict__ counts,
                                                           int B, int cap, int D, vx_fmt_cols f,
                                                           const float *__restrict__ img_wh, float *__restrict__ dst,
                                                           int *__restrict__ total) {
    __shared__ int s_part[256];
    const int b = blockIdx.x;
    int acc = 0;
    for (int i = threadIdx.x; i < b; i += 256) acc += vx_clamp_count(counts[i], cap);
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
        __syncthreads();
    }
    const long long off = s_part[0];
    const int n = vx_clamp_count(counts[b], cap);
    if (b == B - 1 && threadIdx.x == 0 && total) *total = (int)off + n;
    const float w = img_wh ? img_wh[2 * b] : 1.0f, h = img_wh ? img_wh[2 * b + 1] : 1.0f;
    const float *s = src + (long long)b * cap * D;
    float *d = dst + off * f.n;
    for (int i = threadIdx.x; i < n * f.n; i += 256) {
        int row = i / f.n, j = i - row * f.n;
        float v = s[(long long)row * D + f.col[j]];
        int k = f.kind[j];
        if (img_wh && k == 1) v = v * w;
        if (img_wh && k == 2) v = v * h;
        d[i] = v;
    }
}

extern "C" int vx_format_results(const float *d_src, const int32_t *d_counts, int B, int capacity, int D,
                                 const int32_t *h_cols, const uint8_t *h_kinds, int n_cols,
                                 const float *d_image_wh, float *d_dst, int32_t *d_total, void *stream) {
    if (!d_src || !d_counts || !d_dst || !h_cols) return VX_ERR_INVALID_ARG;
    if (B <= 0 || capacity <= 0 || D <= 0 || n_cols <= 0) return VX_ERR_INVALID_ARG;
    if (n_cols > VX_MAX_FORMAT_COLS || D > 256) return VX_ERR_UNSUPPORTED;
    vx_fmt_cols f;
    f.n = n_cols;
    for (int j = 0; j < n_cols; ++j) {
        if (h_cols[j] < 0 || h_cols[j] >= D) return VX_ERR_INVALID_ARG;
        int k = h_kinds ? h_kinds[j] : 0;
        if (k < 0 || k > 2) return VX_ERR_INVALID_ARG;
        f.col[j] = (unsigned char)h_cols[j];
        f.kind[j] = (unsigned char)k;
    }
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_PACK, st);
    hipLaunchKernelGGL(vx_k_format_results, dim3(B), dim3(256), 0, st, d_src, d_counts, B, capacity, D, f,
                       d_image_wh, d_dst, d_total);
    vx_prof_end(VX_PROF_PACK, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}
