// K4 — FPN-SSD / RetinaFace prior-box decode (+ 5-point landmarks), threshold and compaction.
// HBM-bound streaming over f32[B,P,4+2L+NC] rows.
//
// The reference's source for this decoder is absent from the snapshot; the
// contract is the build-authored spec restated in oracle/vortex_oracle.py:prior_box_decode
// (evidence: scripts/export_check/README.md:54-63, :107, :113).  Parity unpinned.
#include "vx_common.h"
#include "vx_profile.h"

struct vx_prior_params {
    int B, P, L, NC, NCOL, NCP, tiles_per_img;
    float v0, v1, thr;
    int use_thr;
};

// One workgroup = 256 consecutive rows.  The tile (256*NCOL contiguous floats) is streamed into LDS
// with coalesced loads, rows padded to an odd stride so the row-per-lane pass is conflict-free.
__global__ __launch_bounds__(256) void vx_k_prior_decode(const float *__restrict__ in,
                                                         const vx_f4 *__restrict__ priors,
                                                         vx_rec *__restrict__ stage, float *__restrict__ stage_lmk,
                                                         int *__restrict__ tile_counts, int *__restrict__ status,
                                                         vx_prior_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ int s_wcount[4];
    const int t = blockIdx.x % p.tiles_per_img, b = blockIdx.x / p.tiles_per_img;
    const int row0 = t * 256;
    const int nrow = min(256, p.P - row0);
    const int NCOL = p.NCOL, NCP = p.NCP;
    const float *g = in + ((long long)b * p.P + row0) * NCOL;
    const int n = nrow * NCOL;
    bool nan = false;
    {
        // 16-B flat loads of the contiguous tile (dword alignment suffices); float4 f = threadIdx.x + 256*k covers
        // elements 4f..4f+3 -> (row, col) advanced incrementally, rows padded to the odd stride NCP in LDS
        typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
        const int n4 = n >> 2;
        int e0 = threadIdx.x * 4;
        int row = e0 / NCOL, col = e0 - row * NCOL;
        const int drow = 1024 / NCOL, dcol = 1024 - drow * NCOL;
        for (int f = threadIdx.x; f < n4; f += 256) {
            v4u q = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(g + 4 * f));
            float v[4] = {q.x, q.y, q.z, q.w};
            int rr = row, cc = col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nan |= (v[j] != v[j]);
                tile[rr * NCP + cc] = v[j];
                if (++cc == NCOL) { cc = 0; ++rr; }
            }
            row += drow; col += dcol;
            if (col >= NCOL) { col -= NCOL; ++row; }
        }
        for (int e = n4 * 4 + threadIdx.x; e < n; e += 256) {       // < 4 leftover floats of a partial last tile
            float v = g[e];
            nan |= (v != v);
            int rr = e / NCOL;
            tile[rr * NCP + (e - rr * NCOL)] = v;
        }
    }
    if (__ballot(nan) != 0 && vx_lane() == 0) atomicOr(status, VX_STATUS_NAN);
    __syncthreads();
    const int r = threadIdx.x;
    const int lane = vx_lane(), wave = threadIdx.x >> 6;
    bool pass = false;
    float conf = 0.f;
    int ci = 0;
    const float *row = tile + r * NCP;
    if (r < nrow) {
        float *lg = tile + r * NCP + 4 + 2 * p.L;
        float m = lg[0];
        for (int k = 1; k < p.NC; ++k) m = lg[k] > m ? lg[k] : m;
        // one exp per logit: e_k replaces the logit in LDS, sum in class order, (max e, first index) on the way
        float e0 = vx_exp(lg[0] - m);
        float sum = 0.f + e0, emax = -1.f;
        int kmax = 1;
        for (int k = 1; k < p.NC; ++k) {
            float e = vx_exp(lg[k] - m);
            lg[k] = e;
            sum = sum + e;
            if (e > emax) { emax = e; kmax = k; }
        }
        // max_k fl(e_k/sum) == fl(max_k e_k / sum): a correctly rounded division is monotonic in its numerator
        conf = emax / sum;
        ci = kmax - 1;
        // "first index of the maximum probability": an earlier class whose e is within an ulp or two of emax can
        // round to the same quotient -- the only case that needs another division
        const float near = emax * 0.999999f;
        for (int k = 1; k < kmax; ++k) {
            float e = lg[k];
            if (e >= near && e / sum == conf) { ci = k - 1; break; }
        }
        pass = p.use_thr ? (conf > p.thr) : true;
    }
    u64 mk = __ballot(pass);
    if (lane == 0) s_wcount[wave] = __popcll(mk);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wcount[w];
    if (pass) {
        int pos = base + vx_mbcnt(mk);
        vx_f4 pr = priors[row0 + r];
        float cx = pr.x + (row[0] * p.v0) * pr.z;
        float cy = pr.y + (row[1] * p.v0) * pr.w;
        float w = pr.z * vx_exp(row[2] * p.v1);
        float h = pr.w * vx_exp(row[3] * p.v1);
        float hw = w / 2.0f, hh = h / 2.0f;
        vx_rec rec;
        rec.a.x = cx - hw; rec.a.y = cy - hh; rec.a.z = cx + hw; rec.a.w = cy + hh;
        rec.b.x = conf; rec.b.y = conf; rec.b.z = __int_as_float(ci); rec.b.w = __int_as_float(row0 + r);
        long long o = (long long)b * p.P + row0 + pos;
        stage[o] = rec;
        if (p.L > 0) {
            float *lm = stage_lmk + o * (2 * p.L);
            for (int q = 0; q < p.L; ++q) {
                lm[2 * q] = pr.x + (row[4 + 2 * q] * p.v0) * pr.z;
                lm[2 * q + 1] = pr.y + (row[5 + 2 * q] * p.v0) * pr.w;
            }
        }
    }
    if (threadIdx.x == 0)
        tile_counts[(long long)b * p.tiles_per_img + t] = s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
}

static inline size_t vx_align256p(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t vx_prior_decode_workspace_bytes(int B, int P, int n_landmarks) {
    if (B <= 0 || P <= 0 || n_landmarks < 0) return 0;
    size_t n = vx_align256p((size_t)B * P * sizeof(vx_rec));
    n += vx_align256p((size_t)B * ((P + 255) / 256) * sizeof(int));
    if (n_landmarks > 0) n += vx_align256p((size_t)B * P * 2 * n_landmarks * sizeof(float));
    return n;
}

extern "C" int vx_prior_decode(const float *d_in, const float *d_priors, int B, int P, int n_landmarks, int n_logits,
                               float variance0, float variance1, float score_threshold, int use_threshold,
                               float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det, int32_t *d_counts,
                               int32_t *d_status, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!d_in || !d_priors || !d_boxes || !d_scores || !d_cls || !d_det || !d_counts || !d_status || !d_workspace)
        return VX_ERR_INVALID_ARG;
    if (B <= 0 || P <= 0 || n_landmarks < 0 || n_logits < 2) return VX_ERR_INVALID_ARG;
    int NCOL = 4 + 2 * n_landmarks + n_logits;
    if (NCOL > 63 || P > VX_MAX_CANDIDATES || n_logits - 1 > VX_MAX_CLASSES) return VX_ERR_UNSUPPORTED;
    if (((uintptr_t)d_in & 3) || ((uintptr_t)d_priors & 15) || ((uintptr_t)d_boxes & 15) ||
        ((uintptr_t)d_workspace & 255) || ((uintptr_t)d_det & 3) || ((uintptr_t)d_cls & 7))
        return VX_ERR_ALIGNMENT;
    if (workspace_bytes < vx_prior_decode_workspace_bytes(B, P, n_landmarks)) return VX_ERR_WORKSPACE;
    int tiles = (P + 255) / 256;
    if ((long long)B * tiles > 0x7fffffffLL) return VX_ERR_UNSUPPORTED;
    char *ws = (char *)d_workspace;
    vx_rec *stage = (vx_rec *)ws;
    size_t off = vx_align256p((size_t)B * P * sizeof(vx_rec));
    int *tile_counts = (int *)(ws + off);
    off += vx_align256p((size_t)B * tiles * sizeof(int));
    float *stage_lmk = n_landmarks > 0 ? (float *)(ws + off) : nullptr;
    vx_prior_params p;
    p.B = B; p.P = P; p.L = n_landmarks; p.NC = n_logits; p.NCOL = NCOL; p.NCP = NCOL | 1;
    p.tiles_per_img = tiles; p.v0 = variance0; p.v1 = variance1; p.thr = score_threshold;
    p.use_thr = use_threshold ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    size_t lds = (size_t)256 * p.NCP * sizeof(float);
    vx_prof_begin(VX_PROF_PRIOR, st);
    hipLaunchKernelGGL(vx_k_prior_decode, dim3((unsigned)(B * tiles)), dim3(256), lds, st, d_in,
                       (const vx_f4 *)d_priors, stage, stage_lmk, tile_counts, d_status, p);
    vx_prof_end(VX_PROF_PRIOR, st);
    vx_finish_params f;
    f.R = P; f.tiles_per_img = tiles; f.D = 6 + 2 * n_landmarks; f.L = n_landmarks; f.uniform_tile_rows = 256;
    int rc = vx_internal_cand_finish(stage, stage_lmk, tile_counts, B, f, d_boxes, d_scores, (long long *)d_cls, d_det,
                                   d_counts, st);
    if (rc != VX_OK) return rc;
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}
