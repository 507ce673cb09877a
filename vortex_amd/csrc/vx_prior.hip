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

// One workgroup = VX_PRIOR_TPW consecutive tiles of 256 rows of one image.  A tile (256*NCOL contiguous floats) is
// streamed with coalesced 16-B non-temporal loads into registers while the previous tile is decoded out of LDS
// (software pipeline: the loads of tile j+1 are in flight during the row pass of tile j), then written to LDS:
//   LAYOUT 1 (NCOL odd, e.g. FPN-SSD 25): the LDS image is the flat tile (one ds_write_b128 per load); an odd row
//            stride makes the row-per-lane pass conflict-free as it is
//   LAYOUT 0 (any NCOL): rows padded to the odd stride NCP = NCOL | 1, four ds_write_b32 per load with the (row, col)
//            of each element advanced incrementally (no per-element division); when NCOL % 4 == 0 (RetinaFace 16) a
//            load never straddles a row and the carry disappears
// A/B (round 2, MI355X): two tiles in flight (second register buffer, 8 tiles per workgroup) 0.095 -> 0.115 ms on the
// RetinaFace batch, 0.078 -> 0.109 ms on FPN-SSD: the conditional prefetch at the end of a group makes the compiler
// wait with vmcnt(0) wherever an older load is consumed (vmcnt retires in order), and the registers cost a wave of
// occupancy.  Parking the tile's priors in LDS before the prefetch (so that no load is consumed behind it): no change.
#define VX_PRIOR_TPW 4
// MAXQ = 16-B loads per thread and tile = ceil(NCOL / 4): 4 (NCOL <= 16), 8 (<= 32) or 16 (<= 63)

template <int LAYOUT, int MAXQ>
__global__ __launch_bounds__(256) void vx_k_prior_decode(const float *__restrict__ in,
                                                         const vx_f4 *__restrict__ priors,
                                                         vx_rec *__restrict__ stage, float *__restrict__ stage_lmk,
                                                         int *__restrict__ tile_counts, int *__restrict__ status,
                                                         vx_prior_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ int s_wcount[4];
    __shared__ int2 s_list[256];
    float *lmk_lds = tile + 256 * (LAYOUT == 1 ? p.NCOL : p.NCP);     // [256][2L] behind the tile
    typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
    const int groups = (p.tiles_per_img + VX_PRIOR_TPW - 1) / VX_PRIOR_TPW;
    const int grp = blockIdx.x % groups, b = blockIdx.x / groups;
    const int t0 = grp * VX_PRIOR_TPW, t1 = min(t0 + VX_PRIOR_TPW, p.tiles_per_img);
    const int NCOL = p.NCOL, NCP = LAYOUT == 1 ? NCOL : p.NCP;
    const int lane = vx_lane(), wave = threadIdx.x >> 6;
    const int r = threadIdx.x;
    const bool quad_rows = (NCOL & 3) == 0;
    bool nan = false;
    v4u q[MAXQ];
    // loads of tile t into registers (whole 16-B groups; the < 4 leftover floats of a partial last tile are read in the store step)
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        const int row0 = t * 256;
        const int n4 = (min(256, p.P - row0) * NCOL) >> 2;
        const float *g = in + ((long long)b * p.P + row0) * NCOL;
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int f = threadIdx.x + 256 * k;
            if (f < n4) q[k] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(g + 4 * f));
        }
    };
    auto store_tile = [&](int t) __attribute__((always_inline)) {
        const int row0 = t * 256;
        const int n = min(256, p.P - row0) * NCOL, n4 = n >> 2;
        int e0 = threadIdx.x * 4;
        int row = e0 / NCOL, col = e0 - row * NCOL;
        const int drow = 1024 / NCOL, dcol = 1024 - drow * NCOL;
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int f = threadIdx.x + 256 * k;
            if (f < n4) {
                const v4u v = q[k];
                nan |= __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);      // v_cmp_u_f32: 2 per load
                if (LAYOUT == 1) {
                    *reinterpret_cast<vx_v4 *>(tile + 4 * f) = vx_v4{v.x, v.y, v.z, v.w};
                } else if (quad_rows) {
                    float *d = tile + row * NCP + col;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    int rr = row, cc = col;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        tile[rr * NCP + cc] = vv[j];
                        if (++cc == NCOL) { cc = 0; ++rr; }
                    }
                }
            }
            row += drow; col += dcol;
            if (col >= NCOL) { col -= NCOL; ++row; }
        }
        const float *g = in + ((long long)b * p.P + row0) * NCOL;
        for (int e = n4 * 4 + threadIdx.x; e < n; e += 256) {       // < 4 leftover floats of a partial last tile
            float v = g[e];
            nan |= (v != v);
            int rr = e / NCOL;
            tile[rr * NCP + (e - rr * NCOL)] = v;
        }
    };
    load_tile(t0);
    for (int t = t0; t < t1; ++t) {
        const int row0 = t * 256;
        const int nrow = min(256, p.P - row0);
        store_tile(t);
        __syncthreads();
        if (t + 1 < t1) load_tile(t + 1);                           // in flight during the row pass below
        bool pass = false;
        float conf = 0.f;
        int ci = 0;
        const float *row = tile + r * NCP;
        if (r < nrow) {
            const float *lg = row + 4 + 2 * p.L;
            float m = lg[0];
            for (int k = 1; k < p.NC; ++k) m = lg[k] > m ? lg[k] : m;
            // one exp per logit, sum in class order, (max e, first index) on the way; `before` = the largest e in
            // front of that index
            float e0 = vx_exp(lg[0] - m);
            float sum = 0.f + e0, emax = -1.f, before = -1.f;
            int kmax = 1;
            for (int k = 1; k < p.NC; ++k) {
                float e = vx_exp(lg[k] - m);
                sum = sum + e;
                if (e > emax) { before = emax; emax = e; kmax = k; }
            }
            // max_k fl(e_k/sum) == fl(max_k e_k / sum): a correctly rounded division is monotonic in its numerator
            conf = emax / sum;
            ci = kmax - 1;
            pass = p.use_thr ? (conf > p.thr) : true;
            // "first index of the maximum probability": an earlier class whose e is within an ulp or two of emax can
            // round to the same quotient -- the only case that needs more divisions (about one row in a million)
            if (pass && before >= emax * 0.999999f) {
                for (int k = 1; k < kmax; ++k) {
                    float e = vx_exp(lg[k] - m);
                    if (e / sum == conf) { ci = k - 1; break; }
                }
            }
        }
        u64 mk = __ballot(pass);
        if (lane == 0) s_wcount[wave] = __popcll(mk);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += s_wcount[w];
        // survivors are decoded densely: each passing row leaves (row, class, confidence) in a list at its compacted
        // position; then thread i decodes survivor i -- with a few survivors per wave (5.8 of 64 on the RetinaFace
        // batch) the decode arithmetic and its stores run once per 64 survivors instead of once per wave
        if (pass) {
            const int pos = base + vx_mbcnt(mk);
            s_list[pos] = make_int2(r | (ci << 8), __float_as_int(conf));
        }
        const int total = s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
        __syncthreads();
        if (r < total) {
            const int2 e = s_list[r];
            const int rr = e.x & 255, cls = e.x >> 8;
            const float cf = __int_as_float(e.y);
            const float *srow = tile + rr * NCP;
            vx_f4 pr = priors[row0 + rr];
            float cx = pr.x + (srow[0] * p.v0) * pr.z;
            float cy = pr.y + (srow[1] * p.v0) * pr.w;
            float w = pr.z * vx_exp(srow[2] * p.v1);
            float h = pr.w * vx_exp(srow[3] * p.v1);
            float hw = w / 2.0f, hh = h / 2.0f;
            vx_rec rec;
            rec.a.x = cx - hw; rec.a.y = cy - hh; rec.a.z = cx + hw; rec.a.w = cy + hh;
            rec.b.x = cf; rec.b.y = cf; rec.b.z = __int_as_float(cls); rec.b.w = __int_as_float(row0 + rr);
            long long o = (long long)b * p.P + row0 + r;
            stage[o] = rec;
            if (p.L > 0) {                                          // landmarks: into the row's own LDS slot, copied out below
                float lmv[2];
                float *ld = lmk_lds + r * (2 * p.L);
                for (int qq = 0; qq < p.L; ++qq) {
                    lmv[0] = pr.x + (srow[4 + 2 * qq] * p.v0) * pr.z;
                    lmv[1] = pr.y + (srow[5 + 2 * qq] * p.v0) * pr.w;
                    ld[2 * qq] = lmv[0]; ld[2 * qq + 1] = lmv[1];
                }
            }
        }
        if (threadIdx.x == 0) tile_counts[(long long)b * p.tiles_per_img + t] = total;
        if (p.L > 0) {
            // the tile's survivors own one contiguous block of stage_lmk: consecutive lanes write consecutive floats
            __syncthreads();
            float *lg_out = stage_lmk + ((long long)b * p.P + row0) * (2 * p.L);
            const int nl = total * 2 * p.L;
            for (int i = threadIdx.x; i < nl; i += 256) lg_out[i] = lmk_lds[i];
        }
        __syncthreads();                                            // the tile, the lists and s_wcount are rewritten by the next trip
    }
    if (__ballot(nan) != 0 && lane == 0) atomicOr(status, VX_STATUS_NAN);
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
    size_t lds = (size_t)256 * (p.NCP + 2 * n_landmarks) * sizeof(float);
    const int groups = (tiles + VX_PRIOR_TPW - 1) / VX_PRIOR_TPW;
    vx_prof_begin(VX_PROF_PRIOR, st);
#define VX_PRIOR_LAUNCH(LAYOUT, MAXQ)                                                                                  \
    hipLaunchKernelGGL((vx_k_prior_decode<LAYOUT, MAXQ>), dim3((unsigned)(B * groups)), dim3(256), lds, st, d_in,             \
                       (const vx_f4 *)d_priors, stage, stage_lmk, tile_counts, d_status, p)
    if (NCOL & 1) {
        if (NCOL <= 16) VX_PRIOR_LAUNCH(1, 4); else if (NCOL <= 32) VX_PRIOR_LAUNCH(1, 8); else VX_PRIOR_LAUNCH(1, 16);
    } else {
        if (NCOL <= 16) VX_PRIOR_LAUNCH(0, 4); else if (NCOL <= 32) VX_PRIOR_LAUNCH(0, 8); else VX_PRIOR_LAUNCH(0, 16);
    }
#undef VX_PRIOR_LAUNCH
    vx_prof_end(VX_PROF_PRIOR, st);
    vx_finish_params f;
    f.R = P; f.tiles_per_img = tiles; f.D = 6 + 2 * n_landmarks; f.L = n_landmarks; f.uniform_tile_rows = 256;
    int rc = vx_internal_cand_finish(stage, stage_lmk, tile_counts, B, f, d_boxes, d_scores, (long long *)d_cls, d_det,
                                   d_counts, st);
    if (rc != VX_OK) return rc;
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}
