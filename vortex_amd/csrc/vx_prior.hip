// K4 — FPN-SSD / RetinaFace prior-box decode (+ 5-point landmarks), threshold and compaction, ONE launch: the dense,
// row-ordered candidate arrays are written by the decode kernel itself (round 3; rounds 1-2 staged per-tile records
// and packed them in a second kernel: 42-54 MB written and read back per batch, 0.04-0.05 ms).
// HBM-bound streaming over f32[B,P,4+2L+NC] rows.
//
// The reference's source for this decoder is absent from the snapshot; the
// contract is the build-authored spec restated in oracle/vortex_oracle.py:prior_box_decode
// (evidence: scripts/export_check/README.md:54-63, :107, :113).  Parity unpinned.
#include "vx_common.h"
#include "vx_profile.h"

struct vx_prior_params {
    int B, P, L, NC, NCOL, NCP, tiles_per_img, groups, groups_reg;
    float v0, v1, thr;
    int use_thr;
};

// A tile = 256 rows (256*NCOL contiguous floats); a group = VX_PRIOR_TPW consecutive tiles of one image = one workgroup.
//
// Output order = row order (the candidate index breaks score ties in the NMS, and the decoder's API returns row-ordered
// tensors), so the rows of a group are written behind those of the groups in front of it:
//   phase 1   the group's rows are scored and the survivors listed in LDS (row, class, confidence: 8 bytes each)
//   hand-off  the workgroup publishes the group's survivor count in one word (count << 1 | 1, an agent-scope store) and
//             sums the words of the groups in front (polling the few that are not there yet).  Group numbers are
//             tickets drawn from a per-image counter at workgroup start (image = blockIdx % B: the workgroups that
//             start together draw from different counters; atomics on ONE address retire at about 5 per
//             microsecond, and 50 workgroups of one image starting together waited up to 10 us for a number), so
//             every group waited for belongs to a running workgroup, and no workgroup waits before it has published: every wait ends, whatever the
//             dispatch order and however few workgroups are resident.
//   phase 2   lane = survivor: the few floats of its row that the box / landmark arithmetic needs are read again (the
//             row was streamed by this very workgroup a moment ago: L2 / Infinity Cache), decoded, and the four output
//             tensors are written densely at base + i -- contiguous, coalesced stores.
// Rounds 1-2 staged per-tile records and packed them in a second kernel (42-54 MB written and read back per batch).
#define VX_PRIOR_TPW 2
#define VX_PRIOR_TICKET_STRIDE 64     // ints between the ticket counters of two images: one 256-byte L2 channel line each

typedef float vx_v4u __attribute__((ext_vector_type(4), aligned(4)));

#define VX_PRIOR_ROWBITS 11           // list entry: row inside the group | class << 11 (groups of up to 2048 rows)
#define VX_PRIOR_MAXGROUP (1 << VX_PRIOR_ROWBITS)
#ifndef VX_PRIOR_REG_RB
#define VX_PRIOR_REG_RB 128           // register kernel: rows per wave and batch
#define VX_PRIOR_REG_NB 2             // batches per wave (A/B on the RetinaFace batch: 256x1 0.1125 ms, 128x2 0.108 ms;
                                      // with ordinary row loads 128x2 0.1038, 128x3 0.1095, 128x4 0.1075, 64x4 0.1058)
#endif
// register kernel: ordinary (not non-temporal) row loads, so the rows of the survivors, which phase 2 reads again, are
// still in the XCD's L2 (PMC: 483 MB read for 414 MB of input with non-temporal loads; A/B RetinaFace 0.1072 -> 0.1044 ms)
#ifdef VX_PRIOR_NT_LD
#define VX_PRIOR_LD(p) __builtin_nontemporal_load(p)
#else
#define VX_PRIOR_LD(p) (*(p))
#endif
#define VX_PRIOR_REG_GROUP (4 * VX_PRIOR_REG_RB * VX_PRIOR_REG_NB)

template <int LIST>
struct vx_prior_smem {
    int2 list[LIST];
    int wcount[4];
    int grp, base;
};

// hand-off + phase 2 of one group (all 256 threads).  `total` survivors are listed in S.list in row order (rows
// relative to the group's first row).  `stage` = LDS for 256 result rows of D = 6 + 2L floats: a lane's result row is
// 24-64 bytes, and storing it from the lane costs one partial-line write request per lane and 8 bytes (measured: the
// 53 MB of a RetinaFace batch took 25 us that way); the rows of a batch are contiguous in the output, so they go
// through LDS and leave as whole 512-byte / 1-KB wave stores.
// lloc (LDS, optional): the four box regressions of every listed survivor, kept by a caller that had the rows in LDS;
// without it they are read again from the input (the row was streamed by this workgroup a moment ago).
template <int LIST>
__device__ __forceinline__ void vx_prior_emit(vx_prior_smem<LIST> &S, const vx_f4 *lloc, float *__restrict__ stage, const vx_prior_params &p,
                                              const float *__restrict__ img, const vx_f4 *__restrict__ priors,
                                              int *__restrict__ fl, int grp, int ngroups, int group_rows, int total,
                                              long long ob, float *__restrict__ boxes, float *__restrict__ scores,
                                              long long *__restrict__ cls, float *__restrict__ det,
                                              int *__restrict__ count_out) {
    const int lane = vx_lane(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) __hip_atomic_store(&fl[grp], (total << 1) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int prow0 = grp * group_rows;
    const int NCOL = p.NCOL, D = 6 + 2 * p.L;                       // D is even
    const bool det8 = ((uintptr_t)det & 7) == 0, det16 = ((D & 3) | (int)((uintptr_t)det & 15)) == 0;
    int j0 = 0;
    do {
        const int j = j0 + threadIdx.x;
        const bool have = j < total;
        vx_f4 bx;
        float cf = 0.f;
        int cid = 0;
        if (have) {
            const int2 e = S.list[j];
            const int prow = prow0 + (e.x & (VX_PRIOR_MAXGROUP - 1));
            cid = e.x >> VX_PRIOR_ROWBITS;
            cf = __int_as_float(e.y);
            const float *srow = img + (long long)prow * NCOL;
            vx_f4 loc;
            if (lloc) loc = lloc[j];
            else { const vx_v4u g4 = *reinterpret_cast<const vx_v4u *>(srow); loc.x = g4.x; loc.y = g4.y; loc.z = g4.z; loc.w = g4.w; }
            const vx_f4 pr = priors[prow];
            const float cx = pr.x + (loc.x * p.v0) * pr.z;
            const float cy = pr.y + (loc.y * p.v0) * pr.w;
            const float w = pr.z * vx_exp(loc.z * p.v1);
            const float h = pr.w * vx_exp(loc.w * p.v1);
            const float hw = w / 2.0f, hh = h / 2.0f;
            bx.x = cx - hw; bx.y = cy - hh; bx.z = cx + hw; bx.w = cy + hh;
            float2 *d2 = reinterpret_cast<float2 *>(stage + threadIdx.x * D);
            d2[0] = make_float2(bx.x, bx.y); d2[1] = make_float2(bx.z, bx.w); d2[2] = make_float2(cf, (float)cid);
#pragma nounroll
            for (int qq = 0; qq < p.L; ++qq)
                d2[3 + qq] = make_float2(pr.x + (srow[4 + 2 * qq] * p.v0) * pr.z, pr.y + (srow[5 + 2 * qq] * p.v0) * pr.w);
        }
        // (the first batch is read and decoded while the groups in front finish their counts)
        if (j0 == 0 && wave == 0) {
            int acc = 0;
            for (int gb = 0; gb < grp; gb += 64) {
                const int g = gb + lane;
                int v = 1;
                if (g < grp) {
                    while (((v = __hip_atomic_load(&fl[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1) == 0)
                        __builtin_amdgcn_s_sleep(2);
                }
                acc += v >> 1;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
            if (lane == 0) {
                S.base = acc;
                if (grp == ngroups - 1) *count_out = acc + total;
            }
        }
        __syncthreads();
        const long long dst0 = ob + S.base + j0;                    // first output row of this batch
        if (have) {
            const long long dst = dst0 + threadIdx.x;
            *reinterpret_cast<vx_f4 *>(boxes + dst * 4) = bx;
            scores[dst] = cf;
            cls[dst] = (long long)cid;
        }
        const int nf = min(256, total - j0) * D;                    // floats of this batch's result rows
        float *dg = det + dst0 * D;
        if (det16) {
#pragma nounroll
            for (int u = threadIdx.x; u < (nf >> 2); u += 256)
                reinterpret_cast<vx_v4 *>(dg)[u] = reinterpret_cast<const vx_v4 *>(stage)[u];
        } else if (det8) {
#pragma nounroll
            for (int u = threadIdx.x; u < (nf >> 1); u += 256)
                reinterpret_cast<float2 *>(dg)[u] = reinterpret_cast<const float2 *>(stage)[u];
        } else {
#pragma nounroll
            for (int u = threadIdx.x; u < nf; u += 256) dg[u] = stage[u];
        }
        j0 += 256;
        if (j0 < total) __syncthreads();                            // the stage is rewritten by the next batch
    } while (j0 < total);
}

// General kernel.  A tile is streamed with coalesced 16-B non-temporal loads into registers while the previous tile is
// scored out of LDS (software pipeline: the loads of tile j+1 are in flight during the row pass of tile j), then
// written to LDS:
//   LAYOUT 1 (NCOL odd, e.g. FPN-SSD 25): the LDS image is the flat tile (one ds_write_b128 per load); an odd row
//            stride makes the row-per-lane pass conflict-free as it is
//   LAYOUT 0 (any NCOL): rows padded to the odd stride NCP = NCOL | 1, four ds_write_b32 per load with the (row, col)
//            of each element advanced incrementally (no per-element division); when NCOL % 4 == 0 a load never
//            straddles a row and the carry disappears
// MAXQ = 16-B loads per thread and tile = ceil(NCOL / 4): 4 (NCOL <= 16), 8 (<= 32) or 16 (<= 63)
// A/B (round 2, MI355X): two tiles in flight (second register buffer) was slower both as a conditional prefetch
// (0.095 -> 0.115 ms) and, in round 3, in a persistent workgroup with unconditional loads (the second buffer costs
// half the occupancy, and the row pass -- LDS latency and the exps -- needs the occupancy more than the loads do).
template <int LAYOUT, int MAXQ>
__global__ __launch_bounds__(256) void vx_k_prior_decode(const float *__restrict__ in,
                                                         const vx_f4 *__restrict__ priors,
                                                         int *__restrict__ tickets, int *__restrict__ flags,
                                                         float *__restrict__ boxes, float *__restrict__ scores,
                                                         long long *__restrict__ cls, float *__restrict__ det,
                                                         int *__restrict__ counts, int *__restrict__ status,
                                                         vx_prior_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ vx_prior_smem<VX_PRIOR_TPW * 256> S;
    __shared__ vx_f4 s_loc[VX_PRIOR_TPW * 256];                     // box regressions of the listed survivors
    const int groups = p.groups;
    const int b = blockIdx.x % p.B;                                 // neighbours in the grid draw from different counters
    if (threadIdx.x == 0) S.grp = atomicAdd(&tickets[b * VX_PRIOR_TICKET_STRIDE], 1);
    __syncthreads();
    const int grp = S.grp;
    const int t0 = grp * VX_PRIOR_TPW, t1 = min(t0 + VX_PRIOR_TPW, p.tiles_per_img);
    const int NCOL = p.NCOL, NCP = LAYOUT == 1 ? NCOL : p.NCP;
    const int lane = vx_lane(), wave = threadIdx.x >> 6;
    const int r = threadIdx.x;
    const bool quad_rows = (NCOL & 3) == 0;
    const long long ob = (long long)b * p.P;
    const float *img = in + ob * NCOL;
    bool nan = false;
    vx_v4u q[MAXQ];
    // loads of tile t into registers (whole 16-B groups; the < 4 leftover floats of a partial last tile are read in the store step)
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        const int row0 = t * 256;
        const int n4 = (min(256, p.P - row0) * NCOL) >> 2;
        const float *g = img + (long long)row0 * NCOL;
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int f = threadIdx.x + 256 * k;
            if (f < n4) q[k] = __builtin_nontemporal_load(reinterpret_cast<const vx_v4u *>(g + 4 * f));
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
                const vx_v4u v = q[k];
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
        const float *g = img + (long long)row0 * NCOL;
        for (int e = n4 * 4 + threadIdx.x; e < n; e += 256) {       // < 4 leftover floats of a partial last tile
            float v = g[e];
            nan |= (v != v);
            int rr = e / NCOL;
            tile[rr * NCP + (e - rr * NCOL)] = v;
        }
    };
    int wg_total = 0;                                               // survivors of this group so far
    load_tile(t0);
    for (int t = t0; t < t1; ++t) {
        const int nrow = min(256, p.P - t * 256);
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
#pragma unroll 4
            for (int k = 1; k < p.NC; ++k) m = lg[k] > m ? lg[k] : m;
            // one exp per logit, sum in class order, (max e, first index) on the way; `before` = the largest e in
            // front of that index
            float e0 = vx_exp(lg[0] - m);
            float sum = 0.f + e0, emax = -1.f, before = -1.f;
            int kmax = 1;
#pragma unroll 4
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
        const u64 mk = __ballot(pass);
        if (lane == 0) S.wcount[wave] = __popcll(mk);
        __syncthreads();
        int base = wg_total;
        for (int w = 0; w < wave; ++w) base += S.wcount[w];
        // the survivor list: (row inside the group | class << 10, confidence) in row order
        if (pass) {
            const int at = base + vx_mbcnt(mk);
            S.list[at] = make_int2(((t - t0) * 256 + r) | (ci << VX_PRIOR_ROWBITS), __float_as_int(conf));
            vx_f4 l4;
            l4.x = row[0]; l4.y = row[1]; l4.z = row[2]; l4.w = row[3];
            s_loc[at] = l4;
        }
        wg_total += S.wcount[0] + S.wcount[1] + S.wcount[2] + S.wcount[3];
        __syncthreads();                                            // the tile and S.wcount are rewritten by the next trip
    }
    // (the tile is free: NCP >= 6 + 2L floats per row)
    vx_prior_emit(S, s_loc, tile, p, img, priors, flags + (long long)b * groups, grp, groups, VX_PRIOR_TPW * 256, wg_total, ob, boxes,
                  scores, cls, det, counts + b);
    if (__ballot(nan) != 0 && lane == 0) atomicOr(status, VX_STATUS_NAN);
}

// Rows of CPR 16-byte chunks whose logits all lie in the last chunk (RetinaFace: 16 floats, the two logits at 14, 15):
// the lane that loaded a row's last chunk scores the row straight from its registers -- no LDS image of the tile, no
// barrier in the stream.  A wave owns RB*NB consecutive rows of the group, in NB batches of RB rows; the RB*CPR/64
// loads of a batch are in flight at once and each is re-issued for the next batch right behind its use (8 KB per wave
// in flight for RB 128, CPR 4); chunk c = k*64 + lane of the wave's rows, so the scoring lanes of load k, then k+1, ...
// run through the rows in order and the wave appends to its own quarter of the list.
// O = position of the first logit inside the chunk, NC = logits (O + NC <= 4).
template <int CPR, int O, int NC>
__global__ __launch_bounds__(256) void vx_k_prior_decode_reg(const float *__restrict__ in,
                                                             const vx_f4 *__restrict__ priors,
                                                             int *__restrict__ tickets, int *__restrict__ flags,
                                                             float *__restrict__ boxes, float *__restrict__ scores,
                                                             long long *__restrict__ cls, float *__restrict__ det,
                                                             int *__restrict__ counts, int *__restrict__ status,
                                                             vx_prior_params p) {
    constexpr int RB = VX_PRIOR_REG_RB, NB = VX_PRIOR_REG_NB, RPW = RB * NB;    // rows per wave: per batch, in all
    __shared__ vx_prior_smem<VX_PRIOR_REG_GROUP> S;
    __shared__ union { int2 wlist[4][RPW]; float stage[256 * 4 * CPR]; } s_u;     // the wave lists, then the result rows
    const int groups = p.groups_reg;
    const int b = blockIdx.x % p.B;                                 // neighbours in the grid draw from different counters
    if (threadIdx.x == 0) S.grp = atomicAdd(&tickets[b * VX_PRIOR_TICKET_STRIDE], 1);
    __syncthreads();
    const int grp = S.grp;
    const int lane = vx_lane(), wave = threadIdx.x >> 6;
    const long long ob = (long long)b * p.P;
    const float *img = in + ob * (4 * CPR);
    const int wrow0 = grp * VX_PRIOR_REG_GROUP + wave * RPW;        // first row of this wave
    const int nchunk = max(0, min(RPW, p.P - wrow0)) * CPR;         // chunks of this wave inside the image
    const float *g = img + (long long)wrow0 * (4 * CPR);
    constexpr int NQ = RB * CPR / 64;                               // loads per lane and batch
    vx_v4u q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int c = k * 64 + lane;
        q[k] = VX_PRIOR_LD(reinterpret_cast<const vx_v4u *>(c < nchunk ? g + 4 * c : img));
    }
    bool nan = false;
    int wcount = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int c = (nb * NQ + k) * 64 + lane;
            const vx_v4u v = q[k];
            if (nb + 1 < NB) {                                      // refill behind the use: the next batch's chunk
                const int cn = c + NQ * 64;
                q[k] = VX_PRIOR_LD(reinterpret_cast<const vx_v4u *>(cn < nchunk ? g + 4 * cn : img));
            }
            const bool inside = c < nchunk;
            nan |= inside & (__builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w));
            const float c4[4] = {v.x, v.y, v.z, v.w};
            bool pass = false;
            float conf = 0.f;
            int ci = 0;
            const int rw = c / CPR;                                 // row inside the wave's rows
            if (c - rw * CPR == CPR - 1 && inside) {
                // the arithmetic of the general kernel's row pass, on registers
                float m = c4[O];
#pragma unroll
                for (int j = 1; j < NC; ++j) m = c4[O + j] > m ? c4[O + j] : m;
                float e0 = vx_exp(c4[O] - m);
                float sum = 0.f + e0, emax = -1.f, before = -1.f;
                int kmax = 1;
#pragma unroll
                for (int j = 1; j < NC; ++j) {
                    float e = vx_exp(c4[O + j] - m);
                    sum = sum + e;
                    if (e > emax) { before = emax; emax = e; kmax = j; }
                }
                conf = emax / sum;
                ci = kmax - 1;
                pass = p.use_thr ? (conf > p.thr) : true;
                if (pass && before >= emax * 0.999999f) {
#pragma unroll
                    for (int j = NC - 1; j >= 1; --j) {
                        float e = vx_exp(c4[O + j] - m);
                        if (j < kmax && e / sum == conf) ci = j - 1;    // the first such index wins
                    }
                }
            }
            const u64 mk = __ballot(pass);
            if (pass) s_u.wlist[wave][wcount + vx_mbcnt(mk)] = make_int2((wave * RPW + rw) | (ci << VX_PRIOR_ROWBITS), __float_as_int(conf));
            wcount += __popcll(mk);
        }
    }
    if (lane == 0) S.wcount[wave] = wcount;
    __syncthreads();
    // the four wave lists, one behind the other
    const int c0 = S.wcount[0], c1 = c0 + S.wcount[1], c2 = c1 + S.wcount[2], total = c2 + S.wcount[3];
    for (int j = threadIdx.x; j < total; j += 256) {
        const int w = (j >= c0) + (j >= c1) + (j >= c2);
        S.list[j] = s_u.wlist[w][j - (w == 0 ? 0 : w == 1 ? c0 : w == 2 ? c1 : c2)];
    }
    __syncthreads();
    vx_prior_emit(S, (const vx_f4 *)nullptr, s_u.stage, p, img, priors, flags + (long long)b * groups, grp, groups, VX_PRIOR_REG_GROUP, total, ob, boxes,
                  scores, cls, det, counts + b);
    if (__ballot(nan) != 0 && lane == 0) atomicOr(status, VX_STATUS_NAN);
}

static inline size_t vx_align256p(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int vx_prior_groups(int P, int rows) { return (P + rows - 1) / rows; }
static inline int vx_prior_max_groups(int P) {
    const int a = vx_prior_groups(P, VX_PRIOR_TPW * 256), b = vx_prior_groups(P, VX_PRIOR_REG_GROUP);
    return a > b ? a : b;
}

extern "C" size_t vx_prior_decode_workspace_bytes(int B, int P, int n_landmarks) {
    if (B <= 0 || P <= 0 || n_landmarks < 0) return 0;
    return vx_align256p((size_t)B * (VX_PRIOR_TICKET_STRIDE + vx_prior_max_groups(P)) * sizeof(int));   // tickets[B][stride] | flags[B][groups]
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
    vx_prior_params p;
    p.B = B; p.P = P; p.L = n_landmarks; p.NC = n_logits; p.NCOL = NCOL; p.NCP = NCOL | 1;
    p.tiles_per_img = tiles; p.v0 = variance0; p.v1 = variance1; p.thr = score_threshold;
    p.use_thr = use_threshold ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int lo = 4 + 2 * n_landmarks;                             // first logit column
    // the logits of a row lie in its last 16-byte chunk: (first logit at 0, 4 logits) or (at 2, 2 logits)
    const bool reg = (NCOL & 3) == 0 && NCOL >= 8 && NCOL <= 16 && lo >= NCOL - 4 && ((uintptr_t)d_in & 15) == 0;
    p.groups = vx_prior_groups(P, VX_PRIOR_TPW * 256);
    p.groups_reg = vx_prior_groups(P, VX_PRIOR_REG_GROUP);
    const int groups = reg ? p.groups_reg : p.groups;
    int *tickets = (int *)d_workspace, *flags = tickets + (size_t)B * VX_PRIOR_TICKET_STRIDE;
    // the tickets and hand-off words start at zero
    if (hipMemsetAsync(d_workspace, 0, (size_t)B * (VX_PRIOR_TICKET_STRIDE + groups) * sizeof(int), st) != hipSuccess)
        return VX_ERR_LAUNCH;
    size_t lds = (size_t)256 * p.NCP * sizeof(float);
    const dim3 grid((unsigned)(B * groups));
    vx_prof_begin(VX_PROF_PRIOR, st);
#define VX_PRIOR_ARGS d_in, (const vx_f4 *)d_priors, tickets, flags, d_boxes, d_scores, (long long *)d_cls, d_det, d_counts, d_status, p
#define VX_PRIOR_LAUNCH(LAYOUT, MAXQ) hipLaunchKernelGGL((vx_k_prior_decode<LAYOUT, MAXQ>), grid, dim3(256), lds, st, VX_PRIOR_ARGS)
#define VX_PRIOR_LAUNCH_REG(CPR, O, NC) hipLaunchKernelGGL((vx_k_prior_decode_reg<CPR, O, NC>), grid, dim3(256), 0, st, VX_PRIOR_ARGS)
    if (reg) {
        const int cpr = NCOL >> 2, o = lo & 3;
        if (o == 0) { if (cpr == 2) VX_PRIOR_LAUNCH_REG(2, 0, 4); else if (cpr == 3) VX_PRIOR_LAUNCH_REG(3, 0, 4); else VX_PRIOR_LAUNCH_REG(4, 0, 4); }
        else        { if (cpr == 2) VX_PRIOR_LAUNCH_REG(2, 2, 2); else if (cpr == 3) VX_PRIOR_LAUNCH_REG(3, 2, 2); else VX_PRIOR_LAUNCH_REG(4, 2, 2); }
    } else if (NCOL & 1) {
        if (NCOL <= 16) VX_PRIOR_LAUNCH(1, 4); else if (NCOL <= 32) VX_PRIOR_LAUNCH(1, 8); else VX_PRIOR_LAUNCH(1, 16);
    } else {
        if (NCOL <= 16) VX_PRIOR_LAUNCH(0, 4); else if (NCOL <= 32) VX_PRIOR_LAUNCH(0, 8); else VX_PRIOR_LAUNCH(0, 16);
    }
    vx_prof_end(VX_PROF_PRIOR, st);
#undef VX_PRIOR_LAUNCH_REG
#undef VX_PRIOR_ARGS
#undef VX_PRIOR_LAUNCH
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}
