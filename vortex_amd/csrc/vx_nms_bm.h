// Whole-chip NMS of large class segments ("bitmask path"): the IoU tests of a segment no longer run on the one CU
// that owns its greedy chain.  Three steps, each its own launch for all large segments of all images of a call:
//
//   prep   (inside the kernel that sorted the segment: vx_k_nms_image / vx_k_part_sweep)
//          the segment's class-offset boxes, already in score order, are ALSO laid out by position (Sort-Tile-Recursive
//          order by two counting sorts, see vx_bm_prep) into 32-byte records {box, area, score rank}; 64 consecutive
//          records form a spatial tile with a bounding box.  The segment is registered in a device-side list, one work
//          item per spatial tile.
//   pairs  vx_k_bm_pairs, a grid over ALL work items (segment, tile B): one wave holds tile B's 64 boxes in its
//          lanes and visits the tiles A it owns a pair with (every unordered tile pair has one owner: the tile
//          with the smaller bounding box, ties to the lower index) whose bounding boxes meet; of tile A only the boxes that
//          reach into B's bounding box are tested (four per trip, the cheap candidate test of the chain sweep, the
//          exact predicate for lanes that fail it).  The reference's predicate is symmetric in the two boxes, so a
//          hit is stored once, as the edge (lower rank -> higher rank): "if the first box is kept, the second is
//          suppressed".  Edges are staged per wave in LDS and appended to the segment's edge list.
//   scan   (vx_k_bm_finish per image / vx_k_bm_scan per partitioned segment) the edges are counting-sorted by the
//          score tile of their suppressor; one wave then walks the score tiles in order: alive = not yet suppressed,
//          the tile's own edges resolve it greedily (bit walk over in-tile rows), the edges of its kept boxes
//          mark later boxes dead.  No IoU arithmetic on this serial part.
//
// Exactness: a pair that is skipped has an empty intersection (one box lies outside the bounding box of the
// other's tile), for which the reference's predicate is false whenever iou_threshold >= 0; every other pair is
// decided by the same routines as the sweeps (vx_async_cand / vx_iou_gt).  Greedy NMS only needs, for every box, the
// set of higher-ranked boxes that suppress it if kept: exactly the edge list.  Needs every box of the segment tame
// (vx_tame) and a threshold inside the bracket's range (vx_thr::fast); a segment that does not qualify, or whose edges
// outgrow its edge buffer (VX_BM_ECAP per box: heavily duplicated boxes), is swept by the chain sweep instead.
//
// Reference semantics: networks/modules/postprocess/utils/nms.py:57-98 (batched_nms), :105-123.
#pragma once

#define VX_BM_MIN 256                 // segments with more boxes than this take the bitmask path (<= 256: vx_sweep_wave)
#define VX_BM_MAXN VX_SORT_CHMAX      // ... up to this many (ranks are 16-bit halves of an edge word; LDS tables of 256 tiles)
#define VX_BM_ECAP 8                  // edge capacity per box of a segment
#define VX_BM_COUNTER_BYTES 256       // counter block: [0] segments, [1] work items, [2] buckets of the run merge, [3] large segments of the class partition, [8..13] debug
#define VX_BM_STAGE 512               // edges staged per wave before a flush
#define VX_BM_MAXEXP VX_SWEEP_MAXBIG  // exported segments per image on the single-launch route
#define VX_BM_TRIP 4

struct __attribute__((aligned(32))) vx_bm_rec { float x1, y1, x2, y2, area; int rank; int pad0, pad1; };
struct __attribute__((aligned(32))) vx_bm_seg {
    int b, lo, len, id;               // image, offset of the segment in the image's sorted arrays, boxes, segment index / class
    int item_base, nt, route, pad;    // first work item (= index of tile 0's bounding box), tiles, 0 = image route / 1 = partitioned route
};

struct vx_bm_ws {
    vx_bm_seg *segs;                  // [max_segs]
    int *counters;                    // [0] segments registered, [1] work items reserved  (zeroed per call)
    int *item_seg;                    // [max_items] work item -> segment
    vx_f4 *tbb;                       // [max_items] bounding boxes of the spatial tiles
    vx_bm_rec *recs;                  // [B * npad_cap] records in spatial order (a segment's records at its own offset)
    unsigned *edges, *edges2;         // [B * npad_cap * VX_BM_ECAP] edge lists: as found / sorted by suppressor tile
    int *ecount;                      // [max_segs] edges found (may exceed the capacity: overflow)
    int *pend;                        // [B][2] image route: segment count and exported-segment count of an image the finish kernel completes
    int *img_exp;                     // [B][VX_BM_MAXEXP] their descriptors
    int npad_cap, max_segs, max_items;
};

__host__ __device__ __forceinline__ int vx_bm_max_segs_per_image(int cap) { return cap / (VX_BM_MIN + 1) + 1; }

// ------------------------------------------------------------------------------------------------ prep
// Spatial order = Sort-Tile-Recursive packing (the bulk-loading order of R-trees), by two counting sorts:
//   size classes   segments of 2048+ boxes split their boxes by the larger side (below 1/8, below 1/4, from 1/4 of the
//                  extent of the centres), smallest class first: a tile then holds boxes of one size -- a single wide box
//                  in a tile of small ones would make the tile meet half the segment
//   slabs          inside a class the boxes are ordered by x (256 bins) and cut into S = ceil(sqrt(tiles)) vertical slabs
//                  of equal box count
//   tiles          inside a slab the boxes are ordered by y (64 bins, direction alternating from slab to slab); 64
//                  consecutive boxes form a tile: a near-square patch with exactly 64 boxes
// Measured on the RetinaFace / FPN-SSD candidates (trips of vx_k_bm_pairs per image / per class segment): Morton order
// of one 32 x 32 grid 7164 / 1053, Morton with size classes 6386 / 1178, Hilbert with size classes 5301 / 969,
// STR with size classes 4767 / 901, STR without 6213 / 851.  (Any order is correct: it only decides which boxes
// share a tile.)
#define VX_BM_XB 256                  // x bins per size class
#define VX_BM_YB 64                   // y bins per slab
#define VX_BM_MAXSLAB 16              // slabs per size class: ceil(sqrt(256 tiles))
#define VX_BM_NKEY2 (3 * VX_BM_MAXSLAB * VX_BM_YB)
#define VX_BM_LEVELS_MIN 2048         // segments below this many boxes keep one size class
#define VX_BM_PREP_TBB(n) ((size_t)(((n) + 63) >> 6) * 16)                 // tile bounding boxes (ordered ints)
#define VX_BM_PREP_KEY2(n) (((size_t)(n) * 2 + 15) & ~(size_t)15)
#define VX_BM_PREP_SCRATCH(n) ((size_t)(3 * VX_BM_XB + 3 + VX_BM_NKEY2 + 1) * 4 + VX_BM_PREP_TBB(n) + VX_BM_PREP_KEY2(n))   // bytes of LDS scratch
#define VX_BM_PREP_SCRATCH_BOXES(n) (VX_BM_PREP_SCRATCH(n) + (size_t)(n) * 16)   // ... with the LDS copy of the boxes

struct vx_bm_prep_smem {
    int ext[4];
    int untame, fail, si, ib;
    int lvl_start[4];                 // first position of each size class in (class, x) order; [3] = n
    int wtot[16];
};

// exclusive scan, in place, of h[0..cnt) by the whole workgroup (cnt <= PER * blockDim.x); ends synchronised
template <int PER>
__device__ __forceinline__ void vx_bm_block_scan(int *h, int cnt, int *wtot) {
    const int lane = vx_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int v[PER], tot = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { const int k = threadIdx.x * PER + q; v[q] = k < cnt ? h[k] : 0; tot += v[q]; }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int pre = (lane < wave && lane < nw) ? wtot[lane] : 0;          // totals of the waves in front (<= 16), summed across lanes
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
    int run = incl - tot + __shfl(pre, 0);
#pragma unroll
    for (int q = 0; q < PER; ++q) { const int k = threadIdx.x * PER + q; if (k < cnt) h[k] = run; run += v[q]; }
    __syncthreads();
}

// Whole workgroup.  sbox = the segment's class-offset boxes in score order (global, written by this workgroup before a
// barrier).  scratch = scratch_bytes >= VX_BM_PREP_SCRATCH(n) bytes of LDS (16-byte aligned); with
// VX_BM_PREP_SCRATCH_BOXES(n) bytes the three passes read the boxes from an LDS copy instead of global memory (each
// pass is a short loop of dependent load -> LDS atomic steps: the load's latency is most of it).  Returns false
// (nothing registered) when a box is not tame.  region = b * npad_cap + lo.
__device__ __forceinline__ bool vx_bm_prep(vx_bm_prep_smem &S, void *scratch, size_t scratch_bytes, const vx_bm_ws w, int b, int lo,
                                        int id, int route, const vx_f4 *__restrict__ sbox, int n, u64 *dbg = nullptr) {
#define VX_PREP_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
    const int lane = vx_lane(), nt = blockDim.x;
    n = __builtin_amdgcn_readfirstlane(n);
    const int NT = (n + 63) >> 6;
    const long long region = (long long)b * w.npad_cap + lo;
    int *tb = reinterpret_cast<int *>(scratch);                      // [NT][4] tile bounding boxes as ordered ints
    int *hx = tb + 4 * NT;                                           // [3 * XB + 3] x histogram -> cursors (per class one spare)
    int *h2 = hx + 3 * VX_BM_XB + 3;                                 // [NKEY2 + 1] (class, slab, y bin) histogram -> cursors
    unsigned short *key2 = reinterpret_cast<unsigned short *>(h2 + VX_BM_NKEY2 + 1);   // [n]
    vx_f4 *lbox = scratch_bytes >= VX_BM_PREP_SCRATCH_BOXES(n)
                      ? reinterpret_cast<vx_f4 *>(reinterpret_cast<char *>(scratch) + VX_BM_PREP_SCRATCH(n)) : nullptr;
    __syncthreads();
    VX_PREP_STAMP(0);
    for (int i = threadIdx.x; i < 3 * VX_BM_XB + 3 + VX_BM_NKEY2; i += nt) hx[i] = 0;
    for (int i = threadIdx.x; i < 4 * NT; i += nt) tb[i] = (i & 2) ? (int)0x80000000 : 0x7fffffff;
    if (threadIdx.x == 0) {
        S.ext[0] = S.ext[1] = 0x7fffffff; S.ext[2] = S.ext[3] = (int)0x80000000;
        S.untame = 0; S.fail = 0;
    }
    __syncthreads();
    {
        float mnx = 3e38f, mny = 3e38f, mxx = -3e38f, mxy = -3e38f;
        bool bad = false;
        for (int i0 = threadIdx.x; i0 < n; i0 += 4 * nt) {             // four loads in flight per thread
            vx_f4 q4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = i0 + k * nt; if (i < n) q4[k] = sbox[i]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * nt;
                if (i < n) {
                    const vx_f4 q = q4[k];
                    if (lbox) lbox[i] = q;
                    bad |= !vx_tame(q, (q.z - q.x) * (q.w - q.y));
                    const float cx = q.x + q.z, cy = q.y + q.w;
                    mnx = fminf(mnx, cx); mxx = fmaxf(mxx, cx); mny = fminf(mny, cy); mxy = fmaxf(mxy, cy);
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, d)); mny = fminf(mny, __shfl_xor(mny, d));
            mxx = fmaxf(mxx, __shfl_xor(mxx, d)); mxy = fmaxf(mxy, __shfl_xor(mxy, d));
        }
        if (lane == 0) {
            atomicMin(&S.ext[0], vx_f2ord(mnx)); atomicMin(&S.ext[1], vx_f2ord(mny));
            atomicMax(&S.ext[2], vx_f2ord(mxx)); atomicMax(&S.ext[3], vx_f2ord(mxy));
        }
        if (__ballot(bad) != 0ull && lane == 0) S.untame = 1;
    }
    __syncthreads();
    VX_PREP_STAMP(1);
    if (S.untame) { __syncthreads(); return false; }
    if (threadIdx.x == 0) {
        const int si = atomicAdd(&w.counters[0], 1), ib = atomicAdd(&w.counters[1], NT);
        S.si = si; S.ib = ib;
        if (si < w.max_segs && ib + NT <= w.max_items) {
            vx_bm_seg sg;
            sg.b = b; sg.lo = lo; sg.len = n; sg.id = id; sg.item_base = ib; sg.nt = NT; sg.route = route; sg.pad = 0;
            w.segs[si] = sg;
            w.ecount[si] = 0;
        } else {
            S.fail = 1;                                // cannot happen with the sizes of vx_bm_carve; never write out of bounds
        }
    }
    const float ex0 = vx_ord2f(S.ext[0]), ey0 = vx_ord2f(S.ext[1]);
    const float exw = fmaxf(vx_ord2f(S.ext[2]) - ex0, 1e-30f), eyw = fmaxf(vx_ord2f(S.ext[3]) - ey0, 1e-30f);   // 2 x the extent
    const float sx = (float)VX_BM_XB / exw, sy = (float)VX_BM_YB / eyw;
    const bool levels = n >= VX_BM_LEVELS_MIN;
    const float s1 = levels ? 0.5f * fmaxf(exw, eyw) * 0.125f : 3e38f, s2 = levels ? 2.0f * s1 : 3e38f;
    auto level_of = [&](const vx_f4 &q) __attribute__((always_inline)) -> int {
        const float sz = fmaxf(q.z - q.x, q.w - q.y);
        return sz < s1 ? 0 : (sz < s2 ? 1 : 2);
    };
    auto xbin_of = [&](const vx_f4 &q) __attribute__((always_inline)) -> int {
        return min(VX_BM_XB - 1, max(0, (int)((q.x + q.z - ex0) * sx)));
    };
    // pass 1: histogram over (size class, x bin)
    for (int i = threadIdx.x; i < n; i += nt) {
        const vx_f4 q = lbox ? lbox[i] : sbox[i];
        atomicAdd(&hx[level_of(q) * (VX_BM_XB + 1) + xbin_of(q)], 1);
    }
    __syncthreads();
    VX_PREP_STAMP(2);
    if (S.fail) { __syncthreads(); return false; }
    // exclusive scan of the 3 x (XB + 1) counters (the spare ones are 0) and of nothing else: 771 <= 2 * 512 threads
    vx_bm_block_scan<2>(hx, 3 * (VX_BM_XB + 1), S.wtot);
    if (threadIdx.x < 3) S.lvl_start[threadIdx.x] = hx[threadIdx.x * (VX_BM_XB + 1)];
    if (threadIdx.x == 3) S.lvl_start[3] = n;
    __syncthreads();
    VX_PREP_STAMP(3);
    // pass 2: position in (class, x) order -> slab; key2 = (class, slab, y bin); histogram over key2
    int per_l[3];
#pragma unroll
    for (int L = 0; L < 3; ++L) {
        const int nl = S.lvl_start[L + 1] - S.lvl_start[L], ntl = (nl + 63) >> 6;
        int sl = (int)ceilf(sqrtf((float)ntl));
        sl = max(1, min(sl, VX_BM_MAXSLAB));
        per_l[L] = max(64, ((ntl + sl - 1) / sl) * 64);                    // boxes per slab (whole tiles)
    }
    for (int i = threadIdx.x; i < n; i += nt) {
        const vx_f4 q = lbox ? lbox[i] : sbox[i];
        const int L = level_of(q);
        const int xr = atomicAdd(&hx[L * (VX_BM_XB + 1) + xbin_of(q)], 1) - S.lvl_start[L];
        const int slab = min(VX_BM_MAXSLAB - 1, xr / (L == 0 ? per_l[0] : (L == 1 ? per_l[1] : per_l[2])));
        int yb = min(VX_BM_YB - 1, max(0, (int)((q.y + q.w - ey0) * sy)));
        if (slab & 1) yb = VX_BM_YB - 1 - yb;
        const int k2 = (L * VX_BM_MAXSLAB + slab) * VX_BM_YB + yb;
        key2[i] = (unsigned short)k2;
        atomicAdd(&h2[k2], 1);
    }
    __syncthreads();
    VX_PREP_STAMP(4);
    vx_bm_block_scan<(VX_BM_NKEY2 + 255) / 256>(h2, VX_BM_NKEY2, S.wtot);
    VX_PREP_STAMP(5);      // 12 per thread covers 256-thread workgroups
    vx_bm_rec *recs = w.recs + region;
    for (int i = threadIdx.x; i < n; i += nt) {         // pass 3: scatter (any order inside a bin), tile bounding boxes
        const vx_f4 q = lbox ? lbox[i] : sbox[i];
        const int pos = atomicAdd(&h2[key2[i]], 1);
        vx_bm_rec r;
        r.x1 = q.x; r.y1 = q.y; r.x2 = q.z; r.y2 = q.w; r.area = (q.z - q.x) * (q.w - q.y); r.rank = i; r.pad0 = 0; r.pad1 = 0;
        recs[pos] = r;
        int *t4 = tb + 4 * (pos >> 6);
        atomicMin(&t4[0], vx_f2ord(q.x)); atomicMin(&t4[1], vx_f2ord(q.y));
        atomicMax(&t4[2], vx_f2ord(q.z)); atomicMax(&t4[3], vx_f2ord(q.w));
    }
    __syncthreads();
    VX_PREP_STAMP(6);
    const int ib = S.ib, si = S.si;
    for (int v = threadIdx.x; v < NT; v += nt) {
        vx_f4 q;
        q.x = vx_ord2f(tb[4 * v]); q.y = vx_ord2f(tb[4 * v + 1]); q.z = vx_ord2f(tb[4 * v + 2]); q.w = vx_ord2f(tb[4 * v + 3]);
        w.tbb[ib + v] = q; w.item_seg[ib + v] = si;
    }
    __syncthreads();
    VX_PREP_STAMP(7);
#undef VX_PREP_STAMP
    return true;
}

// ------------------------------------------------------------------------------------------------ pairs
// The candidate test of the chain sweep, keeping its intermediate values: for two tame boxes c = inter1 - lo * uni1 < 0
// (height left unclamped) proves the reference's predicate false; where c >= 0 both extents are positive, so inter1 and
// uni1 ARE the reference's inter and uni and the bracket / division finish the decision without recomputing them.
__device__ __forceinline__ float vx_bm_cand(const vx_f4 &K, float ka, const vx_f4 &bx, float a, float lo, float &inter, float &uni) {
    const float xx1 = vx_vmax<false>(K.x, bx.x);
    const float yy1 = vx_vmax<false>(K.y, bx.y);
    const float xx2 = vx_vmin<false>(K.z, bx.z);
    const float yy2 = vx_vmin<false>(K.w, bx.w);
    const float w = vx_vmax<false>(xx2 - xx1, 0.0f);
    inter = w * (yy2 - yy1);
    uni = ka + a - inter;
    return inter - lo * uni;
}

#define VX_BM_STAGE_PAIRS 256          // edges staged per wave of vx_k_bm_pairs before a flush

// LDS copy of the boxes of the visited tile A that reach into B's bounding box, compacted (a trip reads four
// consecutive records: one address, immediate offsets; no per-box bit extraction on the scalar unit, which is what
// bounds this kernel) and padded with three records that meet nothing.
struct __attribute__((aligned(32))) vx_bm_lrec { vx_f4 b; float area; int rank; int lane; int pad; };

struct vx_bm_wave_ctx {                // per-wave state of vx_k_bm_pairs (all wave-uniform except the lane's box)
    vx_bm_lrec *srec;                  // [64 + 4]
    unsigned *se;                      // [VX_BM_STAGE_PAIRS] staged edges
    unsigned *eout;                    // the segment's edge list
    int *ecnt;
    int ecap, cnt;
};

__device__ __forceinline__ void vx_bm_flush(vx_bm_wave_ctx &C) {
    const int lane = vx_lane();
    int base = 0;
    if (lane == 0) base = atomicAdd(C.ecnt, C.cnt);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int k = lane; k < C.cnt; k += 64)
        if (base + k < C.ecap) C.eout[base + k] = C.se[k];
    C.cnt = 0;
}

// All trips of one visit: nreach compacted records of tile A against this lane's box (tile B).  DIAG: A is the lane's own
// tile; a record only pairs with later lanes.
template <bool DIAG>
__device__ __forceinline__ void vx_bm_visit(vx_bm_wave_ctx &C, int nreach, const vx_f4 &bx, float ar, int rk, u64 vmask, const vx_thr &T) {
    const int lane = vx_lane();
    for (int j = 0; j < nreach; j += VX_BM_TRIP) {
        vx_bm_lrec K[VX_BM_TRIP];
#pragma unroll
        for (int q = 0; q < VX_BM_TRIP; ++q) K[q] = C.srec[j + q];
        u64 hit[VX_BM_TRIP], any = 0ull;
        float in_[VX_BM_TRIP], un_[VX_BM_TRIP];
#pragma unroll
        for (int q = 0; q < VX_BM_TRIP; ++q) {
            float c = vx_bm_cand(K[q].b, K[q].area, bx, ar, T.lo, in_[q], un_[q]);
            if (DIAG) c = lane > K[q].lane ? c : -1.0f;
            hit[q] = vx_uniform64(__ballot(!(c < 0.0f)) & vmask);
            any |= hit[q];
        }
        if (!any) continue;
#pragma unroll
        for (int q = 0; q < VX_BM_TRIP; ++q) {
            const u64 part = hit[q];
            if (!part) continue;
            // exact decision for the candidate lanes: bracket (inter > hi * uni), the IEEE division where the bracket
            // cannot decide (the candidate test already excluded inter < lo * uni)
            const bool mine = (part >> lane) & 1ull;
            bool sup_l = in_[q] > T.hi * un_[q];
            const u64 und = vx_uniform64(__ballot(mine && !sup_l));
            if (und) sup_l = sup_l || (in_[q] / un_[q]) > T.thr;
            const u64 sup = vx_uniform64(__ballot(sup_l) & part);
            if (sup) {
                if ((sup >> lane) & 1ull) {
                    const unsigned a_ = (unsigned)min(rk, K[q].rank), b_ = (unsigned)max(rk, K[q].rank);
                    C.se[C.cnt + vx_mbcnt(sup)] = (a_ << 16) | b_;
                }
                C.cnt = __builtin_amdgcn_readfirstlane(C.cnt + __popcll(sup));
                if (C.cnt > VX_BM_STAGE_PAIRS - 64) vx_bm_flush(C);
            }
        }
    }
}

__global__ __launch_bounds__(256) void vx_k_bm_pairs(const vx_bm_ws w, float thr_, unsigned long long *iou_count) {
    __shared__ unsigned s_e[4][VX_BM_STAGE_PAIRS];
    __shared__ vx_bm_lrec s_rec[4][64 + 4];
    const int lane = vx_lane(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nitems = min(w.counters[1], w.max_items);
    const vx_thr T = vx_make_thr(thr_);
    vx_bm_wave_ctx C;
    C.se = s_e[wv];
    C.srec = s_rec[wv];
    unsigned long long tests = 0ull;
#ifdef VX_BM_DEBUG
    int d_items = 0, d_visits = 0, d_trips = 0, d_tilepairs = 0;
#define VX_BM_DBG(x) x
#else
#define VX_BM_DBG(x)
#endif
    // Static deal of the work items, one wave each per round.  (A/B on MI355X: a device-side work queue -- eight
    // atomic heads, a look before each atomic -- balanced the tail but took 0.56 ms instead of 0.33 ms on the FPN-SSD
    // batch (27 000 short items) and 0.23 instead of 0.21 ms on RetinaFace: the returning atomics cost more than
    // the imbalance.  Dealing every item to Q waves (wave q visits the owned tiles A with A % Q == q) to shorten the
    // heaviest items: RetinaFace 0.120 / 0.136 / 0.162 / 0.185 ms for Q = 1 / 2 / 3 / 4 -- the kernel is bound by its
    // instruction throughput, not by its tail.  An XCD-contiguous deal (every XCD one run of whole images, so a
    // segment's records are fetched by one L2 instead of all eight: 102 MB read for 14 MB of records) left RetinaFace at
    // 0.120 ms and cost FPN-SSD 0.153 -> 0.187 and the 100k stress 0.386 -> 0.463 ms: the XCDs' shares are uneven, and
    // the reads were never the bound.)
    for (int it = blockIdx.x * 4 + wv; it < nitems; it += gridDim.x * 4) {
        // wave-uniform values are made so explicitly (readfirstlane): loads land in vector registers and everything
        // derived from them would be treated as divergent
        const int si = __builtin_amdgcn_readfirstlane(w.item_seg[it]);
        if (si < 0 || si >= w.max_segs) continue;
        const vx_bm_seg *sgp = w.segs + si;
        const int item_base = __builtin_amdgcn_readfirstlane(sgp->item_base);
        const int Bt = it - item_base, n = __builtin_amdgcn_readfirstlane(sgp->len), NT = __builtin_amdgcn_readfirstlane(sgp->nt);
        const long long region = (long long)__builtin_amdgcn_readfirstlane(sgp->b) * w.npad_cap + __builtin_amdgcn_readfirstlane(sgp->lo);
        const vx_bm_rec *__restrict__ recs = w.recs + region;
        const vx_f4 *__restrict__ tbb = w.tbb + item_base;
        C.eout = w.edges + region * VX_BM_ECAP;
        C.ecap = n * VX_BM_ECAP;
        C.ecnt = w.ecount + si;
        C.cnt = 0;
        struct rec6 { vx_f4 b; float area; int rank; };
        auto load_tile = [&](int A) __attribute__((always_inline)) -> rec6 {
            rec6 r;
            r.b.x = r.b.y = r.b.z = r.b.w = 1e30f; r.area = 0.0f; r.rank = 0;       // a box that meets nothing
            if (A >= 0 && A * 64 + lane < n) {
                const vx_bm_rec *g = recs + A * 64 + lane;
                r.b = *reinterpret_cast<const vx_f4 *>(g);
                const vx_f2 t = *reinterpret_cast<const vx_f2 *>(&g->area);
                r.area = t.x; r.rank = __float_as_int(t.y);
            }
            return r;
        };
        // this lane's box (tile B)
        const rec6 me = load_tile(Bt);
        const vx_f4 bx = me.b;
        const float ar = me.area;
        const int rk = me.rank;
        const u64 vmask = vx_uniform64(__ballot(Bt * 64 + lane < n));
        const int nvalid = __popcll(vmask);
        const vx_f4 bb = tbb[Bt];
        // tiles A this item owns a pair with and whose bounding boxes meet B's (lane = tile; NT <= 256: four masks).
        // Every unordered tile pair has ONE owner, the tile whose boxes sit in the lanes: the one with the smaller
        // bounding box (the boxes of the wide tile are then culled against a small target; the other way round nearly
        // every box of the small tile reaches into the wide one) -- equal areas: by index parity, about half each.
        // A/B on MI355X (RetinaFace b128 / FPN-SSD b64, this kernel): parity only 0.47 / 0.47 ms, smaller box 0.21 / 0.33.
        const float bb_area = (bb.z - bb.x) * (bb.w - bb.y);
        u64 tmq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int Al = c * 64 + lane;
            bool meet = false;
            if (Al < NT) {
                const vx_f4 ta = tbb[Al];
                const float ta_area = (ta.z - ta.x) * (ta.w - ta.y);
                const bool par = (Al < Bt) == (((Al + Bt) & 1) != 0);
                const bool owned = Al == Bt || bb_area < ta_area || (bb_area == ta_area && par);
                meet = owned && !(ta.z <= bb.x || ta.x >= bb.z || ta.w <= bb.y || ta.y >= bb.w);
            }
            tmq[c] = c * 64 < NT ? vx_uniform64(__ballot(meet)) : 0ull;
        }
        auto next_tile = [&]() __attribute__((always_inline)) -> int {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (tmq[c]) { const int A = c * 64 + __builtin_ctzll(tmq[c]); tmq[c] &= tmq[c] - 1ull; return A; }
            return -1;
        };
        int A = __builtin_amdgcn_readfirstlane(next_tile());
        rec6 rc = load_tile(A);
        VX_BM_DBG(++d_items;)
        while (A >= 0) {
            VX_BM_DBG(++d_tilepairs;)
            const int An = __builtin_amdgcn_readfirstlane(next_tile());
            const rec6 rn = load_tile(An);              // in flight during this visit
            const bool diag = A == Bt;
            // boxes of tile A that reach into B's bounding box (lane = box of A; the filler box of a short tile reaches nothing)
            const bool reach = !(rc.b.z <= bb.x || rc.b.x >= bb.z || rc.b.w <= bb.y || rc.b.y >= bb.w);
            const u64 m = vx_uniform64(__ballot(reach));
            const int nreach = __popcll(m);
            if (nreach > (diag ? 1 : 0)) {                // a single box has no partner in its own tile
                VX_BM_DBG(++d_visits; d_trips += (nreach + 3) >> 2;)
                if (reach) {
                    vx_bm_lrec r;
                    r.b = rc.b; r.area = rc.area; r.rank = rc.rank; r.lane = lane; r.pad = 0;
                    C.srec[vx_mbcnt(m)] = r;
                }
                if (lane < VX_BM_TRIP - 1) {              // a short last trip reads records that meet nothing
                    vx_bm_lrec r;
                    r.b.x = r.b.y = r.b.z = r.b.w = 1e30f; r.area = 0.0f; r.rank = 0; r.lane = 64; r.pad = 0;
                    C.srec[nreach + lane] = r;
                }
                __builtin_amdgcn_wave_barrier();
                tests += (unsigned long long)nreach * nvalid;
                if (diag) vx_bm_visit<true>(C, nreach, bx, ar, rk, vmask, T);
                else vx_bm_visit<false>(C, nreach, bx, ar, rk, vmask, T);
                __builtin_amdgcn_wave_barrier();
            }
            A = An;
            rc = rn;
        }
        if (C.cnt) vx_bm_flush(C);
    }
    if (iou_count && tests && lane == 0) atomicAdd(iou_count, tests);
#ifdef VX_BM_DEBUG
    if (lane == 0) {
        atomicAdd(&w.counters[8], d_items); atomicAdd(&w.counters[9], d_tilepairs); atomicAdd(&w.counters[10], d_visits);
        atomicAdd(&w.counters[11], d_trips);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ scan
struct vx_bm_scan_smem {
    int start[VX_BM_MAXN / 64 + 1];
    int cur[VX_BM_MAXN / 64];
    u64 dead[VX_BM_MAXN / 64];
    u64 kept[VX_BM_MAXN / 64];       // the result: kept masks of the score tiles [0, ndone)
    u64 rows[64];
    int ndone;
};

// The chain of vx_bm_scan (one wave): walks the score tiles in order over the edges sorted by suppressor tile.
// LDS_E: the sorted edges are in LDS (reads through the LDS address space: a generic pointer would make every read a
// flat load, 4x the latency of a ds_read on this serial path).
// COLS (needs LDS_E): cols[t * 64 + i] = the boxes of score tile t that suppress box i of the same tile if kept, built
// for all tiles at once by the whole workgroup before the chain starts.  The tile is then resolved in as many steps as
// its longest suppression chain is deep (a box is decided once all its suppressors are: a handful of ballots) instead
// of one scalar step per suppressing box, and the serial path per tile shrinks to: dead mask -> resolve -> dead-mask
// atomics of the kept boxes' edges.
// A/B (round 3, MI355X): the same chain as a PULL -- cross edges sorted by target tile in two lists (from the tile in
// front / from further back), the far list resolved a turn ahead against the kept masks in LDS, the near list against
// last turn's kept mask in a scalar register, both ORed over the wave by DPP, no LDS atomics on the serial path --
// was not faster: vx_k_bm_finish 0.088 -> 0.096 ms on RetinaFace, vx_k_bm_scan 0.048 -> 0.053 ms on FPN-SSD.  One wave
// issues ~250 dependent instructions per tile either way.
typedef __attribute__((address_space(3))) unsigned vx_lds_u32;
typedef __attribute__((address_space(3))) u64 vx_lds_u64;
template <bool LDS_E, bool COLS>
__device__ __forceinline__ void vx_bm_chain(vx_bm_scan_smem &S, const unsigned *edges2, const u64 *cols, int n, int NT, int top_k, u64 *dbg = nullptr) {
    const int lane = vx_lane();
    u64 c_a = 0, c_b = 0, c_c = 0, c_d = 0, cs = dbg ? clock64() : 0;
#define VX_CH_LAP(acc) do { if (dbg) { u64 now_ = clock64(); acc += now_ - cs; cs = now_; } } while (0)
    auto ld = [&](int k) __attribute__((always_inline)) -> unsigned {
        if (LDS_E) return ((const vx_lds_u32 *)edges2)[k];
        return edges2[k];
    };
    // The tile's edges live in registers (4 per lane; a tile with more than 256 edges reads the rest from memory)
    // and the next tile's are requested before this tile is resolved: the chain never waits for a cold load.
    constexpr unsigned NONE = 0xFFFFFFFFu;         // lo = hi = 65535: matches no tile
    auto load4 = [&](int t, unsigned *e) __attribute__((always_inline)) {
        const int s0 = t < NT ? S.start[t] : 0, s1 = t < NT ? S.start[t + 1] : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = s0 + q * 64 + lane; e[q] = k < s1 ? ld(k) : NONE; }
    };
    unsigned ec[4], en[4];
    load4(0, ec);
    u64 colc = COLS ? ((const vx_lds_u64 *)cols)[lane] : 0ull, coln = 0ull;
    int ktot = 0, ndone = NT;
    for (int t = 0; t < NT; ++t) {
        load4(t + 1, en);
        if (COLS && t + 1 < NT) coln = ((const vx_lds_u64 *)cols)[(t + 1) * 64 + lane];
        const int s0 = S.start[t], s1 = S.start[t + 1];
        const int rem = n - t * 64;
        u64 alive = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & ~S.dead[t];
        alive = vx_uniform64(alive);
        VX_CH_LAP(c_a);
        if (COLS) {
            const u64 col = colc & alive;                  // suppressors that are dead already do not count
            if (__ballot(col != 0ull) != 0ull) {
                u64 kept = 0ull, und = alive;
                while (und) {
                    const bool ready = ((und >> lane) & 1ull) && (col & und) == 0ull;
                    const bool k = ready && (col & kept) == 0ull;
                    const u64 rm = vx_uniform64(__ballot(ready));
                    kept |= vx_uniform64(__ballot(k));
                    und &= ~rm;
                }
                alive = kept;
            }
        } else if (s1 > s0 && alive) {
            // the tile's own edges -> in-tile rows, resolved greedily by a bit walk over the boxes that suppress somebody
            S.rows[lane] = 0ull;
            __builtin_amdgcn_wave_barrier();
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned lo = ec[q] >> 16, hi = ec[q] & 0xFFFFu;
                if ((int)(hi >> 6) == t) { atomicOr((unsigned long long *)&S.rows[lo & 63u], 1ull << (hi & 63u)); any = true; }
            }
            for (int k = s0 + 256 + lane; k < s1; k += 64) {
                const unsigned e = ld(k);
                const unsigned lo = e >> 16, hi = e & 0xFFFFu;
                if ((int)(hi >> 6) == t) { atomicOr((unsigned long long *)&S.rows[lo & 63u], 1ull << (hi & 63u)); any = true; }
            }
            if (__ballot(any)) {
                __builtin_amdgcn_wave_barrier();
                const u64 row = S.rows[lane];
                const unsigned rlo = (unsigned)row, rhi = (unsigned)(row >> 32);
                const u64 has = __ballot(row != 0ull);
                u64 todo = alive & has;
                while (todo) {
                    const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                    const u64 mi = ((u64)(unsigned)__builtin_amdgcn_readlane((int)rhi, i) << 32) |
                                   (u64)(unsigned)__builtin_amdgcn_readlane((int)rlo, i);
                    alive &= ~mi;
                    todo = alive & has & ~((2ull << i) - 1ull);
                }
            }
        }
        VX_CH_LAP(c_b);
        int nk = __popcll(alive);
        bool stop = false;
        if (top_k > 0 && ktot + nk >= top_k) {       // only the segment's first top_k kept boxes matter
            int excess = ktot + nk - top_k;
            while (excess-- > 0) alive &= ~(1ull << (63 - __builtin_clzll(alive)));
            nk = top_k - ktot;
            stop = true;
        }
        ktot += nk;
        if (lane == 0) S.kept[t] = alive;
        if (stop) { ndone = t + 1; break; }
        VX_CH_LAP(c_c);
        if (alive && s1 > s0) {                      // edges of the kept boxes: their targets in later tiles die
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned lo = ec[q] >> 16, hi = ec[q] & 0xFFFFu;
                if (ec[q] != NONE && ((alive >> (lo & 63u)) & 1ull) && (int)(hi >> 6) > t)
                    atomicOr((unsigned long long *)&S.dead[hi >> 6], 1ull << (hi & 63u));
            }
            for (int k = s0 + 256 + lane; k < s1; k += 64) {
                const unsigned e = ld(k);
                const unsigned lo = e >> 16, hi = e & 0xFFFFu;
                if (((alive >> (lo & 63u)) & 1ull) && (int)(hi >> 6) > t)
                    atomicOr((unsigned long long *)&S.dead[hi >> 6], 1ull << (hi & 63u));
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) ec[q] = en[q];
        colc = coln;
        VX_CH_LAP(c_d);
    }
    if (dbg && lane == 0) { dbg[3] = c_a; dbg[4] = c_b; dbg[5] = c_c; dbg[6] = c_d; }
#undef VX_CH_LAP
    if (lane == 0) S.ndone = ndone;
}

// Whole workgroup.  edges[0..ecount) of a segment of n boxes -> kept masks in S.kept[0..S.ndone) (cut to the first top_k
// kept boxes when top_k > 0).  Returns false (nothing done) when the edge list overflowed.  Ends synchronised.
// lds_e / lds_cap: LDS room for the sorted edges (used when they fit: the chain then reads LDS instead of waiting for a
// global load per score tile -- RetinaFace b128: vx_k_bm_finish 0.108 -> see DESIGN; edges2 in global memory otherwise).
__device__ __forceinline__ bool vx_bm_scan(vx_bm_scan_smem &S, const unsigned *__restrict__ edges, unsigned *edges2_global,
                                        int ecount, int ecap, int n, int top_k, unsigned *lds_e, int lds_cap, u64 *dbg = nullptr) {
    const int lane = vx_lane(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nt = blockDim.x;
    n = __builtin_amdgcn_readfirstlane(n);
    ecount = __builtin_amdgcn_readfirstlane(ecount);
    if (ecount > ecap) return false;
    const int NT = (n + 63) >> 6, E = ecount;
    unsigned *edges2 = E <= lds_cap ? lds_e : edges2_global;
#define VX_SCAN_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
    __syncthreads();
    VX_SCAN_STAMP(0);
    for (int i = threadIdx.x; i < NT; i += nt) { S.cur[i] = 0; S.dead[i] = 0ull; }
    __syncthreads();
    for (int k = threadIdx.x; k < E; k += nt) atomicAdd(&S.cur[edges[k] >> 22], 1);      // (lo rank >> 6) = bits 22..
    __syncthreads();
    if (wv == 0) {                                      // exclusive scan of <= 256 counters: 4 per lane
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int t = lane * 4 + q; v[q] = t < NT ? S.cur[t] : 0; tot += v[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        int run = incl - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = lane * 4 + q;
            if (t < NT) { S.start[t] = run; S.cur[t] = run; }
            run += v[q];
        }
        if (lane == 63) S.start[NT] = incl;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < E; k += nt) {
        const unsigned e = edges[k];
        edges2[atomicAdd(&S.cur[e >> 22], 1)] = e;
    }
    __syncthreads();                                    // edges2 is visible to this workgroup
    // in-tile suppressor columns of all tiles, behind the sorted edges, when the LDS has the room
    const int c0 = (E + 1) & ~1;
    u64 *cols = (E <= lds_cap && c0 + NT * 128 <= lds_cap) ? reinterpret_cast<u64 *>(lds_e + c0) : nullptr;
    if (cols) {
        for (int i = threadIdx.x; i < NT * 64; i += nt) cols[i] = 0ull;
        __syncthreads();
        for (int k = threadIdx.x; k < E; k += nt) {
            const unsigned e = edges2[k];
            const unsigned lo = e >> 16, hi = e & 0xFFFFu;
            if ((lo >> 6) == (hi >> 6)) atomicOr((unsigned long long *)&cols[hi], 1ull << (lo & 63u));
        }
        __syncthreads();
    }
    VX_SCAN_STAMP(1);
    if (wv == 0) {
        if (cols) vx_bm_chain<true, true>(S, edges2, cols, n, NT, top_k, dbg);
        else if (E <= lds_cap) vx_bm_chain<true, false>(S, edges2, nullptr, n, NT, top_k);
        else vx_bm_chain<false, false>(S, edges2, nullptr, n, NT, top_k);
    }
    VX_SCAN_STAMP(2);
    __syncthreads();
#undef VX_SCAN_STAMP
    return true;
}
