// K5-K7 — class-aware batched NMS for all images of a batch in one pass, no host round trip.
//
//   vx_k_nms_build_keys : key1 = [class:12][descending score:32][index:20] per candidate
//   vx_k_sort_local/_global : per-image bitonic sort of 64-bit keys (LDS chunks up to 16384 keys,
//                         global compare-exchange steps above that); epilogue 1 gathers the
//                         class-offset boxes into sorted order, epilogue 2 writes the outputs
//   vx_k_nms_sweep      : per (image, class segment) greedy suppression: 64x64 IoU tiles resolved
//                         with wavefront ballots; large segments use the block-level sweep
//                         (wave 0 resolves a tile, all waves apply its kept boxes to later tiles)
//   vx_k_part_*         : with 4..256 declared classes the first sort is replaced by a class partition and a
//                         per-class LDS sort inside the sweeping workgroup (see there)
//   vx_k_nms_image      : capacity <= 16384: all of the above for one image in ONE workgroup, keys in LDS
//   second sort         : kept boxes back into global score order (key2 = [score:32][index:32])
//
// Reference: networks/modules/postprocess/utils/nms.py:57-98 (batched_nms: offsets = idxs*2049
// added in fp32, then torchvision.ops.nms), :105-123 (BatchedNMS.forward: index_select by keep).
// The greedy core follows the oracle's restatement of torchvision's CPU kernel (oracle/nms_oracle.c)
// operation for operation; ties in score resolve to the lower candidate index.
//
// Class segmentation is exact w.r.t. the reference's single class-agnostic pass because boxes of
// different classes are disjoint after the 2049*cls offset (|coordinate| < 1024) and a zero
// intersection never exceeds an iou_threshold >= 0.
#include "vx_common.h"
#include "vx_profile.h"

#define VX_SORT_CHMAX 16384
#define VX_STAMP_WORDS (8 + 16 * 8)     // per image: 8 phase stamps of vx_k_nms_image + 8 words per wave of its last column-owner sweep
#define VX_SWEEP_WT 4                 // wave routine: segments up to 64*WT boxes
#define VX_ALIVE_WORDS 2048           // block routine: segments up to 64*2048 = 131072 boxes
#define VX_PART_MAX_CLASSES 256       // class-partitioned path: most classes it takes
#define VX_SWEEP_MAXBIG 128           // large segments queued per workgroup (>= classes per workgroup)

struct vx_nms_params {
    int B, cap, npad_cap, D, n_classes, top_k, out_cap;
    float thr;
};

// Class segmentation == the reference's single pass over offset boxes only while boxes of different classes
// cannot meet after the 2049*cls shift: guaranteed for |coordinate| < 1024 (normalised boxes).  Outside that
// range the call is refused (VX_STATUS_RANGE) unless the caller declared a single class (no shift at all).
__device__ __forceinline__ bool vx_box_in_range(const vx_f4 &b) {
    return !(fabsf(b.x) >= 1024.0f || fabsf(b.y) >= 1024.0f || fabsf(b.z) >= 1024.0f || fabsf(b.w) >= 1024.0f);
}
// ... or every coordinate in [0, 2048] (pixel boxes up to the reference's max_coordinate): class c then occupies
// [2049c, 2049c + 2048], again disjoint from its neighbours.  Mixing the two ranges is not safe, so each violation has
// its own status bit and the result is refused only when both are set.
__device__ __forceinline__ bool vx_box_in_range2(const vx_f4 &b) {
    return b.x >= 0.0f && b.y >= 0.0f && b.z >= 0.0f && b.w >= 0.0f && b.x <= 2048.0f && b.y <= 2048.0f && b.z <= 2048.0f &&
           b.w <= 2048.0f;
}
// status bits for one box (0 when it fits both ranges)
__device__ __forceinline__ int vx_range_bits(const vx_f4 &b) {
    return (vx_box_in_range(b) ? 0 : VX_STATUS_RANGE) | (vx_box_in_range2(b) ? 0 : VX_STATUS_RANGE2);
}
// Sets bits of the call's status word.  The word is one address for the whole launch: normalised boxes that stick out of
// the image set VX_STATUS_RANGE2 in every wave, and thousands of atomics on one address serialise in L2 (measured:
// +0.04 ms on the dense YOLO batch), so the atomic is only issued while the bits are not yet visible.
__device__ __forceinline__ void vx_status_or(int *status, int bits) {
    if (bits == 0) return;
    if ((__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(status, bits);
}
__device__ __forceinline__ void vx_flag_range(int bits, int *status) {       // wave-level OR, at most one atomic per wave
    const bool a = __ballot(bits & VX_STATUS_RANGE) != 0ull, b2 = __ballot(bits & VX_STATUS_RANGE2) != 0ull;
    if ((a || b2) && vx_lane() == 0) vx_status_or(status, (a ? VX_STATUS_RANGE : 0) | (b2 ? VX_STATUS_RANGE2 : 0));
}
// ... into a word of the workgroup's LDS: the workgroup then issues ONE global atomic as its last instruction (a wave's
// later loads queue behind its own atomic, and all workgroups of a launch reach their gather phase together)
__device__ __forceinline__ void vx_flag_range_lds(int bits, int *s_bits) {
    const bool a = __ballot(bits & VX_STATUS_RANGE) != 0ull, b2 = __ballot(bits & VX_STATUS_RANGE2) != 0ull;
    if ((a || b2) && vx_lane() == 0) atomicOr(s_bits, (a ? VX_STATUS_RANGE : 0) | (b2 ? VX_STATUS_RANGE2 : 0));
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vx_k_nms_build_keys(const float *__restrict__ scores,
                                                           const long long *__restrict__ cls,
                                                           const int *__restrict__ counts, vx_nms_params p,
                                                           u64 *__restrict__ keys1, int *__restrict__ count2,
                                                           int *__restrict__ status) {
    int b = blockIdx.y;
    int n = counts ? vx_clamp_count(counts[b], p.cap) : p.cap;
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0 && n > 0) count2[b] = 0;              // n == 0: not this path's image (its count2 may be in use)
    if (k < n) {
        long long c = cls[(long long)b * p.cap + k];
        if (c < 0 || c >= p.n_classes) { atomicOr(status, VX_STATUS_BAD_CLASS); c = c < 0 ? 0 : p.n_classes - 1; }
        keys1[(long long)b * p.npad_cap + k] = vx_key1((int)c, scores[(long long)b * p.cap + k], k);
    }
}

// ------------------------------------------------------------------------------------------------
// Bitonic sort, ascending, of the first npad_b = pow2ceil(n_b) keys of every image (missing keys
// read as the sentinel, which sorts last).  Direction of each compare-exchange comes from the
// global index, so LDS-local phases and global steps compose into one network.
struct vx_sort_params {
    int stride;        // keys per image (npad_cap)
    int CH;            // LDS chunk (power of two <= VX_SORT_CHMAX)
    int k_merge;       // 0: full local sort k=2..CH ; else local tail (j = CH/2..1) of merge stage k
    int fixed_n;       // used when counts == nullptr
    int epilogue;      // 0 none, 1 gather offset boxes, 2 write outputs
};
struct vx_epi_params {
    // epilogue 1
    const float *boxes; vx_f4 *sbox; float *sarea; int cap;
    int *status; int guard;       // guard != 0: flag |coordinate| >= 1024 (see vx_box_in_range)
    const int *sel;               // optional [B]: images with sel[b] == 0 were finished by vx_k_nms_image -> skip
    // epilogue 2
    const float *det; long long *keep_idx; float *out_det; int *keep_counts; int D, top_k, out_cap;
};

__device__ __forceinline__ u64 vx_shfl_xor_u64(u64 v, int j) {
    unsigned lo = __shfl_xor((unsigned)v, j), hi = __shfl_xor((unsigned)(v >> 32), j);
    return ((u64)hi << 32) | lo;
}
// lane ^ 1 and lane ^ 2 stay inside a quad: DPP quad_perm moves on the VALU instead of two ds_bpermute round trips
// through the LDS crossbar (the cross-lane steps of the register sort are bound by that crossbar)
// (lane ^ 16 / lane ^ 32 stay on ds_bpermute: gfx950's v_permlane16/32_swap + select was measured slower here, 58 k vs
// 51 k cycles for the 4096-key sort -- the sort is then VALU-bound and loses the overlap with the LDS crossbar.)
// lane ^ 8 = rotation by 8 inside a row of 16 (row_ror:8); lane ^ 4 = row_shl:4 for banks 0 and 2, row_shr:4 for banks 1 and 3
__device__ __forceinline__ unsigned vx_row_xor8(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned vx_row_xor4(unsigned v) {
    int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
}
template <int J> __device__ __forceinline__ u64 vx_row_xor_u64(u64 v) {
    static_assert(J == 4 || J == 8, "row-local exchanges");
    unsigned lo = J == 8 ? vx_row_xor8((unsigned)v) : vx_row_xor4((unsigned)v);
    unsigned hi = J == 8 ? vx_row_xor8((unsigned)(v >> 32)) : vx_row_xor4((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
template <int J> __device__ __forceinline__ u64 vx_quad_xor_u64(u64 v) {
    static_assert(J == 1 || J == 2, "quad_perm covers lane^1 and lane^2");
    constexpr int ctrl = J == 1 ? 0xB1 : 0x4E;       // quad_perm [1,0,3,2] / [2,3,0,1]
    unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, ctrl, 0xF, 0xF, true);
    unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), ctrl, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}

#define VX_SORT_E 4                            // keys per lane in the register pass
#define VX_SORT_BLK (64 * VX_SORT_E)           // keys per wave in the register pass

// Stages k_from..k_to of the network restricted to partner distances j < VX_SORT_BLK: every wave
// keeps a block of 256 consecutive keys in registers (lane l holds keys l, l+64, l+128, l+192).
// Distances 128 and 64 pair two registers of the same lane, distances <= 32 exchange with lane^j
// (ds_bpermute) — no LDS storage traffic between steps, no barrier until the end, and the four
// chains of a lane are independent (their cross-lane latencies overlap).  Keys e >= m (only when
// m < 256) behave as sentinels.
__device__ __forceinline__ void vx_bitonic_regs(u64 *sk, int m, int base, int k_from, int k_to) {
    const int lane = vx_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nblk = (m + VX_SORT_BLK - 1) / VX_SORT_BLK;
    const int rmax = min(VX_SORT_E, (m + 63) >> 6);
    for (int blk = wave; blk < nblk; blk += nw) {
        const int e0 = blk * VX_SORT_BLK + lane;
        u64 v[VX_SORT_E];
#pragma unroll
        for (int r = 0; r < VX_SORT_E; ++r) v[r] = (e0 + r * 64 < m) ? sk[e0 + r * 64] : VX_KEY_SENTINEL;
        for (int k = k_from; k <= k_to; k <<= 1) {
            // register-register steps (j = 128, 64)
#pragma unroll
            for (int rr = VX_SORT_E / 2; rr >= 1; rr >>= 1) {
                if (rr * 64 <= (k >> 1)) {
#pragma unroll
                    for (int r = 0; r < VX_SORT_E; ++r) {
                        if ((r & rr) == 0 && (r | rr) < rmax) {
                            bool asc = ((base + e0 + r * 64) & k) == 0;
                            u64 a = v[r], c = v[r | rr];
                            bool sw = asc ? (a > c) : (a < c);
                            v[r] = sw ? c : a;
                            v[r | rr] = sw ? a : c;
                        }
                    }
                }
            }
            // cross-lane steps (j = 32, 16 through the LDS crossbar, j = 8 .. 1 by DPP)
            for (int j = min(k >> 1, 32); j > 8; j >>= 1) {
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (int r = 0; r < VX_SORT_E; ++r) {
                    if (r < rmax) {
                        bool asc = ((base + e0 + r * 64) & k) == 0;
                        u64 o = vx_shfl_xor_u64(v[r], j);
                        v[r] = ((asc == lower) == (o < v[r])) ? o : v[r];
                    }
                }
            }
            if (k >= 16) {
                const bool lower = (lane & 8) == 0;
#pragma unroll
                for (int r = 0; r < VX_SORT_E; ++r) {
                    if (r < rmax) {
                        bool asc = ((base + e0 + r * 64) & k) == 0;
                        u64 o = vx_row_xor_u64<8>(v[r]);
                        v[r] = ((asc == lower) == (o < v[r])) ? o : v[r];
                    }
                }
            }
            if (k >= 8) {
                const bool lower = (lane & 4) == 0;
#pragma unroll
                for (int r = 0; r < VX_SORT_E; ++r) {
                    if (r < rmax) {
                        bool asc = ((base + e0 + r * 64) & k) == 0;
                        u64 o = vx_row_xor_u64<4>(v[r]);
                        v[r] = ((asc == lower) == (o < v[r])) ? o : v[r];
                    }
                }
            }
            if (k >= 4) {
                const bool lower = (lane & 2) == 0;
#pragma unroll
                for (int r = 0; r < VX_SORT_E; ++r) {
                    if (r < rmax) {
                        bool asc = ((base + e0 + r * 64) & k) == 0;
                        u64 o = vx_quad_xor_u64<2>(v[r]);
                        v[r] = ((asc == lower) == (o < v[r])) ? o : v[r];
                    }
                }
            }
            {
                const bool lower = (lane & 1) == 0;
#pragma unroll
                for (int r = 0; r < VX_SORT_E; ++r) {
                    if (r < rmax) {
                        bool asc = ((base + e0 + r * 64) & k) == 0;
                        u64 o = vx_quad_xor_u64<1>(v[r]);
                        v[r] = ((asc == lower) == (o < v[r])) ? o : v[r];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < VX_SORT_E; ++r)
            if (e0 + r * 64 < m) sk[e0 + r * 64] = v[r];
    }
    __syncthreads();
}

// Stages k_lo..k_hi of the ascending bitonic network over the m (power of two) keys in LDS whose
// global indices are base..base+m-1.  Partner distances >= 256 go through LDS (one barrier each),
// the rest through vx_bitonic_regs; stages up to k = 256 need no LDS step at all.  Ends synchronised.
__device__ __forceinline__ void vx_bitonic_sort_lds(u64 *sk, int m, int base, int k_lo, int k_hi) {
    const int nt = blockDim.x, half = m >> 1;
    int k = k_lo;
    if (k <= VX_SORT_BLK) {
        int kt = min(k_hi, VX_SORT_BLK);
        vx_bitonic_regs(sk, m, base, k, kt);
        k = kt << 1;
    }
    for (; k <= k_hi; k <<= 1) {
        for (int j = min(k >> 1, half); j >= VX_SORT_BLK; j >>= 1) {
#pragma unroll 2
            for (int t = threadIdx.x; t < half; t += nt) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int q = i | j;
                bool asc = ((base + i) & k) == 0;
                u64 a = sk[i], c = sk[q];
                bool sw = asc ? (a > c) : (a < c);
                if (sw) { sk[i] = c; sk[q] = a; }
            }
            __syncthreads();
        }
        vx_bitonic_regs(sk, m, base, k, k);
    }
}

// LSD radix sort of the n keys in LDS, ascending, digits of up to 9 bits, in place, for keys that arrive in ascending
// order of their low 20 bits (the candidate index) -- then a stable sort on the bits above them is the full 64-bit
// order.  Only the bits that differ between keys are sorted, the score bits [20, 52) and the class bits [52, 64) as
// two ranges of their own, each cut into equal digits (scores above a threshold share their exponent: 25 bits = three
// passes for the RetinaFace batch; dense YOLO: 26 score bits + 7 class bits = 3 + 1 passes, the constant bits between
// them cost nothing).  A wave owns a contiguous run of 64-key rows and keeps them in registers; a pass:
//   rank     per row: the lanes with this lane's digit (one ballot per digit bit), one returning LDS add per distinct
//            digit on the wave's own counters -> position of the key among the wave's keys with that digit
//   scan     the 512 x waves counters in (digit, wave) order, eight per thread
//   scatter  every key to counter base + rank, straight from the registers (all keys were read before the first write)
// against the bitonic network's log2(n)^2 / 2 compare-exchange stages over the padded power of two: 4532 keys take 3
// passes instead of 91 stages over 8192.
// hist: VX_RADIX_HIST_WORDS(waves) words of LDS.  n <= VX_RADIX_ROWS * 64 * waves.  Ends synchronised.
#define VX_RADIX_ROWS 16
#define VX_RADIX_MIN 1024             // fewer keys: the bitonic network (mostly register stages) is as fast
#define VX_RADIX_BITS 9               // A/B (sort cycles, RetinaFace 4532 keys / dense YOLO 2834 keys x 80 classes): 8 bits 47 k / 55 k, 9 bits 49 k / 50 k
#define VX_RADIX_HIST_WORDS(nw) ((1 << VX_RADIX_BITS) * (nw) + 20)
__device__ __forceinline__ bool vx_radix_fits(int n, int nw) { return ((n + 63) >> 6) <= VX_RADIX_ROWS * nw; }
// srank (LDS, optional): srank[candidate index] = the key's position once the score bits are sorted, i.e. its rank in
// the final (score, index) order of the kept list; recorded when the class bits take exactly one pass (returns true).
__device__ __forceinline__ bool vx_radix_sort_lds(u64 *sk, int n, unsigned *hist, unsigned short *srank = nullptr) {
    constexpr int NB = 1 << VX_RADIX_BITS, PER = NB / 64;           // counters per wave, counters per thread in the scan
    const int lane = vx_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rows = ((((n + 63) >> 6) + nw - 1) / nw);             // rows per wave
    const int i0 = wave * rows * 64 + lane;
    unsigned *s_or = hist + NB * nw, *s_wsum = s_or + 4;            // [2] differing bits, [16] wave totals of the scan
    u64 v[VX_RADIX_ROWS];
    if (threadIdx.x < 2) s_or[threadIdx.x] = 0;
    const u64 k0 = sk[0];
    u64 x = 0;
#pragma unroll
    for (int r = 0; r < VX_RADIX_ROWS; ++r) {
        const int i = i0 + r * 64;
        v[r] = (r < rows && i < n) ? sk[i] : k0;
        x |= v[r] ^ k0;
    }
    __syncthreads();
    {
        unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo |= __shfl_xor(lo, d); hi |= __shfl_xor(hi, d); }
        if (lane == 0) { if (lo) atomicOr(&s_or[0], lo); if (hi) atomicOr(&s_or[1], hi); }
    }
    __syncthreads();
    const u64 var = ((((u64)s_or[1]) << 32) | s_or[0]) >> 20 << 20;
    if (var == 0) return false;                                     // one (class, score) for all: already in order
    uint4 *h4 = reinterpret_cast<uint4 *>(hist) + threadIdx.x * (PER / 4);
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) h4[q] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // the two bit ranges: scores, then classes
    const u64 part[2] = {var & ((1ull << 52) - 1ull), var >> 52 << 52};
    const int last_range = part[1] ? 1 : 0;
    bool ranked = false;
    for (int rg = 0; rg < 2; ++rg) {
        if (!part[rg]) continue;
        int shift = __ffsll((long long)part[rg]) - 1;
        int rem = 64 - __clzll((long long)part[rg]) - shift;        // bits of this range
        if (rg == 1 && srank && rem <= VX_RADIX_BITS) {             // v[] holds the keys in (score, index) order here
#pragma unroll
            for (int r = 0; r < VX_RADIX_ROWS; ++r) {
                const int i = i0 + r * 64;
                if (r < rows && i < n) srank[(unsigned)v[r] & 0xFFFFFu] = (unsigned short)i;
            }
            ranked = true;
        }
        for (int np = (rem + VX_RADIX_BITS - 1) / VX_RADIX_BITS; np > 0; --np) {
            const int nbits = (rem + np - 1) / np;
            const unsigned mask = (1u << nbits) - 1u;
            int dest[VX_RADIX_ROWS];                                // counter index | rank << 16, until the scan is done
#pragma unroll
            for (int r = 0; r < VX_RADIX_ROWS; ++r) {
                if (r < rows) {
                    const bool valid = i0 + r * 64 < n;
                    const unsigned d = (unsigned)(v[r] >> shift) & mask;
                    // lanes whose digit equals this lane's: per bit, the ballot of the bit XNOR this lane's bit (as a 0 / -1
                    // word: bit-field extract, two v_xnor, two v_and per digit bit)
                    const u64 pv = __ballot(valid);
                    unsigned plo = (unsigned)pv, phi = (unsigned)(pv >> 32);
#pragma unroll
                    for (int bit = 0; bit < VX_RADIX_BITS; ++bit) {
                        if (bit < nbits) {
                            const int t = ((int)(d << (31 - bit))) >> 31;
                            const u64 bm = __ballot(t != 0);
                            plo &= ~((unsigned)bm ^ (unsigned)t);
                            phi &= ~((unsigned)(bm >> 32) ^ (unsigned)t);
                        }
                    }
                    const u64 peers = ((u64)phi << 32) | plo;
                    const int below = vx_mbcnt(peers);
                    unsigned old = 0;
                    const int ctr = (int)d * nw + wave;
                    if (valid && below == 0) old = atomicAdd(&hist[ctr], (unsigned)__popcll(peers));
                    old = __shfl(old, __ffsll((long long)peers) - 1);    // from the first lane with this digit
                    dest[r] = ctr | ((int)(old + below) << 16);
                }
            }
            __syncthreads();
            {   // exclusive scan of the counters in place: PER consecutive counters per thread
                unsigned c[PER], tot = 0;
#pragma unroll
                for (int q = 0; q < PER / 4; ++q) {
                    const uint4 t = h4[q];
                    c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
                    tot += t.x + t.y + t.z + t.w;
                }
                unsigned incl = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { unsigned y = __shfl_up(incl, d); if (lane >= d) incl += y; }
                if (lane == 63) s_wsum[wave] = incl;
                __syncthreads();
                unsigned pre = lane < wave ? s_wsum[lane] : 0u;     // (wave < 16 <= 64 lanes)
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
                pre = __shfl(pre, 0);
                unsigned run = incl - tot + pre;
#pragma unroll
                for (int q = 0; q < PER / 4; ++q) {
                    uint4 o;
                    o.x = run; run += c[4 * q];
                    o.y = run; run += c[4 * q + 1];
                    o.z = run; run += c[4 * q + 2];
                    o.w = run; run += c[4 * q + 3];
                    h4[q] = o;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < VX_RADIX_ROWS; ++r)
                if (r < rows && i0 + r * 64 < n) sk[hist[dest[r] & 0xFFFF] + (unsigned)(dest[r] >> 16)] = v[r];
            __syncthreads();
            shift += nbits;
            rem -= nbits;
            if (np > 1 || rg < last_range) {                        // another pass: keys in their new order, counters at zero
#pragma unroll
                for (int q = 0; q < PER / 4; ++q) h4[q] = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < VX_RADIX_ROWS; ++r) {
                    const int i = i0 + r * 64;
                    if (r < rows && i < n) v[r] = sk[i];
                }
                __syncthreads();
            }
        }
    }
    return ranked;
}

__global__ __launch_bounds__(1024) void vx_k_sort_local(u64 *__restrict__ keys, const int *__restrict__ counts,
                                                        vx_sort_params sp, vx_epi_params ep) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];
    const int b = blockIdx.y;
    if (ep.sel && ep.sel[b] == 0) return;
    const int n = counts ? vx_clamp_count(counts[b], sp.stride) : sp.fixed_n;
    const int npad = vx_next_pow2(max(n, 1));
    const int base = blockIdx.x * sp.CH;
    if (base >= npad) return;
    const int m = min(sp.CH, npad);
    u64 *g = keys + (long long)b * sp.stride + base;
    const int nt = blockDim.x;
    if (sp.k_merge == 0) {
        for (int i = threadIdx.x; i < m; i += nt) sk[i] = (base + i < n) ? g[i] : VX_KEY_SENTINEL;
    } else {
        if (sp.k_merge > npad) return;
        for (int i = threadIdx.x; i < m; i += nt) sk[i] = g[i];
    }
    __syncthreads();
    const int k_lo = sp.k_merge == 0 ? 2 : sp.k_merge;
    const int k_hi = sp.k_merge == 0 ? m : sp.k_merge;
    vx_bitonic_sort_lds(sk, m, base, k_lo, k_hi);
    const bool final_stage = (sp.k_merge == 0) ? (npad <= sp.CH) : (sp.k_merge == npad);
    // keys go back to global (next stage / sweep read them)
    for (int i = threadIdx.x; i < m; i += nt) g[i] = sk[i];
    if (!final_stage || sp.epilogue == 0) return;
    if (sp.epilogue == 1) {
        // sorted position -> class-offset box  fl32(x + fl32(2049*cls))  (nms.py:94-96) and its area
        int rbits = 0;
        for (int i = threadIdx.x; i < m; i += nt) {
            int gi = base + i;
            if (gi < n) {
                u64 key = sk[i];
                int idx = vx_key1_idx(key);
                float off = (float)(vx_key1_cls(key) * (VX_MAX_COORDINATE + 1));
                vx_f4 bx = *reinterpret_cast<const vx_f4 *>(ep.boxes + ((long long)b * ep.cap + idx) * 4);
                if (ep.guard) rbits |= vx_range_bits(bx);
                bx.x = bx.x + off; bx.y = bx.y + off; bx.z = bx.z + off; bx.w = bx.w + off;
                long long o = (long long)b * sp.stride + gi;
                ep.sbox[o] = bx;
                ep.sarea[o] = (bx.z - bx.x) * (bx.w - bx.y);
            }
        }
        vx_flag_range(rbits, ep.status);
    } else {
        int nk = n;
        if (ep.top_k > 0) nk = min(nk, ep.top_k);
        nk = min(nk, ep.out_cap);
        if (base == 0 && threadIdx.x == 0) ep.keep_counts[b] = nk;
        const int rows = min(m, nk - base);              // rows of this chunk that are output (may be <= 0)
        for (int i = threadIdx.x; i < rows; i += nt)
            ep.keep_idx[(long long)b * ep.out_cap + base + i] = (long long)(sk[i] & 0xFFFFFFFFull);
        // row gather: consecutive threads copy consecutive elements of the output (contiguous stores, each source
        // row read by neighbouring lanes); 8-byte elements when the row length is even
        const float *src = ep.det + (long long)b * ep.cap * ep.D;
        float *dst = ep.out_det + ((long long)b * ep.out_cap + base) * ep.D;
        if ((ep.D & 1) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 7) == 0) {
            const int D2 = ep.D >> 1;
            for (int e = threadIdx.x; e < rows * D2; e += nt) {
                int i = e / D2, q = e - i * D2;
                long long idx = (long long)(sk[i] & 0xFFFFFFFFull);
                reinterpret_cast<float2 *>(dst)[e] = reinterpret_cast<const float2 *>(src + idx * ep.D)[q];
            }
        } else {
            for (int e = threadIdx.x; e < rows * ep.D; e += nt) {
                int i = e / ep.D, q = e - i * ep.D;
                long long idx = (long long)(sk[i] & 0xFFFFFFFFull);
                dst[e] = src[idx * ep.D + q];
            }
        }
    }
}

__global__ __launch_bounds__(256) void vx_k_sort_global(u64 *__restrict__ keys, const int *__restrict__ counts,
                                                        int stride, int fixed_n, int k, int j) {
    const int b = blockIdx.y;
    const int n = counts ? vx_clamp_count(counts[b], stride) : fixed_n;
    const int npad = vx_next_pow2(max(n, 1));
    if (k > npad) return;
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (npad >> 1)) return;
    int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    int q = i | j;
    u64 *g = keys + (long long)b * stride;
    bool asc = (i & k) == 0;
    u64 a = g[i], c = g[q];
    bool sw = asc ? (a > c) : (a < c);
    if (sw) { g[i] = c; g[q] = a; }
}

// ------------------------------------------------------------------------------------------------
// IoU test, operation for operation as oracle/nms_oracle.c (box i = the kept, higher-scoring one).
//
// The reference's predicate is  fl(inter / uni) > thr  with an IEEE division.  The division (12
// instructions) is only executed when the cheap bracket  inter > fl(thr(1+2^-19))*uni  /
// inter < fl(thr(1-2^-19))*uni  cannot decide: inside the bracket the true quotient is more than
// 2^-20 relative (>= 4 ulp) away from thr, so the rounded quotient falls on the same side — the
// result is bit-identical, the undecided band is hit by ~1 pair in 10^5.  max/min are v_max_f32/v_min_f32:
// they differ from std::max/min only for NaN or signed-zero operands, which cannot change w, h.
struct vx_thr { float thr, hi, lo; bool fast; };

__device__ __forceinline__ vx_thr vx_make_thr(float thr) {
    vx_thr t;
    t.thr = thr;
    t.hi = thr * (1.0f + 0x1p-19f);
    t.lo = thr * (1.0f - 0x1p-19f);
    t.fast = thr >= 1e-6f && thr <= 1e6f;      // keeps thr*uni a normal number for 1e-30 < uni < 1e30
    return t;
}

// v_max_f32 / v_min_f32 as written: the compiler's fmaxf/fminf (and fmed3 with an infinite operand, which it
// folds to them) canonicalise both inputs first (v_max_f32 x, x, x) because they came from memory -- 8 extra
// instructions per IoU in the sweep's inner loop.  IU: the first operand is wave-uniform (a readlane result) and
// stays in a scalar register.
template <bool IU> __device__ __forceinline__ float vx_vmax(float a, float b) {
    float r;
    if (IU) asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    else    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool IU> __device__ __forceinline__ float vx_vmin(float a, float b) {
    float r;
    if (IU) asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    else    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool IU>
__device__ __forceinline__ void vx_iou_parts(float ix1, float iy1, float ix2, float iy2, float ia,
                                             float jx1, float jy1, float jx2, float jy2, float ja,
                                             float &inter, float &uni) {
    float xx1 = vx_vmax<IU>(ix1, jx1);
    float yy1 = vx_vmax<IU>(iy1, jy1);
    float xx2 = vx_vmin<IU>(ix2, jx2);
    float yy2 = vx_vmin<IU>(iy2, jy2);
    float w = vx_vmax<false>(xx2 - xx1, 0.0f);
    float h = vx_vmax<false>(yy2 - yy1, 0.0f);
    inter = w * h;
    uni = ia + ja - inter;
}

// bracket test only (T.fast): returns the decision where the bracket decides, sets `und` where it does not
__device__ __forceinline__ bool vx_iou_bracket(float inter, float uni, const vx_thr &T, bool &und) {
    bool hi = inter > T.hi * uni, lo = inter < T.lo * uni;
    und = und || !((hi || lo) && uni > 1e-30f && uni < 1e30f);
    return hi;
}

// A box is "tame" when it is well formed (x2 >= x1, y2 >= y1) with an area in [1e-29, 1e29].  For two tame boxes
// w <= both widths and h <= both heights (rounded differences and products are monotonic), so inter <= min(ia, ja)
// and uni = fl(fl(ia + ja) - inter) lies in [~max(ia, ja), ia + ja] -- inside (1e-30, 1e30) -- and the bracket needs
// no range test.
__device__ __forceinline__ bool vx_tame(const vx_f4 &b, float area) {
    return b.z >= b.x && b.w >= b.y && area >= 1e-29f && area <= 1e29f;
}
__device__ __forceinline__ bool vx_iou_bracket_tame(float inter, float uni, const vx_thr &T, bool &und) {
    bool hi = inter > T.hi * uni, lo = inter < T.lo * uni;
    und = und || !(hi || lo);
    return hi;
}

template <bool IU>
__device__ __forceinline__ bool vx_iou_gt(float ix1, float iy1, float ix2, float iy2, float ia,
                                          float jx1, float jy1, float jx2, float jy2, float ja, const vx_thr &T) {
    float inter, uni;
    vx_iou_parts<IU>(ix1, iy1, ix2, iy2, ia, jx1, jy1, jx2, jy2, ja, inter, uni);
    if (T.fast) {
        bool und = false;
        bool res = vx_iou_bracket(inter, uni, T, und);
        if (und) res = (inter / uni) > T.thr;
        return res;
    }
    return (inter / uni) > T.thr;
}

__device__ __forceinline__ float vx_bcast(float v, int lane_uniform) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// first index in [0,n) with keys[idx] >= target; executed by a whole wave (64-ary search)
__device__ __forceinline__ int vx_lower_bound_wave(const u64 *__restrict__ keys, int n, u64 target) {
    const int lane = vx_lane();
    int lo = 0, hi = n;
    while (hi > lo) {
        long long len = hi - lo;
        int pos = lo + (int)(((long long)(lane + 1) * len) / 65);
        bool ge = keys[pos] >= target;
        u64 m = __ballot(ge);
        if (m == 0) {
            lo = lo + (int)((64LL * len) / 65) + 1;
        } else {
            int f = __builtin_ctzll(m);
            int pf = lo + (int)(((long long)(f + 1) * len) / 65);
            int pp = f > 0 ? lo + (int)(((long long)f * len) / 65) + 1 : lo;
            hi = pf;
            lo = pp;
        }
    }
    return lo;
}

__device__ __forceinline__ void vx_emit_kept(u64 m, bool mine, u64 key1, u64 *__restrict__ keys2,
                                             int *__restrict__ count2) {
    if (m == 0) return;
    int base = 0;
    if (vx_lane() == 0) base = atomicAdd(count2, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (mine) keys2[base + vx_mbcnt(m)] = vx_key2_from_key1(key1);
}

// Ordered emission (class-partitioned path with a run merge instead of a second sort): the kept keys of a segment go
// to keys2[at + rank inside the kept mask], at = the kept boxes of the segment's earlier tiles, so that the
// segment's kept keys form one ascending run.
__device__ __forceinline__ void vx_emit_kept_at(u64 m, bool mine, u64 key1, u64 *__restrict__ keys2, int at) {
    if (mine) keys2[at + vx_mbcnt(m)] = vx_key2_from_key1(key1);
}

__device__ __forceinline__ int vx_wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// kept masks alive[0..ND) of a workgroup sweep -> keys2: through the image's counter (any order), or, with kcount,
// as one ordered run at keys2[0..*kcount)
__device__ __forceinline__ void vx_masks_emit(const u64 *alive, int ND, const u64 *__restrict__ keys,
                                              u64 *__restrict__ keys2, int *count2, int *kcount) {
    const int lane = vx_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (kcount && ND == 0 && threadIdx.x == 0) *kcount = 0;
    for (int u = wave; u < ND; u += nw) {
        const u64 m = alive[u];
        const bool mine = (m >> lane) & 1ull;
        const u64 key = mine ? keys[u * 64 + lane] : 0ull;
        if (kcount) {
            int s = 0;
            for (int q = lane; q < u; q += 64) s += __popcll(alive[q]);
            s = vx_wave_sum_i(s);
            vx_emit_kept_at(m, mine, key, keys2, s);
            if (u == ND - 1 && lane == 0) *kcount = s + __popcll(m);
        } else {
            vx_emit_kept(m, mine, key, keys2, count2);
        }
    }
    __syncthreads();
}

// pair index p = j(j-1)/2 + i (0 <= i < j < 64)  ->  (j << 6) | i, for all 2016 pairs of a 64-box tile
#define VX_NPAIRS64 2016
__device__ __forceinline__ void vx_build_pair_lut(unsigned short *lut) {
    for (int p = threadIdx.x; p < VX_NPAIRS64; p += blockDim.x) {
        int j = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        if (j * (j - 1) / 2 > p) --j;
        if ((j + 1) * j / 2 <= p) ++j;
        int i = p - j * (j - 1) / 2;
        lut[p] = (unsigned short)((j << 6) | i);
    }
}

// Pair-dense wave routine for a segment of n <= 64 boxes (the common per-class size): the
// n(n-1)/2 pairs (i<j) are dealt 64 per step to the lanes, so a 20-box class costs 3 IoU steps
// instead of 20.  Lane l keeps box l; a pair's two boxes arrive by cross-lane reads.  Every
// "i suppresses j" sets bit j of row i of a 64x64 bit matrix in LDS (wmask, 64 words of this wave);
// the greedy order is then resolved with scalar bit operations only.
__device__ __forceinline__ void vx_sweep_pairs(const vx_f4 *__restrict__ sbox, const float *__restrict__ sarea,
                                               const u64 *keys1, int lo, int n, const vx_thr &thr, u64 *wmask,
                                               const unsigned short *pair_lut, u64 *__restrict__ keys2,
                                               int *count2, unsigned long long *iou_count = nullptr, int *kcount = nullptr) {
    const int lane = vx_lane();
    vx_f4 bx = {0.f, 0.f, 0.f, 0.f};
    float ar = 0.f;
    if (lane < n) { bx = sbox[lo + lane]; ar = sarea[lo + lane]; }
    u64 alive = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const int npairs = n * (n - 1) / 2;
    if (iou_count && npairs > 0 && lane == 0) atomicAdd(iou_count, (unsigned long long)npairs);
    if (npairs > 0) {
        wmask[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        for (int p0 = 0; p0 < npairs; p0 += 64) {
            const int p = p0 + lane;
            const bool pv = p < npairs;
            // p = j(j-1)/2 + i, 0 <= i < j : table built once per workgroup (vx_build_pair_lut)
            unsigned ij = pv ? pair_lut[p] : 0u;
            int i = ij & 63, j = ij >> 6;
            float ix1 = __shfl(bx.x, i), iy1 = __shfl(bx.y, i), ix2 = __shfl(bx.z, i), iy2 = __shfl(bx.w, i);
            float ia = __shfl(ar, i);
            float jx1 = __shfl(bx.x, j), jy1 = __shfl(bx.y, j), jx2 = __shfl(bx.z, j), jy2 = __shfl(bx.w, j);
            float ja = __shfl(ar, j);
            bool sup = pv && vx_iou_gt<false>(ix1, iy1, ix2, iy2, ia, jx1, jy1, jx2, jy2, ja, thr);
            if (sup) atomicOr((unsigned long long *)&wmask[i], 1ull << j);
        }
        __builtin_amdgcn_wave_barrier();
        const u64 row = wmask[lane];
        const unsigned rlo = (unsigned)row, rhi = (unsigned)(row >> 32);
        const u64 has = __ballot(row != 0ull);       // only boxes that suppress somebody need a turn
        u64 todo = alive & has;
        while (todo) {
            int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
            u64 mi = ((u64)(unsigned)__builtin_amdgcn_readlane((int)rhi, i) << 32) |
                     (u64)(unsigned)__builtin_amdgcn_readlane((int)rlo, i);
            alive &= ~mi;
            todo = alive & has & ~((2ull << i) - 1ull);
        }
    }
    const bool mine = (alive >> lane) & 1ull;
    u64 key = mine ? keys1[lo + lane] : 0ull;
    if (kcount) {                                    // ordered run at keys2[0..*kcount)
        vx_emit_kept_at(alive, mine, key, keys2, 0);
        if (lane == 0) *kcount = __popcll(alive);
    } else {
        vx_emit_kept(alive, mine, key, keys2, count2);
    }
}

// Wave routine: a segment of n <= 64*WT boxes, everything in registers.
__device__ __forceinline__ void vx_sweep_wave(const vx_f4 *__restrict__ sbox, const float *__restrict__ sarea,
                                              const u64 *__restrict__ keys1, int lo, int n, const vx_thr &thr,
                                              u64 *__restrict__ keys2, int *__restrict__ count2,
                                              unsigned long long *iou_count = nullptr, int *kcount = nullptr) {
    const int lane = vx_lane();
    unsigned long long tests = 0ull;
    float x1[VX_SWEEP_WT], y1[VX_SWEEP_WT], x2[VX_SWEEP_WT], y2[VX_SWEEP_WT], ar[VX_SWEEP_WT];
    bool alive[VX_SWEEP_WT];
#pragma unroll
    for (int u = 0; u < VX_SWEEP_WT; ++u) {
        int o = u * 64 + lane;
        alive[u] = o < n;
        vx_f4 bx = {0.f, 0.f, 0.f, 0.f};
        float a = 0.f;
        if (alive[u]) { bx = sbox[lo + o]; a = sarea[lo + o]; }
        x1[u] = bx.x; y1[u] = bx.y; x2[u] = bx.z; y2[u] = bx.w; ar[u] = a;
    }
#pragma unroll
    for (int t = 0; t < VX_SWEEP_WT; ++t) {
        if (t * 64 < n) {
            u64 todo = __ballot(alive[t]);
            while (todo) {
                int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                float bx1 = vx_bcast(x1[t], i), by1 = vx_bcast(y1[t], i);
                float bx2 = vx_bcast(x2[t], i), by2 = vx_bcast(y2[t], i), ba = vx_bcast(ar[t], i);
                if (iou_count) {                       // live lanes behind box i (measurement only)
                    tests += __popcll(todo & ~((2ull << i) - 1ull));
#pragma unroll
                    for (int u = 1; u < VX_SWEEP_WT; ++u)
                        if (u > t && u * 64 < n) tests += __popcll(__ballot(alive[u]));
                }
#pragma unroll
                for (int u = t; u < VX_SWEEP_WT; ++u) {
                    if (u * 64 < n) {
                        bool sup = vx_iou_gt<true>(bx1, by1, bx2, by2, ba, x1[u], y1[u], x2[u], y2[u], ar[u], thr);
                        if (u == t) sup = sup && (lane > i);
                        alive[u] = alive[u] && !sup;
                    }
                }
                u64 m = __ballot(alive[t]);
                todo = m & ~((2ull << i) - 1ull);
            }
        }
    }
    int at = 0;
#pragma unroll
    for (int u = 0; u < VX_SWEEP_WT; ++u) {
        if (u * 64 < n) {
            u64 m = __ballot(alive[u]);
            u64 key = alive[u] ? keys1[lo + u * 64 + lane] : 0ull;
            if (kcount) { vx_emit_kept_at(m, alive[u], key, keys2, at); at += __popcll(m); }
            else vx_emit_kept(m, alive[u], key, keys2, count2);
        }
    }
    if (kcount && lane == 0) *kcount = at;
    if (iou_count && tests && lane == 0) atomicAdd(iou_count, tests);
}

// ------------------------------------------------------------------------------------------------
// Chain sweep: a class segment of n > 256 boxes (score order; class-offset boxes in global memory, sbox), the whole
// workgroup (W waves, a power of two).  Tile u (64 boxes) belongs to wave u % W.  The greedy order is a chain over
// tiles: the owner of tile t+1 applies the kept boxes of tile t to it, resolves it with a scalar bit walk over the
// tile's precomputed suppression rows (lane i = row i: later boxes of the tile that box i suppresses; built ahead of
// time, off the chain), writes the kept boxes compacted into an LDS ring slot and publishes (S.alive[t+1] now holds
// the kept mask; then S.ready).  Every wave consumes the published tiles in order and applies them to its own later
// tiles, two tiles at a time (boxes streamed from L2 one pair ahead, alive masks in LDS), whenever it gets there.
// No workgroup barrier inside the sweep and the resolver role rotates, so the load balances itself.  Waiting waves
// poll S.ready in LDS; a wave's progress depends on earlier tiles only and all waves of a workgroup are resident,
// so the chain always advances; every wave leaves the loop when S.ready reaches the tile count (last tile resolved,
// or the top_k stop) or when it has no tile left.  The ring is 2W tiles deep: a wave is at most W+1 tiles behind
// the chain (the chain needs it every W tiles), so a slot is never rewritten while somebody still reads it.
//
// IoU test of the apply loop (12 VALU): for two tame boxes (vx_tame) "inter1 < lo*uni1" with the height left
// unclamped proves the reference's predicate false (h < 0: inter1 <= 0 < lo*uni1; h >= 0: the bracket's own "lo"
// side).  Four kept boxes per trip: their candidate values (inter1 - lo*uni1) are max-reduced per lane, so a tile
// costs 4 x 12 VALU + one ballot and one scalar branch per trip (tools/micro/sweep_loop.hip: 1.4-1.7x the lane-tests
// per clock of a branch per pair); only lanes that fail the cheap test -- each box about once, when it dies -- repeat
// the trip with the exact routine (vx_iou_gt: bracket, then the IEEE division), as do untame lanes, tiles with a wild
// kept box, or a threshold outside the bracket's range.
#define VX_ASYNC_TRIP 4
#define VX_ASYNC_MAXTILES_SP 128           // spatial sweep: segments up to 8192 boxes (fewer with fewer waves)

template <int MAXTILES>
struct vx_async_smem_t {
    u64 alive[MAXTILES];                       // tile not resolved yet: lanes still alive; resolved: the kept mask
    unsigned char wild[MAXTILES];              // some kept box of the tile is not tame
    int ready;                                 // tiles resolved so far; the tile count once the sweep is over
    int ndone;                                 // tiles whose kept mask is final (== ready unless the top_k stop hit)
    int ktot;
};
typedef vx_async_smem_t<VX_SORT_CHMAX / 64> vx_async_smem;          // single-launch / partitioned paths: <= 16384 boxes
typedef vx_async_smem_t<2048> vx_async_smem_big;                    // general path: <= 131072 boxes per segment
// LDS ring of compacted kept boxes: 2W tiles x 64 x (16 + 4) bytes (+ one trip of overshoot for the loop's prefetch)
__host__ __device__ __forceinline__ size_t vx_async_ring_bytes(int waves) { return ((size_t)2 * waves * 64 + 8) * 20; }

__device__ __forceinline__ u64 vx_uniform64(u64 v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ float vx_async_cand(const vx_f4 &K, float ka, const vx_f4 &bx, float a, float lo) {
    float xx1 = vx_vmax<false>(K.x, bx.x);
    float yy1 = vx_vmax<false>(K.y, bx.y);
    float xx2 = vx_vmin<false>(K.z, bx.z);
    float yy2 = vx_vmin<false>(K.w, bx.w);
    float w = vx_vmax<false>(xx2 - xx1, 0.0f);
    float inter = w * (yy2 - yy1);
    float uni = ka + a - inter;
    return inter - lo * uni;
}

// lanes of `am` that the wave-uniform box K suppresses (one pair at a time: the row builder)
__device__ __forceinline__ u64 vx_async_test(const vx_f4 &K, float ka, bool kwild, const vx_f4 &bx, float a, u64 unt,
                                             u64 am, const vx_thr &T) {
    u64 slow = __ballot(!(vx_async_cand(K, ka, bx, a, T.lo) < 0.0f));
    slow = (kwild || !T.fast) ? am : ((slow | unt) & am);
    u64 sup = 0;
    if (slow) sup = __ballot(vx_iou_gt<false>(K.x, K.y, K.z, K.w, ka, bx.x, bx.y, bx.z, bx.w, a, T)) & slow;
    return sup;
}

// kept boxes of a ring slot (kc of them, padded to whole trips with boxes that meet nothing) -> N column tiles
template <int N>
__device__ __forceinline__ void vx_async_apply(const vx_f4 *kb, const float *kar, int kc, bool all_slow, const vx_f4 *bx,
                                               const float *ar, u64 *alive, const u64 *unt, const vx_thr &T) {
    vx_f4 K[VX_ASYNC_TRIP], Kn[VX_ASYNC_TRIP];
    vx_v4 ka, kan;
#pragma unroll
    for (int q = 0; q < VX_ASYNC_TRIP; ++q) K[q] = kb[q];
    ka = *reinterpret_cast<const vx_v4 *>(kar);
    for (int j = 0; j < kc; j += VX_ASYNC_TRIP) {
#pragma unroll
        for (int q = 0; q < VX_ASYNC_TRIP; ++q) Kn[q] = kb[j + VX_ASYNC_TRIP + q];       // the ring is padded for this overshoot
        kan = *reinterpret_cast<const vx_v4 *>(kar + j + VX_ASYNC_TRIP);
#pragma unroll
        for (int s = 0; s < N; ++s) {
            float m = vx_async_cand(K[0], ka.x, bx[s], ar[s], T.lo);
            m = vx_vmax<false>(m, vx_async_cand(K[1], ka.y, bx[s], ar[s], T.lo));
            m = vx_vmax<false>(m, vx_async_cand(K[2], ka.z, bx[s], ar[s], T.lo));
            m = vx_vmax<false>(m, vx_async_cand(K[3], ka.w, bx[s], ar[s], T.lo));
            u64 slow = __ballot(!(m < 0.0f));
            slow = all_slow ? alive[s] : ((slow | unt[s]) & alive[s]);
            if (slow) {
                // rare (about once per box, when it dies): the exact predicate for the trip's boxes -- bracket first, the
                // IEEE division only for a pair the bracket cannot decide.  The padding boxes of a last trip meet
                // nothing: inter == 0 fails every form of the test.
                const float kaq[VX_ASYNC_TRIP] = {ka.x, ka.y, ka.z, ka.w};
                bool sup = false, und = false;
                float in_[VX_ASYNC_TRIP], un_[VX_ASYNC_TRIP];
#pragma unroll
                for (int q = 0; q < VX_ASYNC_TRIP; ++q) {
                    vx_iou_parts<false>(K[q].x, K[q].y, K[q].z, K[q].w, kaq[q], bx[s].x, bx[s].y, bx[s].z, bx[s].w, ar[s],
                                        in_[q], un_[q]);
                    sup = vx_iou_bracket(in_[q], un_[q], T, und) || sup;
                }
                if (__ballot(und && ((slow >> vx_lane()) & 1ull)) != 0ull || !T.fast) {
                    sup = false;
#pragma unroll
                    for (int q = 0; q < VX_ASYNC_TRIP; ++q) sup = sup || (in_[q] / un_[q]) > T.thr;
                }
                alive[s] &= ~(__ballot(sup) & slow);
            }
        }
#pragma unroll
        for (int q = 0; q < VX_ASYNC_TRIP; ++q) K[q] = Kn[q];
        ka = kan;
    }
}

// ring = vx_async_ring_bytes(W) bytes of LDS (16-byte aligned).  n <= 64 * MAXTILES.  Ends synchronised with the
// segment's kept masks in S.alive[0 .. S.ndone) (cut to the first top_k kept boxes when top_k > 0).
template <int MAXTILES>
__device__ __forceinline__ void vx_sweep_async(vx_async_smem_t<MAXTILES> &S, void *ring, const vx_f4 *__restrict__ sbox,
                                               int n, const vx_thr &T, int top_k, unsigned long long *iou_count,
                                               u64 *dbg = nullptr) {
    // wave-uniform values are made so explicitly (readfirstlane): the compiler treats threadIdx.x >> 6 and LDS loads
    // as divergent, and the masks / loop bounds derived from them would end up in vector registers
    const int lane = vx_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = blockDim.x >> 6;
    n = __builtin_amdgcn_readfirstlane(n);
    const int NT = (n + 63) >> 6;
    const int RT = 2 * W;                                         // ring depth in tiles
    vx_f4 *rbox = reinterpret_cast<vx_f4 *>(ring);
    float *rar = reinterpret_cast<float *>(rbox + RT * 64 + 8);
    const bool all_slow0 = !T.fast;
    __syncthreads();                                              // S and the ring's LDS are free again
    for (int i = threadIdx.x; i < NT; i += blockDim.x) {
        const int rem = n - i * 64;
        S.alive[i] = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
        S.wild[i] = 0;
    }
    if (threadIdx.x == 0) { S.ready = 0; S.ndone = 0; S.ktot = 0; }
    __syncthreads();
    unsigned long long tests = 0ull;
    // the next tile this wave resolves: its boxes, areas, untame lanes and suppression rows stay in registers
    int nxt = w;
    vx_f4 bxn = {0.f, 0.f, 0.f, 0.f};
    float arn = 0.f;
    u64 untn = 0ull;
    unsigned row_lo = 0u, row_hi = 0u;
    auto load_tile = [&](int u) __attribute__((always_inline)) -> vx_f4 {
        const int o = u * 64 + lane;
        vx_f4 b = {0.f, 0.f, 0.f, 0.f};
        if (u < NT && o < n) b = sbox[o];
        return b;
    };
    // rows of tile nxt (only those of still-alive boxes: the others are never read)
    auto make_rows = [&]() __attribute__((always_inline)) {
        bxn = load_tile(nxt);
        arn = (bxn.z - bxn.x) * (bxn.w - bxn.y);
        row_lo = 0u; row_hi = 0u;
        if (nxt >= NT) return;
        const u64 al = vx_uniform64(S.alive[nxt]);
        untn = __ballot(!vx_tame(bxn, arn)) & al;
        u64 todo = al;
        while (todo) {
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (todo == 0ull) break;                              // the last alive box has nobody behind it
            vx_f4 K;
            K.x = vx_bcast(bxn.x, i); K.y = vx_bcast(bxn.y, i); K.z = vx_bcast(bxn.z, i); K.w = vx_bcast(bxn.w, i);
            const float ka = vx_bcast(arn, i);
            const u64 m = vx_async_test(K, ka, (untn >> i) & 1ull, bxn, arn, untn, todo, T);
            if (lane == i) { row_lo = (unsigned)m; row_hi = (unsigned)(m >> 32); }
        }
        const unsigned long long na = __popcll(al);
        tests += na * (na - (na ? 1 : 0)) / 2;
    };
    // tile t (= nxt), every earlier kept box applied (alive mask a): greedy inside the tile, publish
    auto resolve = [&](int t, u64 a) __attribute__((always_inline)) {
        u64 todo = a;
        while (todo) {
            const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
            const u64 ri = ((u64)(unsigned)__builtin_amdgcn_readlane((int)row_hi, i) << 32) |
                           (u64)(unsigned)__builtin_amdgcn_readlane((int)row_lo, i);
            a &= ~ri;
            todo = a & ~((2ull << i) - 1ull);
        }
        int ktot = __builtin_amdgcn_readfirstlane(S.ktot), nk = __popcll(a);
        bool stop = false;
        if (top_k > 0 && ktot + nk >= top_k) {                   // only the segment's first top_k kept boxes matter
            int excess = ktot + nk - top_k;
            while (excess-- > 0) a &= ~(1ull << (63 - __builtin_clzll(a)));
            nk = top_k - ktot;
            stop = true;
        }
        const int slot = t & (RT - 1);
        if ((a >> lane) & 1ull) {
            const int pos = slot * 64 + vx_mbcnt(a);
            rbox[pos] = bxn;
            rar[pos] = arn;
        }
        if (lane < VX_ASYNC_TRIP) {                              // pad the slot to whole trips with a box that meets nothing
            const int pos = nk + lane;
            if (pos < ((nk + VX_ASYNC_TRIP - 1) & ~(VX_ASYNC_TRIP - 1))) {
                const vx_f4 far = {1e30f, 1e30f, 1e30f, 1e30f};
                rbox[slot * 64 + pos] = far;
                rar[slot * 64 + pos] = 0.0f;
            }
        }
        if (lane == 0) { S.alive[t] = a; S.wild[t] = (a & untn) != 0ull; S.ktot = ktot + nk; S.ndone = t + 1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            __hip_atomic_store(&S.ready, (stop || t + 1 >= NT) ? NT : t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // diagnostics (dbg != nullptr only while profiling): cycles this wave spent waiting / applying / on the chain
    u64 c_wait = 0, c_apply = 0, c_chain = 0, c_rows = 0, c0 = dbg ? clock64() : 0, cs = c0;
#define VX_LAP(acc) do { if (dbg) { u64 now_ = clock64(); acc += now_ - cs; cs = now_; } } while (0)
    make_rows();
    if (w == 0) { resolve(0, vx_uniform64(S.alive[0])); nxt += W; make_rows(); }
    VX_LAP(c_rows);
    for (int t = 0; t + 1 < NT; ++t) {
        if (nxt >= NT) break;                                    // no tile left to resolve or to apply to (nxt > t always)
        int r;
        while ((r = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(&S.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) <= t)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        VX_LAP(c_wait);
        if (r >= NT) break;                                      // every tile is final (last one resolved, or the top_k stop)
        const int slot = t & (RT - 1);
        const int kc = __builtin_amdgcn_readfirstlane(__popcll(S.alive[t]));
        const bool all_slow = all_slow0 || __builtin_amdgcn_readfirstlane((int)S.wild[t]) != 0;
        const vx_f4 *kb = rbox + slot * 64;
        const float *kar = rar + slot * 64;
        int u = nxt;                                             // this wave's first tile behind t
        if (t + 1 == nxt) {                                      // this wave owns tile t+1: the chain's next link
            __builtin_amdgcn_s_setprio(3);
            u64 a = vx_uniform64(S.alive[nxt]);
            tests += (unsigned long long)kc * __popcll(a);
            if (kc > 0 && a) vx_async_apply<1>(kb, kar, kc, all_slow, &bxn, &arn, &a, &untn, T);
            resolve(nxt, a);
            __builtin_amdgcn_s_setprio(0);
            VX_LAP(c_chain);
            nxt += W;
            u = nxt;
        }
        if (kc > 0) {
            // own tiles behind the chain, two per trip; the next pair's boxes are in flight meanwhile
            vx_f4 B[2], Bn[2];
            B[0] = load_tile(u); B[1] = load_tile(u + W);
            for (; u < NT; u += 2 * W) {
                // the top_k stop ends the sweep for everybody: the far tiles are not needed any more
                if (top_k > 0 && __builtin_amdgcn_readfirstlane(
                        __hip_atomic_load(&S.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= NT) break;
                Bn[0] = load_tile(u + 2 * W); Bn[1] = load_tile(u + 3 * W);
                u64 al[2], un[2];
                al[0] = vx_uniform64(S.alive[u]);
                al[1] = u + W < NT ? vx_uniform64(S.alive[u + W]) : 0ull;
                if (al[0] | al[1]) {
                    float ar[2];
                    ar[0] = (B[0].z - B[0].x) * (B[0].w - B[0].y);
                    ar[1] = (B[1].z - B[1].x) * (B[1].w - B[1].y);
                    un[0] = __ballot(!vx_tame(B[0], ar[0])) & al[0];
                    un[1] = __ballot(!vx_tame(B[1], ar[1])) & al[1];
                    tests += (unsigned long long)kc * (__popcll(al[0]) + __popcll(al[1]));
                    const u64 b0 = al[0], b1 = al[1];
                    if (al[1] == 0ull) vx_async_apply<1>(kb, kar, kc, all_slow, B, ar, al, un, T);
                    else if (al[0] == 0ull) vx_async_apply<1>(kb, kar, kc, all_slow, B + 1, ar + 1, al + 1, un + 1, T);
                    else vx_async_apply<2>(kb, kar, kc, all_slow, B, ar, al, un, T);
                    if (lane == 0) {
                        if (al[0] != b0) S.alive[u] = al[0];
                        if (al[1] != b1) S.alive[u + W] = al[1];
                    }
                }
                B[0] = Bn[0]; B[1] = Bn[1];
            }
        }
        VX_LAP(c_apply);
        if (t + 1 + W == nxt) { make_rows(); VX_LAP(c_rows); }   // just resolved a tile: rows of the next own one
    }
    if (iou_count && lane == 0 && tests) atomicAdd(iou_count, tests);        // wave-uniform count of useful lane-tests
    if (dbg && lane == 0 && w < 16) {
        u64 *d = dbg + w * 8;
        d[0] = c_wait; d[1] = c_apply; d[2] = c_chain; d[3] = c_rows; d[4] = clock64() - c0; d[5] = tests;
    }
#undef VX_LAP
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Spatial sweep: the chain sweep's arithmetic with most of its IoU tests culled.  A kept box can only suppress boxes
// it overlaps, but in score order the boxes of a tile are scattered over the whole image.  So the "column" side of
// the sweep is re-tiled by position: the segment's boxes are counting-sorted by the Morton code of their centre cell
// (16 x 16 cells over the segment's extent), 64 consecutive boxes form a spatial tile with a bounding box, and a kept
// box is tested against a spatial tile only if it overlaps that bounding box (otherwise every intersection in the tile
// is empty and the reference's predicate is false for thr > 0).  The chain side keeps the score order:
//   wave 0           owns the chain: in phase t it applies the kept boxes of tile t to tile t+1 (score layout), resolves
//                    tile t+1 greedily and leaves its kept boxes, compacted, in the other half of a double buffer
//   waves 1 .. W-1   share the spatial tiles (round-robin; boxes, score ranks and bounding boxes in LDS): in phase t they
//                    apply the kept boxes of tile t to the boxes of rank >= 64 (t+2) and clear the alive bit (LDS, indexed
//                    by rank) of every box that dies
// one workgroup barrier per phase.  Alive state is kept by rank, so the chain reads the word of its next tile directly.
// Needs every box tame, a threshold inside the bracket's range and 20 bytes of scratch LDS per box (the caller falls
// back to the chain sweep otherwise).
// A/B on MI355X (RetinaFace b128 nms_image / FPN-SSD b64 sweep / stress b32 sweep, ms; baseline 0.648 / 0.61 / 0.605):
//   in-tile resolve by a sequential scan on the chain wave instead of prebuilt rows (8-wave workgroups): - / 0.655 / 0.614
//   s_setprio 3 on the chain wave: 0.647 / 0.622 / 0.608 (no effect: the workgroup is bound by its instruction count,
//     the appliers' trips are 60 % of it, not by the chain's latency)
//   kept boxes of a trip by v_readlane out of the wave's registers instead of LDS reads: 0.800 / 0.749 / 0.725
//   pair-dense visits (the (kept box, live box) pairs of a visit dealt 64 per step to the lanes through two small
//     per-wave index lists in LDS, one test per lane: 83 % useful lane-tests instead of ~30 %): 0.618 / 0.615 / 0.655 --
//     a step costs ~40 instructions (index arithmetic, six per-lane LDS reads) for 64 tests against ~100 for a trip of
//     256 lane-tests, so the instruction count of an applier wave only falls from ~65 k to ~53 k
//   candidate test of two kept boxes per packed fp32 instruction (v_pk_add / v_pk_mul / v_pk_fma_f32; min/max have
//     no packed form): 35 instead of 51 VALU instructions per trip on paper, 0.672 / 0.615 / 0.609 measured (the
//     register-pair shuffling around the packed operands eats the saving)
#define VX_SP_CELLS 256

struct vx_spatial_smem {
    u64 alive[VX_ASYNC_MAXTILES_SP];
    vx_f4 kb[2][64 + VX_ASYNC_TRIP];               // kept boxes of the current / next chain tile (+ null boxes for padding)
    float kar[2][64 + VX_ASYNC_TRIP];
    int kc[2];
    int ktot, stop, ndone;
    vx_f4 tb[VX_ASYNC_MAXTILES_SP];                 // bounding boxes of the spatial tiles
    u64 rows[2][64];                                // suppression rows of the next two chain tiles
    int hist[VX_SP_CELLS];
    int ext[4];                                     // min x, min y, max x, max y of the box centres (order-preserving ints)
    int untame;
};

__device__ __forceinline__ int vx_f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float vx_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__device__ __forceinline__ unsigned vx_morton4(unsigned x, unsigned y) {       // 4 + 4 bits interleaved
    x = (x | (x << 2)) & 0x33u; x = (x | (x << 1)) & 0x55u;
    y = (y | (y << 2)) & 0x33u; y = (y | (y << 1)) & 0x55u;
    return x | (y << 1);
}

// one trip: lanes of `act` that one of the four wave-uniform boxes K suppresses (tame boxes, threshold in range)
__device__ __forceinline__ u64 vx_trip_test(const vx_f4 *K, const vx_v4 &ka, const vx_f4 &bx, float a, u64 act, const vx_thr &T) {
    float m = vx_async_cand(K[0], ka.x, bx, a, T.lo);
    m = vx_vmax<false>(m, vx_async_cand(K[1], ka.y, bx, a, T.lo));
    m = vx_vmax<false>(m, vx_async_cand(K[2], ka.z, bx, a, T.lo));
    m = vx_vmax<false>(m, vx_async_cand(K[3], ka.w, bx, a, T.lo));
    const u64 slow = __ballot(!(m < 0.0f)) & act;
    if (!slow) return 0ull;
    const float kaq[4] = {ka.x, ka.y, ka.z, ka.w};
    bool sup = false, und = false;
    float in_[4], un_[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        vx_iou_parts<false>(K[q].x, K[q].y, K[q].z, K[q].w, kaq[q], bx.x, bx.y, bx.z, bx.w, a, in_[q], un_[q]);
        sup = vx_iou_bracket(in_[q], un_[q], T, und) || sup;
    }
    if (__ballot(und && ((slow >> vx_lane()) & 1ull)) != 0ull) {
        sup = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) sup = sup || (in_[q] / un_[q]) > T.thr;
    }
    return __ballot(sup) & slow;
}

// lds: scratch LDS of lds_bytes (16-byte aligned) for the spatially ordered copy of the segment (20 bytes per box).
// Returns false (nothing done) when the segment does not qualify; else ends synchronised with the kept masks in
// S.alive[0 .. S.ndone).
__device__ __noinline__ bool vx_sweep_spatial(vx_spatial_smem &S, void *lds, size_t lds_bytes, const vx_f4 *__restrict__ sbox,
                                              int n, const vx_thr T /* by value: the routine is a real call */, int top_k,
                                              unsigned long long *iou_count, u64 *dbg = nullptr) {
    const int lane = vx_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = blockDim.x >> 6, nt = blockDim.x;
    n = __builtin_amdgcn_readfirstlane(n);
    const int NT = (n + 63) >> 6;
    const u64 c00 = dbg ? clock64() : 0;
    u64 c_work = 0, c_wait = 0, c_setup = 0, c_rows = 0, cs = 0, n_vis = 0, n_trip = 0;
    if (!T.fast || W < 2 || (64 % W) != 0 || NT > 32 * (W - 1) || NT > VX_ASYNC_MAXTILES_SP || (size_t)NT * 64 * 20 > lds_bytes)
        return false;
    vx_f4 *lbox = reinterpret_cast<vx_f4 *>(lds);                       // [NT*64] boxes in spatial order
    int *lrank = reinterpret_cast<int *>(lbox + NT * 64);               // [NT*64] their score ranks (-1: padding)
    __syncthreads();
    for (int i = threadIdx.x; i < VX_SP_CELLS; i += nt) S.hist[i] = 0;
    for (int i = threadIdx.x; i < NT; i += nt) {
        const int rem = n - i * 64;
        S.alive[i] = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    }
    for (int i = n + threadIdx.x; i < NT * 64; i += nt) lrank[i] = -1;
    if (threadIdx.x == 0) {
        S.ext[0] = S.ext[1] = 0x7fffffff; S.ext[2] = S.ext[3] = (int)0x80000000;
        S.untame = 0; S.ktot = 0; S.stop = 0; S.ndone = 0; S.kc[0] = S.kc[1] = 0;
    }
    if (threadIdx.x < 2 * VX_ASYNC_TRIP) {          // the padding boxes of both buffers meet nothing
        const vx_f4 far = {1e30f, 1e30f, 1e30f, 1e30f};
        S.kb[threadIdx.x >> 2][64 + (threadIdx.x & 3)] = far;
        S.kar[threadIdx.x >> 2][64 + (threadIdx.x & 3)] = 0.0f;
    }
    __syncthreads();
    // extent of the centres, tameness
    {
        float mnx = 3e38f, mny = 3e38f, mxx = -3e38f, mxy = -3e38f;
        bool bad = false;
        for (int i = threadIdx.x; i < n; i += nt) {
            const vx_f4 b = sbox[i];
            bad |= !vx_tame(b, (b.z - b.x) * (b.w - b.y));
            const float cx = b.x + b.z, cy = b.y + b.w;
            mnx = fminf(mnx, cx); mxx = fmaxf(mxx, cx); mny = fminf(mny, cy); mxy = fmaxf(mxy, cy);
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
    if (S.untame) { __syncthreads(); return false; }
    const float ex0 = vx_ord2f(S.ext[0]), ey0 = vx_ord2f(S.ext[1]);
    const float sx = 16.0f / fmaxf(vx_ord2f(S.ext[2]) - ex0, 1e-30f), sy = 16.0f / fmaxf(vx_ord2f(S.ext[3]) - ey0, 1e-30f);
    auto cell_of = [&](const vx_f4 &b) __attribute__((always_inline)) -> int {
        const int cx = min(15, max(0, (int)((b.x + b.z - ex0) * sx))), cy = min(15, max(0, (int)((b.y + b.w - ey0) * sy)));
        return (int)vx_morton4((unsigned)cx, (unsigned)cy);
    };
    for (int i = threadIdx.x; i < n; i += nt) atomicAdd(&S.hist[cell_of(sbox[i])], 1);
    __syncthreads();
    if (w == 0) {                                    // exclusive scan of the 256 cell counts: 4 per lane
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = S.hist[lane * 4 + q]; tot += v[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        int run = incl - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) { S.hist[lane * 4 + q] = run; run += v[q]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += nt) {      // scatter into cell order (any order inside a cell)
        const vx_f4 b = sbox[i];
        const int pos = atomicAdd(&S.hist[cell_of(b)], 1);
        lbox[pos] = b;
        lrank[pos] = i;
    }
    __syncthreads();
    for (int v = w; v < NT; v += W) {                // bounding boxes of the spatial tiles
        vx_f4 q = {3e38f, 3e38f, -3e38f, -3e38f};
        if (lrank[v * 64 + lane] >= 0) q = lbox[v * 64 + lane];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            q.x = fminf(q.x, __shfl_xor(q.x, d)); q.y = fminf(q.y, __shfl_xor(q.y, d));
            q.z = fmaxf(q.z, __shfl_xor(q.z, d)); q.w = fmaxf(q.w, __shfl_xor(q.w, d));
        }
        if (lane == 0) S.tb[v] = q;
    }
    unsigned long long tests = 0ull;
    unsigned gone = 0u;                             // bit s: this wave's s-th spatial tile has nobody left behind the chain
    // chain tiles in score layout: cb = the tile wave 0 resolves next, nb = the tile after it (whose suppression rows
    // every wave helps to build one phase ahead: row i = later boxes of the tile that box i suppresses, by wave i % W)
    vx_f4 cb = {0.f, 0.f, 0.f, 0.f}, nb = {0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int t) __attribute__((always_inline)) -> vx_f4 {
        const int o = t * 64 + lane;
        vx_f4 b = {0.f, 0.f, 0.f, 0.f};
        if (t < NT && o < n) b = sbox[o];
        return b;
    };
    auto build_rows = [&](int t, const vx_f4 &tb_) __attribute__((always_inline)) {
        if (t >= NT) return;
        const float ta = (tb_.z - tb_.x) * (tb_.w - tb_.y);
        const u64 snap = vx_uniform64(S.alive[t]);   // a superset of the tile's alive mask when it is resolved
        u64 mine = 0ull;                             // rows i = w, w + W, ... (W divides 64)
        for (int i = w; i < 64; i += W) mine |= snap & (1ull << i);
        while (mine) {
            const int i = __builtin_ctzll(mine);
            mine &= mine - 1ull;
            const u64 am = snap & ~((2ull << i) - 1ull);
            u64 m = 0ull;
            if (am) {
                vx_f4 K;
                K.x = vx_bcast(tb_.x, i); K.y = vx_bcast(tb_.y, i); K.z = vx_bcast(tb_.z, i); K.w = vx_bcast(tb_.w, i);
                tests += __popcll(am);
                m = vx_async_test(K, vx_bcast(ta, i), false, tb_, ta, 0ull, am, T);
            }
            if (lane == 0) S.rows[t & 1][i] = m;
        }
    };
    // wave 0: tile t (boxes in cb), every earlier kept box applied (mask a): bit walk over the prebuilt rows, kept boxes
    // compacted into buffer t & 1
    auto resolve = [&](int t, u64 a) __attribute__((always_inline)) {
        const float ca = (cb.z - cb.x) * (cb.w - cb.y);
        const u64 row = S.rows[t & 1][lane];
        const unsigned rlo = (unsigned)row, rhi = (unsigned)(row >> 32);
        u64 todo = a;
        while (todo) {
            const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
            const u64 ri = ((u64)(unsigned)__builtin_amdgcn_readlane((int)rhi, i) << 32) |
                           (u64)(unsigned)__builtin_amdgcn_readlane((int)rlo, i);
            a &= ~ri;
            todo = a & ~((2ull << i) - 1ull);
        }
        int ktot = S.ktot, nk = __popcll(a);
        bool stop = t + 1 >= NT;
        if (top_k > 0 && ktot + nk >= top_k) {       // only the segment's first top_k kept boxes matter
            int excess = ktot + nk - top_k;
            while (excess-- > 0) a &= ~(1ull << (63 - __builtin_clzll(a)));
            nk = top_k - ktot;
            stop = true;
        }
        const int buf = t & 1;
        if ((a >> lane) & 1ull) {
            const int pos = vx_mbcnt(a);
            S.kb[buf][pos] = cb;
            S.kar[buf][pos] = ca;
        }
        if (lane < VX_ASYNC_TRIP && nk + lane < 64) {    // null boxes behind the kept ones: the chain's apply loop reads whole trips
            const vx_f4 far = {1e30f, 1e30f, 1e30f, 1e30f};
            S.kb[buf][nk + lane] = far;
            S.kar[buf][nk + lane] = 0.0f;
        }
        if (lane == 0) { S.alive[t] = a; S.kc[buf] = nk; S.ktot = ktot + nk; S.ndone = t + 1; if (stop) S.stop = 1; }
    };
    cb = load_tile(0);
    nb = load_tile(1);
    build_rows(0, cb);
    build_rows(1, nb);
    __syncthreads();
    if (w == 0) resolve(0, vx_uniform64(S.alive[0]));
    cb = nb;
    nb = load_tile(2);
    __syncthreads();
    if (dbg) { cs = clock64(); c_setup = cs - c00; }
    for (int t = 0; t + 1 < NT; ++t) {
        if (S.stop) break;                           // uniform: written before the last barrier
        const int buf = t & 1;
        const int kc = __builtin_amdgcn_readfirstlane(S.kc[buf]);
        // here: cb = tile t+1, nb = tile t+2
        if (w == 0) {
            u64 a = vx_uniform64(S.alive[t + 1]);
            const float ca = (cb.z - cb.x) * (cb.w - cb.y);
            const u64 unt0 = 0ull;
            tests += (unsigned long long)kc * __popcll(a);
            if (kc > 0 && a) vx_async_apply<1>(S.kb[buf], S.kar[buf], kc, false, &cb, &ca, &a, &unt0, T);
            resolve(t + 1, a);
        } else if (kc > 0) {
            const int thr_rank = 64 * (t + 2);
            vx_f4 Kl = {0.f, 0.f, 0.f, 0.f};
            if (lane < kc) Kl = S.kb[buf][lane];
            // this wave's spatial tiles v = (w-1) + s (W-1); the next tile's bounding box is in flight during a visit
            int v = w - 1;
            vx_f4 tbb = S.tb[min(v, NT - 1)];
#pragma nounroll
            for (int s = 0; v < NT; ++s, v += W - 1) {
                const vx_f4 cur = tbb;
                tbb = S.tb[min(v + W - 1, NT - 1)];
                if ((gone >> s) & 1u) continue;
                // kept boxes that reach into the tile's bounding box (lane = kept box)
                u64 m = __ballot(lane < kc && !(Kl.z <= cur.x || Kl.x >= cur.z || Kl.w <= cur.y || Kl.y >= cur.w));
                if (m == 0ull) continue;
                const int r = lrank[v * 64 + lane];
                const vx_f4 bx = lbox[v * 64 + lane];
                const u64 aw = S.alive[max(r, 0) >> 6];
                u64 act = __ballot(r >= thr_rank && ((aw >> (r & 63)) & 1ull));
                if (act == 0ull) { gone |= 1u << s; continue; }      // nobody left behind the chain, now or later
                const float ar = (bx.z - bx.x) * (bx.w - bx.y);
                u64 dead = 0ull;
                if (dbg) ++n_vis;
                while (m != 0ull && act != 0ull) {
                    if (dbg) ++n_trip;
                    int idx[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        idx[q] = 64 + q;                             // a null box unless another kept box is left
                        if (m != 0ull) { idx[q] = __builtin_ctzll(m); m &= m - 1ull; }
                    }
                    vx_f4 K[4];
                    vx_v4 ka;
#pragma unroll
                    for (int q = 0; q < 4; ++q) K[q] = S.kb[buf][idx[q]];
                    ka.x = S.kar[buf][idx[0]]; ka.y = S.kar[buf][idx[1]]; ka.z = S.kar[buf][idx[2]]; ka.w = S.kar[buf][idx[3]];
                    tests += 4ull * __popcll(act);
                    const u64 sup = vx_trip_test(K, ka, bx, ar, act, T);
                    act &= ~sup;
                    dead |= sup;
                }
                if ((dead >> lane) & 1ull) atomicAnd(&S.alive[r >> 6], ~(1ull << (r & 63)));
            }
        }
        if (dbg) { u64 now = clock64(); c_work += now - cs; cs = now; }
        build_rows(t + 2, nb);                       // for the phase after the next one's resolve
        cb = nb;
        nb = load_tile(t + 3);
        if (dbg) { u64 now = clock64(); c_rows += now - cs; cs = now; }
        __syncthreads();
        if (dbg) { u64 now = clock64(); c_wait += now - cs; cs = now; }
    }
    if (iou_count && lane == 0 && tests) atomicAdd(iou_count, tests);
    if (dbg && lane == 0 && w < 16) {
        u64 *d = dbg + w * 8;
        d[0] = c_wait; d[1] = c_work; d[2] = c_setup; d[3] = c_rows; d[4] = clock64() - c00; d[5] = tests; d[6] = n_vis; d[7] = n_trip;
    }
    __syncthreads();
    return true;
}

// kept masks of vx_sweep_spatial -> keys2
__device__ __forceinline__ void vx_spatial_emit(const vx_spatial_smem &S, const u64 *__restrict__ keys,
                                                u64 *__restrict__ keys2, int *count2, int *kcount = nullptr) {
    vx_masks_emit(S.alive, S.ndone, keys, keys2, count2, kcount);
}

// kept masks of vx_sweep_async -> keys2 (keys = the segment's sorted key1s in global memory)
template <int MAXTILES>
__device__ __forceinline__ void vx_async_emit(const vx_async_smem_t<MAXTILES> &S, const u64 *__restrict__ keys,
                                              u64 *__restrict__ keys2, int *count2, int *kcount = nullptr) {
    vx_masks_emit(S.alive, S.ndone, keys, keys2, count2, kcount);
}

#include "vx_nms_bm.h"

__global__ __launch_bounds__(1024) void vx_k_nms_sweep(const u64 *__restrict__ keys1, const vx_f4 *__restrict__ sbox,
                                                      const float *__restrict__ sarea,
                                                      const int *__restrict__ counts, vx_nms_params p,
                                                      u64 *__restrict__ keys2, int *__restrict__ count2,
                                                      unsigned long long *iou_count, int *__restrict__ kept_runs) {
    // kept_runs != nullptr (run merge, see vx_k_runs_merge): the kept keys of class c go in order to
    // keys2[segment start + rank] and their number to kept_runs[b][c]
    __shared__ vx_async_smem_big S;
    __shared__ __attribute__((aligned(16))) unsigned char s_ring[(2 * 16 * 64 + 8) * 20];      // vx_async_ring_bytes(16 waves)
    __shared__ int s_big[VX_SWEEP_MAXBIG][3];
    __shared__ int s_nbig;
    const int b = blockIdx.y;
    const int n = counts ? vx_clamp_count(counts[b], p.cap) : p.cap;
    if (n == 0) return;
    const long long ob = (long long)b * p.npad_cap;
    int *kr = kept_runs ? kept_runs + b * VX_PART_MAX_CLASSES : nullptr;
    const u64 *k1 = keys1 + ob;
    const vx_f4 *sb = sbox + ob;
    const float *sa = sarea + ob;
    u64 *k2 = keys2 + ob;
    int *c2 = count2 + b;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int NC = p.n_classes;
    const vx_thr T = vx_make_thr(p.thr);
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    // phase A: groups of nw consecutive classes are dealt round-robin to the image's workgroups;
    // each wave searches its class segment, sweeps it if small, queues it if large
    // class c -> workgroup c % gridDim.x, wave (c / gridDim.x) % nw: consecutive classes spread over workgroups
    for (int c = wave * gridDim.x + blockIdx.x; c < NC; c += gridDim.x * nw) {
        int lo = vx_lower_bound_wave(k1, n, (u64)c << 52);
        if (lo >= n) break;                              // no class >= c left (keys ascending)
        if (vx_key1_cls(k1[lo]) != c) continue;
        // the last class has no successor: (c + 1) << 52 would wrap to 0 for c = 4095
        int hi = c + 1 < VX_MAX_CLASSES ? vx_lower_bound_wave(k1, n, (u64)(c + 1) << 52) : n;
        int len = hi - lo;
        if (len <= 64 * VX_SWEEP_WT) {
            vx_sweep_wave(sb, sa, k1, lo, len, T, kr ? k2 + lo : k2, c2, iou_count, kr ? kr + c : nullptr);
        } else if (vx_lane() == 0) {
            int q = atomicAdd(&s_nbig, 1);
            s_big[q][0] = lo; s_big[q][1] = len; s_big[q][2] = c;
        }
    }
    __syncthreads();
    // phase B: large segments, whole workgroup each
    const int nbig = s_nbig;
    for (int q = 0; q < nbig; ++q) {
        vx_sweep_async(S, s_ring, sb + s_big[q][0], s_big[q][1], T, p.top_k, iou_count);
        vx_async_emit(S, k1 + s_big[q][0], kr ? k2 + s_big[q][0] : k2, c2, kr ? kr + s_big[q][2] : nullptr);
    }
}

// ------------------------------------------------------------------------------------------------
// Class-partitioned variant of keys -> sort -> sweep for images beyond the single-launch kernel when the caller
// declared a moderate number of classes (VX_PART_MIN_CLASSES..VX_PART_MAX_CLASSES): instead of one bitonic sort
// of all n keys of an image (log^2 n global/LDS passes), the candidates are counted per class, scattered into
// per-class segments (any order) and each segment is sorted in the LDS of the workgroup that then sweeps it.
//   vx_k_part_hist    : per-image class histogram (LDS counters, one global add per class and workgroup)
//   vx_k_part_scatter : segment starts = exclusive scan of the histogram; keys -> keys1[b][segment]
//   vx_k_part_sweep   : one workgroup per (class, image): LDS sort by (score desc, index), gather of the class-offset
//                       boxes, greedy sweep (the routines of vx_k_nms_sweep), kept keys -> keys2.  Launched twice:
//                       segments of up to VX_PART_SMALL keys (16 KiB of LDS, several workgroups per CU), then the
//                       larger ones up to VX_SORT_CHMAX (128 KiB).
// An image with a segment beyond VX_SORT_CHMAX is left to the general path (sel2[b] = its count).
#define VX_PART_MIN_CLASSES 4
#define VX_PART_SMALL 2048
#define VX_PART_TILE 4096

struct vx_part_args {
    vx_nms_params p;
    const float *boxes, *scores;
    const long long *cls;
    const int *sel;        // [B] first-level selection: candidate count of the images left by vx_k_nms_image, else 0
    int *hist, *cursor;    // [B][VX_PART_MAX_CLASSES], zeroed before vx_k_part_hist
    int *segstart;         // [B][VX_PART_MAX_CLASSES + 1]
    int *sel2;             // [B] out: count of the images handed on to the general path, else 0
    u64 *keys1, *keys2;
    vx_f4 *sbox;
    float *sarea;
    int *count2, *status;
    unsigned long long *iou_count;   // optional: useful lane-level IoU tests executed (measurement only)
    int defer;             // != 0 (optimistic call): an image with a segment beyond VX_SORT_CHMAX is only reported
                           //    (VX_STATUS_DEFERRED, keep count 0): the general path is not launched
    int *keep_counts;
    int ordered;           // 1: the kept keys of class c go, in order, to keys2[segstart[c]..) and their number to
                           //    hist[b][c] (the run merge below takes them from there); 0: keys2 through count2
    vx_bm_ws bm;           // bitmask path (vx_nms_bm.h): large segments are exported instead of swept when bm_on != 0
    int bm_on;
    int *big_count, *big_list;   // segments of more than VX_PART_SMALL keys, (b << 8) | c, listed by the first vx_k_part_sweep
    int big_cap;                 // launch for the second (a zeroed counter; nullptr: the second launch covers every (class, image))
};

__device__ __forceinline__ int vx_part_class(long long c, int n_classes, bool &bad) {
    if (c < 0 || c >= n_classes) { bad = true; c = c < 0 ? 0 : n_classes - 1; }
    return (int)c;
}

__global__ __launch_bounds__(1024) void vx_k_part_hist(vx_part_args a) {
    __shared__ int s_h[VX_PART_MAX_CLASSES];
    const int b = blockIdx.y;
    const int n = vx_clamp_count(a.sel[b], a.p.cap);
    const int k0 = blockIdx.x * VX_PART_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.count2[b] = 0;
    if (k0 >= n) return;
    const int NC = a.p.n_classes;
    for (int c = threadIdx.x; c < NC; c += blockDim.x) s_h[c] = 0;
    __syncthreads();
    bool bad = false;
    for (int k = k0 + threadIdx.x; k < min(n, k0 + VX_PART_TILE); k += blockDim.x)
        atomicAdd(&s_h[vx_part_class(a.cls[(long long)b * a.p.cap + k], NC, bad)], 1);
    if (__ballot(bad) != 0 && vx_lane() == 0) atomicOr(a.status, VX_STATUS_BAD_CLASS);
    __syncthreads();
    for (int c = threadIdx.x; c < NC; c += blockDim.x)
        if (s_h[c]) atomicAdd(&a.hist[b * VX_PART_MAX_CLASSES + c], s_h[c]);
}

__global__ __launch_bounds__(1024) void vx_k_part_scatter(vx_part_args a) {
    __shared__ int s_start[VX_PART_MAX_CLASSES + 1];
    __shared__ int s_cnt[VX_PART_MAX_CLASSES];       // this workgroup's count per class, then its next free slot
    __shared__ int s_over;
    const int b = blockIdx.y;
    const int n = vx_clamp_count(a.sel[b], a.p.cap);
    const int k0 = blockIdx.x * VX_PART_TILE;
    if (n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.sel2[b] = 0;
        return;
    }
    if (k0 >= n) return;
    const int NC = a.p.n_classes;
    const int *h = a.hist + b * VX_PART_MAX_CLASSES;
    if (threadIdx.x == 0) s_over = 0;
    for (int c = threadIdx.x; c < NC; c += blockDim.x) s_cnt[c] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {                          // exclusive scan of <= 256 counters: 4 per lane
        const int lane = threadIdx.x;
        int v[4], tot = 0;
        bool over = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int c = lane * 4 + q;
            v[q] = c < NC ? h[c] : 0;
            over |= v[q] > VX_SORT_CHMAX;
            tot += v[q];
        }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        int run = incl - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) { int c = lane * 4 + q; if (c <= NC) s_start[c] = run; run += v[q]; }
        if (lane == 63) s_start[NC] = incl;          // NC == 256: no lane owns index 256 above
        if (__ballot(over) != 0 && lane == 0) s_over = 1;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c <= NC; c += blockDim.x) a.segstart[b * (VX_PART_MAX_CLASSES + 1) + c] = s_start[c];
        if (threadIdx.x == 0) {
            a.sel2[b] = s_over ? n : 0;
            if (s_over && a.defer) { atomicOr(a.status, VX_STATUS_DEFERRED); a.keep_counts[b] = 0; }
        }
    }
    if (s_over) return;                              // the general path redoes this image from its inputs
    const long long ib = (long long)b * a.p.cap;
    const int k1 = min(n, k0 + VX_PART_TILE);
    bool bad = false;
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x)
        atomicAdd(&s_cnt[vx_part_class(a.cls[ib + k], NC, bad)], 1);
    __syncthreads();
    for (int c = threadIdx.x; c < NC; c += blockDim.x) {
        int m = s_cnt[c];
        s_cnt[c] = m ? s_start[c] + atomicAdd(&a.cursor[b * VX_PART_MAX_CLASSES + c], m) : 0;
    }
    __syncthreads();
    u64 *kd = a.keys1 + (long long)b * a.p.npad_cap;
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
        int c = vx_part_class(a.cls[ib + k], NC, bad);
        int pos = atomicAdd(&s_cnt[c], 1);           // any order inside the segment: the segment is sorted next
        kd[pos] = vx_key1(c, a.scores[ib + k], k);
    }
}

// (512-thread variant: three to four workgroups per CU wanted -- 7 waves per SIMD, 72 registers (64 are out of reach); the chain-sweep fall-back spills
// under that budget, the sort / gather / prep that every segment runs do not)
template <int LDS_KEYS, int THREADS>
__device__ __forceinline__ void vx_part_sweep_one(const vx_part_args &a, const int b, const int c, u64 *sk, vx_async_smem &A,
                                                  vx_bm_prep_smem &PREP, int &s_rbits) {
    if (a.sel[b] == 0 || a.sel2[b] != 0) return;
    const int lo = a.segstart[b * (VX_PART_MAX_CLASSES + 1) + c];
    const int len = a.segstart[b * (VX_PART_MAX_CLASSES + 1) + c + 1] - lo;
    // the first launch lists the segments of the second
    if (LDS_KEYS <= VX_PART_SMALL && a.big_count && len > LDS_KEYS && len <= VX_SORT_CHMAX && threadIdx.x == 0) {
        const int i = atomicAdd(a.big_count, 1);
        if (i < a.big_cap) a.big_list[i] = (b << 8) | c;
    }
    // this launch's share: (0, VX_PART_SMALL] or (VX_PART_SMALL, VX_SORT_CHMAX]
    if (len <= 0 || len > LDS_KEYS || (LDS_KEYS > VX_PART_SMALL && len <= VX_PART_SMALL)) return;
    const vx_nms_params &p = a.p;
    const long long ib = (long long)b * p.cap, ob = (long long)b * p.npad_cap;
    const int nt = blockDim.x;
    const int npad = vx_next_pow2(len);
    u64 *kseg = a.keys1 + ob + lo;
    if (threadIdx.x == 0) s_rbits = 0;
    for (int i = threadIdx.x; i < npad; i += nt) sk[i] = i < len ? kseg[i] : VX_KEY_SENTINEL;
    __syncthreads();
    vx_bitonic_sort_lds(sk, npad, 0, 2, npad);
    // sorted position -> class-offset box fl32(x + fl32(2049*cls)) (nms.py:94-96) and its area; the sorted keys go
    // back to the segment's place in global memory (the LDS is reused for the boxes of a large segment)
    vx_f4 *sb = a.sbox + ob + lo;
    float *sa = a.sarea + ob + lo;
    const float off = (float)(c * (VX_MAX_COORDINATE + 1));
    const bool big = len > 64 * VX_SWEEP_WT;
    int oob = 0;
    for (int i = threadIdx.x; i < len; i += nt) {
        const u64 key = sk[i];
        vx_f4 bx = *reinterpret_cast<const vx_f4 *>(a.boxes + (ib + vx_key1_idx(key)) * 4);
        oob |= vx_range_bits(bx);
        bx.x = bx.x + off; bx.y = bx.y + off; bx.z = bx.z + off; bx.w = bx.w + off;
        sb[i] = bx;
        sa[i] = (bx.z - bx.x) * (bx.w - bx.y);
        if (big) kseg[i] = key;
    }
    vx_flag_range_lds(oob, &s_rbits);
    __syncthreads();
    const vx_thr T = vx_make_thr(p.thr);
    u64 *k2 = a.keys2 + ob + (a.ordered ? lo : 0);
    int *c2 = a.count2 + b;
    int *kc = a.ordered ? a.hist + b * VX_PART_MAX_CLASSES + c : nullptr;     // the histogram has been consumed
    if (!big) {
        if (threadIdx.x < 64) vx_sweep_wave(sb, sa, sk, 0, len, T, k2, c2, a.iou_count, kc);
    } else if (a.bm_on && T.fast && vx_bm_prep(PREP, sk, LDS_KEYS > VX_PART_SMALL ? LDS_KEYS * sizeof(u64) : VX_BM_PREP_SCRATCH(LDS_KEYS), a.bm, b, lo, c, 1, sb, len)) {
        // registered for vx_k_bm_pairs; vx_k_bm_scan emits the kept keys
    } else {                                         // untame boxes / threshold outside the bracket's range: chain sweep
        vx_sweep_async(A, sk, sb, len, T, p.top_k, a.iou_count);                       // the ring takes the keys' LDS
        vx_async_emit(A, kseg, k2, c2, kc);
    }
    if (threadIdx.x == 0) vx_status_or(a.status, s_rbits);
}

// Launched twice: <VX_PART_SMALL, 512> over every (class, image) for the segments of up to VX_PART_SMALL keys, then
// <VX_SORT_CHMAX, 1024> for the larger ones -- over the list the first launch made (a bounded grid: a grid of classes x
// images 128-KiB workgroups that nearly all leave at once cost 19 us on the FPN-SSD batch).
template <int LDS_KEYS, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 7 : 4) void vx_k_part_sweep(vx_part_args a) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];
    __shared__ vx_async_smem A;
    __shared__ vx_bm_prep_smem PREP;
    __shared__ int s_rbits;
    if (LDS_KEYS <= VX_PART_SMALL || !a.big_count) {
        vx_part_sweep_one<LDS_KEYS, THREADS>(a, blockIdx.y, blockIdx.x, sk, A, PREP, s_rbits);
        return;
    }
    const int n = min(*a.big_count, a.big_cap);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int e = a.big_list[i];
        vx_part_sweep_one<LDS_KEYS, THREADS>(a, e >> 8, e & 255, sk, A, PREP, s_rbits);
        __syncthreads();
    }
}

// Bitmask path, last step on the class-partitioned route: the registered segments are dealt round-robin to the
// workgroups; each is resolved from its edge list (the chain sweep when the list overflowed) and its kept keys are
// emitted exactly as vx_k_part_sweep would have done.
#define VX_BM_SCAN_LDS_EDGES 4096       // words of LDS of vx_k_bm_scan: sorted edges + suppressor columns (16 KiB >= the chain sweep's ring for 4 waves)
__global__ __launch_bounds__(256, 8) void vx_k_bm_scan(vx_part_args a) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];             // sorted edges / ring of the chain sweep (fallback)
    __shared__ union { vx_async_smem as; vx_bm_scan_smem sc; } U;
    const vx_nms_params &p = a.p;
    const int nsegs = min(a.bm.counters[0], a.bm.max_segs);
    const vx_thr T = vx_make_thr(p.thr);
    for (int si = blockIdx.x; si < nsegs; si += gridDim.x) {
        const vx_bm_seg sg = a.bm.segs[si];
        if (sg.route != 1) continue;
        const int b = sg.b, c = sg.id;
        const long long ob = (long long)b * p.npad_cap, region = ob + sg.lo;
        const u64 *kseg = a.keys1 + region;
        u64 *k2 = a.keys2 + ob + (a.ordered ? sg.lo : 0);
        int *c2 = a.count2 + b;
        int *kc = a.ordered ? a.hist + b * VX_PART_MAX_CLASSES + c : nullptr;
        if (vx_bm_scan(U.sc, a.bm.edges + region * VX_BM_ECAP, a.bm.edges2 + region * VX_BM_ECAP, a.bm.ecount[si],
                       sg.len * VX_BM_ECAP, sg.len, p.top_k, reinterpret_cast<unsigned *>(sk), VX_BM_SCAN_LDS_EDGES)) {
            vx_masks_emit(U.sc.kept, U.sc.ndone, kseg, k2, c2, kc);
        } else {
            vx_sweep_async(U.as, sk, a.sbox + region, sg.len, T, p.top_k, a.iou_count);
            vx_async_emit(U.as, kseg, k2, c2, kc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Run merge (class-partitioned path, ordered emission): the kept keys of an image are n_classes ascending runs (run c
// at keys2[segstart[c] ..), hist[b][c] keys).  Instead of a second sort of all of them, or log2(n_classes) rounds of
// pairwise merges over global memory (round 2: five rounds + a compaction, 0.145 ms on the FPN-SSD batch, each round a
// chain of ~11 dependent L2 reads per key), the output is cut into buckets by key VALUE:
//   vx_k_runs_split : one workgroup per image.  A strided sample of the runs (every stride-th key
//                     samples, at most 1024) is sorted in LDS; every (samples / buckets)-th sample is a splitter; a binary search per
//                     (splitter, run) gives the part of every run that falls into each bucket, their sum the bucket's
//                     place in the output.  A sample's rank is known to within one stride per run, so a bucket holds at
//                     most VX_RUNS_BUCKET + 2 * n_classes * stride keys.
//   vx_k_runs_bucket: one workgroup per (bucket, image): the bucket's keys -- a contiguous piece of every run -- are
//                     gathered into LDS, sorted there, and the output rows are written (the second sort's epilogue).  A
//                     bucket beyond the LDS buffer (possible with hundreds of classes and a long tail of ties in the
//                     sample) is ranked key by key with binary searches over the other runs' pieces instead: slow, exact.
// Keys are unique (the index is part of the key), so the order is total and the result equals the full sort's.
#define VX_RUNS_BUCKET 1024            // keys per bucket aimed at
#define VX_RUNS_MAXB 256               // buckets per image at most (more keys: larger buckets)
#define VX_RUNS_THREADS 1024           // workgroup of vx_k_runs_bucket: about one key per thread -- a bucket is a chain of
                                       // latencies (six dependent global loads, then per merge round ~12 dependent LDS reads
                                       // per key), so keys per thread multiply its duration (8 per thread: 0.09 ms)
#define VX_RUNS_PER_THREAD 2           // positions of a bucket per thread of vx_k_runs_bucket
#define VX_RUNS_CAP (VX_RUNS_THREADS * VX_RUNS_PER_THREAD)   // keys a bucket merges in LDS (two buffers)
#define VX_RUNS_SAMPLES 2048           // LDS sort size of the sample

struct vx_runs_ws {
    int *lb;       // [B][VX_RUNS_MAXB][VX_PART_MAX_CLASSES]: keys of run c below splitter j (j = 1 .. nb-1)
    int *bstart;   // [B][VX_RUNS_MAXB + 1] first output position of every bucket; [nb] = total
    int *nb;       // [B] buckets of the image (0: nothing to do here)
    int *list;     // [B * VX_RUNS_MAXB] the call's buckets, (image << 8) | bucket, in any order ...
    int *nlist;    // ... and their number (zeroed per call)
};

__device__ __forceinline__ bool vx_runs_skip(const vx_part_args &a, int b) {
    return a.sel[b] == 0 || (a.defer && a.sel2[b] != 0);           // finished by vx_k_nms_image / deferred
}

// keys of the sorted run [p, p + n) below x
__device__ __forceinline__ int vx_runs_lower(const u64 *__restrict__ p, int n, u64 x) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(1024) void vx_k_runs_split(vx_part_args a, vx_runs_ws rw, vx_epi_params ep) {
    __shared__ __attribute__((aligned(16))) u64 sk[VX_RUNS_SAMPLES];
    __shared__ int s_R[VX_PART_MAX_CLASSES], s_len[VX_PART_MAX_CLASSES], s_soff[VX_PART_MAX_CLASSES + 1];
    __shared__ u64 s_spl[VX_RUNS_MAXB];
    __shared__ int s_boff[VX_RUNS_MAXB + 1];
    __shared__ int s_total, s_nsamp;
    const int b = blockIdx.x, nt = blockDim.x, lane = vx_lane();
    const int NC = a.p.n_classes;
    if (vx_runs_skip(a, b)) {
        if (threadIdx.x == 0) rw.nb[b] = 0;
        return;
    }
    const long long ob = (long long)b * a.p.npad_cap;
    const u64 *k2 = a.keys2 + ob;
    if (threadIdx.x < 64) {                          // run lengths, their sum (<= 256 runs: 4 per lane)
        int tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane * 4 + q;
            const int len = c < NC ? a.hist[b * VX_PART_MAX_CLASSES + c] : 0;
            if (c < NC) { s_len[c] = len; s_R[c] = a.segstart[b * (VX_PART_MAX_CLASSES + 1) + c]; }
            tot += len;
        }
        tot = vx_wave_sum_i(tot);
        if (lane == 0) s_total = tot;
    }
    for (int j = threadIdx.x; j <= VX_RUNS_MAXB; j += nt) s_boff[j] = 0;
    __syncthreads();
    const int total = s_total;
    int nk = total;
    if (ep.top_k > 0) nk = min(nk, ep.top_k);
    nk = min(nk, ep.out_cap);
    if (threadIdx.x == 0) ep.keep_counts[b] = nk;
    if (total == 0) {
        if (threadIdx.x == 0) rw.nb[b] = 0;
        return;
    }
    int nb = (total + VX_RUNS_BUCKET - 1) / VX_RUNS_BUCKET;
    nb = min(nb, VX_RUNS_MAXB);
    // sample: one key of every block of `stride` keys of every run (the last block may be short), at an offset that differs
    // from run to run (run c: c * stride / n_classes).  With ONE offset for all runs the samples of equally distributed
    // runs come in clumps of n_classes nearly equal keys and the buckets alternate between tiny and three times the aim
    // (simulated: 82 .. 3085 keys for 80 runs of ~700; staggered: 55 .. 1620); sampling block ends also left every run's
    // tail to the last bucket
    const int stride = max(16, (total + 1023) / 1024);
    if (threadIdx.x < 64) {
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = lane * 4 + q; v[q] = c < NC ? (s_len[c] + stride - 1) / stride : 0; tot += v[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        int run = incl - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = lane * 4 + q; if (c <= NC) s_soff[c] = run; run += v[q]; }
        if (lane == 63) { s_soff[NC] = incl; s_nsamp = incl; }
    }
    __syncthreads();
    const int nsamp = s_nsamp;                       // <= total / stride + n_classes <= 1280 < VX_RUNS_SAMPLES
    const int npad = vx_next_pow2(max(nsamp, 1));
    for (int i = threadIdx.x; i < npad; i += nt) {
        u64 key = VX_KEY_SENTINEL;
        if (i < nsamp) {
            int lo = 0, hi = NC;                     // run of sample i: s_soff[lo] <= i < s_soff[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_soff[mid] <= i) lo = mid; else hi = mid; }
            key = k2[s_R[lo] + min((i - s_soff[lo]) * stride + (lo * stride) / NC, s_len[lo] - 1)];
        }
        sk[i] = key;
    }
    __syncthreads();
    vx_bitonic_sort_lds(sk, npad, 0, 2, npad);
    if (nsamp == 0) nb = 1;
    for (int j = 1 + threadIdx.x; j < nb; j += nt) s_spl[j] = sk[(int)(((long long)j * nsamp) / nb)];
    __syncthreads();
    // keys of every run below every splitter
    int *lbt = rw.lb + (long long)b * VX_RUNS_MAXB * VX_PART_MAX_CLASSES;
    for (int e = threadIdx.x; e < (nb - 1) * NC; e += nt) {
        const int j = e / NC + 1, c = e - (j - 1) * NC;
        const int lbv = vx_runs_lower(k2 + s_R[c], s_len[c], s_spl[j]);
        lbt[j * VX_PART_MAX_CLASSES + c] = lbv;
        if (lbv) atomicAdd(&s_boff[j], lbv);
    }
    __syncthreads();
    int *bs = rw.bstart + b * (VX_RUNS_MAXB + 1);
    for (int j = threadIdx.x; j <= nb; j += nt) bs[j] = j == nb ? total : s_boff[j];
    if (threadIdx.x == 0) {
        rw.nb[b] = nb;
        const int at = atomicAdd(rw.nlist, nb);          // the buckets of the call, for vx_k_runs_bucket's bounded grid
        for (int j = 0; j < nb; ++j) rw.list[at + j] = (b << 8) | j;
    }
}

__global__ __launch_bounds__(VX_RUNS_THREADS) void vx_k_runs_bucket(vx_part_args a, vx_runs_ws rw, vx_epi_params ep) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];         // [2 * VX_RUNS_CAP]
    __shared__ int s_R[VX_PART_MAX_CLASSES], s_lo[VX_PART_MAX_CLASSES], s_hi[VX_PART_MAX_CLASSES], s_off[VX_PART_MAX_CLASSES + 1];
    const int nt = blockDim.x, lane = vx_lane();
    const int nlist = min(*rw.nlist, a.p.B * VX_RUNS_MAXB);
    const int NC = a.p.n_classes;
    // the call's buckets are dealt round-robin to a bounded grid (a grid over every possible bucket of every image is
    // mostly workgroups that leave at once, and with 64 KiB of LDS each they queue behind one another)
#ifdef VX_BM_DEBUG
#define VX_RB_STAMP(k) do { if (wi == 5 && threadIdx.x == 0) rw.nlist[14 + (k)] = (int)clock64(); } while (0)
#else
#define VX_RB_STAMP(k)
#endif
    for (int wi = blockIdx.x; wi < nlist; wi += gridDim.x) {
    VX_RB_STAMP(0);
    const int ent = rw.list[wi];
    const int b = ent >> 8, j = ent & 255;
    const int nb = rw.nb[b];
    const int *bs = rw.bstart + b * (VX_RUNS_MAXB + 1);
    const int o0 = bs[j], cnt = bs[j + 1] - o0;
    const int nk = ep.keep_counts[b];
    if (cnt <= 0 || o0 >= nk) continue;
    __syncthreads();                                 // the previous bucket's LDS is free
    const long long ob = (long long)b * a.p.npad_cap;
    const u64 *k2 = a.keys2 + ob;
    const int *lbt = rw.lb + (long long)b * VX_RUNS_MAXB * VX_PART_MAX_CLASSES;
    if (threadIdx.x < 64) {                          // the bucket's piece [lo, hi) of every run and the pieces' prefix
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane * 4 + q;
            v[q] = 0;
            if (c < NC) {
                const int lo = j == 0 ? 0 : lbt[j * VX_PART_MAX_CLASSES + c];
                const int hi = j == nb - 1 ? a.hist[b * VX_PART_MAX_CLASSES + c] : lbt[(j + 1) * VX_PART_MAX_CLASSES + c];
                s_R[c] = a.segstart[b * (VX_PART_MAX_CLASSES + 1) + c];
                s_lo[c] = lo; s_hi[c] = hi;
                v[q] = hi - lo;
            }
            tot += v[q];
        }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        int run = incl - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = lane * 4 + q; if (c <= NC) s_off[c] = run; run += v[q]; }
        if (lane == 63) s_off[NC] = incl;
    }
    __syncthreads();
    VX_RB_STAMP(1);
    auto piece_of = [&](int i) __attribute__((always_inline)) -> int {      // s_off[c] <= i < s_off[c + 1] (empty pieces skipped)
        int lo = 0, hi = NC;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
        return lo;
    };
    auto write_row = [&](int pos, u64 key) __attribute__((always_inline)) {
        const long long idx = (long long)(key & 0xFFFFFFFFull);
        ep.keep_idx[(long long)b * ep.out_cap + pos] = idx;
        const float *det = ep.det + ((long long)b * ep.cap + idx) * ep.D;
        float *out = ep.out_det + ((long long)b * ep.out_cap + pos) * ep.D;
        if ((ep.D & 1) == 0 && ((((uintptr_t)det) | ((uintptr_t)out)) & 7) == 0) {
            for (int q = 0; q < (ep.D >> 1); ++q) reinterpret_cast<float2 *>(out)[q] = reinterpret_cast<const float2 *>(det)[q];
        } else {
            for (int q = 0; q < ep.D; ++q) out[q] = det[q];
        }
    };
    if (cnt <= VX_RUNS_CAP) {
        // the bucket's pieces are sorted runs: merged pairwise in LDS, ceil(log2 n_classes) rounds, every key finds its place
        // by one binary search in the sibling run (the routine of vx_k_nms_image's tail; a bitonic sort of the bucket
        // took 0.22 ms on the FPN-SSD batch, ten times this)
        u64 *src = sk, *dst = sk + VX_RUNS_CAP;
        int rr[VX_RUNS_PER_THREAD];
#pragma unroll
        for (int k = 0; k < VX_RUNS_PER_THREAD; ++k) {
            const int o = threadIdx.x + k * nt;
            rr[k] = 0;
            if (o < cnt) { const int c = piece_of(o); rr[k] = c; src[o] = k2[s_R[c] + s_lo[c] + (o - s_off[c])]; }
        }
        __syncthreads();
        VX_RB_STAMP(2);
        for (int lg = 0; (1 << lg) < NC; ++lg) {
#pragma unroll
            for (int k = 0; k < VX_RUNS_PER_THREAD; ++k) {
                const int o = threadIdx.x + k * nt;
                if (o < cnt) {
                    const int G = (rr[k] >> lg) & ~1;    // groups of 2^lg runs; groups G and G + 1 are merged
                    const int a0 = s_off[min(G << lg, NC)], m0 = s_off[min((G + 1) << lg, NC)], b1 = s_off[min((G + 2) << lg, NC)];
                    const bool second = o >= m0;
                    const u64 x = src[o];
                    int lo2 = second ? a0 : m0, hi2 = second ? m0 : b1;
                    const int sib0 = lo2;
                    while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (src[mid] < x) lo2 = mid + 1; else hi2 = mid; }
                    dst[a0 + (o - (second ? m0 : a0)) + (lo2 - sib0)] = x;
                }
            }
            __syncthreads();
            u64 *t = src; src = dst; dst = t;
        }
        VX_RB_STAMP(3);
        const int rows = min(cnt, nk - o0);
        for (int i = threadIdx.x; i < rows; i += nt) write_row(o0 + i, src[i]);
        __syncthreads();
        VX_RB_STAMP(4);
    } else {
        // beyond the LDS buffer: the place of a key = the keys below it in every run's piece
        for (int i = threadIdx.x; i < cnt; i += nt) {
            const int c = piece_of(i);
            const u64 x = k2[s_R[c] + s_lo[c] + (i - s_off[c])];
            int rank = i - s_off[c];
            for (int c2 = 0; c2 < NC; ++c2)
                if (c2 != c && s_hi[c2] > s_lo[c2]) rank += vx_runs_lower(k2 + s_R[c2] + s_lo[c2], s_hi[c2] - s_lo[c2], x);
            if (o0 + rank < nk) write_row(o0 + rank, x);
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// top_k on the multi-kernel path: instead of sorting every kept key of an image (second sort), a radix select
// over the 64-bit keys (12-bit digits, most significant first) finds the smallest key range that holds the
// first top_k keys and fits the LDS buffer; only those keys are sorted.  Keys are unique (the index is part of
// the key), so the select always converges.  One workgroup per image.
#define VX_TOPK_MAX 2048
#define VX_TOPK_LDS 4096

__global__ __launch_bounds__(1024) void vx_k_topk_select(const u64 *__restrict__ keys2, const int *__restrict__ count2,
                                                        const int *__restrict__ sel, vx_nms_params p, vx_epi_params ep) {
    __shared__ __attribute__((aligned(16))) u64 sk[VX_TOPK_LDS];
    __shared__ int s_h[4096];
    __shared__ int s_wtot[16];
    __shared__ int s_T, s_before, s_cnt, s_n;
    const int b = blockIdx.x, nt = blockDim.x;
    if (sel && sel[b] == 0) return;                  // finished by vx_k_nms_image
    const int n2 = vx_clamp_count(count2[b], p.npad_cap);
    int want = n2;
    if (p.top_k > 0) want = min(want, p.top_k);
    want = min(want, p.out_cap);
    if (threadIdx.x == 0) ep.keep_counts[b] = want;
    if (want == 0) return;
    const u64 *k2 = keys2 + (long long)b * p.npad_cap;
    const int lane = vx_lane(), wave = threadIdx.x >> 6;
    u64 prefix = 0;
    int shift = 64, before = 0;
    if (n2 > VX_TOPK_LDS) {
        while (true) {
            const int bits = min(12, shift);
            const int hi_shift = shift;              // keys taking part: (key >> hi_shift) == prefix
            shift -= bits;
            const unsigned mask = (1u << bits) - 1u;
            for (int i = threadIdx.x; i < 4096; i += nt) s_h[i] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < n2; i += nt) {
                u64 key = k2[i];
                if (hi_shift == 64 || (key >> hi_shift) == prefix) atomicAdd(&s_h[(unsigned)(key >> shift) & mask], 1);
            }
            __syncthreads();
            // the bin that holds the want-th key: 4 bins per thread, wave scan, wave totals
            int v[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = s_h[threadIdx.x * 4 + q]; tot += v[q]; }
            int incl = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            if (lane == 63) s_wtot[wave] = incl;
            __syncthreads();
            int run = before + incl - tot;
            for (int w = 0; w < wave; ++w) run += s_wtot[w];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (run < want && want <= run + v[q]) { s_T = threadIdx.x * 4 + q; s_before = run; s_cnt = v[q]; }
                run += v[q];
            }
            __syncthreads();
            prefix = (hi_shift == 64 ? 0ull : prefix << bits) | (u64)(unsigned)s_T;
            before = s_before;
            const int cnt = s_cnt;
            __syncthreads();                         // s_T / s_before / s_cnt are rewritten by the next level
            if (before + cnt <= VX_TOPK_LDS || shift == 0) break;
        }
    }
    // collect the keys at or below the selected range (everything when the image's keys fit), sort, write
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n2; i += nt) {
        u64 key = k2[i];
        if (shift == 64 || (key >> shift) <= prefix) {
            int pos = atomicAdd(&s_n, 1);
            if (pos < VX_TOPK_LDS) sk[pos] = key;
        }
    }
    __syncthreads();
    const int m = min(s_n, VX_TOPK_LDS);
    const int npad = vx_next_pow2(max(m, 1));
    for (int i = m + threadIdx.x; i < npad; i += nt) sk[i] = VX_KEY_SENTINEL;
    __syncthreads();
    vx_bitonic_sort_lds(sk, npad, 0, 2, npad);
    for (int i = threadIdx.x; i < want; i += nt)
        ep.keep_idx[(long long)b * ep.out_cap + i] = (long long)(sk[i] & 0xFFFFFFFFull);
    const float *src = ep.det + (long long)b * ep.cap * ep.D;
    float *dst = ep.out_det + (long long)b * ep.out_cap * ep.D;
    for (int e = threadIdx.x; e < want * ep.D; e += nt) {
        int i = e / ep.D, q = e - i * ep.D;
        dst[e] = src[(long long)(sk[i] & 0xFFFFFFFFull) * ep.D + q];
    }
}

// ------------------------------------------------------------------------------------------------
// top_k on large inputs: greedy NMS decides a box from higher-scoring boxes only, so the kept boxes among ANY
// score-prefix S of an image's candidates (every candidate outside S scores below every candidate inside) are
// exactly the kept boxes of the full run that lie in S -- and they come first in the output order.  If S yields at
// least top_k kept boxes, the first top_k of them ARE the result, whatever follows S.
//   vx_k_prefix_thr     : one workgroup per image picks a score threshold from a strided sample of 4096 scores (its
//                         r-th best, r chosen so that about vx_prefix_target(top_k) candidates pass; exact radix
//                         select of the sample in LDS).  Any threshold gives a valid prefix -- the sample only aims it.
//   vx_k_prefix_compact : all candidates whose descending-score key is at most the threshold (equal scores stay
//                         together: a true prefix) -> the image's list, in any order (the NMS sorts)
// vx_k_nms_image then runs on the list.  An image whose prefix keeps fewer than top_k boxes, or outgrows the list, is
// redone from all its candidates by the paths behind that kernel.  (Class ids are validated for the candidates that
// take part; the others cannot influence the result.)
#define VX_PREFIX_CAP VX_SORT_CHMAX
#define VX_PREFIX_SAMPLE 4096
#define VX_PREFIX_CHUNK 8192
__host__ __device__ __forceinline__ int vx_prefix_target(int top_k) {
    int m = 4 * top_k;
    return m < 2048 ? 2048 : (m > VX_PREFIX_CAP / 2 ? VX_PREFIX_CAP / 2 : m);
}

__global__ __launch_bounds__(1024) void vx_k_prefix_thr(const float *__restrict__ scores, const int *__restrict__ counts,
                                                       vx_nms_params p, unsigned *__restrict__ pthr, int *__restrict__ pm) {
    __shared__ unsigned s_key[VX_PREFIX_SAMPLE];
    __shared__ int s_h[4096];
    __shared__ int s_wtot[16];
    __shared__ int s_T, s_before;
    const int b = blockIdx.x, lane = vx_lane(), wave = threadIdx.x >> 6;
    const int n = counts ? vx_clamp_count(counts[b], p.cap) : p.cap;
    const int want = vx_prefix_target(p.top_k);
    // pm[b]: -1 = no prefix for this image, else the compaction's cursor (starts at 0)
    if (n <= 2 * want || n <= VX_SORT_CHMAX / 4) {   // small enough as it is
        if (threadIdx.x == 0) pm[b] = -1;
        return;
    }
    const float *sc = scores + (long long)b * p.cap;
    float v[VX_PREFIX_SAMPLE / 1024];
#pragma unroll
    for (int q = 0; q < VX_PREFIX_SAMPLE / 1024; ++q) {
        const int j = threadIdx.x + q * 1024;
        v[q] = sc[(int)(((long long)j * n) / VX_PREFIX_SAMPLE)];
    }
#pragma unroll
    for (int q = 0; q < VX_PREFIX_SAMPLE / 1024; ++q) s_key[threadIdx.x + q * 1024] = vx_desc_key(v[q]);
    // rank (1-based) of the threshold inside the sample: an eighth above the aim, so that a sample that runs a little
    // optimistic still yields enough candidates
    int r = (int)(((long long)want * 9 * VX_PREFIX_SAMPLE) / ((long long)n * 8)) + 3;
    if (r > VX_PREFIX_SAMPLE) r = VX_PREFIX_SAMPLE;
    unsigned prefix = 0;
    int before = 0, shift = 32;
    for (int level = 0; level < 3; ++level) {        // 12 + 12 + 8 bits, most significant first
        const int bits = level < 2 ? 12 : 8;
        const int hi_shift = shift;
        shift -= bits;
        for (int i = threadIdx.x; i < 4096; i += 1024) s_h[i] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < VX_PREFIX_SAMPLE / 1024; ++q) {
            const unsigned key = s_key[threadIdx.x + q * 1024];
            if (hi_shift == 32 || (key >> hi_shift) == prefix) atomicAdd(&s_h[(key >> shift) & ((1u << bits) - 1u)], 1);
        }
        __syncthreads();
        int hv[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { hv[q] = s_h[threadIdx.x * 4 + q]; tot += hv[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int run = before + incl - tot;
        for (int w2 = 0; w2 < wave; ++w2) run += s_wtot[w2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (run < r && r <= run + hv[q]) { s_T = threadIdx.x * 4 + q; s_before = run; }
            run += hv[q];
        }
        __syncthreads();
        prefix = (prefix << bits) | (unsigned)s_T;
        before = s_before;
        __syncthreads();
    }
    if (threadIdx.x == 0) { pthr[b] = prefix; pm[b] = 0; }
}

__global__ __launch_bounds__(1024) void vx_k_prefix_compact(const float *__restrict__ scores, const int *__restrict__ counts,
                                                           vx_nms_params p, const unsigned *__restrict__ pthr,
                                                           int *__restrict__ plist, int *__restrict__ pm) {
    const int b = blockIdx.y, lane = vx_lane();
    const int n = counts ? vx_clamp_count(counts[b], p.cap) : p.cap;
    const int i0 = blockIdx.x * VX_PREFIX_CHUNK;
    // (pm[b] >= 0 is the compaction cursor other workgroups are already adding to: only its sign is read here)
    if (i0 >= n || __hip_atomic_load(&pm[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) return;
    const float *sc = scores + (long long)b * p.cap;
    const unsigned thr = pthr[b];
    int *pl = plist + (long long)b * VX_PREFIX_CAP;
    __shared__ int s_wcnt[16];
    __shared__ int s_base;
    float v[VX_PREFIX_CHUNK / 1024];
#pragma unroll
    for (int q = 0; q < VX_PREFIX_CHUNK / 1024; ++q) {
        const int i = i0 + q * 1024 + threadIdx.x;
        v[q] = i < n ? sc[i] : 0.0f;
    }
    // members of this thread, the wave, the workgroup: ONE global atomic per workgroup reserves the chunk's range
    // (a returning atomic per wave, ~1 300 per image on one word, took 0.26 ms on the 100 000-box batch)
    const int wave = threadIdx.x >> 6;
    unsigned inm = 0;
#pragma unroll
    for (int q = 0; q < VX_PREFIX_CHUNK / 1024; ++q) {
        const int i = i0 + q * 1024 + threadIdx.x;
        if (i < n && vx_desc_key(v[q]) <= thr) inm |= 1u << q;
    }
    const int mine = __popc(inm);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) s_wcnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) { const int c = s_wcnt[w2]; s_wcnt[w2] = tot; tot += c; }
        s_base = tot ? atomicAdd(&pm[b], tot) : 0;
    }
    __syncthreads();
    int pos = s_base + s_wcnt[wave] + incl - mine;
#pragma unroll
    for (int q = 0; q < VX_PREFIX_CHUNK / 1024; ++q) {
        if ((inm >> q) & 1u) {
            if (pos < VX_PREFIX_CAP) pl[pos] = i0 + q * 1024 + threadIdx.x;       // beyond the list: the image is redone from all candidates
            ++pos;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole NMS of one image in ONE workgroup (capacity <= VX_SORT_CHMAX): keys are built and sorted in
// LDS, the class-offset boxes gathered into sorted order, every class segment swept (one wave per
// small segment, the whole workgroup per large one; segment bounds come from 64-ary searches of the
// LDS-resident keys), the kept keys sorted back into score order in the same LDS and the outputs
// written.  One launch for the batch instead of five; work scales with the image's own count.
struct vx_image_args {
    vx_nms_params p;
    const float *boxes, *scores, *det;
    const long long *cls;
    const int *counts;
    vx_f4 *sbox;
    float *sarea;
    u64 *keys1;       // sorted key1s of the image (read back when kept boxes are emitted: the LDS copy is overlaid)
    u64 *keys2;
    unsigned long long *iou_count;   // optional: useful lane-level IoU tests executed (measurement only)
    long long *keep_idx;
    float *out_det;
    int *keep_counts, *status;
    int *seg;         // [B][VX_MAX_CLASSES] class-segment starts in sorted order
    int *segk;        // [B][VX_MAX_CLASSES] kept boxes of each segment (ordered emission, see the run merge below)
    int *sel;         // optional [B]: out, n if the image has more than n_max candidates (left to the multi-kernel path) else 0
    int n_max;
    size_t lds_bytes; // dynamic LDS of the launch
    int defer;        // != 0: such images are only reported (VX_STATUS_DEFERRED, keep_count 0), nobody finishes them
    u64 *stamps;      // [B][8] shader-clock stamps of the phases of block b, or nullptr (set while vx_profile_enable is on)
    vx_bm_ws bm;      // bitmask path (vx_nms_bm.h): large segments are exported instead of swept here when bm_on != 0
    int bm_on;
    const int *plist; // optional [B][VX_PREFIX_CAP]: score-prefix of an image's candidates (vx_k_prefix_select) ...
    const int *pm;    // ... [B] its length, or -1: use all candidates
    // fin_on != 0 (vx_internal_nms_from_stage): the candidate arrays above are not filled yet -- the workgroup first packs
    // its image's per-tile records of the fused YOLO decode into them (what vx_k_cand_finish does) and writes counts[b]
    int fin_on;
    vx_stage_info fin;
    // pack_on != 0: the kernel also packs the kept rows (vx_pack_info; images are assigned by ticket, see vx_image_pack)
    int pack_on;
    vx_pack_info pack;
};

// An image with <= 64 candidates (the common case at the shipped score threshold 0.9: ~10 per image) is one
// wave's work: keys, both sorts (64-lane bitonic networks on shuffles), the greedy sweep and the outputs without
// LDS or workgroup barriers.  Same keys, same IoU test, same ordering rules as the general path below.
__device__ __forceinline__ u64 vx_wave_sort64(u64 key) {
    const int lane = vx_lane();
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            u64 o = j == 1 ? vx_quad_xor_u64<1>(key) : j == 2 ? vx_quad_xor_u64<2>(key) : j == 4 ? vx_row_xor_u64<4>(key)
                    : j == 8 ? vx_row_xor_u64<8>(key) : vx_shfl_xor_u64(key, j);
            bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
            u64 mn = key < o ? key : o, mx = key < o ? o : key;
            key = take_min ? mn : mx;
        }
    }
    return key;
}

__device__ __forceinline__ void vx_nms_image_small(const vx_image_args &a, int b, int n) {
    const vx_nms_params &p = a.p;
    const int lane = vx_lane();
    const long long ib = (long long)b * p.cap;
    u64 key = VX_KEY_SENTINEL;
    bool bad = false;
    if (lane < n) {
        long long c = a.cls[ib + lane];
        if (c < 0 || c >= p.n_classes) { bad = true; c = c < 0 ? 0 : p.n_classes - 1; }
        key = vx_key1((int)c, a.scores[ib + lane], lane);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(a.status, VX_STATUS_BAD_CLASS);
    key = vx_wave_sort64(key);                       // (class, score desc, index); the n valid keys come first
    const bool valid = lane < n;
    vx_f4 bx = {0.f, 0.f, 0.f, 0.f};
    float ar = 0.f;
    int cl = -1;
    int oob = 0;
    if (valid) {
        cl = vx_key1_cls(key);
        float off = (float)(cl * (VX_MAX_COORDINATE + 1));
        bx = *reinterpret_cast<const vx_f4 *>(a.boxes + (ib + vx_key1_idx(key)) * 4);
        oob = vx_range_bits(bx);
        bx.x = bx.x + off; bx.y = bx.y + off; bx.z = bx.z + off; bx.w = bx.w + off;
        ar = (bx.z - bx.x) * (bx.w - bx.y);
    }
    const vx_thr T = vx_make_thr(p.thr);
    u64 alive = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    u64 todo = alive;
    unsigned long long tests = 0ull;
    while (todo) {                                   // wave-uniform: next still-alive box in sorted order
        const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
        if (a.iou_count) tests += __popcll(__ballot(valid && lane > i && cl == __builtin_amdgcn_readlane(cl, i)) & alive);
        const float ix1 = vx_bcast(bx.x, i), iy1 = vx_bcast(bx.y, i), ix2 = vx_bcast(bx.z, i), iy2 = vx_bcast(bx.w, i);
        const float ia = vx_bcast(ar, i);
        const int ci = __builtin_amdgcn_readlane(cl, i);
        bool sup = valid && lane > i && cl == ci && vx_iou_gt<true>(ix1, iy1, ix2, iy2, ia, bx.x, bx.y, bx.z, bx.w, ar, T);
        alive &= ~__ballot(sup);
        todo = alive & ~((2ull << i) - 1ull);
    }
    if (a.iou_count && tests && lane == 0) atomicAdd(a.iou_count, tests);
    const bool kept = (alive >> lane) & 1ull;
    u64 k2 = kept ? vx_key2_from_key1(key) : VX_KEY_SENTINEL;
    k2 = vx_wave_sort64(k2);                         // kept boxes of all classes back into score order
    int nk = __popcll(alive);
    if (p.top_k > 0) nk = min(nk, p.top_k);
    nk = min(nk, p.out_cap);
    if (lane == 0) a.keep_counts[b] = nk;
    if (lane < nk) {
        const int idx = (int)(k2 & 0xFFFFFFFFull);
        a.keep_idx[(long long)b * p.out_cap + lane] = (long long)idx;
        const float *src = a.det + (ib + idx) * p.D;
        float *dst = a.out_det + ((long long)b * p.out_cap + lane) * p.D;
        for (int q = 0; q < p.D; ++q) dst[q] = src[q];
    }
    if (p.n_classes != 1) vx_flag_range(oob, a.status);      // last: nothing of this wave queues behind the atomic
}

#define VX_IMG_MERGE_PER_THREAD 8        // positions of the run merge whose run index a thread keeps in registers
// Tail of the single-launch kernel (also run by vx_k_bm_finish for an image whose large segments went through the
// bitmask path): the kept keys of all segments -> global score order (run merge in LDS, or a second sort) -> outputs.
// n_redo > 0: the image was processed from a score-prefix list; when fewer than top_k boxes were kept it is handed to the
// paths behind this kernel (sel[b] = n_redo, its full candidate count) instead of being written.
// srank != nullptr (LDS, from the radix sort; the LDS behind the keys untouched since): srank[candidate] = rank in the
// final order, so the kept keys are dropped at their rank into `far` and compacted -- no merge rounds.  scratch: LDS for
// nseg + 1 ints + n / 64 + 1 u64.
__device__ __forceinline__ void vx_image_tail(const vx_image_args &a, int b, int n, int nseg, bool merge, u64 *sk,
                                              int *s_gcnt, int *s_count2p, int rbits, int n_redo,
                                              const unsigned short *srank = nullptr, int *scratch = nullptr, u64 *far = nullptr) {
    const vx_nms_params &p = a.p;
    const int nt = blockDim.x, wave = threadIdx.x >> 6, lane = vx_lane();
    const long long ib = (long long)b * p.cap, ob = (long long)b * p.npad_cap;
    u64 *k2 = a.keys2 + ob;
    const int *seg = a.seg + (long long)b * VX_MAX_CLASSES;
    const int *segk = a.segk + (long long)b * VX_MAX_CLASSES;
    int &s_count2 = *s_count2p;
#define VX_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[b * VX_STAMP_WORDS + (k)] = clock64(); } while (0)
    int n2;
    if (merge && srank && nseg > 1) {
        int *R = scratch;
        u64 *bm = reinterpret_cast<u64 *>(scratch + ((nseg + 2) & ~1));
        const int rows = (n + 63) >> 6;
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { int s = threadIdx.x * 4 + q; v[q] = s < nseg ? segk[s] : 0; tot += v[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        if (lane == 63) s_gcnt[wave] = incl;
        for (int i = threadIdx.x; i < rows; i += nt) bm[i] = 0ull;
        __syncthreads();
        int run = incl - tot;
        for (int w2 = 0; w2 < wave; ++w2) run += s_gcnt[w2];
#pragma unroll
        for (int q = 0; q < 4; ++q) { int s = threadIdx.x * 4 + q; if (s < nseg) R[s] = run; run += v[q]; }
        if (threadIdx.x == nt - 1) { R[nseg] = run; s_count2 = run; }
        __syncthreads();
        n2 = s_count2;
        // every kept key to the slot of its rank
        for (int o = threadIdx.x; o < n2; o += nt) {
            int lo2 = 0, hi2 = nseg;                 // invariant R[lo2] <= o < R[hi2]
            while (hi2 - lo2 > 1) { int mid = (lo2 + hi2) >> 1; if (R[mid] <= o) lo2 = mid; else hi2 = mid; }
            const u64 key = k2[seg[lo2] + (o - R[lo2])];
            const int j = srank[(unsigned)key];
            far[j] = key;
            atomicOr((unsigned long long *)&bm[j >> 6], 1ull << (j & 63));
        }
        __syncthreads();
        // compaction in rank order: occupied slots per 64-slot row -> scan -> scatter
        if (wave == 0) {
            int c[4], t4 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { int g = lane * 4 + q; c[q] = g < rows ? __popcll(bm[g]) : 0; t4 += c[q]; }
            int inc2 = t4;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(inc2, d); if (lane >= d) inc2 += y; }
            int r2 = inc2 - t4;
#pragma unroll
            for (int q = 0; q < 4; ++q) { int g = lane * 4 + q; if (g < rows) s_gcnt[g] = r2; r2 += c[q]; }
        }
        __syncthreads();
        for (int g = wave; g < rows; g += (nt >> 6)) {
            const u64 m = bm[g];
            if ((m >> lane) & 1ull) sk[s_gcnt[g] + vx_mbcnt(m)] = far[g * 64 + lane];
        }
        __syncthreads();
    } else if (merge) {
        // run boundaries R[0..nseg] = exclusive scan of the kept counts (4 per thread, wave scan, wave totals)
        u64 *bufA = sk, *bufB = sk + n;
        int *R = reinterpret_cast<int *>(sk + 2 * (size_t)n);
        int v[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { int s = threadIdx.x * 4 + q; v[q] = s < nseg ? segk[s] : 0; tot += v[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        if (lane == 63) s_gcnt[wave] = incl;
        __syncthreads();
        int run = incl - tot;
        for (int w2 = 0; w2 < wave; ++w2) run += s_gcnt[w2];
#pragma unroll
        for (int q = 0; q < 4; ++q) { int s = threadIdx.x * 4 + q; if (s < nseg) R[s] = run; run += v[q]; }
        if (threadIdx.x == nt - 1) { R[nseg] = run; s_count2 = run; }     // nt * 4 >= VX_MAX_CLASSES >= nseg
        __syncthreads();
        n2 = s_count2;
        // runs -> bufA, adjacent; rr = the run of each of this thread's positions (positions keep their run: merged
        // groups are unions of neighbouring runs)
        int rr[VX_IMG_MERGE_PER_THREAD];
#pragma unroll
        for (int k = 0; k < VX_IMG_MERGE_PER_THREAD; ++k) {
            const int o = threadIdx.x + k * nt;
            rr[k] = 0;
            if (o < n2) {
                int lo2 = 0, hi2 = nseg;             // invariant R[lo2] <= o < R[hi2]
                while (hi2 - lo2 > 1) { int mid = (lo2 + hi2) >> 1; if (R[mid] <= o) lo2 = mid; else hi2 = mid; }
                rr[k] = lo2;
                bufA[o] = k2[seg[lo2] + (o - R[lo2])];
            }
        }
        __syncthreads();
        u64 *src = bufA, *dst = bufB;
        // (A/B: advancing four searches of a thread in lockstep to overlap their LDS latencies was slower, 57 k vs
        // 48 k cycles on the dense YOLO batch -- with 16 waves per CU the phase is bound by its instruction count)
        for (int lg = 0; (1 << lg) < nseg; ++lg) {
#pragma unroll
            for (int k = 0; k < VX_IMG_MERGE_PER_THREAD; ++k) {
                const int o = threadIdx.x + k * nt;
                if (o < n2) {
                    const int G = (rr[k] >> lg) & ~1;    // groups of 2^lg runs; groups G and G + 1 are merged
                    const int a0 = R[min(G << lg, nseg)], m0 = R[min((G + 1) << lg, nseg)], b1 = R[min((G + 2) << lg, nseg)];
                    const bool second = o >= m0;
                    const u64 x = src[o];
                    int lo2 = second ? a0 : m0, hi2 = second ? m0 : b1;
                    const int sib0 = lo2;
                    while (lo2 < hi2) { int mid = (lo2 + hi2) >> 1; if (src[mid] < x) lo2 = mid + 1; else hi2 = mid; }
                    dst[a0 + (o - (second ? m0 : a0)) + (lo2 - sib0)] = x;
                }
            }
            __syncthreads();
            u64 *t = src; src = dst; dst = t;
        }
        if (src != sk) {                             // the outputs below read sk
            for (int i = threadIdx.x; i < n2; i += nt) dst[i] = src[i];
            __syncthreads();
        }
    } else {
        // kept keys back into global score order by a second sort
        n2 = s_count2;
        const int npad2 = vx_next_pow2(max(n2, 1));
        __syncthreads();                       // everyone has read s_count2 / finished with keys1 in sk
        for (int i = threadIdx.x; i < npad2; i += nt) sk[i] = i < n2 ? k2[i] : VX_KEY_SENTINEL;
        __syncthreads();
        vx_bitonic_sort_lds(sk, npad2, 0, 2, npad2);
    }
    VX_STAMP(6);
    if (n_redo > n && p.top_k > 0 && n2 < p.top_k) {        // the score prefix kept too few boxes: redo from all candidates
        if (threadIdx.x == 0) {
            a.sel[b] = n_redo;
            a.keep_counts[b] = 0;
            if (a.defer) atomicOr(a.status, VX_STATUS_DEFERRED);
            vx_status_or(a.status, rbits);
        }
        return;
    }
    int nk = n2;
    if (p.top_k > 0) nk = min(nk, p.top_k);
    nk = min(nk, p.out_cap);
    if (threadIdx.x == 0) a.keep_counts[b] = nk;
    const int D = p.D;
    for (int i = threadIdx.x; i < nk; i += nt)
        a.keep_idx[(long long)b * p.out_cap + i] = (long long)(sk[i] & 0xFFFFFFFFull);
    // row gather: consecutive threads copy consecutive floats of the output; four gathers in flight per thread (one at a
    // time, a 1 855-row x 16-float result was a chain of 29 dependent global loads per thread)
    const int tot = nk * D;
    for (int e0 = threadIdx.x; e0 < tot; e0 += 4 * nt) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nt;
            if (e < tot) {
                const int i = e / D, q = e - i * D;
                const int idx = (int)(sk[i] & 0xFFFFFFFFull);
                v[u] = a.det[(ib + idx) * D + q];
            }
        }
        float *dst = a.out_det + (long long)b * p.out_cap * D;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nt;
            if (e < tot) dst[e] = v[u];
        }
    }
    __syncthreads();
    VX_STAMP(7);
    if (threadIdx.x == 0) vx_status_or(a.status, rbits);
#undef VX_STAMP
}

// The compaction pass of the fused YOLO decode inside the NMS kernel (all threads of the workgroup; D == 6): tile counts
// -> exclusive scan -> every record to its dense, row-ordered position in the candidate arrays of image b.  Returns the
// image's candidate count (also written to counts[b]); ends synchronised, the arrays visible to the whole workgroup.
__device__ __forceinline__ int vx_image_finish(const vx_image_args &a, int b) {
    __shared__ int s_off[257];
    __shared__ int s_row0[256];
    __shared__ int s_wt[16];
    const vx_stage_info &f = a.fin;
    const int nt = blockDim.x, lane = vx_lane(), wave = threadIdx.x >> 6, nw = nt >> 6;
    const int t = threadIdx.x;
    int c = 0, row0 = 0;
    if (t < f.tiles) { c = f.tile_counts[(long long)b * f.tiles + t]; row0 = vx_stage_tile_row0(f, t); }
    // a tile of the decode that saw a NaN says so in bit 30 of its count (the decode kernel touches no status word)
    const bool tile_nan = (c & VX_TILE_NAN) != 0;
    c &= ~VX_TILE_NAN;
    if (__ballot(tile_nan) != 0ull && lane == 0) vx_status_or(a.status, VX_STATUS_NAN);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) s_wt[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nw; ++w) { const int v = s_wt[w]; if (w < wave) base += v; total += v; }
    if (t < 256) { s_off[t] = base + incl - c; s_row0[t] = row0; }
    if (t == 0) s_off[256] = total;
    __syncthreads();
    const long long ib = (long long)b * a.p.cap;
    const vx_rec *stage = reinterpret_cast<const vx_rec *>(f.stage) + (long long)b * f.R;
    float *boxes = const_cast<float *>(a.boxes), *scores = const_cast<float *>(a.scores), *det = const_cast<float *>(a.det);
    long long *cls = const_cast<long long *>(a.cls);
    const bool det8 = ((uintptr_t)det & 7) == 0;
    for (int j = threadIdx.x; j < total; j += nt) {
        int lo = 0, hi = 256;                        // largest tile index lt with s_off[lt] <= j
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_off[mid] <= j) lo = mid; else hi = mid; }
        const vx_rec r = stage[s_row0[lo] + (j - s_off[lo])];
        const long long dst = ib + j;
        *reinterpret_cast<vx_f4 *>(boxes + dst * 4) = r.a;
        scores[dst] = r.b.x;
        const int ci = __float_as_int(r.b.z);
        cls[dst] = (long long)ci;
        float *d = det + dst * 6;
        if (det8) {
            float2 *d2 = reinterpret_cast<float2 *>(d);
            d2[0] = make_float2(r.a.x, r.a.y); d2[1] = make_float2(r.a.z, r.a.w); d2[2] = make_float2(r.b.y, (float)ci);
        } else {
            d[0] = r.a.x; d[1] = r.a.y; d[2] = r.a.z; d[3] = r.a.w; d[4] = r.b.y; d[5] = (float)ci;
        }
    }
    if (threadIdx.x == 0) const_cast<int *>(a.counts)[b] = total;
    __syncthreads();                                 // (workgroup-scope release/acquire of the stores above)
    return total;
}

__device__ __forceinline__ void vx_nms_image_body(const vx_image_args &a, const int b) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];
    __shared__ vx_async_smem A;
    __shared__ union { vx_spatial_smem sp; vx_bm_prep_smem prep; } U;      // the spatial sweep and the bitmask prep never overlap
    vx_spatial_smem &SP = U.sp;
    __shared__ int s_big[VX_SWEEP_MAXBIG][3];
    __shared__ int s_nbig, s_count2, s_nseg, s_rbits, s_nexp;
    __shared__ int s_gcnt[VX_SORT_CHMAX / 64];
    __shared__ u64 s_wmask[16][64];
    __shared__ unsigned short s_pair_lut[VX_NPAIRS64];
    const vx_nms_params &p = a.p;
    const int nt = blockDim.x;
    const int n_all = a.fin_on ? vx_image_finish(a, b) : (a.counts ? vx_clamp_count(a.counts[b], p.cap) : p.cap);
    // score-prefix list (top_k on a large input, see vx_k_prefix_select): this kernel works on the list only
    const int pmb = a.pm ? a.pm[b] : -1;
    const int *pl = (pmb > 0 && pmb <= VX_PREFIX_CAP) ? a.plist + (long long)b * VX_PREFIX_CAP : nullptr;
    const int n = pl ? pmb : n_all;
    if (a.bm_on && threadIdx.x == 0) { a.bm.pend[2 * b] = 0; a.bm.pend[2 * b + 1] = 0; }
    if (a.sel) {                                     // capacity beyond the LDS-resident sort: only images that fit
        const bool big = n > a.n_max;
        if (threadIdx.x == 0) {
            a.sel[b] = big ? n : 0;
            if (big && a.defer) { atomicOr(a.status, VX_STATUS_DEFERRED); a.keep_counts[b] = 0; }
        }
        if (big) return;
    }
    if (n == 0) {
        if (threadIdx.x == 0) a.keep_counts[b] = 0;
        return;
    }
    if (n <= 64 && !pl) {                            // before any barrier: waves 1.. leave, wave 0 needs none
        if (threadIdx.x < 64) vx_nms_image_small(a, b, n);
        return;
    }
    const int npad = vx_next_pow2(n);
    const long long ib = (long long)b * p.cap, ob = (long long)b * p.npad_cap;
#define VX_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[b * VX_STAMP_WORDS + (k)] = clock64(); } while (0)
    VX_STAMP(0);
    bool bad = false;
    for (int i = threadIdx.x; i < npad; i += nt) {
        u64 key = VX_KEY_SENTINEL;
        if (i < n) {
            const int idx = pl ? pl[i] : i;
            long long c = a.cls[ib + idx];
            if (c < 0 || c >= p.n_classes) { bad = true; c = c < 0 ? 0 : p.n_classes - 1; }
            key = vx_key1((int)c, a.scores[ib + idx], idx);
        }
        sk[i] = key;
    }
    if (__ballot(bad) != 0 && vx_lane() == 0) atomicOr(a.status, VX_STATUS_BAD_CLASS);
    if (threadIdx.x == 0) { s_nbig = 0; s_count2 = 0; s_rbits = 0; s_nexp = 0; }
    vx_build_pair_lut(s_pair_lut);
    __syncthreads();
    VX_STAMP(1);
    // candidates in index order (no prefix list): the radix sort, its counters behind the keys
    // LDS behind the keys: the radix sort's counters, then (capacity <= 16384 only) the score rank of every candidate
    // and room for the kept keys: the tail then needs no merge rounds (see vx_image_tail)
    unsigned *rhist = reinterpret_cast<unsigned *>(sk + ((n + 1) & ~1));
    unsigned short *srank = reinterpret_cast<unsigned short *>(rhist + ((VX_RADIX_HIST_WORDS(nt >> 6) + 3) & ~3));
    u64 *far = reinterpret_cast<u64 *>(srank + ((p.cap + 7) & ~7));
    bool ranked = false;
    if (!pl && n >= VX_RADIX_MIN && vx_radix_fits(n, nt >> 6) && (size_t)n * sizeof(u64) + 16 + VX_RADIX_HIST_WORDS(nt >> 6) * sizeof(unsigned) <= a.lds_bytes) {
        const bool room = p.cap <= VX_SORT_CHMAX &&
                          (size_t)(reinterpret_cast<char *>(far + n) - reinterpret_cast<char *>(sk)) <= a.lds_bytes;
        ranked = vx_radix_sort_lds(sk, n, rhist, room ? srank : nullptr);
    } else {
        vx_bitonic_sort_lds(sk, npad, 0, 2, npad);
    }
    VX_STAMP(2);
    // sorted position -> class-offset box fl32(x + fl32(2049*cls)) (nms.py:94-96) and its area
    vx_f4 *sb = a.sbox + ob;
    float *sa = a.sarea + ob;
    u64 *kg = a.keys1 + ob;
    int oob = 0;
    for (int i = threadIdx.x; i < n; i += nt) {
        u64 key = sk[i];
        kg[i] = key;
        float off = (float)(vx_key1_cls(key) * (VX_MAX_COORDINATE + 1));
        vx_f4 bx = *reinterpret_cast<const vx_f4 *>(a.boxes + (ib + vx_key1_idx(key)) * 4);
        oob |= vx_range_bits(bx);
        bx.x = bx.x + off; bx.y = bx.y + off; bx.z = bx.z + off; bx.w = bx.w + off;
        sb[i] = bx;
        sa[i] = (bx.z - bx.x) * (bx.w - bx.y);
    }
    if (p.n_classes != 1) vx_flag_range_lds(oob, &s_rbits);
    __syncthreads();
    VX_STAMP(3);
    u64 *k2 = a.keys2 + ob;
    const vx_thr T = vx_make_thr(p.thr);
    const int wave = threadIdx.x >> 6, nw = nt >> 6, lane = vx_lane();
    // ordered list of class-segment starts: boundary flags -> per-64-group counts -> scan -> scatter
    int *seg = a.seg + (long long)b * VX_MAX_CLASSES;
    const int groups = (n + 63) >> 6;
    for (int g = wave; g < groups; g += nw) {
        int i = g * 64 + lane;
        bool flag = i < n && (i == 0 || vx_key1_cls(sk[i - 1]) != vx_key1_cls(sk[i]));
        u64 m = __ballot(flag);
        if (lane == 0) s_gcnt[g] = __popcll(m);
    }
    __syncthreads();
    if (wave == 0) {
        int c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { int g = lane * 4 + q; c[q] = g < groups ? s_gcnt[g] : 0; tot += c[q]; }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        int run = incl - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q) { int g = lane * 4 + q; if (g < groups) s_gcnt[g] = run; run += c[q]; }
        if (lane == 63) s_nseg = incl;
    }
    __syncthreads();
    for (int g = wave; g < groups; g += nw) {
        int i = g * 64 + lane;
        bool flag = i < n && (i == 0 || vx_key1_cls(sk[i - 1]) != vx_key1_cls(sk[i]));
        u64 m = __ballot(flag);
        if (flag) seg[s_gcnt[g] + vx_mbcnt(m)] = i;
    }
    __syncthreads();
    const int nseg = s_nseg;
    // Kept keys: every segment writes its kept keys IN ORDER to k2[lo ..] and their number to segk[s]; the runs are then
    // merged in LDS (ceil(log2 nseg) rounds, one binary search per key) -- a fraction of the work of a second bitonic
    // sort.  Needs two key buffers + the run boundaries in the launch's LDS; otherwise (or with one segment beyond
    // the list) the kept keys go through the image's counter and are sorted.
    const bool merge = nseg <= VX_MAX_CLASSES && n <= VX_IMG_MERGE_PER_THREAD * nt &&
                       ((size_t)2 * n * sizeof(u64) + (size_t)(nseg + 1) * sizeof(int)) <= a.lds_bytes;
    int *segk = a.segk + (long long)b * VX_MAX_CLASSES;
    for (int s = wave; s < nseg; s += nw) {
        int lo = seg[s];
        int hi = s + 1 < nseg ? seg[s + 1] : n;
        int len = hi - lo;
        if (len <= 64) {
            vx_sweep_pairs(sb, sa, sk, lo, len, T, s_wmask[wave], s_pair_lut, merge ? k2 + lo : k2, &s_count2, a.iou_count,
                           merge ? segk + s : nullptr);
        } else if (len <= 64 * VX_SWEEP_WT) {
            vx_sweep_wave(sb, sa, sk, lo, len, T, merge ? k2 + lo : k2, &s_count2, a.iou_count, merge ? segk + s : nullptr);
        } else if (lane == 0) {
            int q = atomicAdd(&s_nbig, 1);
            if (q < VX_SWEEP_MAXBIG) { s_big[q][0] = lo; s_big[q][1] = len; s_big[q][2] = s; }
        }
    }
    __syncthreads();
    VX_STAMP(4);
    // large segments, the whole workgroup each (chain sweep; its kept-box ring takes the keys' LDS -- the sorted keys
    // live on in kg)
    const int nbig = min(s_nbig, VX_SWEEP_MAXBIG);
    // bitmask path: a large tame segment is laid out by position and registered for vx_k_bm_pairs; vx_k_bm_finish then
    // resolves it from its edge list and runs this kernel's tail for the image (needs the ordered-run bookkeeping)
    const bool bm_ok = a.bm_on && merge && T.fast && !pl;      // (a prefix image must be decided here: it may have to be redone)
    for (int q = 0; q < nbig; ++q) {
        const int lo = s_big[q][0], len = s_big[q][1];
        int *kc = merge ? segk + s_big[q][2] : nullptr;
        u64 *kd = merge ? k2 + lo : k2;
        if (bm_ok && len > VX_BM_MIN && len <= VX_BM_MAXN && VX_BM_PREP_SCRATCH(len) <= a.lds_bytes &&
            vx_bm_prep(U.prep, sk, a.lds_bytes, a.bm, b, lo, s_big[q][2], 0, sb + lo, len, a.stamps ? (u64 *)(a.stamps + b * VX_STAMP_WORDS + 24) : nullptr)) {
            if (threadIdx.x == 0) { a.bm.img_exp[b * VX_BM_MAXEXP + s_nexp] = U.prep.si; ++s_nexp; }
            continue;
        }
        if (vx_sweep_spatial(SP, sk, a.lds_bytes, sb + lo, len, T, p.top_k, a.iou_count, a.stamps ? a.stamps + b * VX_STAMP_WORDS + 8 : nullptr)) {
            vx_spatial_emit(SP, kg + lo, kd, &s_count2, kc);
        } else {
            vx_sweep_async(A, sk, sb + lo, len, T, p.top_k, a.iou_count, a.stamps ? a.stamps + b * VX_STAMP_WORDS + 8 : nullptr);
            vx_async_emit(A, kg + lo, kd, &s_count2, kc);
        }
    }
    __syncthreads();
    VX_STAMP(5);
    if (s_nexp > 0) {                                // vx_k_bm_finish completes this image
        if (threadIdx.x == 0) {
            a.bm.pend[2 * b] = nseg; a.bm.pend[2 * b + 1] = s_nexp;
            vx_status_or(a.status, s_rbits);
        }
        return;
    }
    vx_image_tail(a, b, n, nseg, merge, sk, s_gcnt, &s_count2, s_rbits, pl ? n_all : 0,
                  ranked && nbig == 0 ? srank : nullptr, reinterpret_cast<int *>(rhist), far);
#undef VX_STAMP
}

// vx_k_pack_rows inside the NMS kernel.  The packed rows of image b start behind those of the images in front of it, so a
// workgroup publishes its keep count in one word and sums the words in front of it (polling those not there yet); when
// its rows are copied it sets a second word, and the workgroup of the last image waits for all of those before it
// completes the meta words (status) and mirrors them for the host.  No workgroup waits before it has published, and the
// host only takes this path when all B workgroups are resident at once (B <= compute units), so every wait ends.
// (Tickets and a "finished" counter were measured first: 256 returning atomics on one address cost 70 us.)
__device__ __forceinline__ void vx_image_pack(const vx_image_args &a, int b) {
    __shared__ int s_off;
    const vx_nms_params &p = a.p;
    const int lane = vx_lane(), wave = threadIdx.x >> 6, nt = blockDim.x;
    // (one 128-byte line per count word: 255 workgroups polling eight shared lines serialised on one memory channel,
    // 65 us for 256 images)
    int *flags = a.pack.sync + VX_PACK_HDR, *fin = flags + p.B * VX_PACK_STRIDE;
    __syncthreads();                                 // the body's stores (keep_counts[b], the image's out_det rows) are visible
    const int n = vx_clamp_count(a.keep_counts[b], p.out_cap);
    if (threadIdx.x == 0) __hip_atomic_store(&flags[b * VX_PACK_STRIDE], (n << 1) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
        int acc = 0;
        for (int gb = 0; gb < b; gb += 256) {        // four words per lane in flight (B <= 256: one round)
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = gb + k * 64 + lane;
                v[k] = g < b ? __hip_atomic_load(&flags[g * VX_PACK_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = gb + k * 64 + lane;
                while ((v[k] & 1) == 0) {
                    __builtin_amdgcn_s_sleep(8);
                    v[k] = __hip_atomic_load(&flags[g * VX_PACK_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                acc += v[k] >> 1;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) s_off = acc;
    }
    __syncthreads();
    const float *src = a.out_det + (long long)b * p.out_cap * p.D;
    float *dst = a.pack.dst + (long long)s_off * p.D;
    const int nf = n * p.D;
    if ((((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        for (int i = threadIdx.x; i < (nf >> 1); i += nt)
            __builtin_nontemporal_store(reinterpret_cast<const v2 *>(src)[i], reinterpret_cast<v2 *>(dst) + i);
        if ((nf & 1) && threadIdx.x == 0) dst[nf - 1] = src[nf - 1];
    } else {
        for (int i = threadIdx.x; i < nf; i += nt) __builtin_nontemporal_store(src[i], dst + i);
    }
    // "rows copied" word of this image; it carries the image's candidate count for the mirror.  No fence anywhere: an
    // agent-scope release on gfx950 writes the whole L2 back (256 of them cost 65 us); the words that other workgroups
    // read are agent-scope atomics themselves, everything else is read after the kernel.
    const int nc = a.counts ? vx_clamp_count(a.counts[b], p.cap) : p.cap;
    if (b != p.B - 1) {
        __syncthreads();                             // (waits for this workgroup's stores, the status atomic included)
        if (threadIdx.x == 0) __hip_atomic_store(&fin[b], (nc << 1) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // the last image's workgroup: once every other workgroup has finished the meta words are final -- it mirrors them
    // from the published words
    int *mirror = a.pack.mirror;
    for (int g = threadIdx.x; g < p.B; g += nt) {
        int c = (nc << 1) | 1, k = (n << 1) | 1;
        if (g < p.B - 1) {
            while (((c = __hip_atomic_load(&fin[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1) == 0)
                __builtin_amdgcn_s_sleep(8);
            k = __hip_atomic_load(&flags[g * VX_PACK_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mirror) { mirror[g] = c >> 1; mirror[p.B + g] = k >> 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int st = __hip_atomic_load(&a.pack.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.pack.meta[2 * p.B] = st;
        if (mirror) mirror[2 * p.B] = st;
    }
}

__global__ __launch_bounds__(1024) void vx_k_nms_image(vx_image_args a) {
    vx_nms_image_body(a, blockIdx.x);
    if (a.pack_on) vx_image_pack(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Bitmask path, last step on the image route: one workgroup per image that vx_k_nms_image left pending.  Every exported
// segment is resolved from its edge list (vx_bm_scan; the chain sweep when the list overflowed), its kept keys go, in
// order, to the segment's place in keys2 and their number to segk -- exactly what the sweeps of vx_k_nms_image leave --
// and the image is finished by that kernel's tail.
__global__ __launch_bounds__(1024) void vx_k_bm_finish(vx_image_args a) {
    extern __shared__ __attribute__((aligned(16))) u64 sk[];
    __shared__ union { vx_async_smem as; vx_bm_scan_smem sc; } U;
    __shared__ int s_gcnt[VX_SORT_CHMAX / 64];
    __shared__ int s_count2;
    const vx_nms_params &p = a.p;
    const int b = blockIdx.x;
    const int nseg = a.bm.pend[2 * b], nexp = a.bm.pend[2 * b + 1];
    if (nexp <= 0) return;
    const int n = a.counts ? vx_clamp_count(a.counts[b], p.cap) : p.cap;
    const long long ob = (long long)b * p.npad_cap;
    const vx_thr T = vx_make_thr(p.thr);
    int *segk = a.segk + (long long)b * VX_MAX_CLASSES;
    if (threadIdx.x == 0) s_count2 = 0;
#define VX_FSTAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[b * VX_STAMP_WORDS + 8 + (k)] = clock64(); } while (0)
    VX_FSTAMP(0);
    for (int q = 0; q < nexp; ++q) {
        const int si = a.bm.img_exp[b * VX_BM_MAXEXP + q];
        const vx_bm_seg sg = a.bm.segs[si];
        const long long region = ob + sg.lo;
        const u64 *kseg = a.keys1 + region;
        if (vx_bm_scan(U.sc, a.bm.edges + region * VX_BM_ECAP, a.bm.edges2 + region * VX_BM_ECAP, a.bm.ecount[si],
                       sg.len * VX_BM_ECAP, sg.len, p.top_k, reinterpret_cast<unsigned *>(sk), (int)(a.lds_bytes / sizeof(unsigned)),
                       a.stamps ? a.stamps + b * VX_STAMP_WORDS + 16 : nullptr)) {
            VX_FSTAMP(1);
            vx_masks_emit(U.sc.kept, U.sc.ndone, kseg, a.keys2 + region, nullptr, segk + sg.id);
        } else {
            vx_sweep_async(U.as, sk, a.sbox + region, sg.len, T, p.top_k, a.iou_count);
            vx_async_emit(U.as, kseg, a.keys2 + region, nullptr, segk + sg.id);
        }
    }
    __syncthreads();
    VX_FSTAMP(2);
    vx_image_tail(a, b, n, nseg, true, sk, s_gcnt, &s_count2, 0, 0);
    VX_FSTAMP(3);
#undef VX_FSTAMP
}

// ------------------------------------------------------------------------------------------------
static inline size_t vx_align256n(size_t x) { return (x + 255) & ~(size_t)255; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set on first use on each device.
// The per-kernel caches are atomics (threads may enter the library concurrently; setting the attribute twice is harmless).
#include <atomic>
struct vx_attr_cache { std::atomic<int> done[64]; };
static int vx_set_max_lds(vx_attr_cache &c, const void *kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VX_ERR_LAUNCH;
    if (c.done[dev].load(std::memory_order_acquire)) return VX_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return VX_ERR_LAUNCH;
    c.done[dev].store(1, std::memory_order_release);
    return VX_OK;
}

struct vx_nms_ws {
    u64 *keys1; vx_f4 *sbox; float *sarea; u64 *keys2; int *count2; u64 *stamps; int *seg; int *segk; int *sel;
    int *part_hist, *part_cursor, *part_segstart, *sel2;      // class-partitioned path (hist and cursor are adjacent)
    vx_runs_ws runs;                                          // bucket tables of the run merge (vx_k_runs_split)
    vx_bm_ws bm;                                              // bitmask path (capacities beyond the single-launch kernel only)
    int bm_on;
    int *plist, *pm;                                          // score-prefix lists (top_k; same capacities)
    int *big_list; int big_cap;                               // vx_part_args::big_list
    unsigned *pthr;
    size_t bytes;
};

static vx_nms_ws vx_nms_carve(void *ws, int B, int npad_cap) {
    vx_nms_ws w;
    size_t n = (size_t)B * npad_cap;
    char *p = (char *)ws;
    size_t off = 0;
    w.keys1 = (u64 *)(p + off); off += vx_align256n(n * sizeof(u64));
    w.sbox = (vx_f4 *)(p + off); off += vx_align256n(n * sizeof(vx_f4));
    w.sarea = (float *)(p + off); off += vx_align256n(n * sizeof(float));
    w.keys2 = (u64 *)(p + off); off += vx_align256n(n * sizeof(u64));
    w.count2 = (int *)(p + off); off += vx_align256n((size_t)B * sizeof(int));
    w.stamps = (u64 *)(p + off); off += vx_align256n((size_t)B * VX_STAMP_WORDS * sizeof(u64));   // per-image phase clocks (diagnostics)
    w.seg = (int *)(p + off); off += vx_align256n((size_t)B * VX_MAX_CLASSES * sizeof(int));
    w.segk = (int *)(p + off); off += vx_align256n((size_t)B * VX_MAX_CLASSES * sizeof(int));
    w.sel = (int *)(p + off); off += vx_align256n((size_t)B * sizeof(int));
    w.part_hist = (int *)(p + off); off += (size_t)B * VX_PART_MAX_CLASSES * sizeof(int);
    w.part_cursor = (int *)(p + off); off += vx_align256n((size_t)B * VX_PART_MAX_CLASSES * sizeof(int));
    w.part_segstart = (int *)(p + off); off += vx_align256n((size_t)B * (VX_PART_MAX_CLASSES + 1) * sizeof(int));
    w.sel2 = (int *)(p + off); off += vx_align256n((size_t)B * sizeof(int));
    w.runs.nb = (int *)(p + off); off += vx_align256n((size_t)B * sizeof(int));
    w.runs.list = (int *)(p + off); off += vx_align256n((size_t)B * VX_RUNS_MAXB * sizeof(int));
    w.runs.nlist = nullptr;                                   // = bm.counters + 2 (set below, zeroed with the counter block)
    w.runs.bstart = (int *)(p + off); off += vx_align256n((size_t)B * (VX_RUNS_MAXB + 1) * sizeof(int));
    w.runs.lb = (int *)(p + off);
    if (npad_cap > VX_SORT_CHMAX) off += vx_align256n((size_t)B * VX_RUNS_MAXB * VX_PART_MAX_CLASSES * sizeof(int));
    // Bitmask path (vx_nms_bm.h).  Only capacities beyond the single-launch kernel carry it: below that the whole NMS
    // is one launch and stays so.
    w.bm = vx_bm_ws();
    w.bm_on = npad_cap > VX_SORT_CHMAX;
    if (w.bm_on) {
        const int msi = vx_bm_max_segs_per_image(npad_cap);
        w.bm.npad_cap = npad_cap;
        w.bm.max_segs = B * msi;
        w.bm.max_items = B * (npad_cap / 64 + msi);
        w.bm.counters = (int *)(p + off); off += VX_BM_COUNTER_BYTES;
        w.runs.nlist = w.bm.counters + 2;
        w.bm.segs = (vx_bm_seg *)(p + off); off += vx_align256n((size_t)w.bm.max_segs * sizeof(vx_bm_seg));
        w.bm.ecount = (int *)(p + off); off += vx_align256n((size_t)w.bm.max_segs * sizeof(int));
        w.bm.item_seg = (int *)(p + off); off += vx_align256n((size_t)w.bm.max_items * sizeof(int));
        w.bm.tbb = (vx_f4 *)(p + off); off += vx_align256n((size_t)w.bm.max_items * sizeof(vx_f4));
        w.bm.pend = (int *)(p + off); off += vx_align256n((size_t)B * 2 * sizeof(int));
        w.bm.img_exp = (int *)(p + off); off += vx_align256n((size_t)B * VX_BM_MAXEXP * sizeof(int));
        w.bm.recs = (vx_bm_rec *)(p + off); off += vx_align256n(n * sizeof(vx_bm_rec));
        w.bm.edges = (unsigned *)(p + off); off += vx_align256n(n * VX_BM_ECAP * sizeof(unsigned));
        w.bm.edges2 = (unsigned *)(p + off); off += vx_align256n(n * VX_BM_ECAP * sizeof(unsigned));
        w.plist = (int *)(p + off); off += vx_align256n((size_t)B * VX_PREFIX_CAP * sizeof(int));
        w.pm = (int *)(p + off); off += vx_align256n((size_t)B * sizeof(int));
        w.pthr = (unsigned *)(p + off); off += vx_align256n((size_t)B * sizeof(unsigned));
        w.big_cap = B * (npad_cap / VX_PART_SMALL + 1);       // segments of more than VX_PART_SMALL keys
        w.big_list = (int *)(p + off); off += vx_align256n((size_t)w.big_cap * sizeof(int));
    } else {
        w.plist = nullptr; w.pm = nullptr; w.pthr = nullptr; w.big_list = nullptr; w.big_cap = 0;
    }
    w.bytes = off;
    return w;
}

extern "C" size_t vx_batched_nms_workspace_bytes(int B, int capacity) {
    if (B <= 0 || capacity <= 0 || capacity > VX_MAX_CANDIDATES) return 0;
    return vx_nms_carve(nullptr, B, vx_next_pow2(capacity)).bytes;
}

static int vx_sort_launch(u64 *keys, const int *counts, int fixed_n, int B, int npad_cap, int epilogue,
                          const vx_epi_params &ep, hipStream_t st) {
    static vx_attr_cache attr_set;
    // A/B on MI355X (FPN-SSD b64 / stress b32, both sorts): chunks of 16384 keys 0.68 / 0.67 ms, 8192 0.77 / 0.74, 4096 0.76 / 0.89
    int CH = npad_cap < VX_SORT_CHMAX ? npad_cap : VX_SORT_CHMAX;
    size_t lds = (size_t)CH * sizeof(u64);
    // 16384 keys x 8 B = 128 KiB of the CU's 160 KiB LDS
    if (vx_set_max_lds(attr_set, (const void *)vx_k_sort_local, VX_SORT_CHMAX * (int)sizeof(u64)) != VX_OK) return VX_ERR_LAUNCH;
    int threads = CH / 2 < 1024 ? (CH / 2 < 64 ? 64 : CH / 2) : 1024;
    vx_sort_params sp;
    sp.stride = npad_cap; sp.CH = CH; sp.k_merge = 0; sp.fixed_n = fixed_n; sp.epilogue = epilogue;
    dim3 grid_local(npad_cap / CH, B);
    hipLaunchKernelGGL(vx_k_sort_local, grid_local, dim3(threads), lds, st, keys, counts, sp, ep);
    for (long long k = 2LL * CH; k <= npad_cap; k <<= 1) {
        for (long long j = k >> 1; j >= CH; j >>= 1)
            hipLaunchKernelGGL(vx_k_sort_global, dim3((npad_cap / 2 + 255) / 256, B), dim3(256), 0, st, keys, counts,
                               npad_cap, fixed_n, (int)k, (int)j);
        sp.k_merge = (int)k;
        hipLaunchKernelGGL(vx_k_sort_local, grid_local, dim3(threads), lds, st, keys, counts, sp, ep);
    }
    return VX_OK;
}

static int vx_nms_impl(const float *d_boxes, const float *d_scores, const int64_t *d_cls, const float *d_det,
                       const int32_t *d_counts, int B, int capacity, int D, int n_classes,
                       float iou_threshold, int top_k, int64_t *d_keep_idx, float *d_out_det,
                       int32_t *d_keep_counts, int out_capacity, int32_t *d_status, void *d_workspace,
                       size_t workspace_bytes, void *stream, int small_only, const vx_stage_info *fin = nullptr,
                       const vx_pack_info *pack = nullptr) {
    if (!d_boxes || !d_scores || !d_cls || !d_det || !d_keep_idx || !d_out_det || !d_keep_counts || !d_status ||
        !d_workspace)
        return VX_ERR_INVALID_ARG;
    // candidates still staged per tile: only the single-launch kernel packs them (the caller checked the rest)
    if (fin && (vx_next_pow2(capacity) > VX_SORT_CHMAX || fin->tiles > 256 || fin->R != capacity || D != 6 || !d_counts ||
                ((uintptr_t)d_det & 3)))
        return VX_ERR_UNSUPPORTED;
    if (pack && (vx_next_pow2(capacity) > VX_SORT_CHMAX || !pack->dst || !pack->sync || !pack->meta)) return VX_ERR_UNSUPPORTED;
    if (B <= 0 || capacity <= 0 || D <= 0 || out_capacity <= 0) return VX_ERR_INVALID_ARG;
    if (!(iou_threshold >= 0.0f)) return VX_ERR_UNSUPPORTED;     // see header: class segmentation needs thr >= 0
    if (capacity > VX_MAX_CANDIDATES || B > 65535) return VX_ERR_UNSUPPORTED;
    if (n_classes <= 0 || n_classes > VX_MAX_CLASSES) n_classes = VX_MAX_CLASSES;
    int need_out = (top_k > 0 && top_k < capacity) ? top_k : capacity;
    if (out_capacity < need_out) return VX_ERR_INVALID_ARG;
    if (((uintptr_t)d_boxes & 15) || ((uintptr_t)d_workspace & 255) || ((uintptr_t)d_cls & 7) ||
        ((uintptr_t)d_keep_idx & 7))
        return VX_ERR_ALIGNMENT;
    int npad_cap = vx_next_pow2(capacity);
    vx_nms_ws w = vx_nms_carve(d_workspace, B, npad_cap);
    if (workspace_bytes < w.bytes) return VX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    vx_nms_params p;
    p.B = B; p.cap = capacity; p.npad_cap = npad_cap; p.D = D; p.n_classes = n_classes; p.top_k = top_k;
    p.out_cap = out_capacity; p.thr = iou_threshold;

    // Single-launch path: one workgroup per image, keys stay in LDS.  With a capacity beyond the LDS-resident sort
    // it still runs first and finishes every image that has at most VX_SORT_CHMAX candidates (the realistic case:
    // a handful per image whatever the capacity); the multi-kernel path below then only works on the others
    // (sel[b] = their count, 0 = done) -- for an all-small batch its launches find nothing to do.
    const bool all_fit = npad_cap <= VX_SORT_CHMAX;
    const bool partitioned = n_classes >= VX_PART_MIN_CLASSES && n_classes <= VX_PART_MAX_CLASSES;
    vx_image_args ia;
    size_t ia_lds = 0;
    int ia_threads = 1024;
    {
        static vx_attr_cache image_attr_set;
        if (vx_set_max_lds(image_attr_set, (const void *)vx_k_nms_image, VX_SORT_CHMAX * (int)sizeof(u64)) != VX_OK)
            return VX_ERR_LAUNCH;
        vx_image_args a;
        a.p = p; a.boxes = d_boxes; a.scores = d_scores; a.det = d_det; a.cls = (const long long *)d_cls;
        a.counts = d_counts; a.sbox = w.sbox; a.sarea = w.sarea; a.keys2 = w.keys2;
        a.keep_idx = (long long *)d_keep_idx; a.out_det = d_out_det; a.keep_counts = d_keep_counts;
        a.status = d_status; a.stamps = vx_prof_on ? w.stamps : nullptr; a.seg = w.seg; a.segk = w.segk;   // phase clocks only while profiling
        // the optimistic call stops behind this kernel unless classes are declared (then the class-partitioned path is part
        // of the bet) -- and always when a top_k lets the score prefix do the work: an image it cannot finish is deferred
        a.sel = all_fit ? nullptr : w.sel; a.n_max = VX_SORT_CHMAX; a.defer = small_only && (!partitioned || top_k > 0);
        a.keys1 = w.keys1; a.iou_count = vx_iou_counter();
        a.bm = w.bm; a.bm_on = w.bm_on;
        a.plist = nullptr; a.pm = nullptr;
        a.fin_on = fin ? 1 : 0;
        if (fin) a.fin = *fin;
        a.pack_on = pack ? 1 : 0;
        if (pack) a.pack = *pack;
        if (!all_fit && top_k > 0) {                 // top_k on a large input: score prefix first (see vx_k_prefix_select)
            vx_prof_begin(VX_PROF_NMS_KEYS, st);
            hipLaunchKernelGGL(vx_k_prefix_thr, dim3(B), dim3(1024), 0, st, d_scores, d_counts, p, w.pthr, w.pm);
            hipLaunchKernelGGL(vx_k_prefix_compact, dim3((capacity + VX_PREFIX_CHUNK - 1) / VX_PREFIX_CHUNK, B), dim3(1024), 0, st,
                               d_scores, d_counts, p, w.pthr, w.plist, w.pm);
            vx_prof_end(VX_PROF_NMS_KEYS, st);
            a.plist = w.plist; a.pm = w.pm;
        }
        if (w.bm_on && hipMemsetAsync(w.bm.counters, 0, VX_BM_COUNTER_BYTES, st) != hipSuccess) return VX_ERR_LAUNCH;
        const int lds_keys = all_fit ? npad_cap : VX_SORT_CHMAX;
        int threads = lds_keys >= 1024 ? 1024 : 256;
        // the same LDS holds the sort keys and later the kept-box ring of the column-owner sweep
        size_t lds = (size_t)lds_keys * sizeof(u64);
        if (lds < vx_async_ring_bytes(threads / 64)) lds = vx_async_ring_bytes(threads / 64);
        // the spatial sweep of a large class segment wants 20 bytes per box
        {
            size_t want = (size_t)((capacity + 63) / 64) * 64 * 20;
            if (want > (size_t)VX_SORT_CHMAX * sizeof(u64)) want = (size_t)VX_SORT_CHMAX * sizeof(u64);
            if (lds < want) lds = want;
        }
        a.lds_bytes = lds;
        vx_prof_begin(VX_PROF_NMS_IMAGE, st);
        hipLaunchKernelGGL(vx_k_nms_image, dim3(B), dim3(threads), lds, st, a);
        vx_prof_end(VX_PROF_NMS_IMAGE, st);
        if (all_fit) return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
        ia = a; ia_lds = lds; ia_threads = threads;
    }
    // bitmask path: pairs of every registered segment over the whole chip, then the per-image / per-segment scans
    auto bm_launch = [&](const vx_part_args *part) -> int {
        static vx_attr_cache finish_attr_set;
        if (vx_set_max_lds(finish_attr_set, (const void *)vx_k_bm_finish, VX_SORT_CHMAX * (int)sizeof(u64)) != VX_OK) return VX_ERR_LAUNCH;
        int wgs = (w.bm.max_items + 3) / 4;          // one wave per work item, at most the waves the chip holds (256 CUs x 8 WGs)
        if (wgs > 4096) wgs = 4096;
        vx_prof_begin(VX_PROF_BM_PAIRS, st);
        hipLaunchKernelGGL(vx_k_bm_pairs, dim3(wgs), dim3(256), 0, st, w.bm, p.thr, vx_iou_counter());
        vx_prof_end(VX_PROF_BM_PAIRS, st);
        vx_prof_begin(VX_PROF_BM_SCAN, st);
        hipLaunchKernelGGL(vx_k_bm_finish, dim3(B), dim3(ia_threads), ia_lds, st, ia);
        if (part) {
            int sg = w.bm.max_segs < 4096 ? w.bm.max_segs : 4096;
            hipLaunchKernelGGL(vx_k_bm_scan, dim3(sg), dim3(256), VX_BM_SCAN_LDS_EDGES * sizeof(unsigned), st, *part);
        }
        vx_prof_end(VX_PROF_BM_SCAN, st);
        return VX_OK;
    };
    if (small_only && (!partitioned || top_k > 0)) {
        if (bm_launch(nullptr) != VX_OK) return VX_ERR_LAUNCH;
        return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
    }
    d_counts = w.sel;                                  // from here on: only the images the kernel above left

    const bool topk_select = top_k > 0 && top_k <= VX_TOPK_MAX;
    // kept keys as ordered per-class runs + merge rounds instead of the second sort (not with the top_k select,
    // which reads an unordered key set)
    const bool run_merge = partitioned && !topk_select;
    vx_part_args pa = {};
    if (partitioned) {
        static vx_attr_cache part_attr_set;
        if (vx_set_max_lds(part_attr_set, (const void *)vx_k_part_sweep<VX_SORT_CHMAX, 1024>, VX_SORT_CHMAX * (int)sizeof(u64)) != VX_OK)
            return VX_ERR_LAUNCH;
        vx_part_args a;
        a.p = p; a.boxes = d_boxes; a.scores = d_scores; a.cls = (const long long *)d_cls; a.sel = w.sel;
        a.hist = w.part_hist; a.cursor = w.part_cursor; a.segstart = w.part_segstart; a.sel2 = w.sel2;
        a.keys1 = w.keys1; a.keys2 = w.keys2; a.sbox = w.sbox; a.sarea = w.sarea; a.count2 = w.count2; a.status = d_status;
        a.iou_count = vx_iou_counter();
        a.ordered = run_merge ? 1 : 0;
        a.defer = small_only; a.keep_counts = d_keep_counts;
        a.bm = w.bm; a.bm_on = w.bm_on;
        // (the counter block is zeroed once per call, before vx_k_nms_image)
        a.big_count = w.bm_on ? w.bm.counters + 3 : nullptr; a.big_list = w.big_list; a.big_cap = w.big_cap;
        pa = a;
        if (hipMemsetAsync(w.part_hist, 0, (size_t)2 * B * VX_PART_MAX_CLASSES * sizeof(int), st) != hipSuccess)
            return VX_ERR_LAUNCH;
        dim3 tiles((capacity + VX_PART_TILE - 1) / VX_PART_TILE, B);
        vx_prof_begin(VX_PROF_NMS_KEYS, st);
        hipLaunchKernelGGL(vx_k_part_hist, tiles, dim3(1024), 0, st, a);
        hipLaunchKernelGGL(vx_k_part_scatter, tiles, dim3(1024), 0, st, a);
        vx_prof_end(VX_PROF_NMS_KEYS, st);
        vx_prof_begin(VX_PROF_SWEEP, st);
        // dynamic LDS: the segment's keys while sorting, then the scratch of the bitmask prep or the ring of the chain sweep
        size_t small_lds = VX_PART_SMALL * sizeof(u64);
        if (small_lds < VX_BM_PREP_SCRATCH(VX_PART_SMALL)) small_lds = VX_BM_PREP_SCRATCH(VX_PART_SMALL);
        if (small_lds < vx_async_ring_bytes(8)) small_lds = vx_async_ring_bytes(8);
        hipLaunchKernelGGL((vx_k_part_sweep<VX_PART_SMALL, 512>), dim3(n_classes, B), dim3(512), small_lds, st, a);
        if (a.big_count) {
            int gb = w.big_cap < 512 ? w.big_cap : 512;
            hipLaunchKernelGGL((vx_k_part_sweep<VX_SORT_CHMAX, 1024>), dim3(gb), dim3(1024), VX_SORT_CHMAX * sizeof(u64), st, a);
        } else {
            hipLaunchKernelGGL((vx_k_part_sweep<VX_SORT_CHMAX, 1024>), dim3(n_classes, B), dim3(1024), VX_SORT_CHMAX * sizeof(u64), st, a);
        }
        vx_prof_end(VX_PROF_SWEEP, st);
        d_counts = w.sel2;                             // the general path below: images with a segment beyond the LDS sort
    }
    if (bm_launch(partitioned ? &pa : nullptr) != VX_OK) return VX_ERR_LAUNCH;

    // the general path (global sort, one workgroup per class for the sweep) takes the images left over; the optimistic
    // call with declared classes has deferred them instead (vx_k_part_scatter) and skips these launches
    const bool general = !(small_only && partitioned);
    vx_epi_params ep = {};
    ep.boxes = d_boxes; ep.sbox = w.sbox; ep.sarea = w.sarea; ep.cap = capacity;
    ep.status = d_status; ep.guard = n_classes != 1; ep.sel = w.sel;
    ep.det = d_det; ep.keep_idx = (long long *)d_keep_idx; ep.out_det = d_out_det; ep.keep_counts = d_keep_counts;
    ep.D = D; ep.top_k = top_k; ep.out_cap = out_capacity;
    int rc = VX_OK;
    if (general) {
        vx_prof_begin(VX_PROF_NMS_KEYS, st);
        hipLaunchKernelGGL(vx_k_nms_build_keys, dim3((capacity + 255) / 256, B), dim3(256), 0, st, d_scores,
                           (const long long *)d_cls, d_counts, p, w.keys1, w.count2, d_status);
        vx_prof_end(VX_PROF_NMS_KEYS, st);
        vx_prof_begin(VX_PROF_SORT1, st);
        rc = vx_sort_launch(w.keys1, d_counts, capacity, B, npad_cap, 1, ep, st);
        vx_prof_end(VX_PROF_SORT1, st);
        if (rc != VX_OK) return rc;
        // One class per wave and pass; <= 4096/256 = 16 <= VX_SWEEP_MAXBIG classes per workgroup.
        // A/B on MI355X (tools/bench_configs.py), sweep kernel ms for (threads, workgroups per image):
        //   FPN-SSD 20 classes x ~1500 boxes, b64:   (256,5) 2.43  (256,20) 0.88  (512,20) 0.84  (1024,20) 1.01
        //   stress 80 classes x 1250 boxes, b32:     (256,20) 2.10 (256,80) 1.20  (512,80) 1.12  (1024,80) 1.51
        // -> one workgroup per class (8 waves) up to 256 workgroups per image; <= 8 classes: 16 waves per class.
        const int sweep_threads = n_classes <= 8 ? 1024 : 512;
        int X = n_classes <= 256 ? n_classes : 256;
        if (X < 1) X = 1;
        vx_prof_begin(VX_PROF_SWEEP, st);
        hipLaunchKernelGGL(vx_k_nms_sweep, dim3(X, B), dim3(sweep_threads), 0, st, w.keys1, w.sbox, w.sarea, d_counts, p, w.keys2,
                           w.count2, vx_iou_counter(), run_merge ? w.part_hist : nullptr);
        vx_prof_end(VX_PROF_SWEEP, st);
    }
    vx_prof_begin(VX_PROF_SORT2, st);
    if (topk_select) {                                 // only the first top_k keys are needed: select, then sort those
        hipLaunchKernelGGL(vx_k_topk_select, dim3(B), dim3(1024), 0, st, w.keys2, w.count2, w.sel, p, ep);
    } else if (!run_merge) {
        rc = vx_sort_launch(w.keys2, w.count2, 0, B, npad_cap, 2, ep, st);
    } else {
        static vx_attr_cache bucket_attr_set;
        if (vx_set_max_lds(bucket_attr_set, (const void *)vx_k_runs_bucket, 2 * VX_RUNS_CAP * (int)sizeof(u64)) != VX_OK) return VX_ERR_LAUNCH;
        hipLaunchKernelGGL(vx_k_runs_split, dim3(B), dim3(1024), 0, st, pa, w.runs, ep);
        int bwgs = B * ((capacity + VX_RUNS_BUCKET - 1) / VX_RUNS_BUCKET);      // buckets the call can have at most
        if (bwgs > B * VX_RUNS_MAXB) bwgs = B * VX_RUNS_MAXB;
        if (bwgs > 2048) bwgs = 2048;
        hipLaunchKernelGGL(vx_k_runs_bucket, dim3(bwgs), dim3(VX_RUNS_THREADS), 2 * VX_RUNS_CAP * sizeof(u64), st, pa, w.runs, ep);
    }
    vx_prof_end(VX_PROF_SORT2, st);
    if (rc != VX_OK) return rc;
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

extern "C" int vx_batched_nms(const float *d_boxes, const float *d_scores, const int64_t *d_cls, const float *d_det,
                              const int32_t *d_counts, int B, int capacity, int D, int n_classes,
                              float iou_threshold, int top_k, int64_t *d_keep_idx, float *d_out_det,
                              int32_t *d_keep_counts, int out_capacity, int32_t *d_status, void *d_workspace,
                              size_t workspace_bytes, void *stream) {
    return vx_nms_impl(d_boxes, d_scores, d_cls, d_det, d_counts, B, capacity, D, n_classes, iou_threshold, top_k,
                       d_keep_idx, d_out_det, d_keep_counts, out_capacity, d_status, d_workspace, workspace_bytes, stream, 0);
}

// vx_batched_nms whose single-launch kernel starts from the per-tile records of the fused YOLO decode (declared in
// vx_common.h): spares the vx_k_cand_finish launch; the candidate arrays and d_counts are written by the kernel.
int vx_internal_nms_from_stage(const vx_stage_info &info, float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                               int32_t *d_counts, int B, int capacity, int D, int n_classes, float iou_threshold,
                               int top_k, int64_t *d_keep_idx, float *d_out_det, int32_t *d_keep_counts,
                               int out_capacity, int32_t *d_status, void *d_workspace, size_t workspace_bytes,
                               void *stream, const vx_pack_info *pack) {
    return vx_nms_impl(d_boxes, d_scores, d_cls, d_det, d_counts, B, capacity, D, n_classes, iou_threshold, top_k,
                       d_keep_idx, d_out_det, d_keep_counts, out_capacity, d_status, d_workspace, workspace_bytes, stream, 0,
                       &info, pack);
}

// Optimistic variant for capacities beyond the single-launch kernel: ONE launch; an image with more than 16384
// candidates is not processed but reported through VX_STATUS_DEFERRED (its keep count is 0) -- the caller then
// repeats the call with vx_batched_nms.  Saves the ~12 launches of the general path, which find nothing to do
// when every image is small (the normal case for prior-box models: capacity 50 400, a few candidates).
extern "C" int vx_batched_nms_small(const float *d_boxes, const float *d_scores, const int64_t *d_cls, const float *d_det,
                                    const int32_t *d_counts, int B, int capacity, int D, int n_classes,
                                    float iou_threshold, int top_k, int64_t *d_keep_idx, float *d_out_det,
                                    int32_t *d_keep_counts, int out_capacity, int32_t *d_status, void *d_workspace,
                                    size_t workspace_bytes, void *stream) {
    return vx_nms_impl(d_boxes, d_scores, d_cls, d_det, d_counts, B, capacity, D, n_classes, iou_threshold, top_k,
                       d_keep_idx, d_out_det, d_keep_counts, out_capacity, d_status, d_workspace, workspace_bytes, stream, 1);
}

// ------------------------------------------------------------------------------------------------
// vx_batched_nms_noalloc: the same call for callers that cannot allocate (HIP graph capture, tight serving loops,
// harnesses that forbid hipMalloc).  The workspace of vx_batched_nms lives in a file-scope device array; every GPU
// the library runs on holds its own copy at its own address until the process ends, so the address is looked up
// once per device (hipGetSymbolAddress) and only that pointer is ever handed to kernels / memsets -- the array's
// name is a host-side stand-in, not a device address.
#define VX_NOALLOC_SCRATCH_BYTES ((size_t)64 << 20)
static __device__ __align__(256) unsigned char vx_nms_scratch[VX_NOALLOC_SCRATCH_BYTES];

#include <mutex>
static int vx_noalloc_scratch(void **out) {
    static std::mutex mu;
    static void *ptr[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VX_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {                                   // set-up: first call on this device (with it current)
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(vx_nms_scratch)) != hipSuccess || !p) return VX_ERR_LAUNCH;
        ptr[dev] = p;
    }
    *out = ptr[dev];
    return VX_OK;
}

extern "C" size_t vx_batched_nms_noalloc_max_bytes(void) { return VX_NOALLOC_SCRATCH_BYTES; }

extern "C" int vx_batched_nms_noalloc(const float *d_boxes, const float *d_scores, const int64_t *d_cls, const float *d_det,
                                      const int32_t *d_counts, int B, int capacity, int D, int n_classes,
                                      float iou_threshold, int top_k, int64_t *d_keep_idx, float *d_out_det,
                                      int32_t *d_keep_counts, int out_capacity, int32_t *d_status, void *stream) {
    if (B <= 0 || capacity <= 0) return VX_ERR_INVALID_ARG;
    if (capacity > VX_MAX_CANDIDATES) return VX_ERR_UNSUPPORTED;
    const size_t need = vx_nms_carve(nullptr, B, vx_next_pow2(capacity)).bytes;
    if (need > VX_NOALLOC_SCRATCH_BYTES) return VX_ERR_SCRATCH_LIMIT;      // B * pow2ceil(capacity) too large
    void *scratch = nullptr;
    int rc = vx_noalloc_scratch(&scratch);
    if (rc != VX_OK) return rc;
    // every word of the scratch that a kernel expects to start at zero (count2, the class histogram and cursors) is
    // cleared by vx_nms_impl itself on each call; the rest is written before it is read
    return vx_nms_impl(d_boxes, d_scores, d_cls, d_det, d_counts, B, capacity, D, n_classes, iou_threshold, top_k,
                       d_keep_idx, d_out_det, d_keep_counts, out_capacity, d_status, scratch, need, stream, 0);
}

// ------------------------------------------------------------------------------------------------
// Tuple packing (the `tuple(results)` of base_postprocess.py:131-138 without B Python slices):
// rows [0, counts[b]) of every image are copied back to back; offsets[b] = sum(counts[0..b)).
#define VX_PACK_ROWS 256
#define VX_PACK_MAXG 16              // workgroups per image (more for small batches)
__global__ __launch_bounds__(256) void vx_k_pack_rows(const float *__restrict__ src, const int *__restrict__ counts,
                                                      int B, int cap, int D, float *__restrict__ dst,
                                                      int *__restrict__ total, const int *__restrict__ meta_src,
                                                      int *__restrict__ meta_dst, int n_meta) {
    __shared__ int s_part[256];
    // workgroup (g, b) copies the row chunks g, g + gridDim.x, ... (VX_PACK_ROWS rows each) of image b: a bounded grid
    // (the keep counts are not known on the host: one workgroup per possible chunk meant thousands of empty ones for a
    // capacity of 50 400) and enough workgroups per image for a few thousand kept rows
    const int b = blockIdx.y, g = blockIdx.x;
    const int n = vx_clamp_count(counts[b], cap);
    if (g > 0 && g * VX_PACK_ROWS >= n) return;
    // the counts / status words of the call chain, mirrored for the host (meta_dst may be pinned host memory:
    // plain stores, visible after the stream's next event) so the path needs no D2H memcpy
    if (b == 0 && g == 0 && meta_dst)
        for (int i = threadIdx.x; i < n_meta; i += 256) meta_dst[i] = meta_src[i];
    int acc = 0;
    for (int i = threadIdx.x; i < b; i += 256) acc += vx_clamp_count(counts[i], cap);
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
        __syncthreads();
    }
    const long long off = s_part[0];
    if (b == B - 1 && g == 0 && threadIdx.x == 0 && total) *total = (int)off + n;
    for (int r0 = g * VX_PACK_ROWS; r0 < n; r0 += gridDim.x * VX_PACK_ROWS) {
        const int r1 = min(n, r0 + VX_PACK_ROWS);
        const float *s = src + ((long long)b * cap + r0) * D;
        float *d = dst + (off + r0) * D;
        // final output, read by the host side only: non-temporal stores keep it from displacing the next batch's streams
        const int nf = (r1 - r0) * D;
        if ((((uintptr_t)s | (uintptr_t)d) & 7) == 0) {
            typedef float v2 __attribute__((ext_vector_type(2)));
            for (int i = threadIdx.x; i < (nf >> 1); i += 256)
                __builtin_nontemporal_store(reinterpret_cast<const v2 *>(s)[i], reinterpret_cast<v2 *>(d) + i);
            if ((nf & 1) && threadIdx.x == 0) d[nf - 1] = s[nf - 1];
        } else {
            for (int i = threadIdx.x; i < nf; i += 256) __builtin_nontemporal_store(s[i], d + i);
        }
    }
}

extern "C" int vx_pack_rows(const float *d_src, const int32_t *d_counts, int B, int capacity, int D, float *d_dst,
                            int32_t *d_total, const int32_t *d_meta_src, int32_t *meta_dst, int n_meta, void *stream) {
    if (!d_src || !d_counts || !d_dst) return VX_ERR_INVALID_ARG;
    if (B <= 0 || capacity <= 0 || D <= 0) return VX_ERR_INVALID_ARG;
    if (meta_dst && (!d_meta_src || n_meta <= 0)) return VX_ERR_INVALID_ARG;
    if (B > 65535) return VX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_PACK, st);
    int gx = (capacity + VX_PACK_ROWS - 1) / VX_PACK_ROWS;
    const int maxg = 2048 / B > VX_PACK_MAXG ? 2048 / B : VX_PACK_MAXG;     // about 2048 workgroups in all
    if (gx > maxg) gx = maxg;
    hipLaunchKernelGGL(vx_k_pack_rows, dim3(gx, B), dim3(256), 0, st, d_src,
                       d_counts, B, capacity, D, d_dst, d_total, d_meta_src, meta_dst, n_meta);
    vx_prof_end(VX_PROF_PACK, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

// diagnostics: byte offset of the u64[B][8] phase-stamp block inside the NMS workspace
extern "C" size_t vx_batched_nms_stamps_offset(int B, int capacity) {
    if (B <= 0 || capacity <= 0 || capacity > VX_MAX_CANDIDATES) return 0;
    vx_nms_ws w = vx_nms_carve(nullptr, B, vx_next_pow2(capacity));
    return (size_t)((char *)w.stamps - (char *)nullptr);
}

// diagnostics: byte offset of the bitmask path's counter block (int[64]; [0] segments, [1] work items, [8..13] debug
// counters of a -DVX_BM_DEBUG build) inside the NMS workspace, 0 when the capacity has no bitmask path
extern "C" size_t vx_batched_nms_bm_counters_offset(int B, int capacity) {
    if (B <= 0 || capacity <= 0 || capacity > VX_MAX_CANDIDATES) return 0;
    vx_nms_ws w = vx_nms_carve(nullptr, B, vx_next_pow2(capacity));
    return w.bm_on ? (size_t)((char *)w.bm.counters - (char *)nullptr) : 0;
}
