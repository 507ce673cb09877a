// Shared device helpers for libvortex_hip (gfx950 only: wave = 64 lanes).
// Build with -ffp-contract=off: the decoders and the IoU test must round once per
// operation, exactly like the reference's fp32 CPU arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vortex_hip.h"

#define VX_WAVE 64

typedef unsigned long long u64;

struct __attribute__((aligned(16))) vx_f4 { float x, y, z, w; };
typedef float vx_v4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8)))  vx_f2 { float x, y; };
struct __attribute__((packed, aligned(4))) vx_u3 { unsigned a, b, c; };

// staging record written by the decode kernels for every surviving row
//   a = x1,y1,x2,y2   b = score, class_conf, class(bits), row(bits)
struct __attribute__((aligned(16))) vx_rec { vx_f4 a; vx_f4 b; };

__device__ __forceinline__ int vx_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ int vx_mbcnt(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// sigmoid / exp of the YOLO head decode (tolerance-class: <= ~1e-6 relative).
// One definition shared by the standalone and the fused kernel so both give the same bits.
__device__ __forceinline__ float vx_exp(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
__device__ __forceinline__ float vx_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + vx_exp(-x));
}

// YoloV3Layer eval arithmetic for one cell (heads/detection/yolov3.py:281-302)
__device__ __forceinline__ float vx_yolo_xy(float t, float g, float stride) { return (vx_sigmoid(t) + g) * stride; }
__device__ __forceinline__ float vx_yolo_wh(float t, float a, float stride) { return (vx_exp(t) * a) * stride; }

// YoloV3Decoder arithmetic for one surviving row (postprocess/yolov3.py:63-67, :13-16)
__device__ __forceinline__ vx_rec vx_yolo_record(float cx, float cy, float w, float h, float obj,
                                                 float conf, int cls, int row, float img_size) {
    float bx = cx / img_size, by = cy / img_size, bw = w / img_size, bh = h / img_size;
    vx_rec r;
    r.a.x = bx - bw / 2.0f;
    r.a.y = by - bh / 2.0f;
    r.a.z = bx + bw / 2.0f;
    r.a.w = by + bh / 2.0f;
    r.b.x = obj * conf;
    r.b.y = conf;
    r.b.z = __int_as_float(cls);
    r.b.w = __int_as_float(row);
    return r;
}

// descending-score sortable 32-bit image of a float (ties: -0 == +0)
__device__ __forceinline__ unsigned vx_desc_key(float s) {
    if (s == 0.0f) s = 0.0f;
    unsigned u = __float_as_uint(s);
    unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

#define VX_KEY_SENTINEL (~0ull)
// key1 = [cls:12][desc score:32][index:20]
__device__ __forceinline__ u64 vx_key1(int cls, float score, int idx) {
    return ((u64)(unsigned)cls << 52) | ((u64)vx_desc_key(score) << 20) | (u64)(unsigned)idx;
}
__device__ __forceinline__ int vx_key1_idx(u64 k) { return (int)(k & 0xFFFFFu); }
__device__ __forceinline__ int vx_key1_cls(u64 k) { return (int)(k >> 52); }
// key2 = [desc score:32][index:32]  (global keep order: score desc, index asc)
__device__ __forceinline__ u64 vx_key2_from_key1(u64 k) { return (((k >> 20) & 0xFFFFFFFFull) << 32) | (k & 0xFFFFFu); }

// parameters of the per-image compaction pass shared by all decoders (vx_yolo.hip)
struct vx_finish_params {
    int R, tiles_per_img, D, L;           // D = 6 + 2L floats per detection row
    int uniform_tile_rows;                // >0: row0(t) = t*uniform_tile_rows ; 0: fused YOLO tile table
    int tiles_per_wg;                     // tiles packed by one workgroup of vx_k_cand_finish (set at launch)
};
int vx_internal_cand_finish(const void *stage, const float *stage_lmk, const int *tile_counts, int B,
                            vx_finish_params f, float *boxes, float *scores, long long *cls, float *det,
                            int *counts, hipStream_t st);

// What the fused YOLO decode leaves behind for a consumer that packs the per-tile records itself (the single-launch
// NMS kernel does, sparing the vx_k_cand_finish launch): the staging buffers and the tile -> first-row map of the
// three scales (tiles of one image: scale 0 planes 0..2, scale 1, scale 2; see vx_fused_tile_row0).
struct vx_stage_info {
    const void *stage;                    // vx_rec[B][R]
    const int *tile_counts;               // [B][tiles]
    int tiles, R;
    int tile_base[3], tiles_per_plane[3], tile_cells[3], row_base[3], cells[3];
    // set by the CALLER of vx_internal_yolo_fused_stage: `zero_words` ints at `zero_ptr` that the decode kernel clears
    // for the kernel behind it (the chain's status word, the packing hand-off words) -- one fill launch less per batch.
    // The decode kernel itself then needs no zeroed word: a tile that saw a NaN says so in bit 30 of its count.
    int *zero_ptr;
    int zero_words;
};
#define VX_TILE_NAN 0x40000000
__host__ __device__ __forceinline__ int vx_stage_tile_row0(const vx_stage_info &f, int t) {
    const int si = (t >= f.tile_base[2]) ? 2 : (t >= f.tile_base[1] ? 1 : 0);
    const int tt = t - f.tile_base[si];
    const int a = tt / f.tiles_per_plane[si];
    return f.row_base[si] + a * f.cells[si] + (tt - a * f.tiles_per_plane[si]) * f.tile_cells[si];
}
// Packing inside the single-launch NMS kernel (vx_pipeline.hip): the kept rows of the images back to back in `dst`, the
// 2B+1 meta words (counts | keep_counts | status) completed and mirrored by the workgroup that finishes last.
//   sync: int[VX_PACK_SYNC_WORDS(B)], ZERO at launch -- [2] the call chain's status word (the kernels of the chain
//         accumulate their status here), [HDR + b * STRIDE] keep count of image b, published (a line of its own),
//         [HDR + B * STRIDE + b] image b's rows are copied
// Only for B <= the device's compute units (all workgroups resident at once: waiting on a lower image cannot starve it).
#define VX_PACK_HDR 64
#define VX_PACK_STRIDE 32
#define VX_PACK_SYNC_WORDS(B) (VX_PACK_HDR + (size_t)(B) * (VX_PACK_STRIDE + 1))
struct vx_pack_info {
    float *dst;
    int *sync;
    int *meta;                            // device: counts[B] | keep_counts[B] | status
    int *mirror;                          // optional copy of the 2B+1 words (pinned host memory)
};
// vx_yolo_fused_decode without its compaction launch (vx_yolo.hip); vx_batched_nms whose single-launch kernel starts
// from the staged records (vx_nms.hip; capacity == info.R <= 16384, info.tiles <= 256, D == 6, d_counts is written)
int vx_internal_yolo_fused_stage(const float *d_head0, const float *d_head1, const float *d_head2, int B, int S, int C,
                                 const float anchors_px[18], float score_threshold, int use_threshold,
                                 int32_t *d_status, void *d_workspace, size_t workspace_bytes, void *stream,
                                 vx_stage_info *info);
int vx_internal_nms_from_stage(const vx_stage_info &info, float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                               int32_t *d_counts, int B, int capacity, int D, int n_classes, float iou_threshold,
                               int top_k, int64_t *d_keep_idx, float *d_out_det, int32_t *d_keep_counts,
                               int out_capacity, int32_t *d_status, void *d_workspace, size_t workspace_bytes,
                               void *stream, const vx_pack_info *pack);

__host__ __device__ __forceinline__ int vx_next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// per-image row count as read from a caller-supplied device array: [0, cap] whatever the array holds
__device__ __forceinline__ int vx_clamp_count(int c, int cap) { return c < 0 ? 0 : (c > cap ? cap : c); }

// Workgroups go to the 8 XCDs in turn (blockIdx % 8), each with an L2 of its own.  A tile's channel runs are 256 B at
// a 64-byte alignment (a plane of 52 x 52 cells is 10816 B), so every other run shares a 128-byte line with the next
// tile of the plane: with neighbouring tiles on different XCDs both fetch it (PMC: 1.31x the algorithmic reads).  The
// remap gives every XCD one contiguous eighth of the tiles, so neighbours run on one L2 a moment apart.
// Launch 8 * ceil(n / 8) workgroups; a workgroup that gets -1 has no tile.
__device__ __forceinline__ int vx_xcd_tile(int blk, int n) {
    const int chunk = (n + 7) >> 3, t = (blk & 7) * chunk + (blk >> 3);
    return t < min(n, ((blk & 7) + 1) * chunk) ? t : -1;
}
