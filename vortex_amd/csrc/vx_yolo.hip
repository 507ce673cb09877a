// K2 / K3 / fused K2+K3 — YOLOv3 head decode and decoder.  HBM-bound streaming kernels.
//
//   vx_k_yolo_head_decode  : f32[B,3(5+C),G,G] -> rows of f32[B,R,5+C]   (LDS transpose, both sides coalesced)
//   vx_k_yolo_decode_rows  : f32[B,R,5+C] -> surviving-row records        (flat 16-B loads into LDS)
//   vx_k_yolo_fused        : raw heads -> surviving-row records           (channel-major streaming, no LDS)
//   vx_k_cand_finish       : per image, order-preserving compaction of the per-tile records
//
// Reference: networks/modules/heads/detection/yolov3.py:263-304 (YoloV3Layer eval),
// networks/models/detection/yolov3.py:72-76 (concat), networks/modules/postprocess/yolov3.py:29-68.
#include "vx_common.h"
#include "vx_profile.h"



// ------------------------------------------------------------------------------------------------
// K2: one workgroup = 64 consecutive cells of one (image, anchor) plane, all 5+C channels.
// Load: channel-major, 64 consecutive cells per channel (coalesced).  The decoded values go to LDS
// as the FLAT image of the output rows (cell-major, 5+C floats per row), shifted so that LDS and
// global addresses agree mod 16 B; the store phase is then a straight 16-B copy of one contiguous
// 64*(5+C)*4-byte region.  LDS writes have stride 5+C dwords (odd for C=80 -> conflict-free).
struct vx_head_params {
    int B, G, cells, NP, tiles_per_plane;
    float stride;
    float aw[3], ah[3];
    long long out_batch_stride, row_offset;
};

// A/B (round 2, MI355X, bench.py --unfused, average of the three launches; baseline 0.134 ms): 128-cell tiles (512-B runs
// per channel plane, 43 KiB of LDS) 0.181 ms; two 64-cell tiles per workgroup with the loads of the second in flight
// behind the copy-out of the first 0.143 ms.
template <bool VEC>
__device__ __forceinline__ void vx_head_tile(const float *__restrict__ in, float *__restrict__ out, const vx_head_params &p,
                                             int blk, float *__restrict__ tile) {
    int t = blk % p.tiles_per_plane;
    int pa = blk / p.tiles_per_plane;          // b*3 + a
    int a = pa % 3, b = pa / 3;
    int cell0 = t * 64;
    int ncell = min(64, p.cells - cell0);
    const int NP = p.NP;
    long long o0 = (long long)b * p.out_batch_stride + (p.row_offset + (long long)a * p.cells + cell0) * NP;
    int shift = (int)(o0 & 3);
    const float *plane = in + (long long)pa * NP * p.cells + cell0;
    const float aw = p.aw[a], ah = p.ah[a], stride = p.stride;

    if (VEC) {
        // cells % 4 == 0: quads of 4 consecutive cells of one channel (16-B loads)
        int nq = NP * 16;
        for (int q = threadIdx.x; q < nq; q += 256) {
            int ch = q >> 4, c4 = (q & 15) * 4;
            if (c4 < ncell) {
                vx_v4 v = __builtin_nontemporal_load(reinterpret_cast<const vx_v4 *>(plane + (long long)ch * p.cells + c4));
                float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int gc = cell0 + c4 + j;
                    float x = r[j], y;
                    if (ch == 0) y = vx_yolo_xy(x, (float)(gc % p.G), stride);
                    else if (ch == 1) y = vx_yolo_xy(x, (float)(gc / p.G), stride);
                    else if (ch == 2) y = vx_yolo_wh(x, aw, stride);
                    else if (ch == 3) y = vx_yolo_wh(x, ah, stride);
                    else y = vx_sigmoid(x);
                    tile[shift + (c4 + j) * NP + ch] = y;
                }
            }
        }
    } else {
        int ne = NP * 64;
        for (int e = threadIdx.x; e < ne; e += 256) {
            int ch = e >> 6, c = e & 63;
            if (c < ncell) {
                int gc = cell0 + c;
                float x = plane[(long long)ch * p.cells + c], y;
                if (ch == 0) y = vx_yolo_xy(x, (float)(gc % p.G), stride);
                else if (ch == 1) y = vx_yolo_xy(x, (float)(gc / p.G), stride);
                else if (ch == 2) y = vx_yolo_wh(x, aw, stride);
                else if (ch == 3) y = vx_yolo_wh(x, ah, stride);
                else y = vx_sigmoid(x);
                tile[shift + c * NP + ch] = y;
            }
        }
    }
    __syncthreads();
    // flat copy LDS -> global: global float index o0 - shift + i  <->  tile[i]
    int n = ncell * NP;
    int nchunk = (shift + n + 3) >> 2;
    float *gbase = out + (o0 - shift);
    for (int k = threadIdx.x; k < nchunk; k += 256) {
        int i = k * 4;
        if (i >= shift && i + 4 <= shift + n) {
            __builtin_nontemporal_store(*reinterpret_cast<const vx_v4 *>(tile + i), reinterpret_cast<vx_v4 *>(gbase + i));
        } else {
            for (int j = 0; j < 4; ++j)
                if (i + j >= shift && i + j < shift + n) gbase[i + j] = tile[i + j];
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void vx_k_yolo_head_decode(const float *__restrict__ in,
                                                             float *__restrict__ out, vx_head_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int blk = vx_xcd_tile(blockIdx.x, p.B * 3 * p.tiles_per_plane);
    if (blk >= 0) vx_head_tile<VEC>(in, out, p, blk, tile);
}

// The three scales of a YOLOv3 head in ONE launch (YoloV3Layer x3 + the concat of models/detection/yolov3.py:72-76):
// the grid is the tiles of the largest scale, then the next, then the smallest, so the two small scales -- 4.2 and
// 3.7 TB/s as launches of their own (ramp-up and tail of a 0.08 / 0.02 ms kernel) -- fill the tail of the large one.
struct vx_head3_params {
    vx_head_params s[3];
    const float *in[3];
    int blk_end[3];
    int vec[3];
};

__global__ __launch_bounds__(256) void vx_k_yolo_head_decode3(float *__restrict__ out, vx_head3_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int blk = vx_xcd_tile(blockIdx.x, p.blk_end[2]);
    if (blk < 0) return;
    const int k = blk < p.blk_end[0] ? 0 : (blk < p.blk_end[1] ? 1 : 2);
    const int local = blk - (k == 0 ? 0 : p.blk_end[k - 1]);
    if (p.vec[k]) vx_head_tile<true>(p.in[k], out, p.s[k], local, tile);
    else vx_head_tile<false>(p.in[k], out, p.s[k], local, tile);
}

static int vx_head_setup(vx_head_params &p, int B, int G, int C, const float *anchor_wh, float stride,
                         int64_t out_batch_stride, int64_t row_offset, long long *nblk) {
    if (B <= 0 || G <= 0 || C < 0 || out_batch_stride <= 0 || row_offset < 0) return VX_ERR_INVALID_ARG;
    int NP = 5 + C;
    if (NP > 255) return VX_ERR_UNSUPPORTED;
    long long cells = (long long)G * G;
    if ((row_offset + 3 * cells) * NP > out_batch_stride) return VX_ERR_INVALID_ARG;
    p.B = B; p.G = G; p.cells = (int)cells; p.NP = NP;
    p.tiles_per_plane = (int)((cells + 63) / 64);
    p.stride = stride;
    for (int a = 0; a < 3; ++a) { p.aw[a] = anchor_wh[2 * a]; p.ah[a] = anchor_wh[2 * a + 1]; }
    p.out_batch_stride = out_batch_stride; p.row_offset = row_offset;
    *nblk = (long long)B * 3 * p.tiles_per_plane;
    if (*nblk > 0x7fffffffLL) return VX_ERR_UNSUPPORTED;
    return VX_OK;
}

static int vx_yolo_rows(int S);

extern "C" int vx_yolo_head_decode3(const float *d_in0, const float *d_in1, const float *d_in2, float *d_out, int B,
                                    int img_size, int C, const float anchors_px[18], void *stream) {
    if (!d_in0 || !d_in1 || !d_in2 || !d_out || !anchors_px) return VX_ERR_INVALID_ARG;
    if (img_size < 32) return VX_ERR_INVALID_ARG;
    const int mask[3][3] = {{6, 7, 8}, {3, 4, 5}, {0, 1, 2}};      // heads/detection/yolov3.py:319-343
    if ((uintptr_t)d_out & 15) return VX_ERR_ALIGNMENT;
    const float *in[3] = {d_in0, d_in1, d_in2};                     // coarse -> fine, as the reference concatenates them
    const int R = vx_yolo_rows(img_size);
    vx_head3_params p;
    long long end = 0, row = 0;
    long long rows0[3];
    for (int k = 0; k < 3; ++k) { rows0[k] = row; int G = img_size / (32 >> k); row += 3LL * G * G; }
    for (int j = 0; j < 3; ++j) {                                   // grid order: finest (largest) scale first
        const int k = 2 - j;
        if ((uintptr_t)in[k] & 3) return VX_ERR_ALIGNMENT;
        const int G = img_size / (32 >> k);
        const float stride = (float)((double)img_size / (double)G);    // heads/detection/yolov3.py:236
        float awh[6];
        for (int a = 0; a < 3; ++a) {                                  // anchors / stride in fp32 (:244)
            awh[2 * a] = anchors_px[2 * mask[k][a]] / stride;
            awh[2 * a + 1] = anchors_px[2 * mask[k][a] + 1] / stride;
        }
        long long nblk = 0;
        int rc = vx_head_setup(p.s[j], B, G, C, awh, stride, (int64_t)R * (5 + C), rows0[k], &nblk);
        if (rc != VX_OK) return rc;
        end += nblk;
        if (end > 0x7fffffffLL) return VX_ERR_UNSUPPORTED;
        p.in[j] = in[k];
        p.blk_end[j] = (int)end;
        p.vec[j] = ((long long)G * G % 4 == 0) && (((uintptr_t)in[k] & 15) == 0);
    }
    size_t lds = (size_t)(64 * (5 + C) + 4) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_HEAD_DECODE, st);
    hipLaunchKernelGGL(vx_k_yolo_head_decode3, dim3((unsigned)(8 * ((end + 7) / 8))), dim3(256), lds, st, d_out, p);
    vx_prof_end(VX_PROF_HEAD_DECODE, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

extern "C" int vx_yolo_head_decode(const float *d_in, float *d_out, int B, int G, int C,
                                   const float anchor_wh[6], float stride, int64_t out_batch_stride,
                                   int64_t row_offset, void *stream) {
    if (!d_in || !d_out || !anchor_wh) return VX_ERR_INVALID_ARG;
    if (B <= 0 || G <= 0 || C < 0 || out_batch_stride <= 0 || row_offset < 0) return VX_ERR_INVALID_ARG;
    int NP = 5 + C;
    if (NP > 255) return VX_ERR_UNSUPPORTED;
    if (((uintptr_t)d_out & 15) || ((uintptr_t)d_in & 3)) return VX_ERR_ALIGNMENT;
    long long cells = (long long)G * G;
    if ((row_offset + 3 * cells) * NP > out_batch_stride) return VX_ERR_INVALID_ARG;
    vx_head_params p;
    p.B = B; p.G = G; p.cells = (int)cells; p.NP = NP;
    p.tiles_per_plane = (int)((cells + 63) / 64);
    p.stride = stride;
    for (int a = 0; a < 3; ++a) { p.aw[a] = anchor_wh[2 * a]; p.ah[a] = anchor_wh[2 * a + 1]; }
    p.out_batch_stride = out_batch_stride; p.row_offset = row_offset;
    long long nblk = (long long)B * 3 * p.tiles_per_plane;
    if (nblk > 0x7fffffffLL) return VX_ERR_UNSUPPORTED;
    size_t lds = (size_t)(64 * NP + 4) * sizeof(float);
    bool vec = (cells % 4 == 0) && (((uintptr_t)d_in & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_HEAD_DECODE, st);
    const dim3 grid((unsigned)(8 * ((nblk + 7) / 8)));              // vx_xcd_tile
    if (vec) hipLaunchKernelGGL(vx_k_yolo_head_decode<true>, grid, dim3(256), lds, st, d_in, d_out, p);
    else     hipLaunchKernelGGL(vx_k_yolo_head_decode<false>, grid, dim3(256), lds, st, d_in, d_out, p);
    vx_prof_end(VX_PROF_HEAD_DECODE, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// Staging layout shared by the three decoders.  Rows of an image are cut into tiles in row order;
// tile t owns rows [row0(t), row0(t)+n).  Its survivors are written, in row order, at
// stage[b][row0(t) + 0 .. count-1] and tile_counts[b][t] = count.  vx_k_cand_finish then scans the
// counts per image and packs the records densely (order-preserving), producing the four tensors
// the reference's decoder returns.

// K3: tile = 64 rows.  The tile's 64*(5+C) floats are one contiguous region: 16-B loads into LDS
// (same shift trick as K2), NaN test on the way; 4 lanes per row scan the classes (first max wins);
// wave 0 thresholds, ballots and writes the records.
struct vx_rows_params {
    int B, R, C, NP, tiles_per_img;
    float img_size, thr;
    int use_thr;
};

// Software-pipelined: a workgroup walks TPW consecutive tiles; the 16-B loads of tile j+1 are issued into
// registers before tile j is reduced out of LDS, so HBM latency overlaps the class scan instead of preceding it.
// A/B on MI355X, B=256, C=80 (tools/ab_unfused.py): TPW 1 / 2 / 4 / 8 = 4.77 / 5.55 / 4.30 / 3.94 TB/s.
// Round 3: a persistent grid (1792 workgroups, workgroup w walks tiles w, w + grid, ... with the same prefetch) took 0.279 ms
// against 0.166 -- the tiles in flight at any moment are then scattered over the whole tensor instead of one moving front.
#define VX_K3_U 6
template <int TPW>
__global__ __launch_bounds__(256) void vx_k_yolo_decode_rows(const float *__restrict__ pred,
                                                               vx_rec *__restrict__ stage,
                                                               int *__restrict__ tile_counts,
                                                               int *__restrict__ status, vx_rows_params p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ float s_conf[64];
    __shared__ int s_cls[64];
    const int groups_per_img = (p.tiles_per_img + TPW - 1) / TPW;
    // (A/B round 3: the XCD-contiguous workgroup order of the head kernels changes nothing here, 0.1656 vs 0.1655 ms --
    // a tile is one contiguous 21 760-byte region, neighbours share at most its first and last line)
    const int b = blockIdx.x / groups_per_img, t_first = (blockIdx.x % groups_per_img) * TPW;
    const int NP = p.NP;
    bool nan = false;
    vx_v4 v[VX_K3_U];
    bool full[VX_K3_U];
    int shift = 0, n = 0, nchunk = 0;
    const float *gbase = pred;

    auto geometry = [&](int t) {
        int row0 = t * 64;
        int nrow = min(64, p.R - row0);
        long long o0 = ((long long)b * p.R + row0) * NP;
        shift = (int)(o0 & 3);
        n = nrow * NP;
        nchunk = (shift + n + 3) >> 2;
        gbase = pred + (o0 - shift);
    };
    auto issue = [&]() {
#pragma unroll
        for (int u = 0; u < VX_K3_U; ++u) {
            int k = threadIdx.x + u * 256, i = k * 4;
            full[u] = k < nchunk && i >= shift && i + 4 <= shift + n;
            if (full[u]) v[u] = __builtin_nontemporal_load(reinterpret_cast<const vx_v4 *>(gbase + i));
        }
    };

    geometry(t_first);
    issue();
    for (int j = 0; j < TPW; ++j) {
        const int t = t_first + j;
        if (t >= p.tiles_per_img) break;
        const int row0 = t * 64, nrow = min(64, p.R - row0);
        const int cshift = shift;
        // registers -> LDS (edge chunks and rows beyond VX_K3_U*1024 floats: scalar path)
#pragma unroll
        for (int u = 0; u < VX_K3_U; ++u) {
            int k = threadIdx.x + u * 256, i = k * 4;
            if (full[u]) {
                nan |= (v[u].x != v[u].x) | (v[u].y != v[u].y) | (v[u].z != v[u].z) | (v[u].w != v[u].w);
                *reinterpret_cast<vx_v4 *>(tile + i) = v[u];
            } else if (k < nchunk) {
                for (int q = 0; q < 4; ++q)
                    if (i + q >= shift && i + q < shift + n) { float x = gbase[i + q]; nan |= (x != x); tile[i + q] = x; }
            }
        }
        for (int k = threadIdx.x + VX_K3_U * 256; k < nchunk; k += 256) {
            int i = k * 4;
            for (int q = 0; q < 4; ++q)
                if (i + q >= shift && i + q < shift + n) { float x = gbase[i + q]; nan |= (x != x); tile[i + q] = x; }
        }
        __syncthreads();
        if (j + 1 < TPW && t + 1 < p.tiles_per_img) { geometry(t + 1); issue(); }     // next tile in flight
        {
            int r = threadIdx.x >> 2, q = threadIdx.x & 3;
            float best = -INFINITY;
            int bi = 0x7fffffff;
            if (r < nrow) {
                const float *row = tile + cshift + r * NP + 5;
                for (int k = q; k < p.C; k += 4) {
                    float x = row[k];
                    if (x > best || bi == 0x7fffffff) { best = x; bi = k; }
                }
            }
#pragma unroll
            for (int m = 1; m <= 2; m <<= 1) {
                float ob = __shfl_xor(best, m);
                int oi = __shfl_xor(bi, m);
                if (ob > best || (ob == best && oi < bi) || (bi == 0x7fffffff && oi != 0x7fffffff)) { best = ob; bi = oi; }
            }
            if (q == 0 && r < nrow) { s_conf[r] = best; s_cls[r] = bi; }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            int r = threadIdx.x;
            bool pass = false;
            float obj = 0.f;
            const float *row = tile + cshift + r * NP;
            if (r < nrow) {
                obj = row[4];
                pass = p.use_thr ? (obj > p.thr) : true;
            }
            u64 m = __ballot(pass);
            if (pass) {
                int pos = vx_mbcnt(m);
                stage[(long long)b * p.R + row0 + pos] =
                    vx_yolo_record(row[0], row[1], row[2], row[3], obj, s_conf[r], s_cls[r], row0 + r, p.img_size);
            }
            if (r == 0) tile_counts[(long long)b * p.tiles_per_img + t] = __popcll(m);
        }
        __syncthreads();                                   // LDS is rewritten by the next tile
    }
    if (__ballot(nan) != 0 && vx_lane() == 0) atomicOr(status, VX_STATUS_NAN);
}

// ------------------------------------------------------------------------------------------------
// Fused K2+K3.  Work item = one wave = one tile of consecutive cells of one (image, scale, anchor)
// plane.  Lane = 4 consecutive cells (16-B loads, 1 KiB per wave-instruction) when G*G % 4 == 0,
// else 1 cell.  The wave walks the 5+C channel planes once, keeping per cell the raw box/objectness
// logits and the running (max sigmoid, first index) over classes — the [B,R,5+C] tensor is never
// written.  Survivors are ballot-compacted inside the tile (order-preserving).
struct vx_scale_desc {
    const float *ptr;
    int G, cells, tile_cells, tiles_per_plane;
    int tile_base;      // first tile index of this scale within an image
    int row_base;       // first output row of this scale
    float stride;
    float aw[3], ah[3];
};
struct vx_fused_params {
    vx_scale_desc s[3];
    int B, C, NP, R, tiles_per_img;
    float img_size, thr;
    int use_thr;
    int sparse_max;     // threshold-first: tiles with more survivors than this are streamed
    // staged mode (vx_internal_yolo_fused_stage; consumed by vx_image_finish): a NaN is reported in bit 30 of the tile's
    // count instead of an atomic on the status word, and the kernel clears zero_words ints at zero_ptr (vx_stage_info)
    int nan_in_counts, zero_words;
    int *zero_ptr;
};

__device__ __forceinline__ void vx_fused_zero(const vx_fused_params &p) {
    if (p.zero_words > 0)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_words; i += gridDim.x * blockDim.x) p.zero_ptr[i] = 0;
}

__device__ __forceinline__ int vx_fused_tile_row0(const vx_fused_params &p, int t, int &si, int &a, int &cell0) {
    si = (t >= p.s[2].tile_base) ? 2 : (t >= p.s[1].tile_base ? 1 : 0);
    const vx_scale_desc &s = p.s[si];
    int tt = t - s.tile_base;
    a = tt / s.tiles_per_plane;
    cell0 = (tt - a * s.tiles_per_plane) * s.tile_cells;
    return s.row_base + a * s.cells + cell0;
}

template <int V>
struct vx_cells { float v[V]; };

// 16-B load that only needs dword alignment (gfx950 global loads allow it): planes of G*G floats start at any
// multiple of 4 bytes.  nv < 4: the lane owns the last 1..3 cells of a plane -> scalar loads, never past it.
typedef float vx_v4u __attribute__((ext_vector_type(4), aligned(4)));

template <int V, bool NT>
__device__ __forceinline__ vx_cells<V> vx_load_cells(const float *p, int nv = V) {
    vx_cells<V> c;
    if constexpr (V == 4) {
        if (nv == 4) {
            vx_v4u q;
            if constexpr (NT) q = __builtin_nontemporal_load(reinterpret_cast<const vx_v4u *>(p));
            else q = *reinterpret_cast<const vx_v4u *>(p);
            c.v[0] = q.x; c.v[1] = q.y; c.v[2] = q.z; c.v[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) c.v[j] = j < nv ? p[j] : 0.f;
        }
    } else {
        if constexpr (NT) c.v[0] = __builtin_nontemporal_load(p);
        else c.v[0] = *p;
    }
    return c;
}

// VX_FUSED_UN: channel planes per register batch (UN x 1 KiB per wave in flight per batch)
template <int V, int VX_FUSED_UN, bool NT, bool PARTIAL = false>
__device__ __forceinline__ void vx_fused_tile(const vx_fused_params &p, const vx_scale_desc &s, int b, int a,
                                              int cell0, int row0, int tile, vx_rec *stage, int *tile_counts,
                                              int *status) {
    const int lane = vx_lane();
    const int c0 = cell0 + lane * V;                 // first cell of this lane
    const int nv = min(max(s.cells - c0, 0), V);     // valid cells of this lane (the last lane of a plane may own < V)
    const bool live = nv > 0;
    const int nvl = PARTIAL ? nv : V;                // !PARTIAL: every live lane owns V cells -> branch-free vector loads
    const long long cells = s.cells;
    const float *plane = s.ptr + ((long long)(b * 3 + a) * p.NP) * cells + c0;
    float best[V];
    int bi[V];
    bool nan = false;
#pragma unroll
    for (int j = 0; j < V; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    vx_cells<V> tb[5];
#pragma unroll
    for (int ch = 0; ch < 5; ++ch)
#pragma unroll
        for (int j = 0; j < V; ++j) tb[ch].v[j] = 0.f;
    if (live) {
        const int C = p.C;
        const float *cp = plane + 5 * cells;
        // Two register batches in ping-pong: batch B is requested before batch A is consumed, so every
        // wave always has 8..16 KiB of loads in flight.
        vx_cells<V> A[VX_FUSED_UN], Bt[VX_FUSED_UN];
#pragma unroll
        for (int u = 0; u < VX_FUSED_UN; ++u)
            if (u < C) A[u] = vx_load_cells<V, NT>(cp + (long long)u * cells, nvl);
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) tb[ch] = vx_load_cells<V, NT>(plane + (long long)ch * cells, nvl);
        for (int k0 = 0; k0 < C; k0 += 2 * VX_FUSED_UN) {
#pragma unroll
            for (int u = 0; u < VX_FUSED_UN; ++u)
                if (k0 + VX_FUSED_UN + u < C) Bt[u] = vx_load_cells<V, NT>(cp + (long long)(k0 + VX_FUSED_UN + u) * cells, nvl);
#pragma unroll
            for (int u = 0; u < VX_FUSED_UN; ++u) {
                const int k = k0 + u;
                if (k < C) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        float v = A[u].v[j];
                        nan |= (v != v);
                        float sg = vx_sigmoid(v);
                        if (sg > best[j] || k == 0) { best[j] = sg; bi[j] = k; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < VX_FUSED_UN; ++u)
                if (k0 + 2 * VX_FUSED_UN + u < C) A[u] = vx_load_cells<V, NT>(cp + (long long)(k0 + 2 * VX_FUSED_UN + u) * cells, nvl);
#pragma unroll
            for (int u = 0; u < VX_FUSED_UN; ++u) {
                const int k = k0 + VX_FUSED_UN + u;
                if (k < C) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        float v = Bt[u].v[j];
                        nan |= (v != v);
                        float sg = vx_sigmoid(v);
                        if (sg > best[j]) { best[j] = sg; bi[j] = k; }
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 5; ++ch)
#pragma unroll
            for (int j = 0; j < V; ++j) nan |= (tb[ch].v[j] != tb[ch].v[j]);
    }
    const bool wnan = __ballot(nan) != 0;
    if (wnan && !p.nan_in_counts && lane == 0) atomicOr(status, VX_STATUS_NAN);
    float obj[V];
    bool pass[V];
    u64 m[V];
    int below = 0, total = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        obj[j] = live ? vx_sigmoid(tb[4].v[j]) : 0.f;
        pass[j] = j < nv && (p.use_thr ? (obj[j] > p.thr) : true);
        m[j] = __ballot(pass[j]);
        below += vx_mbcnt(m[j]);
        total += __popcll(m[j]);
    }
    int pos = below;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if (pass[j]) {
            int gc = c0 + j;
            float cx = vx_yolo_xy(tb[0].v[j], (float)(gc % s.G), s.stride);
            float cy = vx_yolo_xy(tb[1].v[j], (float)(gc / s.G), s.stride);
            float w = vx_yolo_wh(tb[2].v[j], s.aw[a], s.stride);
            float h = vx_yolo_wh(tb[3].v[j], s.ah[a], s.stride);
            int row = row0 + lane * V + j;
            stage[(long long)b * p.R + row0 + pos] = vx_yolo_record(cx, cy, w, h, obj[j], best[j], bi[j], row, p.img_size);
            ++pos;
        }
    }
    if (lane == 0) tile_counts[(long long)b * p.tiles_per_img + tile] = total | ((wnan && p.nan_in_counts) ? VX_TILE_NAN : 0);
}

// Small batches (fewer tiles than the GPU has wave slots): one wave per tile leaves the kernel bound by a single
// wave's 85-plane sigmoid chain (19.8 us at B=1).  Here a workgroup of 4 waves shares ONE tile: wave w scans the
// classes [w*C/4, (w+1)*C/4), the partial (max sigmoid, first index) pairs meet in LDS and wave 0 merges them in class
// order (strict '>' keeps the first maximum) and finishes the tile exactly as vx_fused_tile does.  Same bits out.
#define VX_SPLIT_WAVES 4
#define VX_SPLIT_MAX_TILES 3584       // up to 14 tiles per CU (B <= 79 at 416): A/B split vs one wave per tile, us at B = 1/16/32/64/96/128: 10/16/26/41/61/82 vs 22/36/41/47/59/75
__global__ __launch_bounds__(256) void vx_k_yolo_fused_split(vx_fused_params p, vx_rec *__restrict__ stage,
                                                             int *__restrict__ tile_counts, int *__restrict__ status) {
    constexpr int V = 4;
    __shared__ float s_best[VX_SPLIT_WAVES][VX_WAVE * V];
    __shared__ int s_bi[VX_SPLIT_WAVES][VX_WAVE * V];
    __shared__ int s_nan[VX_SPLIT_WAVES];
    vx_fused_zero(p);
    const int b = blockIdx.x / p.tiles_per_img, t = blockIdx.x - b * p.tiles_per_img;
    const int lane = vx_lane(), w = threadIdx.x >> 6;
    int si, a, cell0;
    const int row0 = vx_fused_tile_row0(p, t, si, a, cell0);
    const vx_scale_desc &s = p.s[si];
    const int c0 = cell0 + lane * V;
    const int nv = min(max(s.cells - c0, 0), V);
    const bool live = nv > 0;
    const long long cells = s.cells;
    const float *plane = s.ptr + ((long long)(b * 3 + a) * p.NP) * cells + c0;
    const int kb = (int)((long long)p.C * w / VX_SPLIT_WAVES), ke = (int)((long long)p.C * (w + 1) / VX_SPLIT_WAVES);
    float best[V];
    int bi[V];
    bool nan = false;
#pragma unroll
    for (int j = 0; j < V; ++j) { best[j] = -INFINITY; bi[j] = kb; }
    vx_cells<V> tb[5];
#pragma unroll
    for (int ch = 0; ch < 5; ++ch)
#pragma unroll
        for (int j = 0; j < V; ++j) tb[ch].v[j] = 0.f;
    if (live) {
        if (w == 0) {
#pragma unroll
            for (int ch = 0; ch < 5; ++ch) {
                tb[ch] = vx_load_cells<V, false>(plane + (long long)ch * cells, nv);
#pragma unroll
                for (int j = 0; j < V; ++j) nan |= (tb[ch].v[j] != tb[ch].v[j]);
            }
        }
        const float *cp = plane + 5 * cells;
        constexpr int U = 4;
        for (int k0 = kb; k0 < ke; k0 += U) {
            vx_cells<V> A[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + u < ke) A[u] = vx_load_cells<V, false>(cp + (long long)(k0 + u) * cells, nv);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u;
                if (k < ke) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        float v = A[u].v[j];
                        nan |= (v != v);
                        float sg = vx_sigmoid(v);
                        if (sg > best[j] || k == kb) { best[j] = sg; bi[j] = k; }
                    }
                }
            }
        }
    }
    const bool wnan = __ballot(nan) != 0;
    if (wnan && !p.nan_in_counts && lane == 0) atomicOr(status, VX_STATUS_NAN);
    if (lane == 0) s_nan[w] = wnan ? 1 : 0;
#pragma unroll
    for (int j = 0; j < V; ++j) { s_best[w][lane * V + j] = best[j]; s_bi[w][lane * V + j] = bi[j]; }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int q = 1; q < VX_SPLIT_WAVES; ++q) {
        if ((int)((long long)p.C * q / VX_SPLIT_WAVES) == (int)((long long)p.C * (q + 1) / VX_SPLIT_WAVES)) continue;   // empty class range
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float ob = s_best[q][lane * V + j];
            if (ob > best[j]) { best[j] = ob; bi[j] = s_bi[q][lane * V + j]; }
        }
    }
    float obj[V];
    bool pass[V];
    u64 m[V];
    int below = 0, total = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        obj[j] = live ? vx_sigmoid(tb[4].v[j]) : 0.f;
        pass[j] = j < nv && (p.use_thr ? (obj[j] > p.thr) : true);
        m[j] = __ballot(pass[j]);
        below += vx_mbcnt(m[j]);
        total += __popcll(m[j]);
    }
    int pos = below;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if (pass[j]) {
            int gc = c0 + j;
            float cx = vx_yolo_xy(tb[0].v[j], (float)(gc % s.G), s.stride);
            float cy = vx_yolo_xy(tb[1].v[j], (float)(gc / s.G), s.stride);
            float bw = vx_yolo_wh(tb[2].v[j], s.aw[a], s.stride);
            float bh = vx_yolo_wh(tb[3].v[j], s.ah[a], s.stride);
            int row = row0 + lane * V + j;
            stage[(long long)b * p.R + row0 + pos] = vx_yolo_record(cx, cy, bw, bh, obj[j], best[j], bi[j], row, p.img_size);
            ++pos;
        }
    }
    const bool tnan = p.nan_in_counts && (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) != 0;
    if (lane == 0) tile_counts[(long long)b * p.tiles_per_img + t] = total | (tnan ? VX_TILE_NAN : 0);
}

// "Threshold first" tile: the objectness plane is read first (coalesced, 1/(5+C) of the bytes); a tile with
// no survivor is finished.  For each survivor the WAVE gathers the cell's 5+C channel values in one round trip
// (lane = channel, stride = one plane), reduces (max sigmoid, first index) by butterfly and the owner lane
// writes the record -- same arithmetic and same bits as the streaming tile.  HBM traffic: 1 plane + (5+C)
// 128-B lines per survivor; tiles with more than p.sparse_max survivors return false and are streamed.
// The NaN test covers every objectness value but the box/class values of surviving rows only (opt-in mode).
template <int V>
__device__ __forceinline__ bool vx_fused_tile_sparse(const vx_fused_params &p, const vx_scale_desc &s, int b, int a,
                                                     int cell0, int row0, int tile, vx_rec *stage, int *tile_counts,
                                                     int *status) {
    const int lane = vx_lane();
    const int c0 = cell0 + lane * V;
    const int nv = min(max(s.cells - c0, 0), V);
    const bool live = nv > 0;
    const long long cells = s.cells;
    const float *base = s.ptr + ((long long)(b * 3 + a) * p.NP) * cells;      // channel 0, cell 0 of this plane set
    vx_cells<V> to;
#pragma unroll
    for (int j = 0; j < V; ++j) to.v[j] = 0.f;
    if (live) to = vx_load_cells<V, false>(base + 4 * cells + c0, nv);
    bool nan = false;
    float obj[V];
    bool pass[V];
    u64 m[V];
    int below = 0, total = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        nan |= (to.v[j] != to.v[j]);
        obj[j] = live ? vx_sigmoid(to.v[j]) : 0.f;
        pass[j] = j < nv && (obj[j] > p.thr);
        m[j] = __ballot(pass[j]);
        below += vx_mbcnt(m[j]);
        total += __popcll(m[j]);
    }
    if (total > p.sparse_max) return false;                                   // wave-uniform
    int mypos = below;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        u64 mj = m[j];
        while (mj) {                                                          // wave-uniform loop over survivors
            const int L = __ffsll((long long)mj) - 1;
            mj &= mj - 1;
            const float *q = base + (cell0 + L * V + j);
            float best = -INFINITY, v0 = 0.f;
            int bi = 0x7fffffff;
            for (int ch = lane; ch < p.NP; ch += VX_WAVE) {
                float v = q[(long long)ch * cells];
                nan |= (v != v);
                if (ch < VX_WAVE) v0 = v;
                if (ch >= 5) {
                    float sg = vx_sigmoid(v);
                    if (sg > best) { best = sg; bi = ch - 5; }                // ascending ch within a lane: first max
                }
            }
#pragma unroll
            for (int k = 32; k >= 1; k >>= 1) {
                float ob = __shfl_xor(best, k, VX_WAVE);
                int oi = __shfl_xor(bi, k, VX_WAVE);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            float tx = __shfl(v0, 0, VX_WAVE), ty = __shfl(v0, 1, VX_WAVE);
            float tw = __shfl(v0, 2, VX_WAVE), th = __shfl(v0, 3, VX_WAVE);
            if (lane == L) {
                int gc = c0 + j;
                float cx = vx_yolo_xy(tx, (float)(gc % s.G), s.stride);
                float cy = vx_yolo_xy(ty, (float)(gc / s.G), s.stride);
                float w = vx_yolo_wh(tw, s.aw[a], s.stride);
                float h = vx_yolo_wh(th, s.ah[a], s.stride);
                stage[(long long)b * p.R + row0 + mypos] =
                    vx_yolo_record(cx, cy, w, h, obj[j], best, bi, row0 + lane * V + j, p.img_size);
            }
        }
        if (pass[j]) ++mypos;
    }
    const bool wnan = __ballot(nan) != 0;
    if (wnan && !p.nan_in_counts && lane == 0) atomicOr(status, VX_STATUS_NAN);
    if (lane == 0) tile_counts[(long long)b * p.tiles_per_img + tile] = total | ((wnan && p.nan_in_counts) ? VX_TILE_NAN : 0);
    return true;
}

#ifndef VX_FUSED_AB_UN
#define VX_FUSED_AB_UN 2
#endif
#ifndef VX_FUSED_AB_NT
#define VX_FUSED_AB_NT true
#endif
template <int UN, bool NT, bool SPARSE>
__global__ __launch_bounds__(256) void vx_k_yolo_fused(vx_fused_params p, vx_rec *__restrict__ stage,
                                                       int *__restrict__ tile_counts, int *__restrict__ status) {
    // 32-bit index math (the host refuses B * tiles_per_img >= 2^31): no 64-bit division per wave
    // XCD-contiguous workgroup order (vx_xcd_tile): workgroups that share a 128-byte line at their plane-run boundary
    // run on one L2
    vx_fused_zero(p);
    const unsigned waves = (unsigned)p.B * (unsigned)p.tiles_per_img;
    const int wg = vx_xcd_tile((int)blockIdx.x, (int)((waves + 3u) >> 2));
    if (wg < 0) return;
    const unsigned wtile = (unsigned)wg * 4u + (threadIdx.x >> 6);
    if (wtile >= waves) return;
    int b = (int)(wtile / (unsigned)p.tiles_per_img);
    int t = (int)(wtile - (unsigned)b * (unsigned)p.tiles_per_img);
    int si, a, cell0;
    int row0 = vx_fused_tile_row0(p, t, si, a, cell0);
    const vx_scale_desc &s = p.s[si];                 // wave-uniform
    if constexpr (SPARSE) {
        if (vx_fused_tile_sparse<4>(p, s, b, a, cell0, row0, t, stage, tile_counts, status)) return;
    }
    const bool partial = (s.cells & 3) != 0 && cell0 + s.tile_cells >= s.cells;   // wave-uniform: last tile of an odd plane
    if (partial) vx_fused_tile<4, UN, NT, true>(p, s, b, a, cell0, row0, t, stage, tile_counts, status);
    else vx_fused_tile<4, UN, NT, false>(p, s, b, a, cell0, row0, t, stage, tile_counts, status);
}

// ------------------------------------------------------------------------------------------------
// Finish: one workgroup per image.  Exclusive scan of the image's tile counts (256 tiles per round),
// then every record of the round is copied to its dense position (binary search of the scanned
// offsets in LDS).  Writes the decoder's four outputs + the count.
__device__ __forceinline__ int vx_block_excl_scan256(int v, int *s_wave, int &total) {
    // 256 threads = 4 waves
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[w] = x;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += s_wave[i];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(256) void vx_k_cand_finish(const vx_rec *__restrict__ stage,
                                                        const float *__restrict__ stage_lmk,
                                                        const int *__restrict__ tile_counts,
                                                        vx_fused_params fp, vx_finish_params p,
                                                        float *__restrict__ boxes, float *__restrict__ scores,
                                                        long long *__restrict__ cls, float *__restrict__ det,
                                                        int *__restrict__ counts) {
    __shared__ int s_off[257];
    __shared__ int s_row0[256];
    __shared__ int s_wave[4];
    // workgroup (g, b) packs the records of tiles [g*TW, (g+1)*TW) of image b; the records of earlier tiles only
    // enter through their total, which every workgroup sums for itself (a few hundred ints)
    const int b = blockIdx.y, TW = p.tiles_per_wg;
    const int t0 = blockIdx.x * TW;
    int running;
    {
        int acc = 0;
        for (int t = threadIdx.x; t < t0; t += 256) acc += tile_counts[(long long)b * p.tiles_per_img + t];
        (void)vx_block_excl_scan256(acc, s_wave, running);
    }
    {
        int t = t0 + threadIdx.x;
        int c = 0, row0 = 0;
        if (threadIdx.x < TW && t < p.tiles_per_img) {
            c = tile_counts[(long long)b * p.tiles_per_img + t];
            if (p.uniform_tile_rows > 0) row0 = t * p.uniform_tile_rows;
            else { int si, a, cell0; row0 = vx_fused_tile_row0(fp, t, si, a, cell0); }
        }
        int total;
        int ex = vx_block_excl_scan256(c, s_wave, total);
        s_off[threadIdx.x] = ex;
        s_row0[threadIdx.x] = row0;
        if (threadIdx.x == 0) s_off[256] = total;
        __syncthreads();
        for (int j = threadIdx.x; j < total; j += 256) {
            // largest tile index lt with s_off[lt] <= j
            int lo = 0, hi = 256;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (s_off[mid] <= j) lo = mid; else hi = mid;
            }
            int i = j - s_off[lo];
            long long src = (long long)b * p.R + s_row0[lo] + i;
            long long dst = (long long)b * p.R + running + j;
            vx_rec r = stage[src];
            *reinterpret_cast<vx_f4 *>(boxes + dst * 4) = r.a;
            scores[dst] = r.b.x;
            int ci = __float_as_int(r.b.z);
            cls[dst] = (long long)ci;
            float *d = det + dst * p.D;
            if (((p.D & 1) | (int)((uintptr_t)det & 7)) == 0) {
                // rows of an even number of floats: 8-byte stores (a 24-byte row = 3 stores instead of 6)
                float2 *d2 = reinterpret_cast<float2 *>(d);
                d2[0] = make_float2(r.a.x, r.a.y); d2[1] = make_float2(r.a.z, r.a.w); d2[2] = make_float2(r.b.y, (float)ci);
                if (p.L > 0) {
                    const float2 *lm = reinterpret_cast<const float2 *>(stage_lmk + src * (2 * p.L));   // 8*L bytes per row
                    for (int q = 0; q < p.L; ++q) d2[3 + q] = lm[q];
                }
            } else {
                d[0] = r.a.x; d[1] = r.a.y; d[2] = r.a.z; d[3] = r.a.w;
                d[4] = r.b.y; d[5] = (float)ci;
                if (p.L > 0) {
                    const float *lm = stage_lmk + src * (2 * p.L);
                    for (int q = 0; q < 2 * p.L; ++q) d[6 + q] = lm[q];
                }
            }
        }
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) counts[b] = running + total;
    }
}

// 32 tiles (2048 .. 8192 rows) per workgroup: 7 workgroups per image for 50 400 priors, 2 for YOLOv3-416
static inline dim3 vx_finish_grid(vx_finish_params &f, int B) {
    f.tiles_per_wg = f.tiles_per_img < 32 ? f.tiles_per_img : 32;
    return dim3((unsigned)((f.tiles_per_img + f.tiles_per_wg - 1) / f.tiles_per_wg), (unsigned)B);
}

int vx_internal_cand_finish(const void *stage, const float *stage_lmk, const int *tile_counts, int B,
                            vx_finish_params f, float *boxes, float *scores, long long *cls, float *det,
                            int *counts, hipStream_t st) {
    vx_fused_params fp = {};
    if (B > 65535) return VX_ERR_UNSUPPORTED;
    const dim3 grid = vx_finish_grid(f, B);
    vx_prof_begin(VX_PROF_FINISH, st);
    hipLaunchKernelGGL(vx_k_cand_finish, grid, dim3(256), 0, st, (const vx_rec *)stage, stage_lmk, tile_counts, fp,
                       f, boxes, scores, cls, det, counts);
    vx_prof_end(VX_PROF_FINISH, st);
    return VX_OK;
}

// ------------------------------------------------------------------------------------------------
static inline size_t vx_align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t vx_stage_bytes(int B, int R, int tiles_per_img, int L) {
    size_t n = vx_align256((size_t)B * R * sizeof(vx_rec));
    n += vx_align256((size_t)B * tiles_per_img * sizeof(int));
    if (L > 0) n += vx_align256((size_t)B * R * 2 * L * sizeof(float));
    return n;
}

extern "C" size_t vx_yolo_decode_workspace_bytes(int B, int R) {
    if (B <= 0 || R <= 0) return 0;
    return vx_stage_bytes(B, R, (R + 63) / 64, 0);
}

extern "C" int vx_yolo_decode(const float *d_pred, int B, int R, int C, float img_size, float score_threshold,
                              int use_threshold, float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                              int32_t *d_counts, int32_t *d_status, void *d_workspace, size_t workspace_bytes,
                              void *stream) {
    if (!d_pred || !d_boxes || !d_scores || !d_cls || !d_det || !d_counts || !d_status || !d_workspace)
        return VX_ERR_INVALID_ARG;
    if (B <= 0 || R <= 0 || C <= 0) return VX_ERR_INVALID_ARG;
    int NP = 5 + C;
    if (NP > 248 || C > VX_MAX_CLASSES || R > VX_MAX_CANDIDATES) return VX_ERR_UNSUPPORTED;
    if (((uintptr_t)d_pred & 15) || ((uintptr_t)d_boxes & 15) || ((uintptr_t)d_workspace & 255) ||
        ((uintptr_t)d_det & 3) || ((uintptr_t)d_cls & 7))
        return VX_ERR_ALIGNMENT;
    int tiles = (R + 63) / 64;
    if (workspace_bytes < vx_stage_bytes(B, R, tiles, 0)) return VX_ERR_WORKSPACE;
    if ((long long)B * tiles > 0x7fffffffLL) return VX_ERR_UNSUPPORTED;
    vx_rec *stage = (vx_rec *)d_workspace;
    int *tile_counts = (int *)((char *)d_workspace + vx_align256((size_t)B * R * sizeof(vx_rec)));
    hipStream_t st = (hipStream_t)stream;
    vx_rows_params p;
    p.B = B; p.R = R; p.C = C; p.NP = NP; p.tiles_per_img = tiles;
    p.img_size = img_size; p.thr = score_threshold; p.use_thr = use_threshold ? 1 : 0;
    size_t lds = (size_t)(64 * NP + 4) * sizeof(float);
    vx_prof_begin(VX_PROF_DECODE_ROWS, st);
    hipLaunchKernelGGL(vx_k_yolo_decode_rows<2>, dim3((unsigned)(B * ((tiles + 1) / 2))), dim3(256), lds, st, d_pred, stage,
                       tile_counts, d_status, p);
    vx_prof_end(VX_PROF_DECODE_ROWS, st);
    vx_finish_params f;
    f.R = R; f.tiles_per_img = tiles; f.D = 6; f.L = 0; f.uniform_tile_rows = 64;
    int rc = vx_internal_cand_finish(stage, nullptr, tile_counts, B, f, d_boxes, d_scores, (long long *)d_cls, d_det,
                                     d_counts, st);
    if (rc != VX_OK) return rc;
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

static int vx_fused_setup(vx_fused_params &p, const float *h0, const float *h1, const float *h2, int B, int S,
                          int C, const float anchors_px[18]) {
    const float *ptrs[3] = {h0, h1, h2};
    const int div[3] = {32, 16, 8};
    const int mask[3][3] = {{6, 7, 8}, {3, 4, 5}, {0, 1, 2}};   // heads/detection/yolov3.py:319-343
    int tile_base = 0, row_base = 0;
    for (int i = 0; i < 3; ++i) {
        vx_scale_desc &s = p.s[i];
        s.ptr = ptrs[i];
        s.G = S / div[i];
        if (s.G <= 0) return VX_ERR_INVALID_ARG;
        s.cells = s.G * s.G;
        s.tile_cells = 256;     // 64 lanes x 4 cells; 16-B loads need dword alignment only (vx_v4u), a plane's last
                                // lane may own < 4 cells
        s.tiles_per_plane = (s.cells + s.tile_cells - 1) / s.tile_cells;
        s.tile_base = tile_base;
        s.row_base = row_base;
        // stride = img_size / max(grids) as an fp32 tensor (heads/detection/yolov3.py:236)
        s.stride = (float)((double)S / (double)s.G);
        for (int a = 0; a < 3; ++a) {
            // anchor_wh = anchors / stride in fp32 (heads/detection/yolov3.py:244)
            s.aw[a] = anchors_px[2 * mask[i][a]] / s.stride;
            s.ah[a] = anchors_px[2 * mask[i][a] + 1] / s.stride;
        }
        tile_base += 3 * s.tiles_per_plane;
        row_base += 3 * s.cells;
    }
    p.B = B; p.C = C; p.NP = 5 + C; p.R = row_base; p.tiles_per_img = tile_base;
    return VX_OK;
}

static int vx_fused_tiles(int S) {
    int tiles = 0;
    const int div[3] = {32, 16, 8};
    for (int i = 0; i < 3; ++i) {
        int G = S / div[i], cells = G * G;
        // sized for 64-cell tiles (>= the 256-cell tiling the kernels use): one size for every tiling
        tiles += 3 * ((cells + 63) / 64);
    }
    return tiles;
}

static int vx_yolo_rows(int S) {
    int R = 0;
    const int div[3] = {32, 16, 8};
    for (int i = 0; i < 3; ++i) { int G = S / div[i]; R += 3 * G * G; }
    return R;
}

extern "C" size_t vx_yolo_fused_workspace_bytes(int B, int S) {
    if (B <= 0 || S < 32) return 0;
    return vx_stage_bytes(B, vx_yolo_rows(S), vx_fused_tiles(S), 0);
}

// info != nullptr: the decode kernel only; the caller packs the staged records (vx_stage_info)
static int vx_fused_decode_impl(const float *d_head0, const float *d_head1, const float *d_head2, int B, int S,
                                int C, const float anchors_px[18], float score_threshold, int use_threshold,
                                float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                                int32_t *d_counts, int32_t *d_status, void *d_workspace,
                                size_t workspace_bytes, void *stream, vx_stage_info *info) {
    if (!d_head0 || !d_head1 || !d_head2 || !anchors_px || !d_status || !d_workspace) return VX_ERR_INVALID_ARG;
    if (!info && (!d_boxes || !d_scores || !d_cls || !d_det || !d_counts)) return VX_ERR_INVALID_ARG;
    if (B <= 0 || S < 32 || C <= 0) return VX_ERR_INVALID_ARG;
    if (C > VX_MAX_CLASSES || B > 65535) return VX_ERR_UNSUPPORTED;
    if (((uintptr_t)d_head0 & 3) || ((uintptr_t)d_head1 & 3) || ((uintptr_t)d_head2 & 3) || ((uintptr_t)d_workspace & 255))
        return VX_ERR_ALIGNMENT;
    if (!info && (((uintptr_t)d_boxes & 15) || ((uintptr_t)d_det & 3) || ((uintptr_t)d_cls & 7))) return VX_ERR_ALIGNMENT;
    vx_fused_params p = {};
    int rc = vx_fused_setup(p, d_head0, d_head1, d_head2, B, S, C, anchors_px);
    if (rc != VX_OK) return rc;
    if (p.R > VX_MAX_CANDIDATES) return VX_ERR_UNSUPPORTED;
    p.img_size = (float)S; p.thr = score_threshold; p.use_thr = (use_threshold & 1) ? 1 : 0;
    if (info) {                                        // staged mode: vx_image_finish reads the counts (and their NaN bit)
        p.nan_in_counts = 1;
        p.zero_ptr = info->zero_words > 0 ? info->zero_ptr : nullptr;
        p.zero_words = p.zero_ptr ? info->zero_words : 0;
    }
    const bool sparse = p.use_thr && (use_threshold & VX_DECODE_THRESHOLD_FIRST);
    // A/B on MI355X, B=256 (tools/ab_fused.py; streaming = 0.150-0.177 ms at every threshold): 8 -> 0.015 / 0.046 /
    // 0.118 / 0.158 / 0.167 / 0.176 ms at thr 0.9 / 0.7 / 0.5 / 0.3 / 0.2 / 0.06; 16 and 32 lose up to 2.3x at 0.2-0.3;
    // 3 lets a rare streamed tile (one ~38 us dependent chain) set the floor of the sparse case.
    p.sparse_max = 8;
    if (workspace_bytes < vx_stage_bytes(B, p.R, vx_fused_tiles(S), 0)) return VX_ERR_WORKSPACE;
    vx_rec *stage = (vx_rec *)d_workspace;
    int *tile_counts = (int *)((char *)d_workspace + vx_align256((size_t)B * p.R * sizeof(vx_rec)));
    long long waves = (long long)B * p.tiles_per_img;
    long long nblk = (waves + 3) / 4;
    if (waves + 4 > 0x7fffffffLL) return VX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    vx_prof_begin(VX_PROF_FUSED, st);
    // <2, true>: 2 channel planes per register batch in ping-pong (7 waves/SIMD), non-temporal loads.
    // A/B on MI355X, B=256 (tools/ab_fused.py): UN 8/4/3/2/1 with nt = 5.0/5.9/5.9/6.19/5.96 TB/s, UN 4/2 plain = 5.5/5.67;
    // one batch of 4/5/8/10/16/20 planes without ping-pong = 5.88/5.98/5.97/6.02/6.10/5.70.  Ceilings measured on the
    // With the XCD-contiguous workgroup order (round 3): UN 2/3/4 with nt = 6.40/5.92/5.92 TB/s, UN 2/4 plain = 5.80/5.41.
    // same box (tools/micro/read_bw.hip): linear 16-B nt read 7.1 TB/s, this kernel's plane-strided address pattern
    // without arithmetic 6.5 (2 planes in flight) .. 6.9 (12), linear read + sigmoid/max arithmetic 6.8.
    if (sparse)
        hipLaunchKernelGGL((vx_k_yolo_fused<VX_FUSED_AB_UN, VX_FUSED_AB_NT, true>), dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(256), 0, st, p, stage,
                           tile_counts, d_status);
    else if (waves <= VX_SPLIT_MAX_TILES && C >= 2 * VX_SPLIT_WAVES)
        hipLaunchKernelGGL(vx_k_yolo_fused_split, dim3((unsigned)waves), dim3(64 * VX_SPLIT_WAVES), 0, st, p, stage,
                           tile_counts, d_status);
    else
        hipLaunchKernelGGL((vx_k_yolo_fused<VX_FUSED_AB_UN, VX_FUSED_AB_NT, false>), dim3((unsigned)(8 * ((nblk + 7) / 8))), dim3(256), 0, st, p, stage,
                           tile_counts, d_status);
    vx_prof_end(VX_PROF_FUSED, st);
    if (info) {
        info->stage = stage; info->tile_counts = tile_counts; info->tiles = p.tiles_per_img; info->R = p.R;
        for (int i = 0; i < 3; ++i) {
            info->tile_base[i] = p.s[i].tile_base; info->tiles_per_plane[i] = p.s[i].tiles_per_plane;
            info->tile_cells[i] = p.s[i].tile_cells; info->row_base[i] = p.s[i].row_base; info->cells[i] = p.s[i].cells;
        }
        return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
    }
    vx_finish_params f;
    f.R = p.R; f.tiles_per_img = p.tiles_per_img; f.D = 6; f.L = 0; f.uniform_tile_rows = 0;
    vx_prof_begin(VX_PROF_FINISH, st);
    hipLaunchKernelGGL(vx_k_cand_finish, vx_finish_grid(f, B), dim3(256), 0, st, stage, (const float *)nullptr, tile_counts, p, f,
                       d_boxes, d_scores, (long long *)d_cls, d_det, d_counts);
    vx_prof_end(VX_PROF_FINISH, st);
    return hipGetLastError() == hipSuccess ? VX_OK : VX_ERR_LAUNCH;
}

extern "C" int vx_yolo_fused_decode(const float *d_head0, const float *d_head1, const float *d_head2, int B, int S,
                                    int C, const float anchors_px[18], float score_threshold, int use_threshold,
                                    float *d_boxes, float *d_scores, int64_t *d_cls, float *d_det,
                                    int32_t *d_counts, int32_t *d_status, void *d_workspace,
                                    size_t workspace_bytes, void *stream) {
    return vx_fused_decode_impl(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold, d_boxes,
                                d_scores, d_cls, d_det, d_counts, d_status, d_workspace, workspace_bytes, stream, nullptr);
}

int vx_internal_yolo_fused_stage(const float *d_head0, const float *d_head1, const float *d_head2, int B, int S, int C,
                                 const float anchors_px[18], float score_threshold, int use_threshold,
                                 int32_t *d_status, void *d_workspace, size_t workspace_bytes, void *stream,
                                 vx_stage_info *info) {
    if (!info) return VX_ERR_INVALID_ARG;
    return vx_fused_decode_impl(d_head0, d_head1, d_head2, B, S, C, anchors_px, score_threshold, use_threshold, nullptr,
                                nullptr, nullptr, nullptr, nullptr, d_status, d_workspace, workspace_bytes, stream, info);
}
