// Microbenchmark of the NMS apply loop (development aid; not part of the library): one workgroup of WAVES waves,
// each wave holds NT column tiles (64 boxes each) in registers and applies KC kept boxes (LDS) to them, REP times.
// Prints shader cycles per wave-level IoU test for several formulations of the loop body, so the sweep kernels can
// be shaped by measurement:  ./sweep_loop [grid]
// build (the binary is not tracked):  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/sweep_loop.hip -o tools/micro/sweep_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
struct __attribute__((aligned(16))) f4 { float x, y, z, w; };

__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float smax(float a, float b) { float r; asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ float smin(float a, float b) { float r; asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }

struct thr_t { float thr, hi, lo; };

// exact predicate (division)
__device__ __forceinline__ bool exact(const f4 &K, float ka, const f4 &b, float a, float thr) {
    float w = vmax(vmin(K.z, b.z) - vmax(K.x, b.x), 0.f), h = vmax(vmin(K.w, b.w) - vmax(K.y, b.y), 0.f);
    float inter = w * h, uni = ka + a - inter;
    return inter / uni > thr;
}

// V0: candidate test + scalar branch to the exact routine (the current library loop)
__device__ __forceinline__ u64 test_v0(const f4 &K, float ka, const f4 &b, float a, u64 am, const thr_t &T) {
    float w = vmax(vmin(K.z, b.z) - vmax(K.x, b.x), 0.f);
    float inter = w * (vmin(K.w, b.w) - vmax(K.y, b.y));
    float uni = ka + a - inter;
    u64 slow = __ballot(!(inter < T.lo * uni)) & am;
    u64 sup = 0;
    if (slow) sup = __ballot(exact(K, ka, b, a, T.thr)) & slow;
    return sup;
}
// V1: branch-free bracket, per-lane bool accumulators (the round-1 loop)
__device__ __forceinline__ void test_v1(const f4 &K, float ka, const f4 &b, float a, bool &sup, bool &und, const thr_t &T) {
    float w = vmax(vmin(K.z, b.z) - vmax(K.x, b.x), 0.f), h = vmax(vmin(K.w, b.w) - vmax(K.y, b.y), 0.f);
    float inter = w * h, uni = ka + a - inter;
    bool hi = inter > T.hi * uni, lo = inter < T.lo * uni;
    und = und || !(hi || lo);
    sup = sup || hi;
}
// V4: candidate test only, accumulated per lane in a VGPR (no scalar work, no branch): lower bound of the loop body
__device__ __forceinline__ void test_v4(const f4 &K, float ka, const f4 &b, float a, float &acc, const thr_t &T) {
    float w = vmax(vmin(K.z, b.z) - vmax(K.x, b.x), 0.f);
    float inter = w * (vmin(K.w, b.w) - vmax(K.y, b.y));
    float uni = ka + a - inter;
    acc = vmax(acc, inter - T.lo * uni);               // > 0 <=> some kept box is a candidate
}

template <int VAR, int NT>
__global__ __launch_bounds__(1024) void k_loop(const f4 *boxes, int KC, int REP, float thrv, u64 *out_cycles, u64 *out_sink) {
    __shared__ f4 kb[64 + 1];
    __shared__ float kar[64 + 1];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < 65) {
        f4 k = boxes[(blockIdx.x * 977 + threadIdx.x * 131) & 4095];
        kb[threadIdx.x] = k;
        kar[threadIdx.x] = (k.z - k.x) * (k.w - k.y);
    }
    f4 bx[NT];
    float ar[NT];
    u64 alive[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        bx[s] = boxes[(blockIdx.x * 64 * 131 + (w * NT + s) * 64 + lane) & 4095];
        ar[s] = (bx[s].z - bx[s].x) * (bx[s].w - bx[s].y);
        alive[s] = ~0ull;
    }
    thr_t T = {thrv, thrv * (1.0f + 0x1p-19f), thrv * (1.0f - 0x1p-19f)};
    __syncthreads();
    bool sup[NT], und[NT];
    float acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) { sup[s] = false; und[s] = false; acc[s] = -1e30f; }
    const u64 c0 = clock64();
    for (int r = 0; r < REP; ++r) {
        if (VAR == 7) break;
        if (VAR == 0 || VAR == 1 || VAR == 4) {
            f4 K = kb[0];
            float ka = kar[0];
            for (int j = 0; j < KC; ++j) {
                const f4 Kn = kb[j + 1];
                const float kan = kar[j + 1];
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    if (VAR == 0) alive[s] &= ~test_v0(K, ka, bx[s], ar[s], alive[s], T);
                    if (VAR == 1) test_v1(K, ka, bx[s], ar[s], sup[s], und[s], T);
                    if (VAR == 4) test_v4(K, ka, bx[s], ar[s], acc[s], T);
                }
                K = Kn; ka = kan;
            }
        }
        if (VAR == 2) {                                 // V0, two kept boxes per trip
            for (int j = 0; j < KC; j += 2) {
                const f4 K0 = kb[j], K1 = kb[j + 1];
                const float a0 = kar[j], a1 = kar[j + 1];
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    alive[s] &= ~test_v0(K0, a0, bx[s], ar[s], alive[s], T);
                    alive[s] &= ~test_v0(K1, a1, bx[s], ar[s], alive[s], T);
                }
            }
        }
        if (VAR == 3) {                                 // candidate masks OR-ed per trip of 4 kept boxes, one branch per trip
            for (int j = 0; j < KC; j += 4) {
                f4 K[4]; float ka[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { K[q] = kb[j + q]; ka[q] = kar[j + q]; }
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    float m = -1e30f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) test_v4(K[q], ka[q], bx[s], ar[s], m, T);
                    u64 cand = __ballot(!(m < 0.0f)) & alive[s];
                    if (cand) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) alive[s] &= ~(__ballot(exact(K[q], ka[q], bx[s], ar[s], T.thr)) & cand);
                    }
                }
            }
        }
        if (VAR == 5) {                                 // V3 with 8 kept boxes per trip
            for (int j = 0; j < KC; j += 8) {
                f4 K[8]; float ka[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { K[q] = kb[j + q]; ka[q] = kar[j + q]; }
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    float m = -1e30f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) test_v4(K[q], ka[q], bx[s], ar[s], m, T);
                    u64 cand = __ballot(!(m < 0.0f)) & alive[s];
                    if (cand) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) alive[s] &= ~(__ballot(exact(K[q], ka[q], bx[s], ar[s], T.thr)) & cand);
                    }
                }
            }
        }
        if (VAR == 6) {                                 // V5 with the kept box in scalar registers (readfirstlane)
            for (int j = 0; j < KC; j += 4) {
                float kx1[4], ky1[4], kx2[4], ky2[4], ka[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f4 K = kb[j + q];
                    kx1[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, K.x)));
                    ky1[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, K.y)));
                    kx2[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, K.z)));
                    ky2[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, K.w)));
                    ka[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kar[j + q])));
                }
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    float m = -1e30f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float w_ = vmax(smin(kx2[q], bx[s].z) - smax(kx1[q], bx[s].x), 0.f);
                        float inter = w_ * (smin(ky2[q], bx[s].w) - smax(ky1[q], bx[s].y));
                        float uni = ka[q] + ar[s] - inter;
                        m = vmax(m, inter - T.lo * uni);
                    }
                    u64 cand = __ballot(!(m < 0.0f)) & alive[s];
                    if (cand) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f4 K = {kx1[q], ky1[q], kx2[q], ky2[q]};
                            alive[s] &= ~(__ballot(exact(K, ka[q], bx[s], ar[s], T.thr)) & cand);
                        }
                    }
                }
            }
        }
    }
    if (VAR == 7) {                                     // V5 with the arithmetic packed over pairs of kept boxes (v_pk_*_f32)
        typedef float v2f __attribute__((ext_vector_type(2)));
        for (int r = 0; r < REP; ++r) {
            for (int j = 0; j < KC; j += 8) {
                f4 K[8]; float ka[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { K[q] = kb[j + q]; ka[q] = kar[j + q]; }
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    float m = -1e30f;
                    const v2f av = {ar[s], ar[s]}, lov = {T.lo, T.lo};
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        v2f xx1 = {vmax(K[q].x, bx[s].x), vmax(K[q + 1].x, bx[s].x)};
                        v2f yy1 = {vmax(K[q].y, bx[s].y), vmax(K[q + 1].y, bx[s].y)};
                        v2f xx2 = {vmin(K[q].z, bx[s].z), vmin(K[q + 1].z, bx[s].z)};
                        v2f yy2 = {vmin(K[q].w, bx[s].w), vmin(K[q + 1].w, bx[s].w)};
                        v2f wv = xx2 - xx1, hv = yy2 - yy1;
                        wv.x = vmax(wv.x, 0.f); wv.y = vmax(wv.y, 0.f);
                        v2f inter = wv * hv;
                        v2f kav = {ka[q], ka[q + 1]};
                        v2f uni = kav + av - inter;
                        v2f d = inter - lov * uni;
                        m = __builtin_fmaxf(m, __builtin_fmaxf(d.x, d.y));
                    }
                    u64 cand = __ballot(!(m < 0.0f)) & alive[s];
                    if (cand) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) alive[s] &= ~(__ballot(exact(K[q], ka[q], bx[s], ar[s], T.thr)) & cand);
                    }
                }
            }
        }
    }
    const u64 c1 = clock64();
    u64 sink = 0;
#pragma unroll
    for (int s = 0; s < NT; ++s) sink += alive[s] + (sup[s] ? 1 : 0) + (und[s] ? 2 : 0) + (acc[s] > 0.f ? 4 : 0);
    if (lane == 0) {
        out_cycles[blockIdx.x * 16 + w] = c1 - c0;
        out_sink[blockIdx.x * 16 + w] = sink;
    }
}

template <int VAR, int NT>
static void run(const char *name, const f4 *d_boxes, int grid, int waves, int KC, int REP, float thr, u64 *d_cyc, u64 *d_sink) {
    hipLaunchKernelGGL((k_loop<VAR, NT>), dim3(grid), dim3(waves * 64), 0, 0, d_boxes, KC, REP, thr, d_cyc, d_sink);
    hipLaunchKernelGGL((k_loop<VAR, NT>), dim3(grid), dim3(waves * 64), 0, 0, d_boxes, KC, REP, thr, d_cyc, d_sink);
    hipDeviceSynchronize();
    std::vector<u64> h(grid * 16);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) { double c = (double)h[b * 16 + w]; sum += c; if (c > mx) mx = c; }
    double tests = (double)KC * REP * NT;
    printf("%-28s NT=%d waves=%2d  cycles/wave-test: mean %.1f  (slowest wave %.1f)  => %.2f lane-tests/clk/CU\n", name, NT, waves,
           sum / (grid * waves) / tests, mx / tests, tests * 64 * waves / mx);
}

int main(int argc, char **argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 256;
    float thr = 0.45f;
    std::vector<f4> h(4096);
    srand(1);
    for (auto &b : h) {
        float cx = rand() / (float)RAND_MAX, cy = rand() / (float)RAND_MAX;
        float w = 0.02f + 0.2f * rand() / (float)RAND_MAX, hh = 0.02f + 0.2f * rand() / (float)RAND_MAX;
        b = {cx - w / 2, cy - hh / 2, cx + w / 2, cy + hh / 2};
    }
    f4 *d_boxes; u64 *d_cyc, *d_sink;
    hipMalloc(&d_boxes, h.size() * sizeof(f4));
    hipMalloc(&d_cyc, grid * 16 * 8);
    hipMalloc(&d_sink, grid * 16 * 8);
    hipMemcpy(d_boxes, h.data(), h.size() * sizeof(f4), hipMemcpyHostToDevice);
    const int KC = 32, REP = 200;
    for (int waves : {16, 8, 4}) {
        run<0, 1>("V0 cand+branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<0, 2>("V0 cand+branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<0, 4>("V0 cand+branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<1, 1>("V1 bracket, branch-free", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<1, 2>("V1 bracket, branch-free", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<1, 4>("V1 bracket, branch-free", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<2, 2>("V2 V0 x2 kept/trip", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<3, 1>("V3 4 kept/trip, 1 branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<3, 2>("V3 4 kept/trip, 1 branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<3, 4>("V3 4 kept/trip, 1 branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<5, 1>("V5 8 kept/trip, 1 branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<5, 2>("V5 8 kept/trip, 1 branch", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<6, 1>("V6 4 kept/trip, SGPR box", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<6, 2>("V6 4 kept/trip, SGPR box", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<7, 1>("V7 8 kept/trip, packed f32", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<7, 2>("V7 8 kept/trip, packed f32", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<7, 4>("V7 8 kept/trip, packed f32", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<4, 1>("V4 candidate only (floor)", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<4, 2>("V4 candidate only (floor)", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
        run<4, 4>("V4 candidate only (floor)", d_boxes, grid, waves, KC, REP, thr, d_cyc, d_sink);
    }
    return 0;
}
