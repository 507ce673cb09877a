// Read-bandwidth ceiling microbenchmark (development aid; not part of the library).
// Streams N bytes with 16-B loads per lane, UN loads in flight per thread, plain or non-temporal,
// reduces to one value per thread so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_read(const v4 *__restrict__ p, long long n, float *out) {
    long long i = (long long)blockIdx.x * blockDim.x * UN + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x * UN;
    float acc = 0.f;
    for (; i + (long long)(UN - 1) * 256 < n; i += stride) {
        v4 r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) r[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

__device__ __forceinline__ float sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.44269504f)); }

// same stream + the fused kernel's per-element arithmetic (sigmoid, running max/index)
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_read_sig(const v4 *__restrict__ p, long long n, float *out) {
    long long i = (long long)blockIdx.x * blockDim.x * UN + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x * UN;
    float best[4] = {-1.f, -1.f, -1.f, -1.f};
    int bi[4] = {0, 0, 0, 0};
    int k = 0;
    for (; i + (long long)(UN - 1) * 256 < n; i += stride) {
        v4 r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) r[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float v[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { float s = sig(v[j]); if (s > best[j]) { best[j] = s; bi[j] = k; } }
            ++k;
        }
    }
    if (best[0] + best[1] + best[2] + best[3] == 123.456f) out[0] = (float)(bi[0] + bi[1] + bi[2] + bi[3]);
}

// the fused kernel's ADDRESS pattern without its arithmetic: wave = 256 consecutive cells of one plane set,
// walks NP planes (stride = plane bytes), UN planes per batch
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_read_planes(const float *__restrict__ p, int sets, int tiles, int NP,
                                                     long long cells, float *out) {
    long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long long)sets * tiles) return;
    int set = (int)(w / tiles), t = (int)(w % tiles);
    const int c0 = t * 256 + (threadIdx.x & 63) * 4;
    if (c0 + 4 > cells) return;
    const float *q = p + ((long long)set * NP) * cells + c0;
    float acc = 0.f;
    for (int ch = 0; ch + UN <= NP; ch += UN) {
        v4 r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const v4 *a = reinterpret_cast<const v4 *>(q + (long long)(ch + u) * cells);
            r[u] = NT ? __builtin_nontemporal_load(a) : *a;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// one workgroup per plane set: its waves walk the NP contiguous planes together (a linear stream per workgroup)
template <int UN, bool NT>
__global__ void k_read_setwg(const float *__restrict__ p, int NP, long long cells, float *out) {
    const int f = threadIdx.x;                       // float4 index inside a plane
    if (f * 4 + 4 > cells) return;
    const float *q = p + ((long long)blockIdx.x * NP) * cells + f * 4;
    float acc = 0.f;
    for (int ch = 0; ch + UN <= NP; ch += UN) {
        v4 r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const v4 *a = reinterpret_cast<const v4 *>(q + (long long)(ch + u) * cells);
            r[u] = NT ? __builtin_nontemporal_load(a) : *a;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int UN, bool NT>
static void run_setwg(const float *d, float *out, const char *name, int sets, long long cells) {
    const int NP = 84;
    int threads = (int)((cells / 4 + 63) / 64) * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_read_setwg<UN, NT>), dim3(sets), dim3(threads), 0, 0, d, NP, cells, out);
    hipEventRecord(e0);
    const int it = 20;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_read_setwg<UN, NT>), dim3(sets), dim3(threads), 0, 0, d, NP, cells, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    double bytes = (double)sets * NP * cells * 4;
    printf("%-18s set-per-WG (%d thr) sets %d cells %lld: %.4f ms  %.0f GB/s\n", name, threads, sets, cells, ms, bytes / ms / 1e6);
}

template <int UN, bool NT>
static void run_planes(const float *d, float *out, const char *name, int sets = 768, long long cells = 2816) {
    const int NP = 84;                              // NP % UN == 0
    const int tiles = (int)((cells + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = (sets * tiles + 3) / 4;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_read_planes<UN, NT>), dim3(blocks), dim3(256), 0, 0, d, sets, tiles, NP, cells, out);
    hipEventRecord(e0);
    const int it = 20;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_read_planes<UN, NT>), dim3(blocks), dim3(256), 0, 0, d, sets, tiles, NP, cells, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    double bytes = (double)sets * NP * cells * 4;
    printf("%-18s planes pattern sets %d cells %lld: %.4f ms  %.0f GB/s\n", name, sets, cells, ms, bytes / ms / 1e6);
}

template <int UN, bool NT>
static void run_sig(const v4 *d, long long n, float *out, int blocks, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_read_sig<UN, NT>), dim3(blocks), dim3(256), 0, 0, d, n, out);
    hipEventRecord(e0);
    const int it = 20;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_read_sig<UN, NT>), dim3(blocks), dim3(256), 0, 0, d, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    printf("%-18s blocks %6d: %.4f ms  %.0f GB/s (with sigmoid+max)\n", name, blocks, ms, (double)n * 16 / ms / 1e6);
}

template <int UN, bool NT>
static void run(const v4 *d, long long n, float *out, int blocks, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_read<UN, NT>), dim3(blocks), dim3(256), 0, 0, d, n, out);
    hipEventRecord(e0);
    const int it = 20;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_read<UN, NT>), dim3(blocks), dim3(256), 0, 0, d, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    printf("%-18s blocks %6d: %.4f ms  %.0f GB/s\n", name, blocks, ms, (double)n * 16 / ms / 1e6);
}

int main() {
    const long long bytes = 926714880LL;            // the fused decode's input per 256-image batch
    const long long n = bytes / 16;
    v4 *d; float *out;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes);
    run_sig<2, true>(d, n, out, 2048, "UN2 nt sig");
    run_sig<4, true>(d, n, out, 2048, "UN4 nt sig");
    run_sig<4, true>(d, n, out, 8192, "UN4 nt sig");
    run_planes<2, true>((const float *)d, out, "UN2 nt");
    run_planes<4, true>((const float *)d, out, "UN4 nt");
    run_planes<4, false>((const float *)d, out, "UN4 plain");
    run_planes<12, true>((const float *)d, out, "UN12 nt");
    run_planes<2, true>((const float *)d, out, "UN2 nt G52", 768, 2704);
    run_planes<4, true>((const float *)d, out, "UN4 nt G52", 768, 2704);
    run_planes<2, true>((const float *)d, out, "UN2 nt G26", 3072, 676);
    run_planes<4, true>((const float *)d, out, "UN4 nt G26", 3072, 676);
    run_setwg<2, true>((const float *)d, out, "UN2 nt G52", 768, 2704);
    run_setwg<4, true>((const float *)d, out, "UN4 nt G52", 768, 2704);
    run_setwg<6, true>((const float *)d, out, "UN6 nt G52", 768, 2704);
    run_setwg<2, true>((const float *)d, out, "UN2 nt G26", 3072, 676);
    run_setwg<4, true>((const float *)d, out, "UN4 nt G26", 3072, 676);
    run_planes<2, true>((const float *)d, out, "UN2 nt G80", 384, 6400);
    run_planes<2, true>((const float *)d, out, "UN2 nt G40", 1536, 1600);
    run_planes<2, true>((const float *)d, out, "UN2 nt G20", 6144, 400);
    for (int blocks : {2048, 8192}) {
        run<1, false>(d, n, out, blocks, "UN1 plain");
        run<1, true>(d, n, out, blocks, "UN1 nt");
        run<4, false>(d, n, out, blocks, "UN4 plain");
        run<4, true>(d, n, out, blocks, "UN4 nt");
        run<8, true>(d, n, out, blocks, "UN8 nt");
    }
    hipDeviceSynchronize();
    return 0;
}
