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
    const float *q = p + ((long long)set * NP) * cells + (long long)t * 256 + (threadIdx.x & 63) * 4;
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
static void run_planes(const float *d, float *out, const char *name) {
    const int sets = 768, tiles = 11, NP = 84;     // G=52 geometry (cells padded to 11*256), NP % UN == 0
    const long long cells = 2816;
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
    printf("%-18s planes pattern: %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
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
