// Library-level entry points of libvortex_hip.so (see include/vortex_hip.h).
#include "vx_common.h"

extern "C" int vx_version(void) { return 100; }   // 0.1.0

extern "C" const char *vx_error_string(int code) {
    switch (code) {
        case VX_OK: return "ok";
        case VX_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive dimension or bad enum)";
        case VX_ERR_UNSUPPORTED: return "a documented limit of libvortex_hip is exceeded";
        case VX_ERR_WORKSPACE: return "workspace smaller than the matching *_workspace_bytes() query";
        case VX_ERR_LAUNCH: return "HIP reported an error for a kernel launch";
        case VX_ERR_ALIGNMENT: return "a device pointer is not aligned as include/vortex_hip.h requires";
        case VX_ERR_SCRATCH_LIMIT: return "vx_batched_nms_noalloc: B * pow2ceil(capacity) needs more than the 64 MiB static scratch";
        default: return "unknown libvortex_hip error code";
    }
}

// ------------------------------------------------------------------------------------------------
// Profiling hooks (see vx_profile.h / include/vortex_hip.h).
#include "vx_profile.h"
#include <vector>

bool vx_prof_on = false;

namespace {
struct Slot {
    std::vector<hipEvent_t> ev;      // 2 events per launch
    int launches = 0;
    int cap = 0;
};
Slot g_slots[VX_PROF_NSLOTS];
unsigned g_mask = ~0u;               // slots that record events (vx_profile_select)
const char *g_names[VX_PROF_NSLOTS] = {
    "vx_k_normalize/resize", "vx_k_yolo_head_decode", "vx_k_yolo_decode_rows", "vx_k_yolo_fused",
    "vx_k_cand_finish", "vx_k_prior_decode", "vx_k_nms_build_keys", "vx_k_sort(keys1)+gather",
    "vx_k_nms_sweep", "vx_k_sort(keys2)+output", "vx_k_pack_rows", "vx_k_nms_image",
    "vx_k_detr_decode", "vx_k_softmax_max", "vx_k_bm_pairs", "vx_k_bm_finish/scan"};
}  // namespace

void vx_prof_begin_impl(int slot, hipStream_t st) {
    Slot &s = g_slots[slot];
    if (s.launches >= s.cap || !((g_mask >> slot) & 1u)) return;
    (void)hipEventRecord(s.ev[2 * s.launches], st);
}
void vx_prof_end_impl(int slot, hipStream_t st) {
    Slot &s = g_slots[slot];
    if (s.launches >= s.cap || !((g_mask >> slot) & 1u)) return;
    (void)hipEventRecord(s.ev[2 * s.launches + 1], st);
    ++s.launches;
}

extern "C" int vx_profile_enable(int max_launches_per_kernel) {
    if (max_launches_per_kernel <= 0 || max_launches_per_kernel > 65536) return VX_ERR_INVALID_ARG;
    for (int i = 0; i < VX_PROF_NSLOTS; ++i) {
        Slot &s = g_slots[i];
        while ((int)s.ev.size() < 2 * max_launches_per_kernel) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return VX_ERR_LAUNCH;
            s.ev.push_back(e);
        }
        s.cap = max_launches_per_kernel;
        s.launches = 0;
    }
    vx_prof_on = true;
    return VX_OK;
}

extern "C" int vx_profile_select(unsigned slot_mask) {
    g_mask = slot_mask;
    return VX_OK;
}

extern "C" int vx_profile_disable(void) {
    vx_prof_on = false;
    return VX_OK;
}

extern "C" int vx_profile_slots(void) { return VX_PROF_NSLOTS; }

extern "C" int vx_profile_get(int slot, const char **name, int *launches, float *total_ms, float *min_ms,
                              float *max_ms) {
    if (slot < 0 || slot >= VX_PROF_NSLOTS || !name || !launches || !total_ms || !min_ms || !max_ms)
        return VX_ERR_INVALID_ARG;
    Slot &s = g_slots[slot];
    *name = g_names[slot];
    *launches = s.launches;
    float tot = 0.f, mn = 1e30f, mx = 0.f;
    for (int i = 0; i < s.launches; ++i) {
        if (hipEventSynchronize(s.ev[2 * i + 1]) != hipSuccess) return VX_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev[2 * i], s.ev[2 * i + 1]) != hipSuccess) return VX_ERR_LAUNCH;
        tot += ms; mn = ms < mn ? ms : mn; mx = ms > mx ? ms : mx;
    }
    *total_ms = tot; *min_ms = s.launches ? mn : 0.f; *max_ms = mx;
    return VX_OK;
}

// caller-owned device word the NMS sweeps add their executed IoU tests to (nullptr = off, the default)
static unsigned long long *g_iou_counter = nullptr;
unsigned long long *vx_iou_counter() { return g_iou_counter; }
extern "C" int vx_profile_set_iou_counter(uint64_t *d_counter) {
    if ((uintptr_t)d_counter & 7) return VX_ERR_ALIGNMENT;
    g_iou_counter = (unsigned long long *)d_counter;
    return VX_OK;
}

// ------------------------------------------------------------------------------------------------
// Do two streams of this process run kernels side by side?  HIP spreads a process's streams over a few hardware
// queues; two streams that share one run their kernels strictly one after the other, whatever the kernels leave idle.
// The probe runs a one-wave dependent-FMA chain alone on stream_a, then one on each stream, and compares the times.
static __device__ float vx_spin_sink;

__global__ __launch_bounds__(64) void vx_k_spin(int iters) {
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.99999994f, 1.0f);
    if (x == -1.0f) vx_spin_sink = x;                  // never true: keeps the chain alive
}

extern "C" int vx_streams_overlap(void *stream_a, void *stream_b, int *overlap, float *ms_alone, float *ms_both) {
    if (!overlap) return VX_ERR_INVALID_ARG;
    hipStream_t sa = (hipStream_t)stream_a, sb = (hipStream_t)stream_b;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    const int iters = 40000;                              // ~0.1 ms
    float alone = 0.f, both = 0.f;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventCreate(&e2) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(vx_k_spin, dim3(1), dim3(64), 0, sa, iters);   // both queues exist, clocks are up
        hipLaunchKernelGGL(vx_k_spin, dim3(1), dim3(64), 0, sb, iters);
        ok = hipStreamSynchronize(sa) == hipSuccess && hipStreamSynchronize(sb) == hipSuccess;
    }
    // best of three rounds each: a round disturbed by anything else on the GPU only ever reads longer
    for (int r = 0; r < 3 && ok; ++r) {
        float t = 0.f, tb = 0.f;
        ok = hipEventRecord(e0, sa) == hipSuccess;
        hipLaunchKernelGGL(vx_k_spin, dim3(1), dim3(64), 0, sa, iters);
        ok = ok && hipEventRecord(e1, sa) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
             hipEventElapsedTime(&t, e0, e1) == hipSuccess;
        if (ok && (r == 0 || t < alone)) alone = t;
        if (!ok) break;
        ok = hipEventRecord(e0, sa) == hipSuccess;
        hipLaunchKernelGGL(vx_k_spin, dim3(1), dim3(64), 0, sa, iters);
        hipLaunchKernelGGL(vx_k_spin, dim3(1), dim3(64), 0, sb, iters);
        ok = ok && hipEventRecord(e1, sa) == hipSuccess && hipEventRecord(e2, sb) == hipSuccess &&
             hipEventSynchronize(e1) == hipSuccess && hipEventSynchronize(e2) == hipSuccess &&
             hipEventElapsedTime(&t, e0, e1) == hipSuccess && hipEventElapsedTime(&tb, e0, e2) == hipSuccess;
        t = t > tb ? t : tb;
        if (ok && (r == 0 || t < both)) both = t;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (!ok) return VX_ERR_LAUNCH;
    *overlap = both < 1.5f * alone ? 1 : 0;
    if (ms_alone) *ms_alone = alone;
    if (ms_both) *ms_both = both;
    return VX_OK;
}
