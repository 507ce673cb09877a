// Opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled (the default) it is a single predictable branch per launch; nothing is created.
#pragma once
#include <hip/hip_runtime.h>

enum vx_prof_slot {
    VX_PROF_PREPROCESS = 0, VX_PROF_HEAD_DECODE, VX_PROF_DECODE_ROWS, VX_PROF_FUSED, VX_PROF_FINISH,
    VX_PROF_PRIOR, VX_PROF_NMS_KEYS, VX_PROF_SORT1, VX_PROF_SWEEP, VX_PROF_SORT2, VX_PROF_PACK, VX_PROF_NMS_IMAGE,
    VX_PROF_DETR, VX_PROF_SOFTMAX, VX_PROF_BM_PAIRS, VX_PROF_BM_SCAN,
    VX_PROF_NSLOTS
};

extern bool vx_prof_on;
void vx_prof_begin_impl(int slot, hipStream_t st);
void vx_prof_end_impl(int slot, hipStream_t st);

static inline void vx_prof_begin(int slot, hipStream_t st) { if (vx_prof_on) vx_prof_begin_impl(slot, st); }
static inline void vx_prof_end(int slot, hipStream_t st) { if (vx_prof_on) vx_prof_end_impl(slot, st); }

// device counter of useful lane-level IoU tests (NMS sweeps add to it) or nullptr: vx_profile_set_iou_counter
unsigned long long *vx_iou_counter();
