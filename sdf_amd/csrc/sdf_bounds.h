// sdf_bounds.h -- launcher of k_estimate_bounds_w (sdf_bounds.hip).  f64: SDF_PRECISION_F64; full: the tape uses the trigonometric ops;
// work: SDF_BOUNDS_WORK_BYTES of device memory (the waves' per-round exchange words, 32 rounds x 64 waves); tag: 1 .. 65535, different
// from the tag of every call since `work` was last zeroed (the words are NOT cleared per call)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#define SDF_BOUNDS_WORK_BYTES (32 * 64 * 8)
int sdf_launch_bounds(int f64, int full, hipStream_t stream, const uint32_t *code, const void *consts, double *out, void *work, unsigned tag);
