// sdf_bounds.h -- launcher of k_estimate_bounds (sdf_bounds.hip).  f64: SDF_PRECISION_F64; full: the tape uses the trigonometric ops;
// work: 32 x 4 64-bit words of device memory (the four workgroups' per-round exchange; zeroed here)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
int sdf_launch_bounds(int f64, int full, hipStream_t stream, const uint32_t *code, const void *consts, double *out, void *work);
