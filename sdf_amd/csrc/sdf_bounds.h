// sdf_bounds.h -- launcher of k_estimate_bounds (sdf_bounds.hip).  f64: SDF_PRECISION_F64; full: the tape uses the trigonometric ops;
// slots: 0 = (1,1), 1 = (2,2), 2 = (4,4), 3 = (8,8) saved-point / distance register files (the smallest that holds the tape's slots).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// work: 1 + 6 * 32 ints of device memory (the four-workgroup scheme of the (8,8) file: its barrier and hit boxes; zeroed here)
int sdf_launch_bounds(int f64, int full, int slots, hipStream_t stream, const uint32_t *code, const void *consts, double *out, int *work);
