// sdf_weld.hip -- vertex weld of the triangle soup on the device.
//
// Replaces `np.unique(points, axis=0, return_inverse=True)` of the reference's non-STL export path
// (reference sdf/core.py:160-164, `_mesh`): the unique rows of the (3T, 3) float64 soup in
// lexicographic order (x, then y, then z) and, for every soup row, the index of its unique row.
// The reference sorts 3T structured rows on one host core (about 10 s for the 8.8 M rows of the 512^3
// example); here it is three stable LSD radix-sort passes over 64-bit keys (z, then y, then x: rocPRIM
// through hipCUB -- a plain library sort, like a library GEMM), an adjacent-row comparison, a scan and
// a scatter.  -0.0 and +0.0 are the same coordinate, as in NumPy's comparison; which of two such rows
// represents the pair is unspecified there (unstable sort) and is the first soup row here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace sdfk {

// float64 -> u64 whose unsigned order is the float order; both zeros give the same key
__device__ __forceinline__ unsigned long long sortable(double v) {
    if (v == 0.0) v = 0.0;
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | (1ull << 63));
}

__global__ __launch_bounds__(256) void k_weld_keys(const double *__restrict__ pts, const unsigned *__restrict__ perm, long long n, int comp,
                                                   unsigned long long *__restrict__ keys, unsigned *__restrict__ idx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned r = perm ? perm[i] : (unsigned)i;
    keys[i] = sortable(pts[3ull * r + comp]);
    if (!perm) idx[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void k_weld_flags(const double *__restrict__ pts, const unsigned *__restrict__ perm, long long n, int *__restrict__ flags) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int f = 1;
    if (i > 0) {
        const double *a = pts + 3ull * perm[i], *b = pts + 3ull * perm[i - 1];
        f = (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) ? 0 : 1;
    }
    flags[i] = f;
}

__global__ __launch_bounds__(256) void k_weld_scatter(const double *__restrict__ pts, const unsigned *__restrict__ perm, const int *__restrict__ flags,
                                                      const int *__restrict__ uid, long long n, double *__restrict__ uniq, long long *__restrict__ inv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned r = perm[i];
    const long long u = (long long)uid[i] - 1;
    inv[r] = u;
    if (flags[i]) {
        uniq[3 * u] = pts[3ull * r]; uniq[3 * u + 1] = pts[3ull * r + 1]; uniq[3 * u + 2] = pts[3ull * r + 2];
    }
}

#define WCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = (int)e_; goto done; } } while (0)

// pts: n rows of 3 doubles on the device.  On success *d_uniq (3 * *n_unique doubles) and *d_inv (n int64)
// are hipMalloc'ed device buffers owned by the caller.  Returns a hipError_t value (0 = ok).
int weld_device(hipStream_t stream, const double *pts, long long n, double **d_uniq, long long **d_inv, long long *n_unique) {
    int rc = 0;
    *d_uniq = nullptr; *d_inv = nullptr; *n_unique = 0;
    if (n <= 0) return 0;
    if (n >= (1ll << 31)) return (int)hipErrorInvalidValue;
    unsigned long long *k0 = nullptr, *k1 = nullptr;
    unsigned *p0 = nullptr, *p1 = nullptr;
    int *flags = nullptr, *uid = nullptr;
    void *tmp = nullptr;
    size_t tmp_sort = 0, tmp_scan = 0;
    const unsigned grid = (unsigned)((n + 255) / 256);
    int last = 0;
    WCHK(hipMalloc((void **)&k0, (size_t)n * 8)); WCHK(hipMalloc((void **)&k1, (size_t)n * 8));
    WCHK(hipMalloc((void **)&p0, (size_t)n * 4)); WCHK(hipMalloc((void **)&p1, (size_t)n * 4));
    WCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, k0, k1, p0, p1, (int)n, 0, 64, stream));
    WCHK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, (int *)nullptr, (int *)nullptr, (int)n, stream));
    WCHK(hipMalloc(&tmp, tmp_sort > tmp_scan ? tmp_sort : tmp_scan));
    // least significant field first; every pass is stable
    hipLaunchKernelGGL(k_weld_keys, dim3(grid), dim3(256), 0, stream, pts, (const unsigned *)nullptr, n, 2, k0, p0);
    WCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_sort, k0, k1, p0, p1, (int)n, 0, 64, stream));
    hipLaunchKernelGGL(k_weld_keys, dim3(grid), dim3(256), 0, stream, pts, (const unsigned *)p1, n, 1, k0, (unsigned *)nullptr);
    WCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_sort, k0, k1, p1, p0, (int)n, 0, 64, stream));
    hipLaunchKernelGGL(k_weld_keys, dim3(grid), dim3(256), 0, stream, pts, (const unsigned *)p0, n, 0, k0, (unsigned *)nullptr);
    WCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_sort, k0, k1, p0, p1, (int)n, 0, 64, stream));
    // p1 = soup rows in lexicographic order; the key buffers are free now: flags and ids live in them
    flags = reinterpret_cast<int *>(k0); uid = reinterpret_cast<int *>(k1);
    hipLaunchKernelGGL(k_weld_flags, dim3(grid), dim3(256), 0, stream, pts, (const unsigned *)p1, n, flags);
    WCHK(hipcub::DeviceScan::InclusiveSum(tmp, tmp_scan, flags, uid, (int)n, stream));
    WCHK(hipMemcpyAsync(&last, uid + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    WCHK(hipStreamSynchronize(stream));
    *n_unique = last;
    WCHK(hipMalloc((void **)d_uniq, (size_t)last * 24)); WCHK(hipMalloc((void **)d_inv, (size_t)n * 8));
    hipLaunchKernelGGL(k_weld_scatter, dim3(grid), dim3(256), 0, stream, pts, (const unsigned *)p1, (const int *)flags, (const int *)uid, n, *d_uniq, *d_inv);
    WCHK(hipGetLastError());
    WCHK(hipStreamSynchronize(stream));
done:
    for (void *q : {(void *)k0, (void *)k1, (void *)p0, (void *)p1, (void *)tmp}) (void)hipFree(q);
    if (rc) { (void)hipFree(*d_uniq); (void)hipFree(*d_inv); *d_uniq = nullptr; *d_inv = nullptr; }
    return rc;
}

}  // namespace sdfk
