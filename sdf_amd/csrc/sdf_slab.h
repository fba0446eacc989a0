// sdf_slab.h -- the multi-GPU exchange unit and the launchers of its kernels (sdf_plain.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace sdfk { struct MeshCounters; }

// ---- the multi-GPU exchange unit ("slab"): what one rank contributes to the all-gather (sdf_amd/dist.py) ----
// [header 128 B | prefix[cap_items] u64 | xf[cap_items][6] f64 | tris[cap_tris][9] f32], a fixed capacity per call so
// that ONE all-gather of equal-sized slabs moves everything: the counts travel in the header, the triangles in
// marching cubes' own local float32 form (36 B instead of the 72 B of the float64 soup), the per-batch transforms
// next to them.  k_expand turns the gathered slabs into the ordered float64 soup on every rank.
struct SlabHeader {
    long long n_tris, n_items, overflow, n_empty, n_nonempty, n_eval, n_ambiguous, n_sampled, n_pruned, n_work_total;
    long long pad_[6];
};
static_assert(sizeof(SlabHeader) == 128, "slab header");
struct SlabLayout {
    size_t prefix_off, xf_off, tris_off, bytes;
    __host__ __device__ SlabLayout(long long cap_items, long long cap_tris) {
        prefix_off = 128;
        xf_off = prefix_off + (size_t)cap_items * 8;
        tris_off = (xf_off + (size_t)cap_items * 48 + 15) & ~(size_t)15;
        bytes = (tris_off + (size_t)cap_tris * 36 + 255) & ~(size_t)255;
    }
};
struct SlabPtrs { const unsigned char *p[64]; };

// (each returns hipGetLastError() of its launch)
int sdf_launch_pack_slab(unsigned blocks, hipStream_t stream, const sdfk::MeshCounters *ctr, const unsigned long long *status,
                         unsigned char *slab, long long cap_items, long long cap_tris);
int sdf_launch_expand(hipStream_t stream, const SlabPtrs &slabs, int n_slabs, long long cap_items, long long cap_tris, double *out,
                      unsigned long long cap_out);
int sdf_launch_collect_headers(hipStream_t stream, const SlabPtrs &slabs, int n_slabs, long long *out);
