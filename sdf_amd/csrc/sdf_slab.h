// sdf_slab.h -- the multi-GPU exchange unit and the launchers of its kernels (sdf_plain.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace sdfk { struct MeshCounters; }

// ---- the multi-GPU exchange unit ("slab"): what one rank contributes to the all-gather (sdf_amd/dist.py) ----
// [header 128 B | prefix[cap_items] u64 | xf[cap_items][6] f64 | tris[cap_tris] 16 B | raw[raw_cap][9] f32], a fixed
// capacity per call so that ONE all-gather of equal-sized slabs moves everything: the counts travel in the header, the
// per-batch transforms next to the triangles, and a triangle as 16 BYTES instead of the 72 of the float64 soup (r03: 36):
// xGMI is point to point, an all-gather's time is one slab over one link.
//
// A marching-cubes triangle in the batch's local voxel coordinates has its three vertices on edges of ONE cell: of a
// vertex's three float32 coordinates two are integers c or c + 1 and one lies along the edge.  Tri16 keeps the three
// along-edge floats bit for bit and packs the rest into one word: the cell (3 x 6 bits) and per vertex the axis of its
// float (2 bits) and the two offsets (2 bits).  The code is a function of the nine floats alone (slab_encode16) and
// decodes to exactly those floats (slab_decode16) -- any triangle that does not have that shape (a vertex inside a cell:
// the centre vertex of some of Lewiner's tilings of ambiguous cells, ~ 0.1 % of the triangles at most) keeps its nine
// floats in the raw area behind the records and its record says where.  k_expand turns the gathered slabs into the
// ordered float64 soup on every rank; the floats it transforms are the ones marching cubes produced.
struct SlabHeader {
    long long n_tris, n_items, overflow, n_empty, n_nonempty, n_eval, n_ambiguous, n_sampled, n_pruned, n_work_total;
    long long n_raw;            // triangles in the raw area (may exceed its capacity: then `overflow` has bit 8 set and ...
    long long need_tris;        // ... this is the triangle capacity whose raw area would hold them; else = n_tris)
    long long pad_[4];
};
struct Tri16 { unsigned code; float f[3]; };
static_assert(sizeof(Tri16) == 16, "slab triangle record");
enum : unsigned { TRI16_RAW = 1u << 31 };
enum { SLAB_RAW_DIV = 128, SLAB_RAW_MIN = 256 };   // raw capacity of a slab of cap_tris triangles: cap_tris / 128 + 256 (0.8 %; measured share <= 0.3 %)
// nine local float32 -> the record; false: not of the edge shape (the caller stores it raw)
__host__ __device__ inline bool slab_encode16(const float *o, Tri16 &r) {
    int c[3];
    for (int a = 0; a < 3; a++) {
        float m = o[a] < o[3 + a] ? o[a] : o[3 + a];
        m = m < o[6 + a] ? m : o[6 + a];
        const int ci = m >= 0.0f && m < 64.0f ? (int)m : 0;     // (floor; NaN, negative, huge: 0 -- such a triangle ends up raw or encodes all the same)
        c[a] = ci;
    }
    unsigned code = (unsigned)c[0] | ((unsigned)c[1] << 6) | ((unsigned)c[2] << 12);
    bool ok = true;
    for (int k = 0; k < 3; k++) {
        int frac = -1, nfrac = 0;
        unsigned off = 0;
        for (int a = 0; a < 3; a++) {
            const float v = o[3 * k + a];
            const bool lo = v == (float)c[a], hi = v == (float)(c[a] + 1);
            if (!lo && !hi) { nfrac++; frac = a; }
            off |= hi ? 1u << a : 0u;
        }
        if (nfrac > 1) ok = false;
        if (frac < 0) frac = 0;                                  // all three on the lattice: keep the first as "the float"
        const int a1 = frac == 0 ? 1 : 0, a2 = frac == 2 ? 1 : 2;   // the two other axes, ascending
        code |= ((unsigned)frac | (((off >> a1) & 1u) << 2) | (((off >> a2) & 1u) << 3)) << (18 + 4 * k);
        r.f[k] = o[3 * k + frac];
    }
    r.code = code;
    return ok;
}
__host__ __device__ inline void slab_decode16(const Tri16 &r, float *o) {
    const int c[3] = {(int)(r.code & 63u), (int)((r.code >> 6) & 63u), (int)((r.code >> 12) & 63u)};
    for (int k = 0; k < 3; k++) {
        const unsigned v = (r.code >> (18 + 4 * k)) & 15u;
        const int frac = (int)(v & 3u), a1 = frac == 0 ? 1 : 0, a2 = frac == 2 ? 1 : 2;
        o[3 * k + frac] = r.f[k];
        o[3 * k + a1] = (float)(c[a1] + (int)((v >> 2) & 1u));
        o[3 * k + a2] = (float)(c[a2] + (int)((v >> 3) & 1u));
    }
}
static_assert(sizeof(SlabHeader) == 128, "slab header");
struct SlabLayout {
    size_t prefix_off, xf_off, tris_off, raw_off, bytes;
    long long raw_cap;
    __host__ __device__ SlabLayout(long long cap_items, long long cap_tris) {
        prefix_off = 128;
        xf_off = prefix_off + (size_t)cap_items * 8;
        tris_off = (xf_off + (size_t)cap_items * 48 + 15) & ~(size_t)15;
        raw_cap = cap_tris / SLAB_RAW_DIV + SLAB_RAW_MIN;
        raw_off = tris_off + (size_t)cap_tris * 16;
        bytes = (raw_off + (size_t)raw_cap * 36 + 255) & ~(size_t)255;
    }
};
struct SlabPtrs { const unsigned char *p[64]; };

// (each returns hipGetLastError() of its launch)
int sdf_launch_pack_slab(unsigned blocks, hipStream_t stream, const sdfk::MeshCounters *ctr, const unsigned long long *status,
                         unsigned char *slab, long long cap_items, long long cap_tris);
int sdf_launch_expand(hipStream_t stream, const SlabPtrs &slabs, int n_slabs, long long cap_items, long long cap_tris, double *out,
                      unsigned long long cap_out);
int sdf_launch_collect_headers(hipStream_t stream, const SlabPtrs &slabs, int n_slabs, long long *out);
