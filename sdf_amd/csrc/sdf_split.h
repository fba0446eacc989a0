// sdf_split.h -- the meshing pass as TWO kernels of SMALL workgroups (round 5): k_sample + k_march instead of k_mesh.
//
// k_mesh (sdf_device.h) holds a whole compute unit with ONE workgroup of 1024 threads and runs a batch's phases one after the
// other behind workgroup barriers: take the item, sample, count, allocate, emit.  Only the sampling keeps the vector ALU busy;
// round 4 measured 57 % of the wave cycles parked at waitcnt / barriers and 70 % of a round outside the interpreter
// (profiles/r04ad_sync_pmc.json, SDF_MESH_PROF).  Nothing of a batch's latency-bound phases can hide behind another batch's
// arithmetic, because there IS no other batch on the compute unit.
//
// Here the two halves of `_worker` (reference sdf/core.py:45-60) are kernels of their own, each of many independent 256-thread
// workgroups per compute unit, so that one workgroup's round trips (work counter, record, look-back, barriers) overlap with the
// arithmetic of its neighbours -- the occupancy does what hand-written phase overlap inside one workgroup would have to do:
//
//   k_sample   `volume = sdf(P)` (core.py:50-52): four workgroups per CU (the interpreter's 128 VGPRs), each takes work items
//              from the ordered list, evaluates the units k_cull listed (cull_tasks) through the tape interpreter and writes the
//              float32 samples of the SPARSE tile (64 per listed task, TileView's order) plus the tile's sign-bit volume to the
//              item's place in the TILE ARENA in device memory (~ 19 KB per tile of the 512^3 example; a tile that is not
//              culled is written dense).  No LDS tile, no marching state: its LDS is the record, the axes and the sign bits.
//   k_march    `_marching_cubes` + `points * scale + offset` (core.py:54-60): six workgroups per CU (25 KB of LDS, no
//              interpreter registers), each takes items from a counter of its own IN ORDER, brings the sign bits and k_cull's
//              column words into LDS, finds the surface cells (a thread per run of cell rows, ONE scan), their triangle counts
//              (a thread per run of cells, ONE scan; Lewiner's tests where ambiguous), publishes the count, gets its place in
//              the ordered soup by the decoupled look-back of ordered_base (blocking: its predecessors are held by running
//              workgroups that take a few microseconds per item, and the other workgroups of the CU fill the wait), and writes
//              the float64 triangles once, a lane per triangle, transposed through LDS like k_mesh's deferred emission; the
//              corner samples are read from the arena (L2 / MALL).
//
// Bit-identical to k_mesh by construction: the same interpreter on the same points, the same sign rule (value > 0), the same
// cell order, tables and vertex arithmetic (mc_vertex_view, mc33_triangle), the same transform expression.  What it costs: the
// sampled tiles go through device memory once (written by k_sample, read by k_march): + 2 x ~33 MB at 512^3 next to 212 MB of
// soup.  Selected by the host (generate_impl) when k_cull ran (float64, monotone axes); SDF_MESH_SPLIT=0 / sdf_ctx_set_split
// keep k_mesh.  An arena that turns out too small is flagged (overflow bit 4) and the call repeated through k_mesh.
#pragma once
#include "sdf_device.h"

namespace sdfk {

enum { SPLIT_BLOCK = 256 };
enum { MESH_OVERFLOW_TILES = 4u };

// bytes of a work item's tile in the arena: its samples (culled: 64 floats per listed task; else the dense tile), then the
// sign bits ((nvox + 63) / 64 + 2 words: the row extraction reads one word ahead) -- in units of 256 bytes
__host__ __device__ __forceinline__ unsigned tile_data_bytes(int ntl, int nvox) { return (unsigned)(ntl >= 0 ? 256 * ntl : ((4 * nvox + 15) & ~15)); }
__host__ __device__ __forceinline__ unsigned tile_need256(int ntl, int nvox) {
    return (tile_data_bytes(ntl, nvox) + 8u * (unsigned)(((nvox + 63) >> 6) + 2) + 255u) >> 8;
}

// ---- k_sample ------------------------------------------------------------------------------------------------
template <typename T, bool FULL, int NP, int ND, int NS>
__global__ __launch_bounds__(SPLIT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_sample(const uint32_t *__restrict__ code, const T *__restrict__ consts, MeshArgs a) {
    typedef Vec<T, NS> V;
    constexpr int BLOCK = SPLIT_BLOCK, NWAVE = BLOCK / 64;
    __shared__ int bcast[4];
    __shared__ double axes[99];                                            // X, Y, Z of the tile (33 each)
    __shared__ unsigned long long bits[((33 * 33 * 33 + 63) >> 6) + 2];    // the tile's sign bits: value > 0
    __shared__ unsigned short units[CULL_UNIT_CAP + 8];                    // k_cull's list (u16 each, whole tasks)
    __shared__ unsigned sstate[256];                                       // its sub-group states, two bits each
    int tid = threadIdx.x;
    const GridDesc g = a.g;
    const int work_begin = a.ctr->work_begin, work_end = a.ctr->work_end;
    if (tid == 0) {   // the pass's start on the device's own clock (sdf_stats.ms_mesh_device, sclk_mhz)
        const unsigned long long tw = wall_clock64();
        atomicMax(&a.ctr->t_first_inv, ~tw);
        if (blockIdx.x == 0) { a.ctr->clk_cycles = (unsigned long long)clock64(); a.ctr->clk_ticks = tw; }
    }
    for (;;) {
        if (tid == 0) bcast[0] = work_begin + (int)atomicAdd(&a.ctr->work_counter, 1u);
        __syncthreads();
        const int w = uni(bcast[0]);
        if (w >= work_end) break;
        const unsigned *rec = reinterpret_cast<const unsigned *>(a.cull + (size_t)w * CULL_RECORD);
        const int b = uni(a.worklist[w]);
        const unsigned n0 = (unsigned)uni((int)rec[0]) & 0xFFFFu;
        const unsigned long long off256 = (unsigned long long)(unsigned)uni((int)rec[1]);
        const bool culled = n0 != 0xFFFFu;
        const int ntl_cull = culled ? (int)((n0 + 7u) >> 3) : -1;
        int ox, oy, oz, lx, ly, lz;
        batch_origin(g, b, ox, oy, oz, lx, ly, lz);
        const TileTasks tt(lx, ly, lz);
        const int nvox = tt.nvox, lyz = tt.lyz;
        const int nwords = (nvox + 63) >> 6;
        // (an arena that is too small: flagged, the host repeats the call through k_mesh; k_march skips the item alike)
        if (off256 + (unsigned long long)tile_need256(ntl_cull, nvox) > a.tiles_cap256) {   // (uniform)
            if (tid == 0) atomicOr(&a.ctr->overflow, (unsigned)MESH_OVERFLOW_TILES);
            __syncthreads();   // (bcast is rewritten at the top)
            continue;
        }
        float *tile = reinterpret_cast<float *>(a.tiles + off256 * 256ull);
        unsigned long long *tile_bits = reinterpret_cast<unsigned long long *>(a.tiles + off256 * 256ull + tile_data_bytes(ntl_cull, nvox));
        if (culled) {
            const int nw_units = (16 * ntl_cull + 3) >> 2;                              // words of the unit list
            for (int i = tid; i < nw_units; i += BLOCK) reinterpret_cast<unsigned *>(units)[i] = rec[CULL_ULIST / 4 + i];
            sstate[tid] = rec[CULL_SSTATE / 4 + tid];
        }
        if (tid < lx) axes[tid] = g.X[ox + tid];
        else if (tid >= 64 && tid < 64 + ly) axes[33 + tid - 64] = g.Y[oy + tid - 64];
        else if (tid >= 128 && tid < 128 + lz) axes[66 + tid - 128] = g.Z[oz + tid - 128];
        for (int i = tid; i < nwords + 2; i += BLOCK) bits[i] = 0ull;
        __syncthreads();
        const uint32_t *wcode = code + (size_t)b * (size_t)a.tape_stride * 2;
        if (tid == 0) {
            if (a.tape_stride)
                atomicAdd(&a.ctr->n_pruned, (unsigned long long)a.n_instr - reinterpret_cast<const unsigned long long *>(wcode)[a.tape_stride - 1]);
            atomicAdd(&a.ctr->n_sampled, culled ? (unsigned long long)ntl_cull * 64ull : (unsigned long long)nvox);
        }
        const int wave = tid >> 6, lane = tid & 63;
        int ntl = tt.ntask;
        if (culled) {
            ntl = ntl_cull;
            // the sign bits of the samples of DECIDED sub-groups, straight from their states (k_mesh's sign fill: a row of lz
            // samples along z = `pos | pos << 1` of its 16 two-bit states; "positive" = 01); evaluated samples OR theirs in below
            const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
            const int hlast = (c2 - 1) >> 1;
            for (int r = tid; c0 > 0 && c1 > 0 && c2 > 0 && r < lx * ly; r += BLOCK) {
                const int ix = fast_div(r, 1.0f / (float)ly), iy = r - ly * ix;
                const unsigned st = sstate[(min(ix, c0 - 1) >> 1) * 16 + (min(iy, c1 - 1) >> 1)];
                const unsigned pos = st & ~(st >> 1) & 0x55555555u & (unsigned)((4ull << (2 * hlast)) - 1ull);
                unsigned long long rowmask = (unsigned long long)(pos | (pos << 1));
                if ((pos >> (2 * hlast)) & 1u) rowmask |= 1ull << c2;
                if (rowmask) {
                    const int o = r * lz, sh = o & 63;
                    atomicOr(&bits[o >> 6], rowmask << sh);
                    if (sh && (rowmask >> (64 - sh))) atomicOr(&bits[(o >> 6) + 1], rowmask >> (64 - sh));
                }
            }
        }
        // the listed tasks (every task of a tile that is not culled): NS per wave and pass through the interpreter, cast to
        // float32 like skimage's volume cast; sample `lane` of task t of a culled tile goes to tile[64 t + lane]
        for (int t0 = wave * NS; t0 < ntl; t0 += NWAVE * NS) {
            V px, py, pz;
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const int tk = min(t0 + k, ntl - 1);
                int ix, iy, iz;
                if (culled) cull_sample(units, tk, lane, lx, ly, lz, ix, iy, iz); else tt.sample(tk, lane, ix, iy, iz);
                px.v[k] = (T)axes[ix]; py.v[k] = (T)axes[33 + iy]; pz.v[k] = (T)axes[66 + iz];
            }
            const V val = run_tape<T, FULL, NP, ND, NS>(wcode, consts, px, py, pz);
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {   // (the sample index is worked out again rather than kept across the interpreter)
                int ix, iy, iz;
                const bool valid = t0 + k < ntl && (culled ? cull_sample(units, t0 + k, lane, lx, ly, lz, ix, iy, iz) : tt.sample(t0 + k, lane, ix, iy, iz));
                if (valid) {
                    const int i = ix * lyz + iy * tt.lz + iz;
                    const float fv = (float)val.v[k];
                    tile[culled ? 64 * (t0 + k) + lane : i] = fv;
                    if (fv > 0.0f) atomicOr(&bits[i >> 6], 1ull << (i & 63));
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < nwords + 2; i += BLOCK) tile_bits[i] = bits[i];
        // (no barrier here: the next round's writes to LDS come behind the barrier at its top, which every thread reaches
        // only after its share of this copy)
    }
    if (tid == 0 && blockIdx.x == 0) {
        a.ctr->clk_cycles = (unsigned long long)clock64() - a.ctr->clk_cycles;
        a.ctr->clk_ticks = wall_clock64() - a.ctr->clk_ticks;
    }
}

// host-side launcher of k_sample for one float64 family (defined in sdf_mesh_inst.hip next to k_mesh's); slots as for k_mesh
#define SDF_DECLARE_SAMPLE_LAUNCH(NAME) \
    int NAME(int slots, int grid, hipStream_t stream, const uint32_t *code, const double *consts, const MeshArgs &a)
SDF_DECLARE_SAMPLE_LAUNCH(sdf_launch_sample_f64);
SDF_DECLARE_SAMPLE_LAUNCH(sdf_launch_sample_f64_full);

// k_march (sdf_plain.hip): LDS per workgroup decides how many share a compute unit
enum { MARCH_BLOCK = 256, MARCH_LCAP = 3072, MARCH_CELLS = 2048, MARCH_WG_PER_CU = 4 };
int sdf_launch_march(int grid, hipStream_t stream, const MeshArgs &a);

}  // namespace sdfk
