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
// FLAT over the arena: a unit of 256 bytes of a tile's sample area is 64 float32 samples -- task t of a culled tile (eight listed
// units of 2^3 samples, cull_sample's order = TileView's), or samples [64 j, 64 j + 64) of a dense one -- and `owner[u]` (written
// by k_cull next to the place it hands out) says which work item unit u belongs to (-1: a unit of a tile's sign-bit area).  Every
// WAVE works on its own: it draws a run of SAMPLE_RUN consecutive units from the counter, takes them NS at a time through the
// interpreter (two units of the same tile share a pass; a lone one, at a tile boundary, goes alone) and stores unit u's 64 values
// at arena float 64 u + lane.  No LDS, no barrier, no tile-sized work item: the 512^3 example is ~ 98 k units for 4096 waves,
// where whole tiles were 1744 items of very different sizes for 1024 workgroups (measured, r05c: 134 us instead of the ~ 50 us
// the arithmetic needs).  Per-tile bookkeeping (statistics, sign bits) is k_march's.
enum { SAMPLE_RUN = 8 };
template <typename T, bool FULL, int NP, int ND, int NS>
__global__ __launch_bounds__(SPLIT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_sample(const uint32_t *__restrict__ code, const T *__restrict__ consts, MeshArgs a) {
    typedef Vec<T, NS> V;
    static_assert(NS == 1 || NS == 2, "units go through the interpreter one or two at a time");
    const int lane = threadIdx.x & 63;
    const GridDesc g = a.g;
    // (pointers that come out of the by-value argument block are generic to the compiler: said to be global here, so that the
    // loop's loads and stores are global_* instructions, which do not count against the scalar loads' lgkmcnt)
    typedef __attribute__((address_space(1))) const unsigned char gcuchar;
    typedef __attribute__((address_space(1))) const int gcint;
    typedef __attribute__((address_space(1))) const double gcdouble;
    typedef __attribute__((address_space(1))) float gfloat;
    gcuchar *const cull = (gcuchar *)a.cull;
    gcint *const worklist = (gcint *)a.worklist, *const owner = (gcint *)a.owner;
    gfloat *const arena = (gfloat *)a.tiles;
    gcdouble *const gX = (gcdouble *)a.g.X, *const gY = (gcdouble *)a.g.Y, *const gZ = (gcdouble *)a.g.Z;
    const int tape_stride = a.tape_stride;
    MeshCounters *const ctr = a.ctr;
    if (threadIdx.x == 0) {   // the pass's start on the device's own clock (sdf_stats.ms_mesh_device, sclk_mhz)
        const unsigned long long tw = wall_clock64();
        atomicMax(&ctr->t_first_inv, ~tw);
        if (blockIdx.x == 0) { ctr->clk_cycles = (unsigned long long)clock64(); ctr->clk_ticks = tw; }
    }
    const unsigned long long n_units = ctr->tile_cursor;
    if (n_units > a.tiles_cap256) {   // (uniform) the arena is too small: flagged, the host repeats the call with the size k_cull reported
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(&ctr->overflow, (unsigned)MESH_OVERFLOW_TILES);
        return;
    }
    // the tile in hand (uniform; reloaded when a unit of another work item comes up)
    int cur_w = -1, ox = 0, oy = 0, oz = 0, lx = 1, ly = 1, lz = 1, lyz = 1, nvox = 0;
    unsigned long long cur_off = 0;
    bool culled = false;
    float inv_lyz = 1.0f, inv_lz = 1.0f;
    typedef __attribute__((address_space(1))) const unsigned short gcushort;
    gcushort *units = nullptr;
    int cur_b = 0;
    // (runs are dealt out STATICALLY, wave by wave: a work counter in device memory was drawn from by 4096 waves at once and the
    // atomics, serialised at one address behind eight L2s, were most of the kernel -- r05d: 253 us, 87 % of the wave cycles waiting)
    const unsigned n_waves = gridDim.x * (SPLIT_BLOCK / 64), wave_id = blockIdx.x * (SPLIT_BLOCK / 64) + (threadIdx.x >> 6);
    for (unsigned long long run = wave_id;; run += n_waves) {
        const unsigned long long u0 = run * SAMPLE_RUN;
        if (u0 >= n_units) break;
        const int own = (lane < SAMPLE_RUN && u0 + (unsigned long long)lane < n_units) ? owner[u0 + lane] : -1;
        for (int p_ = 0; p_ < SAMPLE_RUN;) {                                  // (uniform; said so explicitly: the values below feed scalar loads)
            const int p = __builtin_amdgcn_readfirstlane(p_);
            const int w = __builtin_amdgcn_readlane(own, p);
            if (w < 0) { p_ = p + 1; continue; }
            const bool pair = NS == 2 && p + 1 < SAMPLE_RUN && __builtin_amdgcn_readlane(own, min(p + 1, SAMPLE_RUN - 1)) == w;
            if (w != cur_w) {
                cur_w = w;
                typedef __attribute__((address_space(1))) const unsigned gcunsigned;
                gcunsigned *rec = (gcunsigned *)(cull + (size_t)w * CULL_RECORD);
                const int b = __builtin_amdgcn_readfirstlane(worklist[w]);
                const unsigned n0 = (unsigned)__builtin_amdgcn_readfirstlane((int)rec[0]) & 0xFFFFu;
                cur_off = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)rec[1]);
                culled = n0 != 0xFFFFu;
                batch_origin(g, b, ox, oy, oz, lx, ly, lz);
                lyz = ly * lz; nvox = lx * lyz;
                inv_lyz = 1.0f / (float)lyz; inv_lz = 1.0f / (float)lz;
                units = (gcushort *)((gcuchar *)rec + CULL_ULIST);
                cur_b = b;
            }
            // (the tile's state is uniform by construction; the compiler cannot see that through the loop)
            ox = uni(ox); oy = uni(oy); oz = uni(oz); lx = uni(lx); ly = uni(ly); lz = uni(lz); lyz = uni(lyz); nvox = uni(nvox);
            cur_off = uni64(cur_off);
            units = (gcushort *)uni64((unsigned long long)units);
            // (`code` stays the base of the tape's address: the interpreter's instruction fetches are scalar loads from a kernel argument)
            const uint32_t *wc = code + (size_t)__builtin_amdgcn_readfirstlane(cur_b) * (size_t)tape_stride * 2;
            V px, py, pz;
            bool valid[NS];
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const int t = (int)(u0 + (unsigned long long)(p + (pair ? k : 0)) - cur_off);   // the unit within its tile (a lone unit fills both slots)
                int ix, iy, iz;
                if (culled) {   // (cull_sample, on the record in device memory)
                    const unsigned u = units[8 * t + (lane >> 3)];
                    const int x = (int)((u >> 9) & 62u) + ((lane >> 2) & 1), y = (int)((u >> 4) & 62u) + ((lane >> 1) & 1), z = (int)((u << 1) & 62u) + (lane & 1);
                    valid[k] = x < lx && y < ly && z < lz;
                    ix = valid[k] ? x : 0; iy = valid[k] ? y : 0; iz = valid[k] ? z : 0;
                } else {
                    const int i = min(64 * t + lane, nvox - 1);
                    valid[k] = 64 * t + lane < nvox;
                    ix = fast_div(i, inv_lyz); const int r = i - ix * lyz; iy = fast_div(r, inv_lz); iz = r - iy * lz;
                }
                px.v[k] = (T)gX[ox + ix]; py.v[k] = (T)gY[oy + iy]; pz.v[k] = (T)gZ[oz + iz];
            }
            const V val = run_tape<T, FULL, NP, ND, NS>(wc, consts, px, py, pz);
            SDF_UNROLL
            for (int k = 0; k < NS; k++)
                if (valid[k] && (k == 0 || pair)) arena[(u0 + (unsigned long long)(p + k)) * 64ull + (unsigned long long)lane] = (float)val.v[k];
            p_ = p + (pair ? 2 : 1);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctr->clk_cycles = (unsigned long long)clock64() - ctr->clk_cycles;
        ctr->clk_ticks = wall_clock64() - ctr->clk_ticks;
    }
}

// host-side launcher of k_sample for one float64 family (defined in sdf_mesh_inst.hip next to k_mesh's); slots as for k_mesh
#define SDF_DECLARE_SAMPLE_LAUNCH(NAME) \
    int NAME(int slots, int grid, hipStream_t stream, const uint32_t *code, const double *consts, const MeshArgs &a)
SDF_DECLARE_SAMPLE_LAUNCH(sdf_launch_sample_f64);
SDF_DECLARE_SAMPLE_LAUNCH(sdf_launch_sample_f64_full);

// k_march (sdf_plain.hip): LDS per workgroup decides how many share a compute unit
enum { MARCH_LCAP = 3072, MARCH_CELLS = 2048 };
// block: 256 (four workgroups per compute unit), 512 (two) or 1024 (one) threads -- grid = that many per compute unit
// two: the counting and the emitting as launches of their own (nothing waits) instead of one launch whose look-back waits
int sdf_launch_march(int block, int two, int n_cu, int nb, hipStream_t stream, const MeshArgs &a);

}  // namespace sdfk
