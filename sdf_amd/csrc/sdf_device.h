// sdf_device.h -- device-side structures and the fused sample+march kernel template (gfx950).
//
//   k_mesh   THE hot kernel: persistent workgroups (one per CU) pull surviving batches from the ordered work list.
//            Per round, for batch k:
//              1. sample   the units of 2^3 samples k_cull listed (cull_tasks) through the tape interpreter (NS tasks of
//                          64 samples per wave and pass -- three in the one-pass kernels, sdf_mesh_inst.hip --, float64), cast to float32 like skimage's volume cast,
//                          into a SPARSE tile in LDS (TileView: only the listed units; a tile that is not culled stays
//                          dense) -- the field never touches HBM (reference `_worker`, sdf/core.py:50-52); the sign bits
//                          of everything else come from the interval pass's sub-group states
//              2. count    one thread per (i0, i1) row of cells builds the row's surface-cell mask from the sign-bit
//                          volume; one thread per surface cell its triangles (Lewiner's tests where the configuration is
//                          ambiguous); two block scans number cells and triangles in skimage's emission order; the count
//                          is published (status[w])
//              3. wait     the batch stays in one of two LDS slots while the workgroup samples batch k + 1
//              4. emit     batch k - 1, one round later: a decoupled look-back over the status words gives its place in the
//                          soup (work items are handed out in order, so every predecessor is held by a running workgroup);
//                          one lane per TRIANGLE: three edge interpolations, the float64 world transform
//                          `points * scale + offset` (core.py:58-60), transposed through LDS so that consecutive lanes
//                          store consecutive coordinates -- meanwhile the last wave draws the next work item and brings
//                          its record and axes in
//            (reference `_marching_cubes`, sdf/core.py:16-18, 54).  What cannot wait -- dense tiles, lists that do not fit,
//            a waiting batch whose predecessors are still not done -- is written at once or parked (r03's scheme).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "sdf_interp.h"
#include "sdf_mc33.h"
#include "sdf_interval.h"
#include "sdf_slab.h"

namespace sdfk {

struct McTables {   // uploaded once per context
    unsigned char ntri[256];   // triangles per sign configuration
    unsigned char amb[256];    // 1 when the configuration is ambiguous (classic table vs Lewiner)
    signed char tri[256][16];  // edge ids, 3 per triangle
    signed char mc33[MC33_FLAT_SIZE];   // Lewiner's tables for the ambiguous configurations (sdf_mc33.h)
};

struct MeshCounters {   // zeroed before every k_mesh run
    unsigned long long n_pruned;      // instructions the interval prepass removed, summed over the batches meshed
    unsigned long long n_eval;
    unsigned int work_counter;
    unsigned int overflow;
    unsigned int n_empty, n_nonempty;
    unsigned long long n_ambiguous;
    unsigned long long n_sampled;     // samples that went through the interpreter (the others were decided by intervals)
    unsigned long long total;         // triangles of this shard (written by the workgroup of the last work item / by k_scan_items)
    unsigned long long cell_cursor;   // two-pass meshing: surface-cell records / triangle-list entries handed out so far
    unsigned long long list_cursor;
    // the kernel's own clock readings (k_mesh): when its first workgroup started (kept as the maximum of ~t so that a
    // zeroed block means "none yet") and its last one ended, in ticks of the constant 100 MHz counter; and what
    // workgroup 0 saw between its start and its end on both counters (shader cycles / 100 MHz ticks = the shader clock
    // the kernel actually ran at)
    unsigned long long t_first_inv, t_last, clk_cycles, clk_ticks;
    unsigned long long n_raw;         // compact output: triangles that went to the slab's raw area (sdf_slab.h)
    // written by k_compact (NOT cleared between meshing retries): the surviving-batch work list
    // and this shard's slice of it, so k_mesh can start without a host round trip
    int nwork, work_begin, work_end;
    int not_mesh2;                    // k_cull: some tile of this shard is not k_mesh2's (left dense, or more than MESH2_NTL_MAX listed tasks)
};
enum { MESH_COUNTERS_RESET_BYTES = 112 };   // the part of MeshCounters cleared before every k_mesh run
// overflow bits: 1 the soup (or an arena) was too small, 2 a look-back timed out, 16 k_mesh2 met a tile it does not hold (the call is
// repeated with k_mesh; 4 and 8 are the slab header's own: items, raw area -- k_pack_slab)
enum { MESH_OVERFLOW_NOT_MESH2 = 16 };
// k_mesh2 (sdf_mesh2.h) takes tiles of at most this many listed tasks (their samples: 48 KB of its 64 KB region)
enum { MESH2_NTL_MAX = 192 };

struct GridDesc {
    const double *X, *Y, *Z;   // device copies of the np.arange axes
    int nx, ny, nz;
    int bs;                    // batch size (cells per axis), samples per axis = bs + 1
    int nbx, nby, nbz;         // batches per axis
};

struct MeshArgs {
    GridDesc g;
    const McTables *mc;
    const int *worklist;           // its length and this shard's slice are in ctr (device side)
    unsigned char *kinds;          // per batch
    unsigned long long *status;    // per work item: look-back word (flag << 62 | triangles), zeroed per run
    double *out;                   // the ordered soup: 9 doubles per triangle, world coordinates
    unsigned long long out_cap;    // triangles
    MeshCounters *ctr;
    int bits_off;                  // byte offset of the sign-bit volume in dynamic LDS
    int list_off;                  // byte offset of the triangle work list in dynamic LDS
    int list_cap;                  // its capacity in entries
    unsigned long long *prof;      // NULL, or 8 phase cycle counters (SDF_MESH_PROF=1 diagnostics)
    int tape_stride;               // 0: `code` is the model's tape; else `code` holds one pruned tape per BATCH (interval
                                   // prepass, sdf_prune.h), `tape_stride` 64-bit words apart, the last word = its length
    int n_instr;                   // instructions of the model's tape (statistics)
    float *park;                   // staging: one slot of park_cap triangles (9 floats each) per workgroup, or NULL
    int park_cap;
    unsigned park_spins;           // polls of the predecessors' counts before a batch is parked
    const unsigned char *cull;     // NULL, or k_cull's records: per work item, the sampling tasks to evaluate (cull_tasks)
    int slot_bytes;                // 0, or the size of one of the two slots of sparse tiles in the dense tile's region (deferred emission, k_mesh)
    int stage_off;                 // byte offset of the transposition area behind the slots (MESH_STAGE_BYTES per wave, up to the end of LDS)
    // compact output (multi-GPU exchange, sdf_generate_compact_async): `out` then holds 9 FLOAT32 per triangle in the
    // batch's local voxel coordinates (what marching cubes itself produces, 36 bytes instead of 72) and xf[] the
    // per-work-item transform (offset[3], scale[3], indexed by w - work_begin) that k_expand applies after the gather
    // (since r04 a triangle of the compact form is a 16-byte record, sdf_slab.h Tri16; the few that do not have the shape
    // go to the slab's raw area: `raw`, nine floats each, handed out by the counter n_raw)
    int compact;
    double *xf;
    int xf_cap;
    float *raw;
    long long raw_cap;
    // two-pass meshing (k_mesh = sample + classify, k_scan_items, k_emit2 = triangles): per work item a descriptor,
    // per surface cell a 36-byte record (cell, configuration, its 8 corner samples), per triangle a 4-byte entry
    // (record << 4 | triangle of the cell) -- handed out from two arenas by atomic cursors, in any order
    int twopass;
    struct ItemDesc *desc;
    unsigned *cells;                    // 9 dwords per record
    unsigned *tlist;
    unsigned long long cells_cap, tlist_cap;
    const int *block_item;              // k_scan_items' index for k_emit2: the work item of triangle 256 b (or NULL)
    // The LAST `tail` work items of the shard are handed out by descending cost instead of by position: order[i] =
    // k_cull's estimate for the i-th item of the tail (listed tasks x instructions); the workgroup that takes the
    // r-th of them looks for the item of rank r.  The kernel ends when its slowest workgroup does, and a workgroup's
    // last item decides how far behind the others it ends; with the expensive items first the stragglers get the
    // cheap ones.  NULL / 0: in order.  tail <= workgroups - 1 keeps the look-back free of deadlock whatever the
    // items do (see k_mesh).
    const int *order;
    int tail;
};
enum { MESH_TAIL_MAX = 255 };

struct ItemDesc {
    unsigned ntri, ncells;
    unsigned long long list_off, cell_off;   // ~0: the arenas were full (flagged: the call is repeated with larger ones)
    double xf[6];                            // offset[3], scale[3] of `points * scale + offset` (reference sdf/core.py:58-60)
};

// dynamic LDS layout of k_mesh
// batches a workgroup may have parked before it has to wait for the oldest one's place.  Four were enough for even
// batches; where the sampling time of a batch varies by an order of magnitude (weave at 2^33: a 244-instruction tape
// of which the interval prepass leaves 10 % here and 60 % there) a workgroup with four full slots spent HALF of the
// kernel waiting for a slow predecessor of its oldest parked batch (SDF_MESH_PROF: placing 14.4 of 29.6 G cycles):
// depth 4: 48.3 ms, 8: 33.6, 16: 27.7, 32: 27.1 ms; the example, gearlike, blobby, pawn do not care, knurling gains 4 %.
enum { MESH_PARK_DEPTH = 16 };
// ([0, 64) and [128, 192): the two buffers the block scans of the count phase alternate between; [64, 128): bcast)
enum { MESH_LDS_SUMS2 = 128, MESH_LDS_NTRI = 192, MESH_LDS_AXES = MESH_LDS_NTRI + 256, MESH_LDS_PEND = MESH_LDS_AXES + 800, MESH_LDS_TRI = MESH_LDS_PEND + 64 * MESH_PARK_DEPTH,
       MESH_LDS_VOL = MESH_LDS_TRI + 256 * 5 * 2 };   // PEND: per parked batch 6 doubles + 2 ints; TRI: the triangle table, one 16-bit word per (configuration, triangle)

__device__ __forceinline__ void batch_origin(const GridDesc &g, int b, int &ox, int &oy, int &oz, int &lx, int &ly, int &lz) {
    // itertools.product(Xs, Ys, Zs): Z fastest (reference sdf/core.py:119)
    const int ibz = b % g.nbz, iby = (b / g.nbz) % g.nby, ibx = b / (g.nbz * g.nby);
    ox = ibx * g.bs; oy = iby * g.bs; oz = ibz * g.bs;
    lx = min(g.bs + 1, g.nx - ox); ly = min(g.bs + 1, g.ny - oy); lz = min(g.bs + 1, g.nz - oz);
}

// i / d for 0 <= i < 2^16, d >= 1, through the float pipe (exact: (i + 0.5) / d is never closer
// than 0.5 / d to an integer, far above the float rounding error of the product)
__device__ __forceinline__ int fast_div(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

template <int BLOCK>
__device__ __forceinline__ int block_exclusive_scan(int v, int *wave_sums, int &total) {
    // wave64 inclusive scan in the vector ALU (data-parallel primitives: four shifts inside the rows of 16 lanes, then
    // the row totals across), then the wave totals through LDS.  (Six `__shfl_up` steps -- ds_bpermute, an LDS round
    // trip each -- used to make a scan cost 1.6 k cycles of a 256-thread workgroup; a lane that has no source, or whose
    // row is masked out, adds the 0 of `old`.)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);   // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);   // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);   // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);   // row_shr:8
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { const int s = wave_sums[w]; if (w < wid) base += s; tot += s; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

// the same with ONE barrier: the caller alternates between two buffers from scan to scan (a thread that writes buffer A for scan i + 2 has
// passed the barrier of scan i + 1, which every thread reaches only after its reads of scan i)
template <int BLOCK>
__device__ __forceinline__ int block_exclusive_scan1(int v, int *wave_sums, int &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { const int s = wave_sums[w]; if (w < wid) base += s; tot += s; }
    total = tot;
    return base + inc - v;
}

// the same for a predicate: a ballot and a masked bit count instead of six shuffles
template <int BLOCK>
__device__ __forceinline__ int block_exclusive_count(bool p, int *wave_sums, int &total) {
    const unsigned long long m = __ballot(p);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) wave_sums[wid] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { const int s = wave_sums[w]; if (w < wid) base += s; tot += s; }
    total = tot;
    __syncthreads();
    return base + below;
}

// sign bits of the four samples (o0,o1) in {0,1}^2 of one i2-plane, bit (2*o0+o1) set when > 0
__device__ __forceinline__ unsigned plane_bits(const float *v, int s0, int s1) {
    return (v[0] > 0.0f ? 1u : 0u) | (v[s1] > 0.0f ? 2u : 0u) | (v[s0] > 0.0f ? 4u : 0u) | (v[s0 + s1] > 0.0f ? 8u : 0u);
}
// plane bit j -> configuration bit 2j (o2 = 0) ; shift left by one for o2 = 1
__device__ __forceinline__ unsigned spread4(unsigned s) { return (s & 1u) | ((s & 2u) << 1) | ((s & 4u) << 2) | ((s & 8u) << 3); }

// sign configuration of the cell at column i2 of a cell row from the four row bit strings
// (q = 2 * o0 + o1): bit c = 4 * o0 + 2 * o1 + o2 of the configuration = bit (i2 + o2) of row q
__device__ __forceinline__ unsigned cell_config(const unsigned long long *rb, int i2) {
    unsigned cfg = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) cfg |= ((unsigned)(rb[q] >> i2) & 3u) << (2 * q);
    // rows give bit pairs (o2 = 0 in bit 0, o2 = 1 in bit 1) at position 2q: already c = 2q + o2
    return cfg;
}

#ifndef SDF_TAPE_WARM
#define SDF_TAPE_WARM 1
#endif
#ifndef SDF_FAST_VERTEX
#define SDF_FAST_VERTEX 1
#endif
// The coordinate of a marching-cubes vertex along its edge: samples vlo (at grid index ibase) and vhi (at ibase + 1).
// skimage's placement: with w = 1/(eps+|v|), t = w_hi / (w_lo + w_hi), evaluated in float64 on the
// float32 samples, stored as float32.
__device__ __forceinline__ float mc_edge_pos(double vlo, double vhi, double ibase) {
    const double eps = 2.220446049250313e-16;
    const double a = eps + fabs(vlo), b = eps + fabs(vhi);
    float pf;
#if SDF_FAST_VERTEX
    // skimage's t = whi / (wlo + whi) with wlo = 1 / a, whi = 1 / b is a / (a + b) up to its three roundings (<= 5e-16
    // absolute, t <= 1), and only float32(i + t) is kept.  ONE division by reciprocal + Newton steps instead of three
    // correctly rounded ones (33 of the ~115 instructions of a vertex); the candidate q = i + t' is within 1e-14 of the
    // reference's float64 sum, so wherever float32(q - D) == float32(q + D) for D = 2^-44 = 5.7e-14 -- rounding is
    // monotone -- that float IS the reference's.  Otherwise (a float32 rounding boundary inside the 1e-13 window:
    // ~1e-6 of the vertices; NaN / infinite samples) the three divisions run.  Bit-identical by construction.
    const double sab = a + b;
    double r = __builtin_amdgcn_rcp(sab);
    r = fma(fma(-sab, r, 1.0), r, r);
    r = fma(fma(-sab, r, 1.0), r, r);
    double tq = a * r;
    tq = fma(fma(-sab, tq, a), r, tq);
    const double q = ibase + tq;
    const float f_lo = (float)(q - 0x1p-44), f_hi = (float)(q + 0x1p-44);
    pf = f_lo;
    if (__builtin_expect(!(f_lo == f_hi), 0))
#endif
    {
        const double wlo = 1.0 / a, whi = 1.0 / b;
        const double t = whi / (wlo + whi);
        pf = (float)(ibase + t);
    }
    return pf;
}
// One marching-cubes vertex on edge e of the cell at (i0,i1,i2); v points at the cell's corner 0
// in a volume with strides (s0, s1, 1).
__device__ __forceinline__ void mc_vertex(const float *v, int s0, int s1, int i0, int i1, int i2, int e, float *o) {
    const int axis = e >> 2, oa = (e >> 1) & 1, ob = e & 1;
    int o0, o1, o2, stride;
    if (axis == 0) { o0 = 0; o1 = oa; o2 = ob; stride = s0; }
    else if (axis == 1) { o0 = oa; o1 = 0; o2 = ob; stride = s1; }
    else { o0 = oa; o1 = ob; o2 = 0; stride = 1; }
    const int base = o0 * s0 + o1 * s1 + o2;
    const float pf = mc_edge_pos((double)v[base], (double)v[base + stride], (double)(axis == 0 ? i0 : (axis == 1 ? i1 : i2)));
    o[0] = axis == 0 ? pf : (float)(i0 + o0); o[1] = axis == 1 ? pf : (float)(i1 + o1); o[2] = axis == 2 ? pf : (float)(i2 + o2);
}

// ---- the tile of one batch as the marching phases of k_mesh see it -----------------------------
// dense:  every sample of the (<= 33)^3 tile, x-major (what k_mesh sampled until r03).
// sparse: ONLY the samples of the units k_cull listed (cull_tasks: units of 2^3 samples, the rest of the tile matters by
//         its sign bits alone), eight floats per listed unit in the order of the list; `colinfo` (k_cull's column words:
//         listed u2 of column (u0, u1) | index of the column's first listed unit << 17) finds a sample's unit.  A tile
//         of the 512^3 example lists ~ 450 of its 4913 units: 14 KB instead of 144 KB, which is what lets TWO tiles live
//         in the CU's LDS -- the batch being sampled and the previous one, whose triangles are written one batch later,
//         when its predecessors have long published their counts (k_mesh: deferred emission).
// Marching cubes reads values only at the corners of cells with a sign change; those lie in undecided sub-groups, whose
// samples all belong to listed units (cull_tasks) -- `at` is never asked for a sample the sparse form does not hold.
struct TileView {
    const float *smp;
    const unsigned *colinfo;
    int lyz, lz;
    bool sparse;                 // (workgroup-uniform)
    __device__ __forceinline__ float at(int ix, int iy, int iz) const {
        if (!sparse) return smp[ix * lyz + iy * lz + iz];
        const unsigned c = colinfo[(ix >> 1) * 17 + (iy >> 1)];
        const int k = (int)(c >> 17) + __popc(c & ((1u << (iz >> 1)) - 1u));
        return smp[8 * k + ((ix & 1) << 2) + ((iy & 1) << 1) + (iz & 1)];
    }
    // the 8 corner samples of cell (i0, i1, i2) as a 2 x 2 x 2 volume (strides 4, 2, 1)
    __device__ __forceinline__ void cell(int i0, int i1, int i2, float *c8) const {
#pragma unroll
        for (int q = 0; q < 8; q++) c8[q] = at(i0 + (q >> 2), i1 + ((q >> 1) & 1), i2 + (q & 1));
    }
};
__device__ __forceinline__ void mc_vertex_view(const TileView &vw, int i0, int i1, int i2, int e, float *o) {
    const int axis = e >> 2, oa = (e >> 1) & 1, ob = e & 1;
    const int o0 = axis == 0 ? 0 : oa, o1 = axis == 0 ? oa : (axis == 1 ? 0 : ob), o2 = axis == 2 ? 0 : ob;
    const int x = i0 + o0, y = i1 + o1, z = i2 + o2;
    const float vlo = vw.at(x, y, z), vhi = vw.at(x + (axis == 0 ? 1 : 0), y + (axis == 1 ? 1 : 0), z + (axis == 2 ? 1 : 0));
    const float pf = mc_edge_pos((double)vlo, (double)vhi, (double)(axis == 0 ? i0 : (axis == 1 ? i1 : i2)));
    o[0] = axis == 0 ? pf : (float)x; o[1] = axis == 1 ? pf : (float)y; o[2] = axis == 2 ? pf : (float)z;
}

// compact output: triangle number `pos` of the shard's slab from its nine local floats
struct Tri16Sink {                 // where the records go: the slab's record area, its raw area and the raw area's counter
    double *out;
    float *raw;
    long long raw_cap;
    unsigned long long *n_raw;
};
__device__ __forceinline__ void store_tri16(const Tri16Sink &a, unsigned long long pos, const float *o) {
    Tri16 r;
    if (__builtin_expect(!slab_encode16(o, r), 0)) {
        const unsigned long long idx = atomicAdd(a.n_raw, 1ull);
        if (idx < (unsigned long long)a.raw_cap) {
            float *dst = a.raw + idx * 9ull;
            for (int q = 0; q < 9; q++) dst[q] = o[q];
        }
        r.code = TRI16_RAW; r.f[0] = __uint_as_float((unsigned)idx); r.f[1] = 0.0f; r.f[2] = 0.0f;
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<u4 *>(a.out)[pos] = u4{r.code, __float_as_uint(r.f[0]), __float_as_uint(r.f[1]), __float_as_uint(r.f[2])};
}
__device__ __forceinline__ void store_tri16(const MeshArgs &a, unsigned long long pos, const float *o) {
    store_tri16(Tri16Sink{a.out, a.raw, a.raw_cap, &a.ctr->n_raw}, pos, o);
}

// ---- ordered allocation: exclusive prefix of the triangle counts over the work list -----------
// status word of work item w: flag << 62 | value; flag 0 = nothing yet, 1 = value is the item's own
// count ("aggregate"), 2 = value is the inclusive prefix up to and including w.  Decoupled look-back
// (Merrill & Garland): publish the aggregate, then walk back 64 predecessors at a time, summing
// aggregates until an inclusive prefix is met; publish the own inclusive prefix.  Work items are
// taken from the counter in order, so every predecessor is held by a RUNNING workgroup: the walk
// never waits on a workgroup that is itself waiting for a CU.  Words are exchanged with agent-scope
// atomics (per-CU L1 and per-XCD L2 are not coherent for plain accesses); every spin is bounded.
// Called by wave 0 of the workgroup (all 64 lanes).
//
// A workgroup whose predecessors are still sampling does not wait for them: it writes the batch's
// triangles in compact form (9 floats each) into its own staging slot ("parks" the batch), goes on
// with the next batch, and moves the parked triangles to their place once that batch is sampled
// too (k_mesh).  Only the aggregate has to be published early; the walk can happen any time later.
// a workgroup-uniform value the compiler cannot see as one (it came through LDS): back into scalar registers
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
#define MESH_NOT_READY (~0ull - 1ull)
#define MESH_FLAG_AGG (1ull << 62)
#define MESH_FLAG_PFX (2ull << 62)
#define MESH_VAL_MASK ((1ull << 62) - 1)
__device__ __forceinline__ unsigned wave_sum_u32_dpp(unsigned x) {   // (every lane's x < 2^26: the sum of 64 fits)
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    // three limbs of 21 bits through the vector ALU's row shifts (v < 2^62: triangle counts) -- six `__shfl_xor` steps on both halves were
    // twelve ds_bpermute round trips on the look-back's critical path, with fifteen waves waiting at the barrier behind it (r05ab)
    const unsigned long long a = wave_sum_u32_dpp((unsigned)(v & 0x1FFFFFull)), b = wave_sum_u32_dpp((unsigned)((v >> 21) & 0x1FFFFFull)),
                             c = wave_sum_u32_dpp((unsigned)((v >> 42) & 0xFFFFFull));
    return a + (b << 21) + (c << 42);
}
__device__ __forceinline__ void publish_count(unsigned long long *status, int w, int w_begin, unsigned long long total) {
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_store(&status[w], (w == w_begin ? MESH_FLAG_PFX : MESH_FLAG_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// this lane's word of the first look-back window of work item w (the 64 items in front of it).  The
// caller issues it EARLY -- before the cell counting, whose work then hides the round trip to the
// device-coherent level -- and hands it to ordered_base afterwards.
__device__ __forceinline__ unsigned long long lookback_prefetch(const unsigned long long *status, int w, int w_begin) {
    const int j = w - 1 - (int)(threadIdx.x & 63);
    return j >= w_begin ? __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : MESH_FLAG_PFX;
}
// the exclusive prefix of work item w; ~0 on timeout; with max_spins small: MESH_NOT_READY when a
// predecessor has not published its count yet.  `first` = lookback_prefetch(w), possibly stale
// (a stale word can only say "not ready yet").
// Behind the prefetched window the walk takes FOUR windows (256 predecessors) per round trip: with deferred emission a
// batch publishes its inclusive prefix a whole batch after its count, so ~ two items per workgroup -- 500 words -- carry
// only a count at any time, and one window per round trip made the walk eight dependent trips to the coherent level
// (6 k cycles per batch).  The lanes add up what they see; ONE wave reduction at the end.
__device__ __forceinline__ unsigned long long ordered_base(unsigned long long *status, int w, int w_begin, unsigned long long total,
                                                           unsigned max_spins, unsigned long long first) {
    const int lane = threadIdx.x & 63;
    if (w == w_begin) return 0;
    unsigned long long acc = 0;   // this lane's share of the exclusive prefix
    int idx = w - 1;
    bool use_first = true;
    for (unsigned spins = 0; spins < max_spins;) {   // (only waiting counts as a spin, walking back does not)
        constexpr int NW = 4;
        unsigned long long sw[NW];
        const int nwin = use_first ? 1 : NW;
        if (use_first) sw[0] = first;
        else {
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int j = idx - 64 * k - lane;
                sw[k] = j >= w_begin ? __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : MESH_FLAG_PFX;   // in front of the shard: prefix 0
            }
        }
        use_first = false;
        bool wait = false;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            if (k >= nwin || wait) break;   // (uniform)
            const unsigned flag = (unsigned)(sw[k] >> 62);
            const unsigned long long pending = __ballot(flag == 0), is_pfx = __ballot(flag == 2);
            if (is_pfx) {
                const int p = __ffsll((long long)is_pfx) - 1;                 // nearest predecessor with a prefix
                if (pending & ((1ull << p) - 1ull)) { wait = true; break; }
                acc += lane <= p ? (sw[k] & MESH_VAL_MASK) : 0ull;
                const unsigned long long excl = wave_sum_u64(acc);
                if (lane == 0) __hip_atomic_store(&status[w], MESH_FLAG_PFX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return excl;
            }
            if (pending) { wait = true; break; }
            acc += sw[k] & MESH_VAL_MASK;
            idx -= 64;
        }
        if (wait) { __builtin_amdgcn_s_sleep(1); spins++; }
    }
    return max_spins < (1u << 24) ? MESH_NOT_READY : ~0ull;
}
#define MESH_SPIN_FOREVER (1u << 24)   // (bounded all the same: a timeout is reported as an error)

// triangle j of an ambiguous cell: re-runs the selection (cheaper than carrying the tiling through
// the LDS work list for the few cells concerned), applies skimage's face flip for
// gradient_direction='descent' and places the three vertices
__device__ __noinline__ void mc33_triangle(const float *corner, int s0, int s1, int i0, int i1, int i2,
                                           const signed char *tab, int j, float *o) {
    double lv[8];
    int off;
    mc33_load_cell(corner, s0, s1, lv);
    const int n = mc33_cell(lv, tab, &off);
    if (j >= n) { for (int q = 0; q < 9; q++) o[q] = 0.0f; return; }
    const signed char lew_edge[12] = {8, 5, 9, 4, 10, 7, 11, 6, 0, 1, 3, 2};   // MC33_EDGE
    for (int q = 0; q < 3; q++) {
        const int e = tab[off + 3 * j + 2 - q];
        if (e == 12) mc33_centre_vertex(lv, i0, i1, i2, o + 3 * q);
        else mc_vertex(corner, s0, s1, i0, i1, i2, lew_edge[e], o + 3 * q);
    }
}

// ---- sampling tasks --------------------------------------------------------------------------
// The samples of a tile are evaluated in TASKS of 64 (one per wave and sample slot).  A full 33^3
// tile is cut into the 8^3 cubes of 4^3 samples plus its three far faces (563 tasks, dense); any
// other tile into runs of 64 consecutive samples.  Cubes keep a task's samples close together in all
// three directions, which is what lets whole tasks drop out in cull_tasks.
struct TileTasks {
    int lx, ly, lz, lyz, nvox;
    float inv_lyz, inv_lz;
    bool regular;
    int ntask;
    __device__ __forceinline__ TileTasks(int lx_, int ly_, int lz_) : lx(lx_), ly(ly_), lz(lz_) {
        lyz = ly * lz; nvox = lx * lyz;
        inv_lyz = 1.0f / (float)lyz; inv_lz = 1.0f / (float)lz;
        regular = lx == 33 && ly == 33 && lz == 33;
        ntask = regular ? 563 : (nvox + 63) >> 6;
    }
    // sample `lane` of a task; false: the task has no such sample (ix, iy, iz are valid indices all the same)
    __device__ __forceinline__ bool sample(int task, int lane, int &ix, int &iy, int &iz) const {
        if (!regular) {
            const int i = min(task * 64 + lane, nvox - 1);
            ix = fast_div(i, inv_lyz); const int r = i - ix * lyz; iy = fast_div(r, inv_lz); iz = r - iy * lz;
            return task * 64 + lane < nvox;
        }
        if (task < 512) { ix = 4 * (task >> 6) + (lane >> 4); iy = 4 * ((task >> 3) & 7) + ((lane >> 2) & 3); iz = 4 * (task & 7) + (lane & 3); return true; }
        if (task < 530) { const int p = min((task - 512) * 64 + lane, 1088); ix = 32; iy = fast_div(p, 1.0f / 33.0f); iz = p - 33 * iy; return (task - 512) * 64 + lane < 1089; }
        if (task < 547) { const int p = min((task - 530) * 64 + lane, 1055); iy = 32; ix = fast_div(p, 1.0f / 33.0f); iz = p - 33 * ix; return (task - 530) * 64 + lane < 1056; }
        const int p = (task - 547) * 64 + lane; iz = 32; ix = p >> 5; iy = p & 31; return true;
    }
};

// Where can the surface not be?  Three levels of interval arithmetic (sdf_interval.h) over boxes of the tile's cells: the
// (up to) 4^3 boxes of 8^3 cells, the groups of 4^3 cells inside the boxes that could not be decided, the SUB-GROUPS of 2^3
// cells inside the groups that could not be decided; one thread per box.  A sub-group whose interval excludes zero has no
// surface cell and its samples only matter by their sign: a sample is evaluated iff it belongs to an undecided sub-group
// (a sample belongs to up to 8 of them: along an axis, sub-groups (i - 1) / 2 and i / 2).  Results are the same bit for
// bit: marching cubes reads values only at the corners of cells with a sign change, and every such cell lies in an
// undecided sub-group.  (Two decided neighbours share a face of samples, so they cannot carry opposite signs.)  The
// bound 1e-30 keeps the sign through the cast to float32.
//
// What is evaluated is listed in UNITS of 2^3 samples [2u, 2u + 1] per axis (u = 0 .. 16; unit 16 is the lone sample 32 of
// a full axis), eight units to a task of 64 lanes: a unit is listed iff one of the sub-groups {u - 1, u}^3 is undecided.
// (Until r03 the last level were the groups of 4^3 cells and a task a cube of 4^3 samples: tools/cull3study.py -- the
// interpreter saw 21 - 28 % of the surviving batches' samples; with this level 10 - 14 %.  Every tile shape goes the same way:
// the units are clipped to the tile.)
//
// Writes the record of the batch -- u16 number of listed units (0xFFFF: not culled), the units (u16 each: u0 << 10 | u1 << 5 |
// u2, ascending, padded with 0xFFFF to whole tasks), the sub-group states -- and returns the number of TASKS, or -1 when the tile is not culled
// (degenerate tile, the interval state of the tape does not fit in LDS, or so many units that listing them saves nothing).
// Listing more units than necessary is harmless, missing one is not.  This is the body of k_cull (sdf_hip.hip), a kernel
// of its own in front of k_mesh: inside k_mesh the interval pass would run on half the waves of a workgroup that holds a
// whole CU, and its registers would compete with the interpreter's.
// record / scratch: [0] u16 unit count | CULL_ULIST: units | CULL_SSTATE: 16^3 sub-group states (0 unknown, 1 positive,
// 2 negative), TWO BITS each: word h0 * 16 + h1 holds the 16 states along h2 | CULL_COLINFO || scratch only, from CULL_RECORD:
// box states (64 B), group states (512 B), the groups to evaluate (512 u16); the sub-group states as BYTES while the levels
// run lie over the unit list (CULL_SBYTES: packed into the record before the units are listed)
// CULL_COLINFO: per COLUMN (u0, u1) of units one word, `listed u2 (17 bits) | index of the column's first listed unit << 17`:
// with it k_mesh finds a sample of a listed unit in a tile that stores ONLY the listed units (TileView, sparse form)
enum { CULL_UNIT_CAP = 3072, CULL_ULIST = 8, CULL_SBYTES = CULL_ULIST, CULL_SSTATE = CULL_ULIST + 2 * CULL_UNIT_CAP, CULL_COLINFO = CULL_SSTATE + 1024,
       CULL_RECORD = CULL_COLINFO + 1160,
       CULL_MSTATE = CULL_RECORD, CULL_GSTATE = CULL_MSTATE + 64, CULL_ELIST = CULL_GSTATE + 512, CULL_PACC = CULL_ELIST + 1024,
       CULL_SCRATCH = CULL_PACC + 64 };
static_assert(CULL_RECORD % 8 == 0 && CULL_SSTATE % 8 == 0 && CULL_COLINFO % 8 == 0, "the record is copied in words; its state rows are read as u64");
// sample `lane` of task `task` of a culled tile (units: the record's list); false: no such sample (ix, iy, iz are valid
// indices all the same)
__device__ __forceinline__ bool cull_sample(const unsigned short *units, int task, int lane, int lx, int ly, int lz, int &ix, int &iy, int &iz) {
    const unsigned u = units[8 * task + (lane >> 3)];                    // u0 << 10 | u1 << 5 | u2; the padding 0xFFFF lies outside every tile
    const int x = (int)((u >> 9) & 62u) + ((lane >> 2) & 1), y = (int)((u >> 4) & 62u) + ((lane >> 1) & 1), z = (int)((u << 1) & 62u) + (lane & 1);
    const bool ok = x < lx && y < ly && z < lz;
    ix = ok ? x : 0; iy = ok ? y : 0; iz = ok ? z : 0;
    return ok;
}
template <int BLOCK, bool FULL, bool RARE>
__device__ __forceinline__ int cull_tasks(const uint32_t *wcode, const double *consts, int n_instr_w, int lx, int ly, int lz,
                                       const double *axes, double *ia_state, int ia_bytes, unsigned char *scratch, int *wave_sums,
                                       int ia_np, int ia_nd, unsigned long long *prof = nullptr, int levels = 3) {
    const int tid = threadIdx.x;
    // SDF_MESH_PROF: cycles of thread 0 per phase, collected in LDS (12 words behind the group list) and added to the
    // global counters by the caller when the workgroup is done (an atomic per phase would stall the phases it measures:
    // the workgroups of a launch run in step)
    long long tprev = prof ? clock64() : 0;
    unsigned *pacc = reinterpret_cast<unsigned *>(scratch + CULL_PACC);
    if (prof && tid == 0) for (int k = 0; k < 12; k++) pacc[k] = 0;
#define SDF_CULL_PROF(K) do { if (prof && tid == 0) { const long long tn = clock64(); pacc[(K) - 16] += (unsigned)(tn - tprev); tprev = tn; } } while (0)
    const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
    const int per_pass = min(BLOCK, ia_bytes / ((6 * ia_np + 2 * ia_nd) * 8)) & ~63;
    if (c0 <= 0 || c1 <= 0 || c2 <= 0 || per_pass < 64) return -1;
    unsigned short *ulist = reinterpret_cast<unsigned short *>(scratch + CULL_ULIST);
    unsigned char *sstate = scratch + CULL_SBYTES;   // per sub-group of 2^3 cells, [h0][h1][h2], 16 per axis (a byte each, over the unit list)
    static_assert(2 * CULL_UNIT_CAP >= 4096, "the byte states lie over the unit list");
    unsigned char *mstate = scratch + CULL_MSTATE;   // per box of 8^3 cells
    unsigned char *gstate = scratch + CULL_GSTATE;   // per group of 4^3 cells
    unsigned short *elist = reinterpret_cast<unsigned short *>(scratch + CULL_ELIST);   // up to 512 groups: to evaluate, then the undecided ones
    // the interval of the model over the box of cells [x0, x1) x [y0, y1) x [z0, z1) (clipped to the tile) -> 0 / 1 / 2
    auto box_state = [&](int x0, int x1, int y0, int y1, int z0, int z1) -> unsigned char {
        Ival bx{axes[x0], axes[min(x1, c0)]}, by{axes[33 + y0], axes[33 + min(y1, c1)]}, bz{axes[66 + z0], axes[66 + min(z1, c2)]};
        if (bx.lo > bx.hi) { const double t = bx.lo; bx.lo = bx.hi; bx.hi = t; }     // (a descending axis)
        if (by.lo > by.hi) { const double t = by.lo; by.lo = by.hi; by.hi = t; }
        if (bz.lo > bz.hi) { const double t = bz.lo; bz.lo = bz.hi; bz.hi = t; }
        if (ia::bad(bx) || ia::bad(by) || ia::bad(bz)) { bx = by = bz = ia::top(); }
        const Ival v = ia_run_tape<false, FULL, RARE>(wcode, consts, nullptr, nullptr, n_instr_w, bx, by, bz, true,
                                          IaShared{ia_state, ia_np, per_pass}, ia_nd, nullptr);
        return v.lo > 1e-30 ? 1 : (v.hi < -1e-30 ? 2 : 0);
    };
    // ONE loop over the three levels (phase 0: the boxes, one wave; 1: the groups inside undecided boxes; 2: the sub-groups
    // inside undecided groups; per_pass boxes at a time) so that the interval interpreter is instantiated once per kernel.
    int phase = 0, e0 = 0, nev = 0;
    for (;;) {
        if (phase == 1 && e0 >= nev) {                                                 // (uniform) groups done: on to the sub-groups
            __syncthreads();
            SDF_CULL_PROF(19);
            int nund = 0;
            for (int g0 = 0; levels >= 3 && g0 < 512; g0 += BLOCK) {   // the undecided groups, compacted (elist is free: every listed group has been evaluated)
                const int gi = g0 + tid;
                const bool und = gi < 512 && gstate[gi] == 0;
                int n;
                const int pos = nund + block_exclusive_count<BLOCK>(und, wave_sums, n);
                if (und) elist[pos] = (unsigned short)gi;
                nund += n;
            }
            for (int h = tid; h < 4096; h += BLOCK) {   // a sub-group inherits the state of its group; one outside the tile counts as decided
                const int h0 = h >> 8, h1 = (h >> 4) & 15, h2 = h & 15;
                const bool exists = 2 * h0 < c0 && 2 * h1 < c1 && 2 * h2 < c2;
                sstate[h] = exists ? gstate[((h0 >> 1) * 8 + (h1 >> 1)) * 8 + (h2 >> 1)] : 1;
            }
            __syncthreads();
            phase = 2; e0 = 0; nev = 8 * nund;   // (two levels: none -- the sub-groups keep their groups' states)
        }
        if (phase == 2 && e0 >= nev) break;                                            // (uniform)
        const bool run = phase == 0 ? tid < 64 : (tid < per_pass && e0 + (tid & ~63) < nev);   // (whole waves)
        int target = 0, x0, y0, z0, ext;
        bool live;
        if (phase == 0) {
            const int m0 = tid >> 4, m1 = (tid >> 2) & 3, m2 = tid & 3;
            live = 8 * m0 < c0 && 8 * m1 < c1 && 8 * m2 < c2;
            x0 = live ? 8 * m0 : 0; y0 = live ? 8 * m1 : 0; z0 = live ? 8 * m2 : 0; ext = 8;
        } else if (phase == 1) {
            live = e0 + tid < nev;
            target = run ? (int)elist[min(e0 + tid, nev - 1)] : 0;
            x0 = 4 * (target >> 6); y0 = 4 * ((target >> 3) & 7); z0 = 4 * (target & 7); ext = 4;
        } else {
            const int e = min(e0 + tid, nev - 1);
            const int gq = run ? (int)elist[e >> 3] : 0, ch = e & 7;
            const int h0 = 2 * (gq >> 6) + (ch >> 2), h1 = 2 * ((gq >> 3) & 7) + ((ch >> 1) & 1), h2 = 2 * (gq & 7) + (ch & 1);
            live = e0 + tid < nev && 2 * h0 < c0 && 2 * h1 < c1 && 2 * h2 < c2;
            target = (h0 * 16 + h1) * 16 + h2;
            x0 = live ? 2 * h0 : 0; y0 = live ? 2 * h1 : 0; z0 = live ? 2 * h2 : 0; ext = 2;
        }
        if (run) {
            const unsigned char st = box_state(x0, x0 + ext, y0, y0 + ext, z0, z0 + ext);
            if (phase == 0) mstate[tid & 63] = live ? st : 1;
            else if (live) (phase == 1 ? gstate : sstate)[target] = st;
        }
        if (phase != 0) { if (prof && tid == 0) pacc[7]++; e0 += per_pass; continue; }
        SDF_CULL_PROF(17);
        __syncthreads();
        for (int g0 = 0; g0 < 512; g0 += BLOCK) {   // a group inherits the state of its box; the undecided ones are listed
            const int gi = g0 + tid;
            const int a0 = gi >> 6, a1 = (gi >> 3) & 7, a2 = gi & 7;
            const bool exists = gi < 512 && 4 * a0 < c0 && 4 * a1 < c1 && 4 * a2 < c2;
            const unsigned char ms = exists ? mstate[((a0 >> 1) * 4 + (a1 >> 1)) * 4 + (a2 >> 1)] : 1;
            if (gi < 512) gstate[gi] = ms;
            int n;
            const int pos = nev + block_exclusive_count<BLOCK>(ms == 0, wave_sums, n);
            if (ms == 0) elist[pos] = (unsigned short)gi;
            nev += n;
        }
        __syncthreads();
        SDF_CULL_PROF(18);
        if (prof && tid == 0) pacc[6] = (unsigned)nev;
        phase = 1; e0 = 0;
    }
    __syncthreads();
    SDF_CULL_PROF(20);
    // the undecided sub-groups as bit rows: ub[h0 * 16 + h1] has bit h2 set (over the group states and the group list, done with)
    unsigned short *ub = reinterpret_cast<unsigned short *>(scratch + CULL_GSTATE);    // 256 u16 = 512 B
    for (int r = tid; r < 256; r += BLOCK) {
        const unsigned long long lo8 = *reinterpret_cast<const unsigned long long *>(sstate + 16 * r);
        const unsigned long long hi8 = *reinterpret_cast<const unsigned long long *>(sstate + 16 * r + 8);
        unsigned m = 0, pk = 0;
        SDF_UNROLL for (int k = 0; k < 8; k++) { m |= ((lo8 >> (8 * k)) & 255ull) == 0ull ? 1u << k : 0u; pk |= (unsigned)((lo8 >> (8 * k)) & 3ull) << (2 * k); }
        SDF_UNROLL for (int k = 0; k < 8; k++) { m |= ((hi8 >> (8 * k)) & 255ull) == 0ull ? 256u << k : 0u; pk |= (unsigned)((hi8 >> (8 * k)) & 3ull) << (2 * k + 16); }
        ub[r] = (unsigned short)m;
        reinterpret_cast<unsigned *>(scratch + CULL_SSTATE)[r] = pk;   // the record's form: two bits per sub-group
    }
    __syncthreads();
    // one thread per COLUMN (u0, u1) of units: unit u2 of the column is listed iff one of the sub-groups {u0 - 1, u0} x
    // {u1 - 1, u1} x {u2 - 1, u2} is undecided; the listed units in ascending order
    const int nu0 = (lx + 1) >> 1, nu1 = (ly + 1) >> 1, nu2 = (lz + 1) >> 1;           // units per axis (<= 17)
    int nlisted = 0;
    unsigned *colinfo = reinterpret_cast<unsigned *>(scratch + CULL_COLINFO);
    for (int q0 = 0; q0 < 17 * 17; q0 += BLOCK) {
        const int q = q0 + tid, u0 = q / 17, u1 = q - 17 * u0;
        unsigned um = 0;
        if (q < 17 * 17 && u0 < nu0 && u1 < nu1) {
            unsigned m = 0;
            for (int d0 = 0; d0 < 2; d0++)
                for (int d1 = 0; d1 < 2; d1++) {
                    const int h0 = u0 - d0, h1 = u1 - d1;
                    if (h0 >= 0 && h0 < 16 && h1 >= 0 && h1 < 16) m |= ub[h0 * 16 + h1];
                }
            um = (m | (m << 1)) & ((1u << nu2) - 1u);
        }
        int n;
        int pos = nlisted + block_exclusive_scan<BLOCK>(__popc(um), wave_sums, n);
        nlisted += n;
        if (q < 17 * 17) colinfo[q] = um | ((unsigned)min(pos, 32767) << 17);
        if (nlisted <= CULL_UNIT_CAP) {                                                // (uniform)
            while (um) {
                const int u2 = __ffs((int)um) - 1;
                um &= um - 1u;
                ulist[pos++] = (unsigned short)((u0 << 10) | (u1 << 5) | u2);
            }
        }
    }
    SDF_CULL_PROF(24);
    if (nlisted > CULL_UNIT_CAP) { __syncthreads(); return -1; }                        // (uniform) nearly everything: the dense path is cheaper
    if (tid < 8 && nlisted + tid < ((nlisted + 7) & ~7)) ulist[nlisted + tid] = 0xFFFFu;   // whole tasks
    if (tid == 0) reinterpret_cast<unsigned short *>(scratch)[0] = (unsigned short)nlisted;
    __syncthreads();
    SDF_CULL_PROF(25);
#undef SDF_CULL_PROF
    return (nlisted + 7) >> 3;
}

// Stores of the soup.  Marking them non-temporal (so that 200 MB of output would not flush the parked
// triangles out of the XCD's L2) was measured and is OFF: 8-byte non-temporal stores are not merged in
// L2, WRITE_SIZE doubled (368 -> 749 MB per launch) and the step got 14 % slower.
#ifndef SDF_SOUP_NT
#define SDF_SOUP_NT 0
#endif
#if SDF_SOUP_NT
#define SDF_SOUP_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define SDF_SOUP_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif

// ---- deferred emission: the slots of sparse tiles ----------------------------------------------
// The region of the dense tile ([MESH_LDS_VOL, bits_off) of dynamic LDS) holds TWO slots when tiles are sparse; a slot:
//   [0, 64)        the batch as the emission needs it later: offset[3], scale[3] (double), work item, triangles, tasks
//   [64, 1232)     k_cull's column words (TileView::colinfo)
//   [1232, ..)     64 floats per listed task (eight units of 2^3 samples), then the batch's triangle list
//   between the two slots and the sign bits: per wave 32 x 9 floats through which the triangles of a waiting batch are
//   transposed, so that consecutive lanes store consecutive coordinates of the soup (whole cache lines per store
//   instruction instead of 72-byte strides).  The sign bits and the work area behind it stay free during that emission:
//   the LAST wave of the workgroup uses the time to take the next work item and bring its record and axes in (k_mesh).
enum { MESH_SLOT_HDR = 1232, MESH_SLOT_COLINFO = 64, MESH_STAGE_BYTES = 32 * 9 * 4 };   // (a wave transposes its 64 triangles in two halves)

template <typename T, bool FULL, int NP, int ND, int NS, int BLOCK, bool TWOPASS = false>
__global__ __launch_bounds__(BLOCK) void k_mesh(const uint32_t *__restrict__ code, const T *__restrict__ consts, MeshArgs a_byval) {
    // (the argument block is read from the kernel-argument segment where it is used -- a scalar load from constant memory, two
    // words of pointer to keep -- instead of living in ~ 80 scalar registers that the round structure then spills)
    typedef __attribute__((address_space(4))) const MeshArgs KArgs;
    // (the block follows the two pointers in the kernel-argument segment: a new leading parameter, or an alignment of MeshArgs above
    // 16 bytes, would move it)
    static_assert(sizeof(const uint32_t *) + sizeof(const T *) == 16 && alignof(MeshArgs) <= 16, "MeshArgs sits at byte 16 of the kernel arguments");
    KArgs *ap = (KArgs *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + 16);
    (void)a_byval;
#define KA (*ap)
#define SDF_SINK (Tri16Sink{KA.out, KA.raw, KA.raw_cap, &KA.ctr->n_raw})
    typedef Vec<T, NS> V;
    constexpr int RPT = 1024 / BLOCK;   // (i0, i1) rows of cells per thread (a tile has <= 32 x 32 rows)
    constexpr int MESH_CELL_CHUNKS = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *wave_sums = reinterpret_cast<int *>(smem);                 // 16 ints
    int *bcast = wave_sums + 16;                                    // 16 ints of scratch
    unsigned char *ntri_lds = smem + MESH_LDS_NTRI;                 // 256 B: ntri | ambiguous << 7
    double *axes = reinterpret_cast<double *>(smem + MESH_LDS_AXES);  // 3 * 33 doubles (X, Y, Z of the tile)
    float *vol = reinterpret_cast<float *>(smem + MESH_LDS_VOL);    // (bs+1)^3 floats: the dense tile
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(smem + KA.bits_off);   // 1 bit per sample: value > 0
    // behind the sign bits: k_cull's record of the batch while it is sampled; the cell table and triangle list of a DENSE tile
    unsigned *wlist = reinterpret_cast<unsigned *>(smem + KA.list_off);
    // `tid` is made opaque to the optimiser at every phase boundary (SDF_FRESH): whatever a phase derives
    // from it is worked out again there instead of being kept -- i.e. spilled -- across the interpreter
    int tid = threadIdx.x;
#define SDF_FRESH() asm volatile("" : "+v"(tid))
    const GridDesc g = a_byval.g;
    // the triangle table in LDS: edge ids e0 | e1 << 4 | e2 << 8 of triangle j of configuration cfg at [5 cfg + j] (a
    // non-ambiguous configuration has at most 5).  The emission used to fetch them as three byte loads from the 4 KB table in
    // device memory per triangle: 4.6 % of the kernel by knock-out (r04aj)
    unsigned short *tri_lds = reinterpret_cast<unsigned short *>(smem + MESH_LDS_TRI);

    if (tid < 256) ntri_lds[tid] = (unsigned char)(KA.mc->ntri[tid] | (KA.mc->amb[tid] << 7));
    if (!TWOPASS)
        for (int i = tid; i < 256 * 5; i += BLOCK) {
            const int cfg = i / 5, j = i - 5 * cfg;
            const signed char *t3 = &KA.mc->tri[cfg][3 * j];
            tri_lds[i] = (unsigned short)(((unsigned)t3[0] & 15u) | (((unsigned)t3[1] & 15u) << 4) | (((unsigned)t3[2] & 15u) << 8));
        }

    for (int i = tid; i < (int)((KA.list_off - KA.bits_off) >> 3); i += BLOCK) bits[i] = 0ull;   // (every round leaves them cleared for the next)
    const int work_begin = KA.ctr->work_begin, work_end = KA.ctr->work_end;
    if (KA.prof && tid == 0) KA.prof[64 + 4 * blockIdx.x] = wall_clock64();        // (timeline of the workgroup, 100 MHz)
    if (tid == 0) {   // the kernel's start on the device's own clock (sdf_stats.ms_mesh_device, sclk_mhz)
        const unsigned long long tw = wall_clock64();
        atomicMax(&KA.ctr->t_first_inv, ~tw);
        if (blockIdx.x == 0) { KA.ctr->clk_cycles = (unsigned long long)clock64(); KA.ctr->clk_ticks = tw; }
    }
    long long tprev = KA.prof ? clock64() : 0;
#define SDF_PROF(K) do { if (KA.prof && tid == 0) { const long long tn = clock64(); atomicAdd(&KA.prof[K], (unsigned long long)(tn - tprev)); tprev = tn; } } while (0)
    // position-dependent bookkeeping of work item w_ (thread 0)
    auto settle = [&](int w_, unsigned long long excl, unsigned long long total_) {
        if (excl == ~0ull) atomicOr(&KA.ctr->overflow, 2u);           // look-back timed out (never expected)
        else if (excl + total_ > KA.out_cap) atomicOr(&KA.ctr->overflow, 1u);
        if (w_ == work_end - 1 && excl != ~0ull) KA.ctr->total = excl + total_;
    };
    // The parked batches of this workgroup: a FIFO of up to MESH_PARK_DEPTH, each in its own staging slot (all
    // values workgroup-uniform).  Sampling times differ a lot between batches (pruned tapes, culled tiles): with
    // ONE slot a workgroup that met a slow predecessor twice in a row stood still -- 12 % of the kernel's
    // cycles were spent in the look-back of the parked batch.  A batch is placed as soon as its predecessors
    // have published (checked once per batch of this workgroup, oldest first); only a full FIFO waits.
    float *my_park = KA.park ? KA.park + (size_t)blockIdx.x * (size_t)MESH_PARK_DEPTH * (size_t)KA.park_cap * 9 : nullptr;
    int pq_head = 0, pq_count = 0;
    unsigned char *pend_base = smem + MESH_LDS_PEND;   // entry k: double xf[6] (offset[3], scale[3]), int w, int total
    auto pend_xf = [&](int k) { return reinterpret_cast<double *>(pend_base + 64 * k); };
    auto pend_wt = [&](int k) { return reinterpret_cast<int *>(pend_base + 64 * k + 48); };
    // place the parked batches whose predecessors have published, oldest first; `must`: wait for ALL of them
    auto place_parked = [&](unsigned long long pre_pend, bool must, int need) {   // need: free entries wanted afterwards
        bool first = true;
        while (pq_count > 0) {
            const int pend_w = uni(pend_wt(pq_head)[0]), pend_total = uni(pend_wt(pq_head)[1]);
            const bool block = must || pq_count > MESH_PARK_DEPTH - need;
            if (tid < 64) {
                const unsigned long long pre = first ? pre_pend : lookback_prefetch(KA.status, pend_w, work_begin);
                const unsigned long long excl = ordered_base(KA.status, pend_w, work_begin, (unsigned long long)pend_total,
                                                             block ? MESH_SPIN_FOREVER : KA.park_spins, pre);
                if (tid == 0) {
                    if (excl != MESH_NOT_READY) settle(pend_w, excl, (unsigned long long)pend_total);
                    reinterpret_cast<unsigned long long *>(bcast + 4)[0] = excl;
                }
            }
            first = false;
            __syncthreads();
            const unsigned long long pbase = uni64(reinterpret_cast<unsigned long long *>(bcast + 4)[0]);
            if (pbase == MESH_NOT_READY) { __syncthreads(); break; }   // (bcast is reused)
            if (pbase != ~0ull && pbase + (unsigned long long)pend_total <= KA.out_cap) {
                double *dst0 = KA.out + pbase * 9ull;
                const float *src = my_park + (size_t)pq_head * (size_t)KA.park_cap * 9;
                const double *xf = pend_xf(pq_head);
                const double pof0 = xf[0], pof1 = xf[1], pof2 = xf[2], psc0 = xf[3], psc1 = xf[4], psc2 = xf[5];
                // consecutive lanes move consecutive coordinates (4-byte loads, 8-byte stores: whole cache lines per
                // instruction on both sides); coordinate e belongs to axis e % 3, so the axis of a thread's k-th
                // coordinate is (its first axis + k * (BLOCK % 3)) % 3.  Eight loads in flight per thread.
                // (16-byte loads / stores, four coordinates per lane and instruction, with all of a batch's loads in flight
                // before the first store: measured in r02, placing 16.7 -> 15.4 Mcycles, k_mesh unchanged within noise, and
                // the 16 floats held per lane cost the headline variant its spill-free register allocation.  Rejected.)
                static_assert(BLOCK % 3 == 1 || BLOCK % 3 == 2, "axis rotation below");
                const int n9 = pend_total * 9;
                const double sc[3] = {psc0, psc1, psc2}, of[3] = {pof0, pof1, pof2};
                constexpr int U = 8;
                if (KA.compact) {   // (uniform) the slab takes 16-byte records: a lane per parked triangle
                    for (int t = tid; t < pend_total; t += BLOCK) {
                        float f[9];
                        SDF_UNROLL for (int q = 0; q < 9; q++) f[q] = src[(size_t)t * 9 + q];
                        store_tri16(SDF_SINK, pbase + (unsigned long long)t, f);
                    }
                } else
                for (int e0 = tid; e0 < n9; e0 += BLOCK * U) {
                    float f[U];
                    SDF_UNROLL for (int k = 0; k < U; k++) f[k] = src[min(e0 + k * BLOCK, n9 - 1)];
                    int ax = e0 % 3;
                    SDF_UNROLL
                    for (int k = 0; k < U; k++) {
                        const double s_ = ax == 0 ? sc[0] : (ax == 1 ? sc[1] : sc[2]), o_ = ax == 0 ? of[0] : (ax == 1 ? of[1] : of[2]);
                        if (e0 + k * BLOCK < n9) SDF_SOUP_STORE(dst0 + e0 + k * BLOCK, (double)f[k] * s_ + o_);
                        ax += BLOCK % 3; if (ax >= 3) ax -= 3;
                    }
                }
            }
            pq_head = (pq_head + 1) % MESH_PARK_DEPTH; pq_count--;
            __syncthreads();   // (bcast is reused)
        }
    };
    // ---- deferred emission ----
    // A batch whose predecessors are still sampling when it has been counted used to be PARKED: its triangles went to a
    // staging slot as 9 floats each and were moved to their place one batch later -- 97 % of the batches of the 512^3
    // example, 2.4 x the soup's bytes through the memory system and a tenth of the kernel spent moving them.  With sparse
    // tiles (TileView) the counted batch simply STAYS in LDS -- its listed samples, its column words, its triangle list,
    // in one of two slots -- while the workgroup samples and counts its next batch in the other slot; then, one batch
    // later, the look-back finds every predecessor published and the triangles are written ONCE, as float64, where they
    // belong.  What cannot be deferred -- a tile that is not culled or lists too many units (it takes the whole region:
    // a waiting batch is written first), a batch whose triangle list does not fit -- goes the old way, and so does a
    // deferred batch whose predecessors are STILL not done one batch later (it is parked then; nothing ever waits).
    // (Taking the next work item's INDEX early -- thread 0 issuing the atomic and the work-list load behind the sampling
    // phase -- was built and measured in r02: the time at the top of the loop did not move (it is the record / axis loads,
    // not the atomic), and the two values carried across the phases cost 6 - 9 more spilled registers: 0.288 -> 0.300 ms.
    // What does pay is below: a whole WAVE brings the item, its record and its axes into LDS during the emission.)
    const int slot_bytes = TWOPASS ? 0 : KA.slot_bytes;
    auto slot_base = [&](int s_) { return smem + MESH_LDS_VOL + (size_t)s_ * (size_t)slot_bytes; };
    int dq_slot = -1;          // the slot of the counted batch whose triangles are still to be written (-1: none)
    bool carry = false;        // the work item in `w` was taken in the previous round (which only wrote the waiting batch)
    int w = 0;
    // The next work item, taken EARLY: while the other waves write the waiting batch's triangles, the last wave draws the
    // next item from the counter and brings what the round after needs -- batch index, header, k_cull's record, the
    // tile's axes -- into LDS (bcast[8..11], the work area, `axes`: all idle then).  Taking an item, its header and its
    // record were three dependent round trips to device memory at the top of every round (13 k cycles of 76 k per batch
    // at 512^3, SDF_MESH_PROF); now they run under the emission.  Items are still handed out in order of the counter, a
    // workgroup merely holds its next one a little earlier; items of the cost-ordered tail are drawn but not loaded (their
    // rank decides which batch the draw stands for).
    bool nx_valid = false;     // bcast[8] holds the counter value of the next item (drawn by the last round)
    bool have = false;         // ... and its batch index / header are in bcast[9..10], record and axes in LDS
    for (;;) {
        SDF_FRESH();
        asm volatile("" : "+s"(ap));
        dq_slot = uni(dq_slot); pq_head = uni(pq_head); pq_count = uni(pq_count);   // (uniform by construction)
        if (!carry) {
            if (nx_valid) {   // (uniform) drawn during the last round's emission; visible since that round's last barrier
                if (tid == 0) { const int idx = bcast[8]; bcast[0] = work_begin + idx; bcast[1] = idx; }
                have = uni(bcast[11]) != 0;
            } else {
                if (tid == 0) { const int idx = (int)atomicAdd(&KA.ctr->work_counter, 1u); bcast[0] = work_begin + idx; bcast[1] = idx; }
                have = false;
            }
            nx_valid = false;
            SDF_PROF(42);
            __syncthreads();
            w = bcast[0];

            // Items are handed out in list order, so that the predecessors of a batch are always held by running
            // workgroups -- except inside the tail, which goes by descending cost (MeshArgs::order): the r-th workgroup to
            // arrive there takes the item of rank r (every thread ranks one item among the tail's <= 255 costs; ties by
            // position).  That is safe: a batch publishes its COUNT before anything that can wait, so the look-back
            // needs every earlier batch to be sampled, no more; a workgroup that waits (full FIFO, a batch too large to
            // park) holds one item of the tail at most that others wait for, and the tail has fewer items than there are
            // workgroups, hence some workgroup is always free to take the item everybody waits for.
            {
                const int n_work = work_end - work_begin, tail = min(KA.tail, n_work), r = bcast[1] - (n_work - tail);
                if (KA.order && r >= 0 && r < tail && !have) {   // (uniform; an item that came with its record is not of the tail)
                    int *cost = reinterpret_cast<int *>(wlist), *rnk = cost + 256;   // (the work area is idle here)
                    for (int i = tid; i < 256; i += BLOCK) { cost[i] = i < tail ? KA.order[i] : -1; rnk[i] = 0; }
                    __syncthreads();
                    {   // item i = tid % 256 against a quarter (half) of the others, the partial ranks added up in LDS
                        constexpr int PARTS = BLOCK / 256, SPAN = 256 / PARTS;
                        const int i = tid & 255, j0 = (tid >> 8) * SPAN, ci = cost[i];
                        int part = 0;
                        for (int j = j0; j < j0 + SPAN; j++) { const int cj = cost[j]; part += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
                        if (part) atomicAdd(&rnk[i], part);
                    }
                    __syncthreads();
                    if (tid < tail && rnk[tid] == r) bcast[0] = work_end - tail + tid;
                    __syncthreads();
                    w = bcast[0];
                    __syncthreads();   // (the work area takes the batch's record next)
                }
            }
        }
        w = uni(w);
        SDF_PROF(43);
        carry = false;
        const bool finished = w >= work_end;
        if (finished && KA.prof && tid == 0 && KA.prof[64 + 4 * blockIdx.x + 1] == 0) KA.prof[64 + 4 * blockIdx.x + 1] = wall_clock64();
        if (finished && dq_slot < 0) break;
        // ---- what kind of tile?  (the header word of k_cull's record: a uniform load) ----
        bool flush_only = finished;   // this round only writes the waiting batch
        int b = 0, ntl_cull = -1;     // listed tasks of a culled tile (-1: not culled)
        bool sparse = false;
        if (!finished) {
            int b_v;
            unsigned n0_v;
            if (have) { b_v = bcast[9]; n0_v = (unsigned)bcast[10]; }
            else {
                b_v = KA.worklist[w];                                                  // (both loads in flight before either is waited for)
                n0_v = KA.cull ? reinterpret_cast<const unsigned *>(KA.cull + (size_t)w * CULL_RECORD)[0] : 0xFFFFu;
            }
            b = uni(b_v);
            const unsigned n0 = (unsigned)uni((int)n0_v) & 0xFFFFu;
            if (n0 != 0xFFFFu) ntl_cull = (int)((n0 + 7u) >> 3);
            sparse = ntl_cull >= 0 && slot_bytes > 0 && MESH_SLOT_HDR + 256 * ntl_cull + 4 * MESH_CELL_CHUNKS * BLOCK <= slot_bytes;
            if (!sparse && dq_slot >= 0) { flush_only = true; carry = true; }   // a dense tile takes the slots' region
        }
        const bool culled = ntl_cull >= 0;
        SDF_PROF(44);
        const int cur_slot = dq_slot == 0 ? 1 : 0;
        // the current batch: where its samples, its cell table / triangle list live
        unsigned char *cs = slot_base(sparse ? cur_slot : 0);
        float *smp = reinterpret_cast<float *>(cs + MESH_SLOT_HDR);
        unsigned *clist = sparse ? reinterpret_cast<unsigned *>(cs + MESH_SLOT_HDR + 256 * ntl_cull) : wlist;
        const int lcap = sparse ? min((slot_bytes - MESH_SLOT_HDR - 256 * ntl_cull) >> 2, 16384) : KA.list_cap;
        int lx = 2, ly = 2, lz = 2, lyz = 4;
        int row_tris[RPT], row_off[RPT], row_cell0[RPT];
        unsigned row_mask[RPT];
        unsigned long long row_bits[RPT][4];   // sign bits of the four sample rows (o0, o1) of a cell row
        // (left uninitialised on purpose: written by the counting of THIS batch, read by its own emission only -- zeroes here
        // would be carried through the interpreter in registers)
        int total = 0, c1 = 1;
        float inv_c1 = 1.0f;
        bool list_ready = false, emit_cur = false;
        unsigned long long pre_own = 0, pre_dq = 0, pre_pend = 0;
        if (!flush_only) {
        // k_cull's record of the batch (cull_tasks) travels next to the axes: units and sub-group states into the work area
        // (idle until the cells of a dense tile are listed), the column words into the batch's slot
        int ox, oy, oz;
        batch_origin(g, b, ox, oy, oz, lx, ly, lz);
        if (have) {   // (uniform) record and axes came in during the last round's emission; the column words wait in the work area
            if (sparse) for (int i = tid; i < 289; i += BLOCK) reinterpret_cast<unsigned *>(cs + MESH_SLOT_COLINFO)[i] = wlist[CULL_COLINFO / 4 + i];
        } else {
            if (culled) {
                const unsigned *rec = reinterpret_cast<const unsigned *>(KA.cull + (size_t)w * CULL_RECORD);
                const int nwords = (CULL_ULIST + 16 * ntl_cull + 3) >> 2;
                for (int i = tid; i < nwords; i += BLOCK) wlist[i] = rec[i];
                for (int i = tid; i < 256; i += BLOCK) wlist[CULL_SSTATE / 4 + i] = rec[CULL_SSTATE / 4 + i];
                if (sparse) for (int i = tid; i < 289; i += BLOCK) reinterpret_cast<unsigned *>(cs + MESH_SLOT_COLINFO)[i] = rec[CULL_COLINFO / 4 + i];
            }
            if (tid < lx) axes[tid] = g.X[ox + tid];
            else if (tid >= 64 && tid < 64 + ly) axes[33 + tid - 64] = g.Y[oy + tid - 64];
            else if (tid >= 128 && tid < 128 + lz) axes[66 + tid - 128] = g.Z[oz + tid - 128];
        }
        __syncthreads();
        SDF_PROF(0);

        // ---- 1. sample: volume = sdf(P).reshape(shape), cast to float32 (core.py:50-52) ----
        // (with the interval prepass on, this batch has its own tape with the irrelevant instructions removed)
        // (`code` stays the base of every address so that the loads remain scalar loads from a read-only
        // kernel argument; b comes out of LDS, hence the readfirstlane)
        const uint32_t *wcode = code + (size_t)__builtin_amdgcn_readfirstlane(b) * (size_t)KA.tape_stride * 2;
        if (KA.tape_stride && tid == 0)
            atomicAdd(&KA.ctr->n_pruned, (unsigned long long)KA.n_instr - reinterpret_cast<const unsigned long long *>(wcode)[KA.tape_stride - 1]);
        const TileTasks tt(lx, ly, lz);
        const int nvox = tt.nvox;
        lyz = tt.lyz;
        const int wave = tid >> 6, lane = tid & 63;
        constexpr int NWAVE = BLOCK / 64;
        // ---- 1a. sub-groups of 2^3 cells whose interval excludes the surface are not sampled: k_cull left the
        // list of units to evaluate and the sign of the decided sub-groups (cull_tasks) ----
        long long tsub = KA.prof ? clock64() : 0;
#define SDF_SUBPROF(K) do { if (KA.prof && tid == 0) { const long long tn = clock64(); atomicAdd(&KA.prof[K], (unsigned long long)(tn - tsub)); tsub = tn; } } while (0)
        const unsigned short *units = reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(wlist) + CULL_ULIST);
        const unsigned *sstate = wlist + CULL_SSTATE / 4;   // 16 x 16 words of 16 two-bit states
        int ntl = tt.ntask;
        if (culled) {
            ntl = ntl_cull;
            // Only SIGNS matter at the samples of decided sub-groups (marching cubes reads values at the corners
            // of cells with a sign change, and those lie in undecided sub-groups): their bits of the sign-bit
            // volume are set here straight from the sub-group states -- no float is written for them -- and the
            // evaluated samples OR theirs in as they are stored (1c).  A sample owned by an undecided sub-group
            // starts at 0; an evaluated sample owned by a decided sub-group has the sub-group's sign anyway.
            const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
            const int nwords = (nvox + 63) >> 6;
            (void)nwords;   // (the words -- + 2: the row extraction reads one word ahead -- were cleared when the previous batch had been
                            // counted, or at the kernel's start: below; until r05ab here, in front of a barrier of its own)
            // <cull-sign-fill>  (tests/native/cull_tasks_host.py cuts this loop out for the host test)
            // a thread per row of lz samples along z.  Sub-group h of the row owns samples 2 h and 2 h + 1 (the last one, hlast,
            // the boundary sample c2 too): its state sits at bits 2 h, 2 h + 1 of the row's word, "positive" = 01, so the
            // samples' bits are the positive states' low bits doubled.
            const int hlast = (c2 - 1) >> 1;
            for (int r = tid; r < lx * ly; r += BLOCK) {
                const int ix = fast_div(r, 1.0f / (float)ly), iy = r - ly * ix;
                const unsigned st = sstate[(min(ix, c0 - 1) >> 1) * 16 + (min(iy, c1 - 1) >> 1)];
                const unsigned pos = st & ~(st >> 1) & 0x55555555u & (unsigned)((4ull << (2 * hlast)) - 1ull);   // (sub-groups beyond the tile count as decided: not here)
                unsigned long long rowmask = (unsigned long long)(pos | (pos << 1));
                if ((pos >> (2 * hlast)) & 1u) rowmask |= 1ull << c2;
                if (rowmask) {
                    const int o = r * lz, sh = o & 63;
                    atomicOr(&bits[o >> 6], rowmask << sh);
                    if (sh && (rowmask >> (64 - sh))) atomicOr(&bits[(o >> 6) + 1], rowmask >> (64 - sh));
                }
            }
            // </cull-sign-fill>
            // (no barrier: the evaluation below only ORs into the same words)
        }
        if (tid == 0) atomicAdd(&KA.ctr->n_sampled, culled ? (unsigned long long)ntl * 64ull : (unsigned long long)nvox);
        SDF_SUBPROF(9);
        // ---- 1c. evaluate the listed tasks: NS per wave and pass.  A sparse tile keeps sample `lane` of task t at
        // smp[64 t + lane] (unit 8 t + lane / 8 of the list, sample lane % 8 of the unit: TileView::at) ----
        // <sample-loop>  (tests/native/cull_tasks_host.py cuts this loop out for the host test)
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
                    if (sparse) smp[64 * (t0 + k) + lane] = fv; else vol[i] = fv;
                    if (culled && fv > 0.0f) atomicOr(&bits[i >> 6], 1ull << (i & 63));
                }
            }
        }
        // </sample-loop>
        __syncthreads();
        SDF_FRESH();
        SDF_SUBPROF(10);
        // ---- 1d. the sign-bit volume: one word per 64 consecutive samples (the marching phases classify
        // cells from these bits instead of re-reading 8 floats per cell) ----
        for (int word = wave * 4; !culled && word < ((nvox + 63) >> 6); word += NWAVE * 4) {   // (four reads in flight per lane)
            float f[4];
            SDF_UNROLL for (int k = 0; k < 4; k++) f[k] = vol[min((word + k) * 64 + lane, nvox - 1)];
            SDF_UNROLL
            for (int k = 0; k < 4; k++) {
                const unsigned long long m = __ballot((word + k) * 64 + lane < nvox && f[k] > 0.0f);
                if (lane == 0 && word + k < ((nvox + 63) >> 6)) bits[word + k] = m;
            }
        }
        if (!culled) {
            if (tid < 2) bits[((nvox + 63) >> 6) + tid] = 0ull;   // the row extraction reads one word ahead
            __syncthreads();
        }
        SDF_SUBPROF(11);
#undef SDF_SUBPROF
        SDF_PROF(1);
        SDF_FRESH();
        const TileView cvw{sparse ? smp : vol, reinterpret_cast<const unsigned *>(cs + MESH_SLOT_COLINFO), lyz, lz, sparse};

        // ---- 2. count: a thread owns the i2-rows of cells (i0, i1) = row tid + k * BLOCK ----
        // (wave 0 first asks for the predecessors' status words -- of this batch, of the waiting one and of the oldest
        // parked one -- so that the answers arrive while the cells are counted)
        if (tid < 64 && !TWOPASS) {
            if (!sparse) pre_own = lookback_prefetch(KA.status, w, work_begin);   // (a sparse tile's batch waits a round: asked for then)
            if (dq_slot >= 0) pre_dq = lookback_prefetch(KA.status, reinterpret_cast<const int *>(slot_base(dq_slot) + 48)[0], work_begin);
            if (pq_count > 0) pre_pend = lookback_prefetch(KA.status, pend_wt(pq_head)[0], work_begin);
        }
        const int c0 = lx - 1, c2 = lz - 1;
        c1 = ly - 1;
        const int nrows = (c0 > 0 && c1 > 0 && c2 > 0) ? c0 * c1 : 0;
        inv_c1 = 1.0f / (float)max(c1, 1);
        int my_amb = 0;
        // sign strings of the four sample rows around cell row (i0, i1) and the mask of its surface cells
        auto row_signs = [&](int i0, int i1, unsigned long long *rb) -> unsigned {
            SDF_UNROLL
            for (int q = 0; q < 4; q++) {   // q = 2 * o0 + o1
                const int o = (i0 + (q >> 1)) * lyz + (i1 + (q & 1)) * lz;
                const unsigned long long w0 = bits[o >> 6], w1 = bits[(o >> 6) + 1];
                const int sh = o & 63;
                rb[q] = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
            }
            // a cell (columns i2, i2 + 1) has a surface unless its 8 bits are all equal
            const unsigned long long any = rb[0] | rb[1] | rb[2] | rb[3];
            const unsigned long long all = rb[0] & rb[1] & rb[2] & rb[3];
            const unsigned long long ones = all & (all >> 1), zeros = ~any & ~(any >> 1);
            return (unsigned)(~(ones | zeros)) & (c2 >= 32 ? 0xFFFFFFFFu : ((1u << c2) - 1u));
        };
        // triangles of an ambiguous cell: Lewiner's tests on its 8 corner samples pick the tiling (rare)
        auto amb_count = [&](int i0, int i1, int i2) -> int {
            float c8[8];
            double lv[8];
            int off;
            cvw.cell(i0, i1, i2, c8);
            mc33_load_cell(c8, 4, 2, lv);
            return mc33_cell(lv, KA.mc->mc33, &off);
        };
        // ---- 2a. surface cells per row, their running count over the rows (the order of the soup) ----
        int ncells = 0;
        // (the scans of this phase take ONE barrier each: they alternate between two buffers, block_exclusive_scan1)
        int scan_ix = 0;
        int *wave_sums_b = reinterpret_cast<int *>(smem + MESH_LDS_SUMS2);
#define SDF_SCAN(V, TOT) block_exclusive_scan1<BLOCK>((V), (scan_ix++ & 1) ? wave_sums_b : wave_sums, (TOT))
        SDF_UNROLL
        for (int k = 0; k < RPT; k++) {
            const int r = tid + k * BLOCK;
            unsigned mask = 0;
            SDF_UNROLL for (int q = 0; q < 4; q++) row_bits[k][q] = 0ull;
            if (r < nrows) {
                const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                mask = row_signs(i0, i1, row_bits[k]);
            }
            row_mask[k] = mask;
            int tot;
            row_cell0[k] = ncells + SDF_SCAN(__popc(mask), tot);
            ncells += tot;
        }
        ncells = uni(ncells);
        SDF_PROF(45);
        // ---- 2b. ONE THREAD PER SURFACE CELL (up to MESH_CELL_CHUNKS * BLOCK of them).  The row's thread only
        // SCATTERS its cells -- (row, column) into a table at the cell's running index, a few
        // ALU instructions and one LDS write each, nothing to wait for; then thread s takes cell s: looks up its
        // triangles and, after a scan, writes them into the triangle list right away.  A thread per ROW used to
        // walk its surface cells one dependent LDS look-up after the other, here and again when the list was
        // built: a surface that runs along the rows (a flat face: 32 surface cells in each of a few rows, none
        // in the others) made both phases wait for a handful of lanes -- a tenth of the kernel.  Batches with
        // more cells, or more triangles than the list holds, take the per-row path below.  (The table lives in
        // the list region: every thread has read its entry before the scan's barriers, the list is written
        // behind them.) ----
        unsigned cinfo[MESH_CELL_CHUNKS];
        int cn[MESH_CELL_CHUNKS], coff[MESH_CELL_CHUNKS];
        bool per_cell = false;            // the per-cell path ran (cinfo / cn / coff are valid)
        if (ncells <= MESH_CELL_CHUNKS * BLOCK && lcap >= MESH_CELL_CHUNKS * BLOCK) {
            SDF_UNROLL
            for (int k = 0; k < RPT; k++) {
                const int r = tid + k * BLOCK;
                unsigned m = row_mask[k];
                int pos = row_cell0[k];
                while (m) {
                    const int i2 = __ffs((int)m) - 1;
                    m &= m - 1u;
                    clist[pos++] = (unsigned)r | ((unsigned)i2 << 10);   // (the cell's thread works out the configuration: below)
                }
            }
            __syncthreads();
            per_cell = true;
            SDF_UNROLL
            for (int k = 0; k < MESH_CELL_CHUNKS; k++) {
                const int sidx = tid + k * BLOCK;
                int n = 0;
                unsigned info = 0;
                if (sidx < ncells) {
                    const unsigned ce = clist[sidx];
                    const int r = (int)(ce & 1023u), i2 = (int)((ce >> 10) & 31u), i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                    // the sign configuration from the row's four sign strings, read again here: in the scatter above it
                    // cost the thread of a row ~40 instructions per surface cell, one cell after the other -- 32 in a row
                    // along a flat face -- while every other lane waited (3.6 k cycles per batch)
                    unsigned long long rb[4];
                    row_signs(i0, i1, rb);
                    const unsigned cfg = cell_config(rb, i2);
                    const unsigned e = ntri_lds[cfg];
                    if (e & 128u) { n = amb_count(i0, i1, i2); my_amb++; }
                    else n = (int)(e & 7u);
                    // entry: cell (15 bits) | ambiguous (1) | configuration (8) | triangle in cell (4)
                    info = ((unsigned)((i0 << 10) | (i1 << 5) | i2) << 13) | ((e & 128u) << 5) | (cfg << 4);
                }
                int tot = 0;
                coff[k] = total;
                if (k * BLOCK < ncells) { coff[k] += SDF_SCAN(n, tot); total += tot; }   // (uniform)
                cinfo[k] = info; cn[k] = n;
            }
            if (TWOPASS) {
                list_ready = true;        // (no list in LDS: the entries go to the arena below)
            } else if (total <= lcap) {
                SDF_UNROLL
                for (int k = 0; k < MESH_CELL_CHUNKS; k++)
                    for (int j = 0; j < cn[k]; j++) clist[coff[k] + j] = cinfo[k] | (unsigned)j;
                list_ready = true;        // (made visible by the barriers of the allocation below)
            } else {
                per_cell = false;
            }
        }
        SDF_PROF(46);
        SDF_UNROLL for (int k = 0; k < RPT; k++) { row_tris[k] = 0; row_off[k] = 0; }
        if (!list_ready) {
            // ---- 2c. per-row counting: a thread owns the i2-rows of cells (i0, i1) = row tid + k * BLOCK ----
            total = 0; my_amb = 0;
            SDF_UNROLL
            for (int k = 0; k < RPT; k++) {
                const int r = tid + k * BLOCK;
                int n = 0;
                if (r < nrows) {
                    const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                    unsigned m = row_mask[k];
                    while (m) {
                        const int i2 = __ffs((int)m) - 1;
                        m &= m - 1u;
                        const unsigned e = ntri_lds[cell_config(row_bits[k], i2)];
                        if (e & 128u) { n += amb_count(i0, i1, i2); my_amb++; }
                        else n += (int)(e & 7u);
                    }
                }
                row_tris[k] = n;
                int tot;
                row_off[k] = total + SDF_SCAN(n, tot);
                total += tot;
            }
        }
        total = uni(total);
        // the sign bits are dead from here on (the emission reads samples and the rows' strings in registers): cleared NOW, for the next
        // culled tile's sign fill -- behind this round's remaining barriers instead of in front of a barrier of its own
        for (int i = tid; i < (int)((KA.list_off - KA.bits_off) >> 3); i += BLOCK) bits[i] = 0ull;
        // ---- the batch's count is public from here on; bookkeeping that needs no position ----
        if (tid < 64 && !TWOPASS) publish_count(KA.status, w, work_begin, (unsigned long long)total);
        if (KA.compact && tid == 0 && w - work_begin < KA.xf_cap) {   // the batch's transform travels with the compact soup
            double *xf = KA.xf + (size_t)(w - work_begin) * 6;
            xf[0] = axes[0]; xf[1] = axes[33]; xf[2] = axes[66];
            xf[3] = axes[1] - axes[0]; xf[4] = axes[34] - axes[33]; xf[5] = axes[67] - axes[66];
        }
        if (tid == 0) {
            atomicAdd(total ? &KA.ctr->n_nonempty : &KA.ctr->n_empty, 1u);
            atomicAdd(&KA.ctr->n_eval, (unsigned long long)nvox);
            KA.kinds[b] = total ? 2 : 1;
        }
        if (my_amb) atomicAdd(&KA.ctr->n_ambiguous, (unsigned long long)my_amb);
        if constexpr (TWOPASS) {
            // ---- two-pass meshing: this kernel stops at the classification.  What k_emit2 needs to produce the batch's
            // triangles -- per surface cell its configuration and 8 corner samples, per triangle which cell and which of
            // the cell's triangles -- goes to the arenas, at offsets handed out by two atomic cursors (no order, hence
            // nothing to wait for: no look-back, no parking, no placing); k_scan_items then numbers the triangles in
            // work-list order and k_emit2 writes them, at full occupancy, straight to their final place ----
            if (tid == 0) {
                unsigned long long cb = 0, lb = 0;
                if (total) {
                    cb = atomicAdd(&KA.ctr->cell_cursor, (unsigned long long)ncells);
                    lb = atomicAdd(&KA.ctr->list_cursor, (unsigned long long)total);
                    if (cb + (unsigned long long)ncells > KA.cells_cap || lb + (unsigned long long)total > KA.tlist_cap) {
                        atomicOr(&KA.ctr->overflow, 1u);
                        cb = lb = ~0ull;
                    }
                }
                reinterpret_cast<unsigned long long *>(bcast + 2)[0] = cb;
                reinterpret_cast<unsigned long long *>(bcast + 4)[0] = lb;
                ItemDesc d;
                d.ntri = (unsigned)total; d.ncells = (unsigned)ncells; d.list_off = lb; d.cell_off = cb;
                d.xf[0] = axes[0]; d.xf[1] = axes[33]; d.xf[2] = axes[66];
                d.xf[3] = axes[1] - axes[0]; d.xf[4] = axes[34] - axes[33]; d.xf[5] = axes[67] - axes[66];
                KA.desc[w] = d;
            }
            __syncthreads();
            const unsigned long long cell_base = reinterpret_cast<unsigned long long *>(bcast + 2)[0];
            const unsigned long long list_base = reinterpret_cast<unsigned long long *>(bcast + 4)[0];
            if (total && cell_base != ~0ull) {
                auto put_cell = [&](unsigned long long idx, unsigned info, int i0, int i1, int i2) {
                    unsigned *rec = KA.cells + idx * 9ull;
                    float c8[8];
                    cvw.cell(i0, i1, i2, c8);
                    rec[0] = info;
                    SDF_UNROLL
                    for (int q = 0; q < 8; q++) rec[1 + q] = __float_as_uint(c8[q]);   // corner q = 4 * o0 + 2 * o1 + o2
                };
                if (per_cell) {
                    SDF_UNROLL
                    for (int k = 0; k < MESH_CELL_CHUNKS; k++) {
                        const int sidx = tid + k * BLOCK;
                        if (sidx < ncells) {
                            const int cell = (int)(cinfo[k] >> 13);
                            put_cell(cell_base + (unsigned long long)sidx, cinfo[k], cell >> 10, (cell >> 5) & 31, cell & 31);
                            for (int j = 0; j < cn[k]; j++) KA.tlist[list_base + (unsigned long long)(coff[k] + j)] = ((unsigned)sidx << 4) | (unsigned)j;
                        }
                    }
                } else {
                    SDF_UNROLL
                    for (int k = 0; k < RPT; k++) {
                        if (row_tris[k] == 0) continue;
                        const int r = tid + k * BLOCK;
                        const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                        unsigned m = row_mask[k];
                        int sidx = row_cell0[k], pos = row_off[k];
                        while (m) {
                            const int i2 = __ffs((int)m) - 1;
                            m &= m - 1u;
                            const unsigned cfg = cell_config(row_bits[k], i2);
                            const unsigned en = ntri_lds[cfg];
                            int n = (int)(en & 7u);
                            if (en & 128u) n = amb_count(i0, i1, i2);
                            put_cell(cell_base + (unsigned long long)sidx, ((unsigned)((i0 << 10) | (i1 << 5) | i2) << 13) | ((en & 128u) << 5) | (cfg << 4), i0, i1, i2);
                            for (int j = 0; j < n; j++, pos++) KA.tlist[list_base + (unsigned long long)pos] = ((unsigned)sidx << 4) | (unsigned)j;
                            sidx++;
                        }
                    }
                }
            }
            SDF_PROF(2);
            __syncthreads();   // vol / bcast are reused by the next batch
            continue;
        }
        // ---- this batch: its triangles are written one batch later (it stays in its slot), or right away ----
        if (sparse && list_ready) {
            if (tid == 0) {
                double *xf = reinterpret_cast<double *>(cs);
                xf[0] = axes[0]; xf[1] = axes[33]; xf[2] = axes[66];
                xf[3] = axes[1] - axes[0]; xf[4] = axes[34] - axes[33]; xf[5] = axes[67] - axes[66];
                int *m = reinterpret_cast<int *>(cs + 48);
                m[0] = w; m[1] = total; m[2] = ntl_cull;
            }
        } else emit_cur = true;
        SDF_PROF(2);
        }   // (!flush_only)
        SDF_FRESH();
        __syncthreads();   // (the slot's header)

        // ---- 3 + 4. the triangles: first the waiting batch's (pass 0), then this batch's if it cannot wait (pass 1).  Ordered
        // allocation (wave 0): the batch takes its position if every predecessor has published its count -- the waiting
        // batch's have had a whole batch's time -- else it is PARKED (staging slot, 9 floats per triangle, placed later).
        // Then the per-triangle work list in LDS, one lane per triangle. ----
        const bool sparse_next = !flush_only && !emit_cur;   // this batch becomes the waiting one
        // ---- parked batches are older than anything here: their predecessors have long published, place them (and make room
        // for what this round may park: the waiting batch, this batch) ----
        { const long long tp0 = KA.prof ? clock64() : 0;
        if (flush_only && pq_count > 0 && tid < 64) pre_pend = lookback_prefetch(KA.status, pend_wt(pq_head)[0], work_begin);
        place_parked(pre_pend, false, (emit_cur ? 1 : 0) + (dq_slot >= 0 ? 1 : 0));
        if (KA.prof && tid == 0) atomicAdd(&KA.prof[6], (unsigned long long)(clock64() - tp0)); }
        SDF_PROF(47);
        // (two copies of this code, one per kind of batch, rather than one loop over both: the rows' sign strings and offsets
        // that only a list built in passes needs would otherwise stay in registers through the waiting batch's emission)
        auto emit_stage = [&](auto is_dq_tag) {
            constexpr bool is_dq = decltype(is_dq_tag)::value;
            const unsigned char *ds = slot_base(max(dq_slot, 0));
            const int e_w = is_dq ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(ds + 48)[0]) : w;
            const int e_total = is_dq ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(ds + 48)[1]) : total;
            const int e_ntl = is_dq ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(ds + 48)[2]) : 0;
            const bool e_ready = is_dq || list_ready;
            const int e_lcap = is_dq ? 16384 : lcap;
            unsigned *lst = is_dq ? const_cast<unsigned *>(reinterpret_cast<const unsigned *>(ds + MESH_SLOT_HDR + 256 * e_ntl)) : clist;
            const TileView vw{is_dq ? reinterpret_cast<const float *>(ds + MESH_SLOT_HDR) : (sparse ? smp : vol),
                              reinterpret_cast<const unsigned *>((is_dq ? ds : cs) + MESH_SLOT_COLINFO), lyz, lz, is_dq || sparse};
            // points * scale + offset (reference sdf/core.py:58-60): scale = first axis step of the batch, offset = its
            // first sample, per axis
            const double *exf = reinterpret_cast<const double *>(ds);
            const double of0 = is_dq ? exf[0] : axes[0], of1 = is_dq ? exf[1] : axes[33], of2 = is_dq ? exf[2] : axes[66];
            const double sc0 = is_dq ? exf[3] : axes[1] - of0, sc1 = is_dq ? exf[4] : axes[34] - of1, sc2 = is_dq ? exf[5] : axes[67] - of2;
            const bool may_park = KA.park && e_total <= KA.park_cap;
            const bool block = !may_park || (is_dq && finished);   // (at the end of the list there is nothing else to do but wait)
            if (tid < 64) {
                const unsigned long long pre = is_dq ? (flush_only ? lookback_prefetch(KA.status, e_w, work_begin) : pre_dq)
                                                     : (sparse ? lookback_prefetch(KA.status, e_w, work_begin) : pre_own);
                const unsigned long long excl = ordered_base(KA.status, e_w, work_begin, (unsigned long long)e_total, block ? MESH_SPIN_FOREVER : KA.park_spins, pre);
                if (tid == 0) {
                    if (excl != MESH_NOT_READY) settle(e_w, excl, (unsigned long long)e_total);
                    reinterpret_cast<unsigned long long *>(bcast + 2)[0] = excl;
                    bcast[12] = 0;   // (the waiting batch's triangles are handed to the waves in chunks of 64: below)
                }
            }
            __syncthreads();
            // ---- the last wave takes the next work item while the others start on the triangles (see `nx_valid` above) ----
            const bool prefetch = is_dq && !finished && !carry && KA.stage_off > 0;   // (uniform; `carry`: the next item is in hand already)
            if (prefetch && tid >= BLOCK - 64) {
                const int ln = tid & 63;
                int idx = 0;
                if (ln == 0) idx = (int)atomicAdd(&KA.ctr->work_counter, 1u);
                idx = uni(idx);
                const int nw_ = work_begin + idx;
                const int n_work = work_end - work_begin, tail = min(KA.tail, n_work);
                const bool in_tail = KA.order && idx >= n_work - tail;
                int loaded = 0, nb_ = 0;
                unsigned nn0 = 0xFFFFu;
                if (nw_ < work_end && !in_tail) {   // (wave-uniform)
                    const unsigned *rec = reinterpret_cast<const unsigned *>(KA.cull + (size_t)nw_ * CULL_RECORD);
                    const int nbv = KA.worklist[nw_];
                    const unsigned n0v = KA.cull ? rec[0] : 0xFFFFu;
                    nb_ = uni(nbv);
                    nn0 = (unsigned)uni((int)n0v) & 0xFFFFu;
#if SDF_TAPE_WARM
                    // the next batch's own tape (interval prepass, one copy per batch in device memory) through the scalar cache NOW:
                    // the interpreter asks for one instruction ahead only, so every new 64-byte line -- 8 instructions -- would
                    // otherwise be a round trip to L2 that all sixteen waves of the workgroup sit out together
                    if (KA.tape_stride) {
                        const unsigned long long *tw = reinterpret_cast<const unsigned long long *>(code + (size_t)nb_ * (size_t)KA.tape_stride * 2);
                        unsigned long long acc_w = 0;
                        for (int l = 0; l < KA.tape_stride; l += 8) acc_w ^= tw[l];
                        asm volatile("" :: "s"(acc_w));
                    }
#endif
                    int nox, noy, noz, nlx, nly, nlz;
                    batch_origin(g, nb_, nox, noy, noz, nlx, nly, nlz);
                    if (ln < nlx) axes[ln] = g.X[nox + ln];                       // (the waiting batch's transform sits in its slot)
                    if (ln < nly) axes[33 + ln] = g.Y[noy + ln];
                    if (ln < nlz) axes[66 + ln] = g.Z[noz + ln];
                    if (nn0 != 0xFFFFu) {
                        const int nwords = (int)((CULL_ULIST + 2u * ((nn0 + 7u) & ~7u) + 3u) >> 2);
                        for (int i = ln; i < nwords; i += 64) wlist[i] = rec[i];
                        for (int i = ln; i < (CULL_RECORD - CULL_SSTATE) / 4; i += 64) wlist[CULL_SSTATE / 4 + i] = rec[CULL_SSTATE / 4 + i];   // states + column words
                    }
                    loaded = 1;
                }
                if (ln == 0) { bcast[8] = idx; bcast[9] = nb_; bcast[10] = (int)nn0; bcast[11] = loaded; }
            }
            if (prefetch) nx_valid = true;
            const unsigned long long base = uni64(reinterpret_cast<unsigned long long *>(bcast + 2)[0]);
            SDF_PROF(48);
            const bool parking = base == MESH_NOT_READY;
            const bool fits = parking || (base != ~0ull && base + (unsigned long long)e_total <= KA.out_cap);
            const int park_slot = (pq_head + pq_count) % MESH_PARK_DEPTH;   // (the FIFO has room: a full one was waited for above)
            if (parking) {
                if (tid == 0) {
                    double *xf = pend_xf(park_slot);
                    xf[0] = of0; xf[1] = of1; xf[2] = of2; xf[3] = sc0; xf[4] = sc1; xf[5] = sc2;
                    pend_wt(park_slot)[0] = e_w; pend_wt(park_slot)[1] = e_total;
                }
                pq_count++;
                if (KA.prof && tid == 0) atomicAdd(&KA.prof[7], 1ull);
            }
            if (KA.prof && tid == 0 && is_dq) atomicAdd(&KA.prof[12], 1ull);
            for (int lo = 0; fits && lo < e_total; lo += e_lcap) {
                const int ecn = min(e_lcap, e_total - lo);
                SDF_UNROLL
                for (int k = 0; k < RPT; k++) {
                    if (is_dq || e_ready || row_tris[k] == 0 || row_off[k] >= lo + ecn || row_off[k] + row_tris[k] <= lo) continue;
                    const int r = tid + k * BLOCK;
                    const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                    int pos = row_off[k] - lo;
                    unsigned m = row_mask[k];
                    while (m) {
                        const int i2 = __ffs((int)m) - 1;
                        m &= m - 1u;
                        const unsigned cfg = cell_config(row_bits[k], i2);
                        const unsigned en = ntri_lds[cfg];
                        int n = (int)(en & 7u);
                        if (en & 128u) {
                            float c8[8];
                            double lv[8];
                            int off;
                            vw.cell(i0, i1, i2, c8);
                            mc33_load_cell(c8, 4, 2, lv);
                            n = mc33_cell(lv, KA.mc->mc33, &off);
                        }
                        // entry: cell (15 bits) | ambiguous (1) | configuration (8) | triangle in cell (4)
                        const unsigned e = ((unsigned)((i0 << 10) | (i1 << 5) | i2) << 13) | ((en & 128u) << 5) | (cfg << 4);
                        for (int j = 0; j < n; j++, pos++)
                            if (pos >= 0 && pos < ecn) lst[pos] = e | (unsigned)j;
                    }
                }
                if (!is_dq) __syncthreads();   // (the list rebuilt above; a waiting batch's list has been in its slot for a round --
                                               // and the last wave, busy with the next work item, must not be waited for here)
                SDF_PROF(3);
                double *dst0 = KA.out + (parking ? 0ull : base + (unsigned long long)lo) * 9ull;
                float *park0 = my_park + ((size_t)park_slot * (size_t)KA.park_cap + (size_t)lo) * 9;
                const bool staged = is_dq && KA.stage_off > 0 && !parking;   // (uniform)
                // (whole waves: the transposition below is wave-wide.  The waiting batch's triangles go to the waves in chunks of 64
                // as they come for them -- the last wave joins late, it has taken the next work item first)
                for (int t0 = tid & ~63;; t0 += BLOCK) {
                    if (is_dq) {
                        int ch = 0;
                        if ((tid & 63) == 0) ch = atomicAdd(&bcast[12], 1);
                        t0 = 64 * uni(ch);
                    }
                    if (t0 >= ecn) break;
                    const int t = t0 + (tid & 63);
                    const bool live = t < ecn;
                    const unsigned e = lst[live ? t : ecn - 1];
                    const int j = (int)(e & 15u), cfg = (int)((e >> 4) & 255u), cell = (int)(e >> 13);
                    const int i0 = cell >> 10, i1 = (cell >> 5) & 31, i2 = cell & 31;
                    float o[9];
                    if (e & 4096u) {
                        // (the out-of-line call gets an array of its OWN: `o` handed to it would have to live in scratch memory
                        // for every triangle, not only the ambiguous ones -- 36 bytes written and read back per triangle,
                        // r04h: a quarter of the kernel's write traffic)
                        float c8[8], oa[9];
                        vw.cell(i0, i1, i2, c8);
                        mc33_triangle(c8, 4, 2, i0, i1, i2, KA.mc->mc33, j, oa);
                        SDF_UNROLL for (int q = 0; q < 9; q++) o[q] = oa[q];
                    } else {
                        const unsigned tt3 = tri_lds[5 * cfg + min(j, 4)];
                        mc_vertex_view(vw, i0, i1, i2, (int)(tt3 & 15u), o);
                        mc_vertex_view(vw, i0, i1, i2, (int)((tt3 >> 4) & 15u), o + 3);
                        mc_vertex_view(vw, i0, i1, i2, (int)(tt3 >> 8), o + 6);
                    }
                    if (KA.compact && !parking) {   // (uniform) the exchange's 16-byte record, straight from the registers
                        if (live) store_tri16(SDF_SINK, base + (unsigned long long)(lo + t), o);
                    } else if (staged) {
                        // through LDS: lane l holds triangle t0 + l (9 floats); afterwards lane l stores coordinates 64 k + l,
                        // k = 0 .. 8, of the wave's 576: consecutive lanes, consecutive addresses.  Coordinate c belongs to axis
                        // c % 3 and 64 % 3 == 1: the axis of a lane's k-th coordinate is (l + k) % 3.
                        float *stg = reinterpret_cast<float *>(smem + KA.stage_off) + (tid >> 6) * (MESH_STAGE_BYTES / 4);
                        const int ln = tid & 63;
                        const int a0 = ln % 3;
                        const double s_[3] = {a0 == 0 ? sc0 : (a0 == 1 ? sc1 : sc2), a0 == 0 ? sc1 : (a0 == 1 ? sc2 : sc0), a0 == 0 ? sc2 : (a0 == 1 ? sc0 : sc1)};
                        const double o_[3] = {a0 == 0 ? of0 : (a0 == 1 ? of1 : of2), a0 == 0 ? of1 : (a0 == 1 ? of2 : of0), a0 == 0 ? of2 : (a0 == 1 ? of0 : of1)};
                        SDF_UNROLL
                        for (int h = 0; h < 2; h++) {   // the wave's 64 triangles in two halves of 32 (288 coordinates = 4.5 per lane)
                            if (live && (ln >> 5) == h) { SDF_UNROLL for (int q = 0; q < 9; q++) stg[(ln & 31) * 9 + q] = o[q]; }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const int nval = min(32, ecn - t0 - 32 * h) * 9;   // (<= 0: nothing)
                            double *dstw = dst0 + (size_t)(t0 + 32 * h) * 9;
                            SDF_UNROLL for (int k = 0; k < 5; k++) { const int c = 64 * k + ln; if (c < nval) SDF_SOUP_STORE(dstw + c, (double)stg[c] * s_[k % 3] + o_[k % 3]); }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();   // (the area is rewritten by the next half)
                        }
                    } else if (!live) {
                    } else if (parking) {   // 36 bytes per lane: two 16-byte stores (4-byte aligned) and one of 4
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        float *dst = park0 + (size_t)t * 9;
                        *reinterpret_cast<f4u *>(dst) = f4u{o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f4u *>(dst + 4) = f4u{o[4], o[5], o[6], o[7]};
                        dst[8] = o[8];
                    } else {
                        double *dst = dst0 + (size_t)t * 9;
                        SDF_UNROLL
                        for (int q = 0; q < 9; q += 3) {
                            SDF_SOUP_STORE(dst + q, (double)o[q] * sc0 + of0);
                            SDF_SOUP_STORE(dst + q + 1, (double)o[q + 1] * sc1 + of1);
                            SDF_SOUP_STORE(dst + q + 2, (double)o[q + 2] * sc2 + of2);
                        }
                    }
                }
                __syncthreads();   // list / tile are reused
                SDF_PROF(4);
            }
            __syncthreads();   // (bcast is reused by the next stage; the slot by the next batch)
        };
        if (emit_cur) emit_stage(std::integral_constant<bool, false>());       // (uniform)
        if (dq_slot >= 0) emit_stage(std::integral_constant<bool, true>());    // (uniform)
        dq_slot = sparse_next ? cur_slot : -1;
        SDF_PROF(49);
        if (finished) break;
    }
    SDF_FRESH();
    place_parked(pq_count > 0 && tid < 64 ? lookback_prefetch(KA.status, pend_wt(pq_head)[0], work_begin) : 0ull, true, 0);
    SDF_PROF(5);
    if (KA.prof && tid == 0) KA.prof[64 + 4 * blockIdx.x + 2] = wall_clock64();
    if (tid == 0) {
        const unsigned long long tw = wall_clock64();
        atomicMax(&KA.ctr->t_last, tw);
        if (blockIdx.x == 0) { KA.ctr->clk_cycles = (unsigned long long)clock64() - KA.ctr->clk_cycles; KA.ctr->clk_ticks = tw - KA.ctr->clk_ticks; }
    }
#undef SDF_FRESH
#undef SDF_PROF
}
#undef KA
#undef SDF_SINK

// host-side launcher of one (T, FULL) family, defined in sdf_mesh_inst.hip (one translation
// unit per family so the variants compile in parallel).  slots: 0 = (1,1), 1 = (2,2), 2 = (4,2), 3 = (2,4), 4 = (4,4), 5 = (8,8)
// register files; shape: 0 = 1024 threads x 1 sample per lane, 1 = 512 x 2, anything else = 1024 x 2 (where instantiated).
#define SDF_DECLARE_MESH_LAUNCH(NAME, T) \
    int NAME(int slots, int shape, int twopass, int grid, size_t lds, hipStream_t stream, const uint32_t *code, const T *consts, const MeshArgs &a)
SDF_DECLARE_MESH_LAUNCH(sdf_launch_mesh_f64, double);
SDF_DECLARE_MESH_LAUNCH(sdf_launch_mesh_f64_full, double);

}  // namespace sdfk
