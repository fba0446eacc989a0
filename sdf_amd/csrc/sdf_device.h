// sdf_device.h -- device-side structures and the fused sample+march kernel template (gfx950).
//
//   k_mesh   THE hot kernel: persistent workgroups (one per CU: the tile owns most of the CU's
//            160 KiB LDS) pull surviving batches from the ordered work list.  Per batch:
//              1. sample   the (<=33)^3 tile through the tape interpreter (NS samples per lane,
//                          float64 or float32), cast to float32 like skimage's volume cast, and
//                          store it in LDS -- the field never touches HBM
//                          (reference `_worker`, sdf/core.py:50-52)
//              2. count    one thread per (i0, i1) row of cells walks i2, builds the 8-bit sign
//                          configuration from LDS and sums triangles per row; a block-wide
//                          wave-shuffle prefix scan turns the row counts into offsets
//              3. compact  surface cells expand into a per-triangle work list in LDS
//                          (cell, configuration, triangle-in-cell), in skimage's emission order
//              4. emit     one lane per TRIANGLE: three edge interpolations from the LDS tile, the
//                          float64 world transform `points * scale + offset` (core.py:58-60) and
//                          72 contiguous bytes per lane straight into the ORDERED output soup.
//                          The batch's place in the soup is the exclusive prefix of the triangle
//                          counts of all earlier work items, obtained without a second pass by a
//                          decoupled look-back over per-item status words (work items are handed
//                          out in order, so every predecessor is already being worked on)
//            (reference `_marching_cubes`, sdf/core.py:16-18, 54)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdf_interp.h"
#include "sdf_mc33.h"

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
    unsigned long long total;         // triangles of this shard (written by the workgroup of the last work item)
    // written by k_compact (NOT cleared between meshing retries): the surviving-batch work list
    // and this shard's slice of it, so k_mesh can start without a host round trip
    int nwork, work_begin, work_end, pad_;
};
enum { MESH_COUNTERS_RESET_BYTES = 48 };   // the part of MeshCounters cleared before every k_mesh run

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
};

// dynamic LDS layout of k_mesh
enum { MESH_LDS_NTRI = 128, MESH_LDS_AXES = 384, MESH_LDS_PEND = 1184, MESH_LDS_VOL = 1248 };

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
    // wave64 inclusive scan by shuffles, then the wave totals through LDS
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { const int s = wave_sums[w]; if (w < wid) base += s; tot += s; }
    total = tot;
    __syncthreads();
    return base + inc - v;
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

// One marching-cubes vertex on edge e of the cell at (i0,i1,i2); v points at the cell's corner 0
// in a volume with strides (s0, s1, 1).  skimage's placement: with w = 1/(eps+|v|),
// t = w_hi / (w_lo + w_hi), evaluated in float64 on the float32 samples, stored as float32.
__device__ __forceinline__ void mc_vertex(const float *v, int s0, int s1, int i0, int i1, int i2, int e, float *o) {
    const int axis = e >> 2, oa = (e >> 1) & 1, ob = e & 1;
    int o0, o1, o2, stride;
    if (axis == 0) { o0 = 0; o1 = oa; o2 = ob; stride = s0; }
    else if (axis == 1) { o0 = oa; o1 = 0; o2 = ob; stride = s1; }
    else { o0 = oa; o1 = ob; o2 = 0; stride = 1; }
    const int base = o0 * s0 + o1 * s1 + o2;
    const double vlo = (double)v[base], vhi = (double)v[base + stride];
    const double eps = 2.220446049250313e-16;
    const double wlo = 1.0 / (eps + fabs(vlo)), whi = 1.0 / (eps + fabs(vhi));
    const double t = whi / (wlo + whi);
    double p0 = (double)(i0 + o0), p1 = (double)(i1 + o1), p2 = (double)(i2 + o2);
    if (axis == 0) p0 = (double)i0 + t; else if (axis == 1) p1 = (double)i1 + t; else p2 = (double)i2 + t;
    o[0] = (float)p0; o[1] = (float)p1; o[2] = (float)p2;
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
#define MESH_NOT_READY (~0ull - 1ull)
#define MESH_FLAG_AGG (1ull << 62)
#define MESH_FLAG_PFX (2ull << 62)
#define MESH_VAL_MASK ((1ull << 62) - 1)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, d, 64), hi = __shfl_xor((unsigned)(v >> 32), d, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
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
__device__ __forceinline__ unsigned long long ordered_base(unsigned long long *status, int w, int w_begin, unsigned long long total,
                                                           unsigned max_spins, unsigned long long first) {
    const int lane = threadIdx.x & 63;
    if (w == w_begin) return 0;
    unsigned long long excl = 0;
    int idx = w - 1;
    bool use_first = true;
    for (unsigned spins = 0; spins < max_spins;) {   // (only waiting counts as a spin, walking back does not)
        const int j = idx - lane;
        unsigned long long sw = first;
        if (!use_first)
            sw = j >= w_begin ? __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : MESH_FLAG_PFX;   // in front of the shard: prefix 0
        use_first = false;
        const unsigned flag = (unsigned)(sw >> 62);
        const unsigned long long pending = __ballot(flag == 0), is_pfx = __ballot(flag == 2);
        if (is_pfx) {
            const int p = __ffsll((long long)is_pfx) - 1;                 // nearest predecessor with a prefix
            if (pending & ((1ull << p) - 1ull)) { __builtin_amdgcn_s_sleep(1); spins++; continue; }
            excl += wave_sum_u64(lane <= p ? (sw & MESH_VAL_MASK) : 0ull);
            if (lane == 0) __hip_atomic_store(&status[w], MESH_FLAG_PFX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return excl;
        }
        if (pending) { __builtin_amdgcn_s_sleep(1); spins++; continue; }
        excl += wave_sum_u64(sw & MESH_VAL_MASK);
        idx -= 64;
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

template <typename T, bool FULL, int NP, int ND, int NS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_mesh(const uint32_t *__restrict__ code, const T *__restrict__ consts, MeshArgs a) {
    typedef Vec<T, NS> V;
    constexpr int RPT = 1024 / BLOCK;   // (i0, i1) rows of cells per thread (a tile has <= 32 x 32 rows)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *wave_sums = reinterpret_cast<int *>(smem);                 // 16 ints
    int *bcast = wave_sums + 16;                                    // 16 ints of scratch
    unsigned char *ntri_lds = smem + MESH_LDS_NTRI;                 // 256 B: ntri | ambiguous << 7
    double *axes = reinterpret_cast<double *>(smem + MESH_LDS_AXES);  // 3 * 33 doubles (X, Y, Z of the tile)
    float *vol = reinterpret_cast<float *>(smem + MESH_LDS_VOL);    // (bs+1)^3 floats
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(smem + a.bits_off);   // 1 bit per sample: value > 0
    unsigned *list = reinterpret_cast<unsigned *>(smem + a.list_off);
    const int tid = threadIdx.x;
    const GridDesc g = a.g;
    const signed char *tri_tab = &a.mc->tri[0][0];

    if (tid < 256) ntri_lds[tid] = (unsigned char)(a.mc->ntri[tid] | (a.mc->amb[tid] << 7));

    const int work_begin = a.ctr->work_begin, work_end = a.ctr->work_end;
    long long tprev = a.prof ? clock64() : 0;
#define SDF_PROF(K) do { if (a.prof && tid == 0) { const long long tn = clock64(); atomicAdd(&a.prof[K], (unsigned long long)(tn - tprev)); tprev = tn; } } while (0)
    // position-dependent bookkeeping of work item w_ (thread 0)
    auto settle = [&](int w_, unsigned long long excl, unsigned long long total_) {
        if (excl == ~0ull) atomicOr(&a.ctr->overflow, 2u);           // look-back timed out (never expected)
        else if (excl + total_ > a.out_cap) atomicOr(&a.ctr->overflow, 1u);
        if (w_ == work_end - 1 && excl != ~0ull) a.ctr->total = excl + total_;
    };
    // the parked batch of this workgroup (all values workgroup-uniform)
    float *my_park = a.park ? a.park + (size_t)blockIdx.x * (size_t)a.park_cap * 9 : nullptr;
    int pend_w = -1, pend_total = 0;
    double *pend_xf = reinterpret_cast<double *>(smem + MESH_LDS_PEND);   // its offset[3], scale[3]
    auto place_parked = [&](unsigned long long pre_pend) {
        if (pend_w < 0) return;
        if (tid < 64) {
            const unsigned long long excl = ordered_base(a.status, pend_w, work_begin, (unsigned long long)pend_total, MESH_SPIN_FOREVER, pre_pend);
            if (tid == 0) {
                settle(pend_w, excl, (unsigned long long)pend_total);
                reinterpret_cast<unsigned long long *>(bcast + 4)[0] = excl;
            }
        }
        __syncthreads();
        const unsigned long long pbase = reinterpret_cast<unsigned long long *>(bcast + 4)[0];
        if (pbase != ~0ull && pbase + (unsigned long long)pend_total <= a.out_cap) {
            double *dst0 = a.out + pbase * 9ull;
            const double pof0 = pend_xf[0], pof1 = pend_xf[1], pof2 = pend_xf[2], psc0 = pend_xf[3], psc1 = pend_xf[4], psc2 = pend_xf[5];
            // 12 coordinates (four points) per thread and pass: three 16-byte loads in flight, then the
            // stores; coordinate e belongs to axis e % 3
            const int n9 = pend_total * 9, nchunk = n9 / 12;
            const double sc[3] = {psc0, psc1, psc2}, of[3] = {pof0, pof1, pof2};
            for (int ch = tid; ch < nchunk; ch += BLOCK) {
                const float4 *src = reinterpret_cast<const float4 *>(my_park) + (size_t)ch * 3;
                const float4 v0 = src[0], v1 = src[1], v2 = src[2];
                const float f[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
                double *d = dst0 + (size_t)ch * 12;
                SDF_UNROLL for (int q = 0; q < 12; q++) d[q] = (double)f[q] * sc[q % 3] + of[q % 3];
            }
            for (int e = nchunk * 12 + tid; e < n9; e += BLOCK) dst0[e] = (double)my_park[e] * sc[e % 3] + of[e % 3];
        }
        pend_w = -1;
        __syncthreads();   // (bcast is reused)
    };
    for (;;) {
        if (tid == 0) bcast[0] = work_begin + (int)atomicAdd(&a.ctr->work_counter, 1u);
        __syncthreads();
        const int w = bcast[0];
        if (w >= work_end) break;
        const int b = a.worklist[w];
        int ox, oy, oz, lx, ly, lz;
        batch_origin(g, b, ox, oy, oz, lx, ly, lz);
        if (tid < lx) axes[tid] = g.X[ox + tid];
        else if (tid >= 64 && tid < 64 + ly) axes[33 + tid - 64] = g.Y[oy + tid - 64];
        else if (tid >= 128 && tid < 128 + lz) axes[66 + tid - 128] = g.Z[oz + tid - 128];
        __syncthreads();
        SDF_PROF(0);

        // ---- 1. sample: volume = sdf(P).reshape(shape), cast to float32 (core.py:50-52) ----
        // (with the interval prepass on, this batch has its own tape with the irrelevant instructions removed)
        // (`code` stays the base of every address so that the loads remain scalar loads from a read-only
        // kernel argument; w comes out of LDS, hence the readfirstlane)
        const uint32_t *wcode = code + (size_t)__builtin_amdgcn_readfirstlane(b) * (size_t)a.tape_stride * 2;
        if (a.tape_stride && tid == 0)
            atomicAdd(&a.ctr->n_pruned, (unsigned long long)a.n_instr - reinterpret_cast<const unsigned long long *>(wcode)[a.tape_stride - 1]);
        const int nvox = lx * ly * lz;
        const int lyz = ly * lz;
        const float inv_lyz = 1.0f / (float)lyz, inv_lz = 1.0f / (float)lz;
        for (int i0 = 0; i0 < nvox; i0 += BLOCK * NS) {
            V px, py, pz;
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const int i = min(i0 + k * BLOCK + tid, nvox - 1);
                const int ix = fast_div(i, inv_lyz), r = i - ix * lyz, iy = fast_div(r, inv_lz), iz = r - iy * lz;
                px.v[k] = (T)axes[ix]; py.v[k] = (T)axes[33 + iy]; pz.v[k] = (T)axes[66 + iz];
            }
            const V val = run_tape<T, FULL, NP, ND, NS>(wcode, consts, px, py, pz);
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const int i = i0 + k * BLOCK + tid;
                const float f = (float)val.v[k];
                if (i < nvox) vol[i] = f;
                // the wave's 64 consecutive samples -> one word of the sign-bit volume (the marching
                // phases classify cells from these bits instead of re-reading 8 floats per cell)
                const unsigned long long m = __ballot(i < nvox && f > 0.0f);
                if ((tid & 63) == 0 && i < nvox) bits[i >> 6] = m;
            }
        }
        if (tid < 2) bits[((nvox + 63) >> 6) + tid] = 0ull;   // the row extraction reads one word ahead
        __syncthreads();
        SDF_PROF(1);

        // ---- 2. count: a thread owns the i2-rows of cells (i0, i1) = row tid + k * BLOCK ----
        // (wave 0 first asks for the predecessors' status words -- of this batch and of the parked one --
        // so that the answers arrive while the cells are counted)
        unsigned long long pre_own = 0, pre_pend = 0;
        if (tid < 64) {
            pre_own = lookback_prefetch(a.status, w, work_begin);
            if (pend_w >= 0) pre_pend = lookback_prefetch(a.status, pend_w, work_begin);
        }
        const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
        const int nrows = (c0 > 0 && c1 > 0 && c2 > 0) ? c0 * c1 : 0;
        const float inv_c1 = 1.0f / (float)max(c1, 1);
        int row_tris[RPT], row_off[RPT];
        unsigned row_mask[RPT];
        unsigned long long row_bits[RPT][4];   // sign bits of the four sample rows (o0, o1) of a cell row
        int total = 0, my_amb = 0;
        SDF_UNROLL
        for (int k = 0; k < RPT; k++) {
            const int r = tid + k * BLOCK;
            int n = 0;
            unsigned mask = 0;
            SDF_UNROLL for (int q = 0; q < 4; q++) row_bits[k][q] = 0ull;
            if (r < nrows) {
                const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                SDF_UNROLL
                for (int q = 0; q < 4; q++) {   // q = 2 * o0 + o1
                    const int o = (i0 + (q >> 1)) * lyz + (i1 + (q & 1)) * lz;
                    const unsigned long long w0 = bits[o >> 6], w1 = bits[(o >> 6) + 1];
                    const int sh = o & 63;
                    row_bits[k][q] = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
                }
                // a cell (columns i2, i2 + 1) has a surface unless its 8 bits are all equal
                const unsigned long long any = row_bits[k][0] | row_bits[k][1] | row_bits[k][2] | row_bits[k][3];
                const unsigned long long all = row_bits[k][0] & row_bits[k][1] & row_bits[k][2] & row_bits[k][3];
                const unsigned long long ones = all & (all >> 1), zeros = ~any & ~(any >> 1);
                mask = (unsigned)(~(ones | zeros)) & (c2 >= 32 ? 0xFFFFFFFFu : ((1u << c2) - 1u));
                unsigned m = mask;
                while (m) {
                    const int i2 = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const unsigned e = ntri_lds[cell_config(row_bits[k], i2)];
                    if (e & 128u) {   // ambiguous configuration: Lewiner's tests pick the tiling (rare)
                        double lv[8];
                        int off;
                        mc33_load_cell(vol + i0 * lyz + i1 * lz + i2, lyz, lz, lv);
                        n += mc33_cell(lv, a.mc->mc33, &off);
                        my_amb++;
                    } else {
                        n += (int)(e & 7u);
                    }
                }
            }
            row_tris[k] = n; row_mask[k] = mask;
            int tot;
            row_off[k] = total + block_exclusive_scan<BLOCK>(n, wave_sums, tot);
            total += tot;
        }
        // ---- the batch's count is public from here on; bookkeeping that needs no position ----
        if (tid < 64) publish_count(a.status, w, work_begin, (unsigned long long)total);
        if (tid == 0) {
            atomicAdd(total ? &a.ctr->n_nonempty : &a.ctr->n_empty, 1u);
            atomicAdd(&a.ctr->n_eval, (unsigned long long)nvox);
            a.kinds[b] = total ? 2 : 1;
        }
        if (my_amb) atomicAdd(&a.ctr->n_ambiguous, (unsigned long long)my_amb);
        // ---- a parked batch is older than this one: its predecessors have long published, place it ----
        { const long long tp0 = a.prof ? clock64() : 0;
        place_parked(pre_pend);
        if (a.prof && tid == 0) atomicAdd(&a.prof[6], (unsigned long long)(clock64() - tp0)); }
        // ---- ordered allocation (wave 0): take the position if every predecessor has published its
        // count, else park this batch instead of waiting for them ----
        const bool may_park = a.park && total <= a.park_cap;
        if (tid < 64) {
            const unsigned long long excl = ordered_base(a.status, w, work_begin, (unsigned long long)total, may_park ? a.park_spins : MESH_SPIN_FOREVER, pre_own);
            if (tid == 0) {
                if (excl != MESH_NOT_READY) settle(w, excl, (unsigned long long)total);
                reinterpret_cast<unsigned long long *>(bcast + 2)[0] = excl;
            }
        }
        __syncthreads();
        const unsigned long long base = reinterpret_cast<unsigned long long *>(bcast + 2)[0];
        const bool parking = base == MESH_NOT_READY;
        const bool fits = parking || (base != ~0ull && base + (unsigned long long)total <= a.out_cap);
        // points * scale + offset (reference sdf/core.py:58-60): scale = first axis step of the
        // batch, offset = its first sample, per axis
        const double of0 = axes[0], of1 = axes[33], of2 = axes[66];
        const double sc0 = axes[1] - of0, sc1 = axes[34] - of1, sc2 = axes[67] - of2;
        if (parking) {
            pend_w = w; pend_total = total;
            if (tid == 0) { pend_xf[0] = of0; pend_xf[1] = of1; pend_xf[2] = of2; pend_xf[3] = sc0; pend_xf[4] = sc1; pend_xf[5] = sc2; }
            if (a.prof && tid == 0) atomicAdd(&a.prof[7], 1ull);
        }
        SDF_PROF(2);

        // ---- 3 + 4. per-triangle work list in LDS, then one lane per triangle ----
        for (int lo = 0; fits && lo < total; lo += a.list_cap) {
            const int cn = min(a.list_cap, total - lo);
            SDF_UNROLL
            for (int k = 0; k < RPT; k++) {
                if (row_tris[k] == 0 || row_off[k] >= lo + cn || row_off[k] + row_tris[k] <= lo) continue;
                const int r = tid + k * BLOCK;
                const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                const float *row = vol + i0 * lyz + i1 * lz;
                int pos = row_off[k] - lo;
                unsigned m = row_mask[k];
                while (m) {
                    const int i2 = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const unsigned cfg = cell_config(row_bits[k], i2);
                    const unsigned en = ntri_lds[cfg];
                    int n = (int)(en & 7u);
                    if (en & 128u) {
                        double lv[8];
                        int off;
                        mc33_load_cell(row + i2, lyz, lz, lv);
                        n = mc33_cell(lv, a.mc->mc33, &off);
                    }
                    // entry: cell (15 bits) | ambiguous (1) | configuration (8) | triangle in cell (4)
                    const unsigned e = ((unsigned)((i0 << 10) | (i1 << 5) | i2) << 13) | ((en & 128u) << 5) | (cfg << 4);
                    for (int j = 0; j < n; j++, pos++)
                        if (pos >= 0 && pos < cn) list[pos] = e | (unsigned)j;
                }
            }
            __syncthreads();
            SDF_PROF(3);
            double *dst0 = a.out + (parking ? 0ull : base + (unsigned long long)lo) * 9ull;
            float *park0 = my_park + (size_t)lo * 9;
            for (int t = tid; t < cn; t += BLOCK) {
                const unsigned e = list[t];
                const int j = (int)(e & 15u), cfg = (int)((e >> 4) & 255u), cell = (int)(e >> 13);
                const int i0 = cell >> 10, i1 = (cell >> 5) & 31, i2 = cell & 31;
                const float *corner = vol + i0 * lyz + i1 * lz + i2;
                float o[9];
                if (e & 4096u) {
                    mc33_triangle(corner, lyz, lz, i0, i1, i2, a.mc->mc33, j, o);
                } else {
                    const signed char *tt = tri_tab + cfg * 16 + 3 * j;
                    mc_vertex(corner, lyz, lz, i0, i1, i2, tt[0], o);
                    mc_vertex(corner, lyz, lz, i0, i1, i2, tt[1], o + 3);
                    mc_vertex(corner, lyz, lz, i0, i1, i2, tt[2], o + 6);
                }
                if (parking) {
                    float *dst = park0 + (size_t)t * 9;
                    SDF_UNROLL for (int q = 0; q < 9; q++) dst[q] = o[q];
                } else {
                    double *dst = dst0 + (size_t)t * 9;
                    SDF_UNROLL
                    for (int q = 0; q < 9; q += 3) {
                        dst[q] = (double)o[q] * sc0 + of0;
                        dst[q + 1] = (double)o[q + 1] * sc1 + of1;
                        dst[q + 2] = (double)o[q + 2] * sc2 + of2;
                    }
                }
            }
            __syncthreads();   // list / vol are reused
            SDF_PROF(4);
        }
        __syncthreads();   // vol / bcast are reused by the next batch
    }
    place_parked(pend_w >= 0 && tid < 64 ? lookback_prefetch(a.status, pend_w, work_begin) : 0ull);
    SDF_PROF(5);
#undef SDF_PROF
}

// host-side launcher of one (T, FULL) family, defined in sdf_mesh_inst.hip (one translation
// unit per family so the variants compile in parallel).  slots: 0 = (2,2), 1 = (4,4), 2 = (8,8)
// register files; shape: 0 = 1024 threads x 1 sample per lane, 1 = 512 x 2, anything else = 1024 x 2.
#define SDF_DECLARE_MESH_LAUNCH(NAME, T) \
    int NAME(int slots, int shape, int grid, size_t lds, hipStream_t stream, const uint32_t *code, const T *consts, const MeshArgs &a)
SDF_DECLARE_MESH_LAUNCH(sdf_launch_mesh_f64, double);
SDF_DECLARE_MESH_LAUNCH(sdf_launch_mesh_f64_full, double);
SDF_DECLARE_MESH_LAUNCH(sdf_launch_mesh_f32, float);
SDF_DECLARE_MESH_LAUNCH(sdf_launch_mesh_f32_full, float);

}  // namespace sdfk
