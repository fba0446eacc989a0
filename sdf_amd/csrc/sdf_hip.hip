// sdf_hip.hip -- the tape-interpreter kernels (k_eval_*, k_skip, k_prune_list, k_cull) + the C ABI of libsdf_hip.so (gfx950
// only).  k_mesh is instantiated in sdf_mesh_inst.hip, k_estimate_bounds in sdf_bounds.hip; every kernel that is not an
// interpreter (k_compact, k_scan_items, k_emit2, k_pack_slab, k_expand, k_mc_*, k_field_*, k_cast_f32, k_stl) in sdf_plain.hip.
//
// Kernels (one call of sdf_generate enqueues k_skip -> k_compact [-> k_prune_list] -> k_cull -> k_mesh
// [-> k_scan_items -> k_emit2] on one stream, without a host round trip in between)
//   k_eval_points / k_eval_grid   f(P): the tape interpreter alone; k_eval_points_ext: with user closures (L_EXTERN)
//   k_eval_tiles                  the float32 volumes of a chunk of batches of more than 33^3 samples (batch_size > 32: generate_big)
//   k_estimate_bounds             the reference's `_estimate_bounds` loop (sdf/core.py:62-82) as one launch (sdf_bounds.hip)
//   k_skip                        the reference's `_skip` predicate for every batch at once
//                                 (reference sdf/core.py:28-43), 9 lanes per batch, 7 batches per wave; its surplus
//                                 workgroups run the interval pruning pass of the same batches (sdf_prune.h)
//   k_compact                     ordered work list of the surviving batches; clears the counters / look-back words
//   k_prune_list                  the pruning pass for the survivors only (grids with many batches)
//   k_cull / k_cull_lean          per surviving batch: the sampling tasks that have to be evaluated, by interval
//                                 arithmetic over groups of 4^3 cells (cull_tasks, sdf_device.h)
//   k_mesh (sdf_device.h)         THE hot kernel: one persistent workgroup per CU pulls batches
//                                 from the work list; samples the (<=33)^3 tile through the tape
//                                 interpreter (float64 -> float32 like skimage's cast) straight
//                                 into LDS (143,748 B of gfx950's 160 KiB), classifies the cells,
//                                 finds the batch's place in the ordered soup by a look-back over
//                                 the earlier batches' counts and writes the float64 world-space
//                                 triangles in reference order (reference `_worker`,
//                                 sdf/core.py:45-60, and `points.extend`, :141)
//   k_scan_items / k_emit2        two-pass meshing (long tapes): k_mesh stops at the classification, these number
//                                 and write the triangles
//   k_pack_slab / k_expand        multi-GPU exchange: a shard's soup as a fixed-capacity slab / the gathered slabs
//                                 expanded into the ordered float64 soup
//   k_mc_rows / k_mc_emit         marching cubes of a caller-supplied volume (`_marching_cubes`);
//   k_cast_f32 / k_field_*        the same for the batches of a host-evaluated field (user closures)
//   k_stl                         50-byte STL records (reference sdf/stl.py:4-24)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sdf_hip.h"
#include "mc_table.h"
#include "sdf_device.h"
#include "sdf_mesh2.h"
#include "sdf_prune.h"
#include "sdf_slab.h"
#include "sdf_expand_host.h"
#include "sdf_plain.h"
#include "sdf_bounds.h"

using namespace sdfk;

// ============================================================================================
// device side (the fused sample+march kernel k_mesh lives in sdf_device.h)
// ============================================================================================

template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_eval_points(const uint32_t *__restrict__ code, const T *__restrict__ consts,
                                                     const double *__restrict__ pts, long long n, int dim,
                                                     double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T x = (T)pts[i * dim], y = (T)pts[i * dim + 1], z = dim > 2 ? (T)pts[i * dim + 2] : T(0);
    out[i] = (double)run_tape1<T, FULL>(code, consts, x, y, z);
}

template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_eval_grid(const uint32_t *__restrict__ code, const T *__restrict__ consts,
                                                   const double *__restrict__ X, const double *__restrict__ Y,
                                                   const double *__restrict__ Z, int nx, int ny, int nz,
                                                   double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)nx * ny * nz;
    if (i >= n) return;
    const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((long long)nz * ny));
    out[i] = (double)run_tape1<T, FULL>(code, consts, (T)X[ix], (T)Y[iy], (T)Z[iz]);
}

// `volume = sdf(P).reshape(...)`, cast to float32 as skimage does (reference sdf/core.py:50-54), for a chunk of whole tiles in
// device memory: batch_size > 32, whose (batch_size + 1)^3 tile does not fit the LDS of a compute unit (generate_big).  One
// lane per sample; blockIdx.y = the tile (FieldTile: its place in `vol`, its extents), `org` its first sample per axis.
template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_eval_tiles(const uint32_t *__restrict__ code, const T *__restrict__ consts,
                                                    const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                                                    const FieldTile *__restrict__ tiles, const int *__restrict__ org, float *__restrict__ vol) {
    const FieldTile tl = tiles[blockIdx.y];
    const long long n = (long long)tl.n0 * tl.n1 * tl.n2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int iz = (int)(i % tl.n2), iy = (int)((i / tl.n2) % tl.n1), ix = (int)(i / ((long long)tl.n2 * tl.n1));
    const int *o = org + 3 * blockIdx.y;
    vol[tl.vol_off + i] = (float)run_tape1<T, FULL>(code, consts, (T)X[o[0] + ix], (T)Y[o[1] + iy], (T)Z[o[2] + iz]);
}

// f(P) for tapes with user closures (L_EXTERN leaves): `dump` writes every leaf's current point for the host
// to call the closure on, the second pass reads the closures' values (sdf_interp.h ExtIO)
template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_eval_points_ext(const uint32_t *__restrict__ code, const T *__restrict__ consts,
                                                         const double *__restrict__ pts, long long n, int dim,
                                                         double *__restrict__ ext, int dump, double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T x = (T)pts[i * dim], y = (T)pts[i * dim + 1], z = dim > 2 ? (T)pts[i * dim + 2] : T(0);
    const T v = run_tape1_ext<T, FULL>(code, consts, x, y, z, ExtIO{ext, n, i, dump != 0});
    if (!dump) out[i] = (double)v;
}

// (k_estimate_bounds -- `_estimate_bounds`, reference sdf/core.py:62-82, as one launch -- lives in sdf_bounds.hip, built next to this unit)

// reference sdf/core.py:28-43.  9 lanes per batch, 7 batches per wave (lane 63 idles): lane 0 of a batch = centre,
// lanes 1..8 = corners in itertools.product((x0,x1),(y0,y1),(z0,z1)) order.  kinds[b] = 0 (skipped) or 255 (pending).
// (16 lanes per batch, 7 of them idle, until r02p: the kernel is the tape at 9 points per batch and nothing else.)
enum { SKIP_BATCHES_PER_BLOCK = 7 * (256 / 64) };
// Workgroups >= pa.first_block run the interval pass of the same batches instead (sdf_prune.h).
template <typename T, bool FULL, bool RARE>
__device__ __forceinline__ void skip_body(const uint32_t *__restrict__ code, const T *__restrict__ consts, GridDesc g,
                                              int nbatches, unsigned char *__restrict__ kinds, const PruneArgs &pa,
                                              const double *__restrict__ c64, const uint16_t *__restrict__ rstart,
                                              const uint16_t *__restrict__ lstart, double *prune_lds, int skip_b0) {
    if ((int)blockIdx.x >= pa.first_block) {   // (uniform)
        prune_block<FULL, RARE>(code, c64, rstart, lstart, pa, g, nbatches, (int)blockIdx.x - pa.first_block, prune_lds);
        return;
    }
    const int lane = threadIdx.x & 63, slot = min(lane / 9, 6), l = lane - 9 * slot;    // (lane 63: l = 9, idle)
    // (the skip test of batches [skip_b0, nbatches): a rank of a multi-GPU job tests its share only, sdf_skip_kinds)
    const int b = skip_b0 + ((int)blockIdx.x * (256 / 64) + (int)(threadIdx.x >> 6)) * 7 + slot;
    const bool live = b < nbatches && l < 9;
    int ox = 0, oy = 0, oz = 0, lx = 1, ly = 1, lz = 1;
    if (b < nbatches) batch_origin(g, b, ox, oy, oz, lx, ly, lz);
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0;
    if (b < nbatches) {
        x0 = g.X[ox]; x1 = g.X[ox + lx - 1]; y0 = g.Y[oy]; y1 = g.Y[oy + ly - 1]; z0 = g.Z[oz]; z1 = g.Z[oz + lz - 1];
    }
    const double cx = (x0 + x1) / 2, cy = (y0 + y1) / 2, cz = (z0 + z1) / 2;
    double px = cx, py = cy, pz = cz;
    if (l >= 1) { const int k = l - 1; px = (k & 4) ? x1 : x0; py = (k & 2) ? y1 : y0; pz = (k & 1) ? z1 : z0; }
    T v = T(0);
    if (live) v = run_tape1<T, FULL>(code, consts, (T)px, (T)py, (T)pz);
    const int base = 9 * slot;
    const T vc = __shfl(v, base, 64);          // centre
    const T v1 = __shfl(v, base + 1, 64);      // values[0]
    const bool pos = v1 > T(0);
    const bool ok = pos ? (v > T(0)) : (v < T(0));
    const unsigned long long m = __ballot(ok);
    const bool all_same = ((m >> (base + 1)) & 0xFFull) == 0xFFull;
    if (b < nbatches && l == 0) {
        const double r = fabs((double)vc);
        const double d = sqrt(((cx - x0) * (cx - x0) + (cy - y0) * (cy - y0)) + (cz - z0) * (cz - z0));
        const bool skip = !(r <= d) && all_same;
        kinds[b] = skip ? 0 : 255;
    }
}

// (two entry points: tapes with one of the less common leaves of ia_leaf_rare get the interval pass that knows
// them, the others keep the leaner one -- a kernel's registers and scratch are those of its hungriest callee)
template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_skip(const uint32_t *__restrict__ code, const T *__restrict__ consts, GridDesc g,
                                              int nbatches, unsigned char *__restrict__ kinds, PruneArgs pa,
                                              const double *__restrict__ c64, const uint16_t *__restrict__ rstart, const uint16_t *__restrict__ lstart,
                                              int skip_b0) {
    extern __shared__ double prune_lds[];
    skip_body<T, FULL, false>(code, consts, g, nbatches, kinds, pa, c64, rstart, lstart, prune_lds, skip_b0);
}
template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_skip_rare(const uint32_t *__restrict__ code, const T *__restrict__ consts, GridDesc g,
                                                   int nbatches, unsigned char *__restrict__ kinds, PruneArgs pa,
                                                   const double *__restrict__ c64, const uint16_t *__restrict__ rstart, const uint16_t *__restrict__ lstart,
                                                   int skip_b0) {
    extern __shared__ double prune_lds[];
    skip_body<T, FULL, true>(code, consts, g, nbatches, kinds, pa, c64, rstart, lstart, prune_lds, skip_b0);
}

// The interval pass of sdf_prune.h as a kernel of its own, over this shard's work list: for grids with many
// batches, most of which the skip test removes (weave at 2^33: 266 k batches, 38 k survive), pruning only the
// survivors is worth the extra launch behind k_compact; small grids keep it fused into k_skip's launch, where
// it costs nothing on the critical path.
template <bool FULL, bool RARE>
__global__ __launch_bounds__(PRUNE_BLOCK) void k_prune_list(const uint32_t *__restrict__ code, GridDesc g, int nbatches, PruneArgs pa,
                                                               const double *__restrict__ c64, const uint16_t *__restrict__ rstart,
                                                               const uint16_t *__restrict__ lstart) {
    extern __shared__ double prune_list_lds[];
    if ((long long)blockIdx.x * (PRUNE_BLOCK / 8) >= (long long)(pa.ctr->work_end - pa.ctr->work_begin)) return;   // (uniform)
    prune_block<FULL, RARE>(code, c64, rstart, lstart, pa, g, nbatches, (int)blockIdx.x, prune_list_lds);
}

// One workgroup per work item of this shard: which sampling tasks of the batch have to be evaluated
// (cull_tasks, sdf_device.h).  The record goes to global memory; k_mesh picks it up.
#define CULL_BLOCK 256
template <bool FULL, bool RARE, int CB = CULL_BLOCK>
__device__ __forceinline__ void cull_body(const uint32_t *__restrict__ code, const double *__restrict__ consts, GridDesc g,
                                                     const int *__restrict__ worklist, const MeshCounters *__restrict__ ctr,
                                                     int tape_stride, int n_instr, int ia_np, int ia_nd, int ia_bytes,
                                                     unsigned char *__restrict__ out, unsigned char *cull_smem, unsigned long long *prof,
                                                     int *__restrict__ order, int tail_max, int levels) {
    int *wave_sums = reinterpret_cast<int *>(cull_smem);                       // 64 B
    double *axes = reinterpret_cast<double *>(cull_smem + 64);                 // 3 * 33 doubles
    unsigned char *scratch = cull_smem + 896;                                  // CULL_SCRATCH bytes
    double *ia_state = reinterpret_cast<double *>(cull_smem + 896 + CULL_SCRATCH);
    const int tid = threadIdx.x;
    const long long tstart = prof ? clock64() : 0;
    // (one workgroup per BATCH is launched -- the host does not know the length of the work list -- and the surplus ones
    // leave here.  Striding over the list with a grid sized for the compute units was measured in r02: the loop costs the
    // kernel 10 - 20 % (example 60 -> 70 us, gearlike 2^30 prepass 0.36 -> 0.43 ms), the empty workgroups nothing.)
    // (the grid's pointers and sizes out of the kernel-argument segment NOW, next to the counters' load: the compiler
    // fetches a by-value struct's fields where they are first used, i.e. one dependent scalar-memory trip at a time)
    asm volatile("" :: "s"(g.X), "s"(g.Y), "s"(g.Z), "s"(g.nx), "s"(g.ny), "s"(g.nz), "s"(g.nby), "s"(g.nbz), "s"(g.bs));
    const int w = ctr->work_begin + (int)blockIdx.x;
    if (w >= ctr->work_end) return;
    long long t_ctr = 0, t_b = 0;
    if (prof) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_ctr) : "s"(w) : "memory");
    const int b = __builtin_amdgcn_readfirstlane(worklist[w]);   // (uniform: the tape is then read with scalar loads)
    if (prof) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b) : "s"(b) : "memory");
    int ox, oy, oz, lx, ly, lz;
    batch_origin(g, b, ox, oy, oz, lx, ly, lz);
    long long t_org = 0;
    if (prof) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_org) : "s"(ox), "s"(oy), "s"(oz), "s"(lx) : "memory");
    for (int i = tid; i < 99; i += CB) {   // (ONE load per lane whatever the axis: three branches were three loads in a row for the first wave)
        const int ax = i < 33 ? 0 : (i < 66 ? 1 : 2), k = i - 33 * ax;
        const double *src = ax == 0 ? g.X + ox : (ax == 1 ? g.Y + oy : g.Z + oz);
        if (k < (ax == 0 ? lx : (ax == 1 ? ly : lz))) axes[i] = src[k];
    }
    long long t_ax = 0, t_bar = 0;
    if (prof) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_ax) : "s"(b) : "memory");
    __syncthreads();
    if (prof) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_bar) : "s"(b) : "memory");
    const uint32_t *wcode = code + (size_t)b * (size_t)tape_stride * 2;
    const int n_instr_w = tape_stride ? (int)reinterpret_cast<const unsigned long long *>(wcode)[tape_stride - 1] : n_instr;
    long long tstart1 = 0;   // (profiling: the clock once the work item, its axes and the length of its tape have arrived)
    if (prof) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstart1) : "s"(n_instr_w) : "memory");
    const int ntl = cull_tasks<CB, FULL, RARE>(wcode, consts, n_instr_w, lx, ly, lz, axes, ia_state, ia_bytes, scratch, wave_sums, ia_np, ia_nd, prof, levels);
    const long long tw = prof ? clock64() : 0;
    if (tid == 0 && ntl < 0) reinterpret_cast<unsigned short *>(scratch)[0] = (unsigned short)0xFFFF;   // (else: the number of listed units, cull_tasks)
    // (what the host needs to know before it may hand the NEXT call of this tape on this grid to k_mesh2: rarely written)
    if (tid == 0 && (ntl > MESH2_NTL_MAX || (ntl < 0 && lx > 1 && ly > 1 && lz > 1))) const_cast<MeshCounters *>(ctr)->not_mesh2 = 1;
    __syncthreads();
    {   // the record: header + the listed units (whole tasks), and the sub-group states
        unsigned *rec = reinterpret_cast<unsigned *>(out + (size_t)w * CULL_RECORD);
        const unsigned *src = reinterpret_cast<const unsigned *>(scratch);
        const int nwords = ntl < 0 ? 1 : (CULL_ULIST + 16 * ntl + 3) >> 2;
        for (int i = tid; i < nwords; i += CB) rec[i] = src[i];
        if (ntl >= 0) for (int i = tid; i < (CULL_RECORD - CULL_SSTATE) / 4; i += CB) rec[CULL_SSTATE / 4 + i] = src[CULL_SSTATE / 4 + i];   // sub-group states + column words
    }
    // ---- every work item of the tail of the list leaves its cost estimate for k_mesh, which hands the tail out by
    // descending cost (MeshArgs::order) ----
    if (order && tid == 0) {
        const int tail = min(tail_max, ctr->work_end - ctr->work_begin), tpos = w - (ctr->work_end - tail);
        if (tpos >= 0) order[tpos] = (ntl < 0 ? 563 : ntl) * max(n_instr_w, 1);
    }
    if (prof && tid == 0) {
        const unsigned *pacc = reinterpret_cast<const unsigned *>(scratch + CULL_PACC);
        atomicAdd(&prof[32 + (ntl < 0 ? 9 : min(ntl >> 6, 8))], 1ull);   // histogram of the listed tasks per work item, bins of 64
        atomicAdd(&prof[16], (unsigned long long)(tstart1 - tstart));
        atomicAdd(&prof[26], (unsigned long long)(t_ctr - tstart));      // ... of which: until the shard's range is known
        atomicAdd(&prof[27], (unsigned long long)(t_b - t_ctr));         // ... until the batch index is
        atomicAdd(&prof[28], (unsigned long long)(t_ax - t_org));        // ... until this wave's axis values are in LDS
        atomicAdd(&prof[30], (unsigned long long)(t_org - t_b));         // ... (before that: the batch's origin from its index)
        atomicAdd(&prof[29], (unsigned long long)(t_bar - t_ax));        // ... its wait at the barrier (the workgroup's other waves)
        atomicAdd(&prof[21], (unsigned long long)(clock64() - tw));
        for (int k = 1; k < 10; k++) if (k != 5) atomicAdd(&prof[16 + k], (unsigned long long)pacc[k]);
    }
}

template <bool FULL, bool RARE, int CB = CULL_BLOCK>
__global__ __launch_bounds__(CB) void k_cull(const uint32_t *__restrict__ code, const double *__restrict__ consts, GridDesc g,
                                                     const int *__restrict__ worklist, const MeshCounters *__restrict__ ctr,
                                                     int tape_stride, int n_instr, int ia_np, int ia_nd, int ia_bytes,
                                                     unsigned char *__restrict__ out, unsigned long long *prof,
                                                     int *__restrict__ order, int tail_max, int levels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cull_smem[];
    cull_body<FULL, RARE, CB>(code, consts, g, worklist, ctr, tape_stride, n_instr, ia_np, ia_nd, ia_bytes, out, cull_smem, prof, order, tail_max, levels);
}
// the variant for tapes without trigonometry and without the rarer leaves: 70 VGPRs without spilling, seven waves per
// SIMD (the others take 99 - 104; holding them to five or six waves was measured in r02p: no faster)
__global__ __launch_bounds__(CULL_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_cull_lean(const uint32_t *__restrict__ code, const double *__restrict__ consts, GridDesc g,
                                                     const int *__restrict__ worklist, const MeshCounters *__restrict__ ctr,
                                                     int tape_stride, int n_instr, int ia_np, int ia_nd, int ia_bytes,
                                                     unsigned char *__restrict__ out, unsigned long long *prof,
                                                     int *__restrict__ order, int tail_max, int levels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cull_smem[];
    cull_body<false, false>(code, consts, g, worklist, ctr, tape_stride, n_instr, ia_np, ia_nd, ia_bytes, out, cull_smem, prof, order, tail_max, levels);
}
// (experiment: the same with two waves per workgroup -- twelve workgroups fit a CU, every work item of the 512^3
// example is resident at once instead of in two rounds)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_cull_lean128(const uint32_t *__restrict__ code, const double *__restrict__ consts, GridDesc g,
                                                     const int *__restrict__ worklist, const MeshCounters *__restrict__ ctr,
                                                     int tape_stride, int n_instr, int ia_np, int ia_nd, int ia_bytes,
                                                     unsigned char *__restrict__ out, unsigned long long *prof,
                                                     int *__restrict__ order, int tail_max, int levels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cull_smem[];
    cull_body<false, false, 128>(code, consts, g, worklist, ctr, tape_stride, n_instr, ia_np, ia_nd, ia_bytes, out, cull_smem, prof, order, tail_max, levels);
}

// (every kernel that is not a tape interpreter -- the compaction, marching cubes of caller-supplied volumes, the two-pass
// meshing's scan and emission, the slab kernels, the STL records -- lives in sdf_plain.hip: see build.sh for why)

// ============================================================================================
// host side: the C ABI
// ============================================================================================

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return 1; }
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));          \
    } while (0)

// Waiting for the device WITHOUT going to sleep.  hipStreamSynchronize / hipEventSynchronize block on an interrupt
// after a short active wait; on a virtualised host the wake-up costs milliseconds (BENCH_r02: a synchronous 512^3 call
// took 2.2 ms on the driver's box against 0.4 ms of device work).  The calls of this library last 0.3 - 40 ms, so the
// host polls the completion signal (hipStreamQuery / hipEventQuery read it directly) for up to g_spin_us microseconds
// and only then falls back to the blocking wait.  SDF_WAIT_SPIN_US=0 restores the blocking behaviour.
static long g_spin_us = [] { const char *e = getenv("SDF_WAIT_SPIN_US"); return e ? atol(e) : 100000L; }();
template <typename Query, typename Block>
static hipError_t spin_then_block(Query query, Block block) {
    if (g_spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned n = 0;; n++) {
            const hipError_t e = query();
            if (e != hipErrorNotReady) return e;
            if ((n & 63u) == 63u &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > g_spin_us)
                break;
            __builtin_ia32_pause();
        }
    }
    return block();
}
// Entering the library: select the context's device and DROP whatever error another library left in this thread's
// "last error" slot (hipGetLastError is sticky per thread: a failed probe inside RCCL or torch -- seen after
// destroy_process_group: "invalid device ordinal" -- would otherwise be reported by our next launch check)
static hipError_t set_device(int device) {
    const hipError_t e = hipSetDevice(device);
    (void)hipGetLastError();
    return e;
}
static hipError_t stream_wait(hipStream_t s) {
    return spin_then_block([&] { return hipStreamQuery(s); }, [&] { return hipStreamSynchronize(s); });
}
static hipError_t event_wait(hipEvent_t ev) {
    return spin_then_block([&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); });
}

// Every allocation of this translation unit goes through these two, so that the tests can make the n-th one fail
// (sdf_test_fail_alloc) and check that every error path hands back what it had taken.
static int g_fail_alloc_in = 0;      // > 0: the g_fail_alloc_in-th allocation from now fails once
static bool g_alloc_hook_hit = false;
static bool alloc_fails_now() { g_alloc_hook_hit = g_fail_alloc_in > 0 && --g_fail_alloc_in == 0; return g_alloc_hook_hit; }
static hipError_t dev_malloc(void **p, size_t bytes) { if (alloc_fails_now()) { *p = nullptr; return hipErrorOutOfMemory; } return hipMalloc(p, bytes); }
static hipError_t host_malloc(void **p, size_t bytes) { if (alloc_fails_now()) { *p = nullptr; return hipErrorOutOfMemory; } return hipHostMalloc(p, bytes, hipHostMallocDefault); }

// Device allocations are recycled through a small per-device free list: hipMalloc / hipFree cost
// tens of microseconds each (and hipFree synchronises), which at ~1 ms per generate call was 10 %
// of the step when every mesh allocated and freed its seven buffers.
static const bool g_pool_trace = getenv("SDF_POOL_TRACE") != nullptr;   // every hipMalloc / hipFree behind the pool, to stderr
struct DevPool {
    struct Blk { void *p; size_t bytes; int device; };
    std::vector<Blk> free_list;
    std::mutex mu;
    void *take(size_t need, int device, size_t *got) {
        std::lock_guard<std::mutex> g(mu);
        int best = -1;
        for (int i = 0; i < (int)free_list.size(); i++) {
            const Blk &b = free_list[i];
            if (b.device != device || b.bytes < need || b.bytes > std::max<size_t>(4 * need, 1 << 16)) continue;
            if (best < 0 || b.bytes < free_list[best].bytes) best = i;
        }
        if (best < 0) return nullptr;
        void *p = free_list[best].p;
        *got = free_list[best].bytes;
        free_list.erase(free_list.begin() + best);
        return p;
    }
    void give(void *p, size_t bytes, int device) {
        std::lock_guard<std::mutex> g(mu);
        // (up to eight calls in flight x up to twelve buffers each come back at once: a list shorter than that evicts -- hipFree, a
        // device synchronisation -- blocks the very next call allocates again)
        if (free_list.size() >= 160) {   // evict the oldest block
            if (g_pool_trace) fprintf(stderr, "[sdf pool] evict %zu bytes (hipFree)\n", free_list.front().bytes);
            (void)hipFree(free_list.front().p);
            free_list.erase(free_list.begin());
        }
        free_list.push_back({p, bytes, device});
    }
    void drop_device(int device) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < free_list.size();) {
            if (free_list[i].device == device) { (void)hipFree(free_list[i].p); free_list.erase(free_list.begin() + i); }
            else i++;
        }
    }
};
static DevPool g_pool;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int device = -1;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        release();
        int dev = 0;
        (void)hipGetDevice(&dev);
        size_t want = std::max(need, (size_t)256);
        want = (want + 255) & ~(size_t)255;
        size_t got = 0;
        if (void *q = g_pool.take(want, dev, &got)) { p = q; bytes = got; device = dev; return 0; }
        if (g_pool_trace) fprintf(stderr, "[sdf pool] miss %zu bytes (hipMalloc)\n", want);
        hipError_t e = dev_malloc(&p, want);
        if (e != hipSuccess && !g_alloc_hook_hit) {   // give the cached blocks back to the driver and retry once
            g_pool.drop_device(dev);
            e = dev_malloc(&p, want);
        }
        if (e != hipSuccess) { p = nullptr; return fail(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); }
        bytes = want; device = dev;
        return 0;
    }
    void release() { if (p) g_pool.give(p, bytes, device); p = nullptr; bytes = 0; }
};

#define SDF_STAGE_BYTES (1u << 20)
// Calls in flight on one context (sdf_generate_to_device_async): each owns a slot = its pinned staging
// (axes on the way in, counters on the way out) and its events; a slot is reused only after the call that
// held it has completed.
// (eight since r04: with six calls in flight the 512^3 example steps in 0.237 ms, with four in 0.248, same box alternating;
// a lane's park slots -- 1.2 GB -- are allocated when the lane is first used)
#ifndef SDF_CALL_SLOTS
#define SDF_CALL_SLOTS 8
#endif
struct CallSlot {
    hipEvent_t e0 = nullptr, e2 = nullptr, e3 = nullptr, e4 = nullptr;   // start, prepass end, k_mesh start (re-runs), k_mesh end
    hipEvent_t done = nullptr;                                            // behind the counters' copy to the host
    bool busy = false;
    struct sdf_mesh *owner = nullptr;                                     // the in-flight mesh whose counters / events the slot holds
    hipStream_t stream = nullptr;                                         // the lane asynchronous calls of this slot run on
    DevBuf park;                                                          // ... and its k_mesh staging slots
};
#define SDF_BATCH_SIZE_MAX 512   // (513^3 float32 = 540 MB per tile: generate_big takes one tile per submission there)
#define SDF_PARK_TRIS 8192   // triangles per workgroup staging slot of k_mesh (36 bytes each); larger batches wait instead
                             // (16 slots per workgroup: 1.2 GB per call lane, allocated on a lane's first use; with 4096
                             // per slot weave at 2^33 has batches that cannot park: 30.3 instead of 27.7 ms)

struct sdf_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev[8] = {};
    int n_cu = 256;
    size_t lds_max = 0;
    DevBuf scratch_in, scratch_out, rows, rows_off, mc;
    DevBuf ext;                       // closure points / values of sdf_eval_*extern* (L_EXTERN leaves)
    DevBuf field_vals, field_vol, field_tiles;   // sdf_generate_field: a chunk's sampled values (f64), volumes (f32), tile table
    int mesh2 = 0;                    // SDF_MESH2: k_mesh2 (two workgroups of 512 threads per CU) 0 never (default: measured in r06, bit-identical
                                      // and 4 - 6 % slower than k_mesh at 512^3, profiles/r06e_two_wg.json) / 1 whenever the tape has a variant
                                      // (a tile it does not hold is flagged and the call repeated) / -1 when the last call of the tape
                                      // on the same grid says every tile is its
    DevBuf prof;                      // SDF_MESH_PROF=1: per-phase cycle counters of k_mesh (diagnostics)
    int prune = 1;                    // SDF_PRUNE=0 switches the interval prepass off (diagnostics)
    int parking = 1;                  // SDF_PARK=0: k_mesh waits for its predecessors instead of parking a batch (diagnostics)
    int cull = 1;                     // SDF_CULL=0: k_mesh samples every voxel of a batch instead of deciding cell groups by intervals
    int prune_list_min = 8192;        // SDF_PRUNE_LIST_MIN: from this many batches on the interval prepass runs behind k_compact, over the work list
    int park_spins = 1;               // SDF_PARK_SPINS: polls before parking (tuning; measured: waiting never pays)
    DevBuf park;                      // k_mesh's staging slots, one per CU (allocated by the first sdf_generate)
    int mesh_slots = -1;              // SDF_MESH_SLOTS override of the register-file variant (tuning)
    std::vector<DevBuf> arena_pool;   // soup buffers handed back by destroyed meshes
    std::vector<DevBuf> counter_pool; // 64-byte MeshCounters blocks handed back by destroyed meshes
    void *h_stage = nullptr;          // pinned host staging, SDF_CALL_SLOTS x SDF_STAGE_BYTES
    CallSlot slots[SDF_CALL_SLOTS];
    unsigned slot_seq = 0;
    int slot_streams = 1;             // SDF_SLOT_STREAMS=0: asynchronous calls stay on the context's stream (diagnostics)
    int cull_block = 0;               // SDF_CULL_BLOCK=64 / 128 / 256: threads per work item of k_cull (0: the default of the variant)
    int tail_order = 1;               // SDF_TAIL_ORDER=0: k_mesh takes the whole work list in order
    int twopass = -1;                 // SDF_MESH_TWOPASS=0 / 1: force the one-pass k_mesh (look-back + parking) resp. k_mesh / k_scan_items / k_emit2
    int defer = 1;                    // SDF_DEFER=0: k_mesh keeps every tile dense and writes (or parks) a batch's triangles right after counting it
    int cull_levels = 0;              // SDF_CULL_LEVELS=2 / 3: interval levels of k_cull (3: + sub-groups of 2^3 cells); 0: by the tape (see generate_impl)
    DevBuf bounds_work;               // k_estimate_bounds_w: the waves' exchange words (tagged per call, sdf_bounds.hip)
    unsigned bounds_tag = 0, bounds_tag0 = 0;
    // sdf_generate_records: what the last call of a MODEL (content hash) on a grid needed -- triangles, raw-area triangles -- so that the
    // next one, possibly through a fresh tape object of the same model, can size its slab without a host round trip
    struct RecHint { unsigned long long tris = 0, raw = 0; };
    std::map<std::pair<unsigned long long, unsigned long long>, RecHint> rec_hints;
    void *h_rec = nullptr;            // pinned staging of sdf_mesh_emit_host_workers: a slab's head, raw area and records on their way to the host threads
    size_t h_rec_bytes = 0;
    std::vector<hipEvent_t> rec_ev;   // ... one event per piece of the copy
};

struct sdf_tape {
    sdf_ctx *ctx = nullptr;
    uint32_t *d_code = nullptr;
    double *d_c64 = nullptr;
    float *d_c32 = nullptr;
    uint32_t n_words = 0, n_consts = 0;
    bool full = false;
    uint32_t n_p = 0, n_d = 0;
    uint16_t *d_rstart = nullptr, *d_lstart = nullptr;   // operand ranges of the prunable combines (or NULL)
    bool ia_complete = false;                            // every op has an interval form (sdf_interval.h ia_has_form)
    bool ia_rare = false;                                // ... one of them a leaf of ia_leaf_rare (the k_cull variant that knows them)
    uint32_t n_extern = 0;                               // user closures the tape reads through L_EXTERN leaves (sdf_eval_points_extern_*)
    unsigned long long hint_key = 0, hint_total_tris = 0;   // arena sizing: last call of this tape
    // which meshing kernel the next call on the same grid takes (k_mesh2: two workgroups per CU): 0 unknown, 1 every tile of the
    // last call was k_mesh2's (k_cull's verdict, MeshCounters::not_mesh2), 2 not so, 3 k_mesh2 ran and flagged a tile (sticky)
    unsigned long long mesh2_key = 0;
    int mesh2_state = 0;
    unsigned long long content_hash = 0;                    // FNV-1a of the code words and the constants' bits: what identifies the MODEL,
                                                            // on every rank alike and whatever address the tape object lands on (sdf_comm.inc)
};

struct sdf_mesh {
    sdf_ctx *ctx = nullptr;
    sdf_stats st = {};
    GridDesc g = {};
    DevBuf axes, kinds, worklist, status, out, prune, tapes, cull, order;
    DevBuf desc, cellrecs, trilist;   // two-pass meshing: per work item / per surface cell / per triangle (sdf_device.h ItemDesc)
    DevBuf blockidx;                  // ... and per 256 triangles of the soup: the work item of the first of them
    bool pruned = false;
    bool used_mesh2 = false;       // the meshing pass was k_mesh2's (two workgroups per CU)
    hipStream_t stream = nullptr;  // the stream the generating call ran on (the context's, or a call slot's lane)
    DevBuf counters;               // this call's MeshCounters block (pooled in the context)
    int work_begin = 0, work_end = 0;
    void *emitted_to = nullptr;    // caller buffer the soup was gathered into by sdf_generate_to_device
    // sdf_generate_to_device_async: everything sdf_mesh_wait needs to finish the call
    struct Pending {
        bool active = false;
        sdf_tape *tape = nullptr;
        int slot = 0, nb = 0, bs = 0, sparse = 0, precision = 0;
        bool pruning = false, own_start = false, compact = false;
        uint32_t n_instr = 0;
        unsigned long long key = 0;
        void *d_out = nullptr;
        int64_t cap_out = 0, shard_index = 0, shard_count = 1;
        std::vector<double> axes;      // host copy (a soup that does not fit is re-run synchronously)
        int nx = 0, ny = 0, nz = 0;
    } pend;
    double *weld_pts = nullptr;    // sdf_mesh_weld: unique rows / row -> unique row (hipMalloc'ed by sdf_weld.hip)
    long long *weld_inv = nullptr;
    long long weld_n = -1;
    // sdf_generate_records: the triangles were written as 16-byte records into a slab of the library's (sdf_slab.h); the float64 soup
    // is made on the host threads (sdf_mesh_emit_host_workers) or, for the readers that want it on the device, by k_expand on demand
    bool records = false;
    DevBuf slab;
    long long slab_items = 0, slab_tris = 0, n_raw = 0;
    bool rec_overflow = false;     // the slab (or its raw area) was too small: rec_need_tris is the capacity that holds the call
    long long rec_need_tris = 0;
};

namespace sdfk {
int weld_device(hipStream_t stream, const double *pts, long long n, double **d_uniq, long long **d_inv, long long *n_unique);   // sdf_weld.hip
}

static bool tape_needs_full(const uint32_t *code, uint32_t n_words, const double *consts) {
    auto trig_ease = [](int id) {
        return id == EASE_in_sine || id == EASE_out_sine || id == EASE_in_out_sine || id == EASE_in_expo ||
               id == EASE_out_expo || id == EASE_in_out_expo || id == EASE_in_elastic || id == EASE_out_elastic ||
               id == EASE_in_out_elastic;
    };
    for (uint32_t i = 0; i + 1 < n_words; i += 2) {
        const uint32_t op = code[i] & 255u;
        const double *c = consts + (code[i + 1] & 0xFFFFFFu) + 1;
        switch (op) {
        case OP_TWIST: case OP_BEND: case OP_BEND_RADIAL: case OP_WRAP_AROUND: case OP_CIRC_PREP: case OP_CIRC_SET:
        case OP_TRANS_RAD_PRE:
            return true;
        case OP_BEND_LINEAR: if (trig_ease((int)c[10])) return true; break;
        case OP_TRANS_LIN_PRE: if (trig_ease((int)c[7])) return true; break;
        case OP_EXTTO_PRE: if (trig_ease((int)c[1])) return true; break;
        default: break;
        }
    }
    return false;
}

static int validate_tape(const uint32_t *code, uint32_t n_words, uint32_t n_consts, uint32_t n_p, uint32_t n_d) {
    if (n_words < 2 || (n_words & 1)) return fail("tape: code must be a non-empty list of 2-word instructions");
    if (n_p > SDF_NP_SLOTS || n_d > SDF_ND_SLOTS) return fail("tape: model needs more register slots than this build provides");
    if ((code[n_words - 2] & 255u) != OP_END) return fail("tape: missing END");
    for (uint32_t i = 0; i < n_words; i += 2) {
        const uint32_t w0 = code[i], w1 = code[i + 1], op = w0 & 255u, post = (w0 >> 8) & 7u, sa = w0 >> 24, sb = w1 >> 24;
        if (op >= OP_COUNT) return fail("tape: unknown opcode");
        if (post > POST_BLEND) return fail("tape: unknown post-combine");
        if (w0 & 0x800000u) return fail("tape: reserved bit set");
        if ((w0 & 0x000800u) && ((w0 >> 12) & 7u) >= std::max(n_p, 1u)) return fail("tape: reload slot out of range");
        if ((w0 & 0x008000u) && ((w0 >> 16) & 7u) >= std::max(n_p, 1u)) return fail("tape: save slot out of range");
        if ((w0 & 0x080000u) && ((w0 >> 20) & 7u) >= std::max(n_d, 1u)) return fail("tape: push slot out of range");
        if (sa >= SDF_NP_SLOTS || sb >= SDF_NP_SLOTS) return fail("tape: slot out of range");
        if ((w1 & 0xFFFFFFu) >= n_consts) return fail("tape: constant offset out of range");
        if (op == OP_END && i != n_words - 2) return fail("tape: END before the last instruction");
    }
    return 0;
}

static int ctx_init(sdf_ctx *c);

extern "C" {

int sdf_abi_version(void) { return SDF_ABI_VERSION; }

#ifndef SDF_BUILD_INFO
#define SDF_BUILD_INFO "unknown toolchain (not built by csrc/build.sh)"
#endif
const char *sdf_build_info(void) { return SDF_BUILD_INFO; }
const char *sdf_last_error(void) { return g_err.c_str(); }

int sdf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return 0; }
    return n;
}

int sdf_test_fail_alloc(int nth) { g_fail_alloc_in = nth > 0 ? nth : 0; return 0; }

int sdf_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes) {
    if (!free_bytes || !total_bytes) return fail("sdf_device_mem_info: NULL argument");
    HIPCHK(set_device(device));
    HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}

int sdf_ctx_create(int device, sdf_ctx **out) {
    if (!out) return fail("sdf_ctx_create: out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("sdf_ctx_create: no such device");
    HIPCHK(set_device(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(std::string("sdf_ctx_create: this library is built for gfx950 only, device is ") + prop.gcnArchName);
    sdf_ctx *c = new sdf_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    c->lds_max = prop.sharedMemPerBlock;
    if (ctx_init(c)) {          // (whatever was created so far goes back)
        const std::string keep = g_err;
        sdf_ctx_destroy(c);
        g_err = keep;
        return 1;
    }
    *out = c;
    return 0;
}

static int ctx_init(sdf_ctx *c) {
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto &e : c->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(host_malloc(&c->h_stage, (size_t)SDF_CALL_SLOTS * SDF_STAGE_BYTES + 4096));   // (+ the bounds estimate's result, sdf_estimate_bounds)
    for (auto &cs : c->slots) {
        HIPCHK(hipEventCreate(&cs.e0)); HIPCHK(hipEventCreate(&cs.e2)); HIPCHK(hipEventCreate(&cs.e3)); HIPCHK(hipEventCreate(&cs.e4));
        HIPCHK(hipEventCreateWithFlags(&cs.done, hipEventDisableTiming));
        HIPCHK(hipStreamCreateWithFlags(&cs.stream, hipStreamNonBlocking));
    }
    if (const char *e = getenv("SDF_SLOT_STREAMS")) c->slot_streams = atoi(e);
    if (const char *e = getenv("SDF_CULL_BLOCK")) c->cull_block = atoi(e);
    if (const char *e = getenv("SDF_TAIL_ORDER")) c->tail_order = atoi(e);
    if (const char *e = getenv("SDF_MESH_TWOPASS")) c->twopass = atoi(e);
    McTables t;
    memcpy(t.ntri, MC_NTRI, 256);
    memcpy(t.amb, MC_AMBIGUOUS, 256);
    memcpy(t.tri, MC_TRI, sizeof(t.tri));
    memcpy(t.mc33, MC33_FLAT, sizeof(t.mc33));
    if (c->mc.ensure(sizeof(t))) return 1;
    HIPCHK(hipMemcpy(c->mc.p, &t, sizeof(t), hipMemcpyHostToDevice));
    if (const char *e = getenv("SDF_BOUNDS_TAG0")) c->bounds_tag0 = (unsigned)atoi(e) & 0xFFFFu;   // (tests: the first tag of the exchange words)
    if (const char *e = getenv("SDF_MESH2")) c->mesh2 = atoi(e);
    if (const char *e = getenv("SDF_MESH_SLOTS")) c->mesh_slots = atoi(e);
    if (const char *e = getenv("SDF_PRUNE")) c->prune = atoi(e);
    if (const char *e = getenv("SDF_PARK")) c->parking = atoi(e);
    if (const char *e = getenv("SDF_PRUNE_LIST_MIN")) c->prune_list_min = std::max(atoi(e), 0);
    if (const char *e = getenv("SDF_CULL")) c->cull = atoi(e);
    if (const char *e = getenv("SDF_DEFER")) c->defer = atoi(e) ? 1 : 0;
    if (const char *e = getenv("SDF_CULL_LEVELS")) c->cull_levels = atoi(e);
    if (const char *e = getenv("SDF_PARK_SPINS")) c->park_spins = std::max(atoi(e), 1);
    if (const char *e = getenv("SDF_MESH_PROF")) { if (atoi(e) && c->prof.ensure(512 + 4096 * 32)) return 1; }
    return 0;
}

int sdf_ctx_destroy(sdf_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)stream_wait(c->stream);
    for (auto &cs : c->slots) { if (cs.stream) (void)stream_wait(cs.stream); cs.park.release(); }
    for (DevBuf *b : {&c->scratch_in, &c->scratch_out, &c->rows, &c->rows_off, &c->mc, &c->prof, &c->park, &c->ext, &c->field_vals,
                      &c->field_vol, &c->field_tiles, &c->bounds_work})
        b->release();
    for (auto &b : c->arena_pool) b.release();
    for (auto &b : c->counter_pool) b.release();
    g_pool.drop_device(c->device);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &cs : c->slots) for (hipEvent_t e : {cs.e0, cs.e2, cs.e3, cs.e4, cs.done}) if (e) (void)hipEventDestroy(e);
    for (auto &cs : c->slots) if (cs.stream) (void)hipStreamDestroy(cs.stream);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    for (hipEvent_t e : c->rec_ev) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

int sdf_ctx_set_stream(sdf_ctx *c, void *s) {
    if (!c) return fail("sdf_ctx_set_stream: ctx is NULL");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return 0;
}

int sdf_ctx_set_prune(sdf_ctx *c, int enabled) {
    if (!c) return fail("sdf_ctx_set_prune: ctx is NULL");
    c->prune = enabled ? 1 : 0;
    return 0;
}

int sdf_ctx_set_cull(sdf_ctx *c, int enabled) {
    if (!c) return fail("sdf_ctx_set_cull: ctx is NULL");
    c->cull = enabled ? 1 : 0;
    return 0;
}

int sdf_ctx_set_defer(sdf_ctx *c, int on) {
    if (!c) return fail("sdf_ctx_set_defer: ctx is NULL");
    c->defer = on ? 1 : 0;
    return 0;
}
int sdf_ctx_set_mesh2(sdf_ctx *c, int mode) {
    if (!c) return fail("sdf_ctx_set_mesh2: ctx is NULL");
    if (mode < -1 || mode > 1) return fail("sdf_ctx_set_mesh2: -1 (by the previous call), 0 (never) or 1 (whenever the tape has a variant)");
    c->mesh2 = mode;
    return 0;
}
int sdf_ctx_set_cull_levels(sdf_ctx *c, int levels) {
    if (!c) return fail("sdf_ctx_set_cull_levels: ctx is NULL");
    if (levels != 0 && levels != 2 && levels != 3) return fail("sdf_ctx_set_cull_levels: 0 (the library's choice), 2 or 3");
    c->cull_levels = levels;
    return 0;
}

int sdf_ctx_set_tail_order(sdf_ctx *c, int on) {
    if (!c) return fail("sdf_ctx_set_tail_order: ctx is NULL");
    c->tail_order = on ? 1 : 0;
    return 0;
}
int sdf_ctx_set_twopass(sdf_ctx *c, int mode) {
    if (!c) return fail("sdf_ctx_set_twopass: ctx is NULL");
    c->twopass = mode < 0 ? -1 : (mode ? 1 : 0);
    return 0;
}

// hand the device memory the library keeps for reuse (blocks of destroyed meshes, soup and counter pools) back to the
// driver: for a caller that switches to a job of a very different size -- the cached blocks of the old job do not fit the
// new one and would be evicted one by one, each hipFree a device synchronisation in the middle of the new job's calls
int sdf_ctx_trim(sdf_ctx *c) {
    if (!c) return fail("sdf_ctx_trim: ctx is NULL");
    HIPCHK(set_device(c->device));
    HIPCHK(stream_wait(c->stream));
    for (auto &cs : c->slots) if (cs.stream) HIPCHK(stream_wait(cs.stream));
    for (auto &b : c->arena_pool) b.release();
    c->arena_pool.clear();
    g_pool.drop_device(c->device);
    return 0;
}

int sdf_ctx_synchronize(sdf_ctx *c) {
    if (!c) return fail("sdf_ctx_synchronize: ctx is NULL");
    HIPCHK(set_device(c->device));
    HIPCHK(stream_wait(c->stream));
    for (auto &cs : c->slots) if (cs.stream) HIPCHK(stream_wait(cs.stream));
    return 0;
}

int sdf_tape_create(sdf_ctx *c, const uint32_t *code, uint32_t n_words, const double *consts, uint32_t n_consts,
                    uint32_t n_p, uint32_t n_d, sdf_tape **out) {
    if (!c || !code || !consts || !out) return fail("sdf_tape_create: NULL argument");
    if (validate_tape(code, n_words, n_consts, n_p, n_d)) return 1;
    HIPCHK(set_device(c->device));
    sdf_tape *t = new sdf_tape();
    struct Guard { sdf_tape *t; ~Guard() { if (t) { const std::string keep = g_err; sdf_tape_destroy(t); g_err = keep; } } } guard{t};
    t->ctx = c; t->n_words = n_words; t->n_consts = n_consts;
    t->full = tape_needs_full(code, n_words, consts);
    {
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(code, (size_t)n_words * 4); mix(consts, (size_t)n_consts * 8); mix(&n_p, 4); mix(&n_d, 4);
        t->content_hash = h;
    }
    t->ia_complete = true;
    for (uint32_t i = 0; i < n_words; i += 2) {
        t->ia_complete = t->ia_complete && ia_has_form(code[i] & 255u);
        t->ia_rare = t->ia_rare || ia_is_rare_leaf(code[i] & 255u);
        if ((code[i] & 255u) == OP_L_EXTERN) {
            if ((code[i + 1] & 0xFFFFFFu) + 1 >= n_consts) return fail("tape: closure index out of the constant pool");
            const double k = consts[(code[i + 1] & 0xFFFFFFu) + 1];
            if (!(k >= 0.0 && k < 65536.0 && k == (double)(uint32_t)k)) return fail("tape: bad closure index");
            t->n_extern = std::max(t->n_extern, (uint32_t)k + 1u);
        }
    }
    t->n_p = n_p; t->n_d = n_d;
    // two more constants behind the tape's own: a K slot and +0.0, the operand of the `acc + (+0.0)` that a
    // decided smooth combine turns into (sdf_interval.h compact_tape); an instruction with constant
    // offset n_consts reads it as c[0]
    std::vector<double> c64(consts, consts + n_consts);
    c64.push_back(0.0); c64.push_back(0.0);
    std::vector<float> c32(c64.size());
    for (size_t i = 0; i < c64.size(); i++) c32[i] = (float)c64[i];
    std::vector<uint32_t> pcode(code, code + n_words);   // + one more END: the interpreter looks one instruction ahead
    pcode.push_back(code[n_words - 2]); pcode.push_back(code[n_words - 1]);
    HIPCHK(dev_malloc((void **)&t->d_code, pcode.size() * sizeof(uint32_t)));
    HIPCHK(dev_malloc((void **)&t->d_c64, c64.size() * sizeof(double)));
    HIPCHK(dev_malloc((void **)&t->d_c32, c32.size() * sizeof(float)));
    HIPCHK(hipMemcpy(t->d_code, pcode.data(), pcode.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t->d_c64, c64.data(), c64.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t->d_c32, c32.data(), c32.size() * sizeof(float), hipMemcpyHostToDevice));
    guard.t = nullptr;
    *out = t;
    return 0;
}

int sdf_tape_set_prune_info(sdf_tape *t, const uint16_t *rstart, const uint16_t *lstart, uint32_t n_instr) {
    if (!t || !rstart || !lstart) return fail("sdf_tape_set_prune_info: NULL argument");
    if (n_instr * 2 != t->n_words) return fail("sdf_tape_set_prune_info: one entry per instruction expected");
    for (uint32_t i = 0; i < n_instr; i++) {
        const bool none = rstart[i] == 0xFFFF || lstart[i] == 0xFFFF;
        if (!none && !(lstart[i] <= rstart[i] && rstart[i] <= i)) return fail("sdf_tape_set_prune_info: operand range out of order");
    }
    HIPCHK(set_device(t->ctx->device));
    if (!t->d_rstart) HIPCHK(dev_malloc((void **)&t->d_rstart, n_instr * 2));
    if (!t->d_lstart) HIPCHK(dev_malloc((void **)&t->d_lstart, n_instr * 2));
    HIPCHK(hipMemcpy(t->d_rstart, rstart, n_instr * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t->d_lstart, lstart, n_instr * 2, hipMemcpyHostToDevice));
    return 0;
}

int sdf_tape_destroy(sdf_tape *t) {
    if (!t) return 0;
    (void)hipSetDevice(t->ctx->device);
    // kernels that read the tape may still run on the context's stream, on a call slot's lane or on a communicator's lanes:
    // wait for the DEVICE (a tape is destroyed once per model, not per call)
    (void)hipDeviceSynchronize();
    if (t->d_code) (void)hipFree(t->d_code);
    if (t->d_c64) (void)hipFree(t->d_c64);
    if (t->d_c32) (void)hipFree(t->d_c32);
    if (t->d_rstart) (void)hipFree(t->d_rstart);
    if (t->d_lstart) (void)hipFree(t->d_lstart);
    delete t;
    return 0;
}

}  // extern "C"

// dispatch over (precision, FULL)
#define LAUNCH_TAPE(KERNEL, grid, block, shmem, t, precision, ...)                                              \
    do {                                                                                                        \
        if ((precision) == SDF_PRECISION_F64) {                                                                 \
            if ((t)->full) hipLaunchKernelGGL((KERNEL<double, true>), grid, block, shmem, (t)->ctx->stream, (t)->d_code, (t)->d_c64, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<double, false>), grid, block, shmem, (t)->ctx->stream, (t)->d_code, (t)->d_c64, __VA_ARGS__); \
        } else {                                                                                                \
            if ((t)->full) hipLaunchKernelGGL((KERNEL<float, true>), grid, block, shmem, (t)->ctx->stream, (t)->d_code, (t)->d_c32, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<float, false>), grid, block, shmem, (t)->ctx->stream, (t)->d_code, (t)->d_c32, __VA_ARGS__); \
        }                                                                                                       \
    } while (0)

#define LAUNCH_TAPE_ON(STREAM, KERNEL, grid, block, shmem, t, precision, ...)                                   \
    do {                                                                                                        \
        if ((precision) == SDF_PRECISION_F64) {                                                                 \
            if ((t)->full) hipLaunchKernelGGL((KERNEL<double, true>), grid, block, shmem, STREAM, (t)->d_code, (t)->d_c64, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<double, false>), grid, block, shmem, STREAM, (t)->d_code, (t)->d_c64, __VA_ARGS__); \
        } else {                                                                                                \
            if ((t)->full) hipLaunchKernelGGL((KERNEL<float, true>), grid, block, shmem, STREAM, (t)->d_code, (t)->d_c32, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<float, false>), grid, block, shmem, STREAM, (t)->d_code, (t)->d_c32, __VA_ARGS__); \
        }                                                                                                       \
    } while (0)

extern "C" {

int sdf_eval_points(sdf_tape *t, const void *d_pts, int64_t n, int dim, void *d_out, int precision) {
    if (!t || !d_pts || !d_out) return fail("sdf_eval_points: NULL argument");
    if (dim != 2 && dim != 3) return fail("sdf_eval_points: dim must be 2 or 3");
    if (precision != SDF_PRECISION_F64 && precision != SDF_PRECISION_F32) return fail("sdf_eval_points: bad precision");
    if (t->n_extern) return fail("sdf_eval_points: the tape reads user closures (L_EXTERN): use sdf_eval_extern_points_host / sdf_eval_points_extern_host");
    if (n <= 0) return 0;
    HIPCHK(set_device(t->ctx->device));
    const unsigned grid = (unsigned)((n + 255) / 256);
    LAUNCH_TAPE(k_eval_points, dim3(grid), dim3(256), 0, t, precision, (const double *)d_pts, (long long)n, dim, (double *)d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

int sdf_eval_points_host(sdf_tape *t, const double *h_pts, int64_t n, int dim, double *h_out, int precision) {
    if (!t || !h_pts || !h_out) return fail("sdf_eval_points_host: NULL argument");
    if (dim != 2 && dim != 3) return fail("sdf_eval_points_host: dim must be 2 or 3");
    if (precision != SDF_PRECISION_F64 && precision != SDF_PRECISION_F32) return fail("sdf_eval_points_host: bad precision");
    if (n <= 0) return 0;
    sdf_ctx *c = t->ctx;
    HIPCHK(set_device(c->device));
    if (c->scratch_in.ensure((size_t)n * dim * 8) || c->scratch_out.ensure((size_t)n * 8)) return 1;
    HIPCHK(hipMemcpyAsync(c->scratch_in.p, h_pts, (size_t)n * dim * 8, hipMemcpyHostToDevice, c->stream));
    if (sdf_eval_points(t, c->scratch_in.p, n, dim, c->scratch_out.p, precision)) return 1;
    HIPCHK(hipMemcpyAsync(h_out, c->scratch_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    return 0;
}

int sdf_eval_grid_host(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                       double *h_out, int precision) {
    if (!t || !X || !Y || !Z || !h_out) return fail("sdf_eval_grid_host: NULL argument");
    if (precision != SDF_PRECISION_F64 && precision != SDF_PRECISION_F32) return fail("sdf_eval_grid_host: bad precision");
    if (t->n_extern) return fail("sdf_eval_grid_host: the tape reads user closures (L_EXTERN): evaluate it with the *_extern_* entry points");
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
    sdf_ctx *c = t->ctx;
    HIPCHK(set_device(c->device));
    const size_t n = (size_t)nx * ny * nz;
    if (c->scratch_in.ensure((size_t)(nx + ny + nz) * 8) || c->scratch_out.ensure(n * 8)) return 1;
    double *dX = (double *)c->scratch_in.p, *dY = dX + nx, *dZ = dY + ny;
    HIPCHK(hipMemcpyAsync(dX, X, (size_t)nx * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dY, Y, (size_t)ny * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dZ, Z, (size_t)nz * 8, hipMemcpyHostToDevice, c->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    LAUNCH_TAPE(k_eval_grid, dim3(grid), dim3(256), 0, t, precision, (const double *)dX, (const double *)dY, (const double *)dZ,
                nx, ny, nz, (double *)c->scratch_out.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_out, c->scratch_out.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    return 0;
}

int sdf_estimate_bounds(sdf_tape *t, double *h_out6, int precision) {
    if (!t || !h_out6) return fail("sdf_estimate_bounds: NULL argument");
    if (precision != SDF_PRECISION_F64 && precision != SDF_PRECISION_F32) return fail("sdf_estimate_bounds: bad precision");
    if (t->n_extern) return fail("sdf_estimate_bounds: the tape reads user closures (L_EXTERN): probe it through the *_extern_* entry points");
    sdf_ctx *c = t->ctx;
    HIPCHK(set_device(c->device));
    if (c->scratch_out.ensure(4096)) return 1;
    // the waves' exchange words are tagged per call instead of zeroed per call (sdf_bounds.hip): cleared when the buffer is new and when
    // the 16-bit tag wraps
    if (!c->bounds_work.p || c->bounds_tag >= 65535u) {
        if (c->bounds_work.ensure(SDF_BOUNDS_WORK_BYTES)) return 1;
        HIPCHK(hipMemsetAsync(c->bounds_work.p, 0, SDF_BOUNDS_WORK_BYTES, c->stream));
        c->bounds_tag = c->bounds_tag0; c->bounds_tag0 = 0;
    }
    const unsigned tag = ++c->bounds_tag;
    {
        const int rc = sdf_launch_bounds(precision == SDF_PRECISION_F64 ? 1 : 0, t->full ? 1 : 0, c->stream, (const uint32_t *)t->d_code,
                                         precision == SDF_PRECISION_F64 ? (const void *)t->d_c64 : (const void *)t->d_c32, (double *)c->scratch_out.p,
                                         c->bounds_work.p, tag);
        if (rc) return fail(std::string("k_estimate_bounds launch: ") + hipGetErrorString((hipError_t)rc));
    }
    // (the seven doubles land in pinned memory behind the call slots' staging: a copy into pageable memory goes through the runtime's
    // own staging and a second host copy)
    double *h = (double *)((char *)c->h_stage + (size_t)SDF_CALL_SLOTS * SDF_STAGE_BYTES);
    HIPCHK(hipMemcpyAsync(h, c->scratch_out.p, 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    if (h[6] == 2.0) return fail("sdf_estimate_bounds: the probe workgroups did not meet at their barrier (device busy): use the host loop");
    if (h[6] != 0.0) return fail("zero-size array to reduction operation maximum which has no identity");   // (NumPy's words, reference sdf/core.py:80)
    memcpy(h_out6, h, 48);
    return 0;
}

int sdf_tape_extern_count(sdf_tape *t) { return t ? (int)t->n_extern : 0; }

// phase 1 of f(P) for a tape with user closures: the point every L_EXTERN leaf sees, per sample
int sdf_eval_extern_points_host(sdf_tape *t, const double *h_pts, int64_t n, int dim, double *h_ext_pts, int precision) {
    if (!t || !h_pts || !h_ext_pts) return fail("sdf_eval_extern_points_host: NULL argument");
    if (dim != 2 && dim != 3) return fail("sdf_eval_extern_points_host: dim must be 2 or 3");
    if (precision != SDF_PRECISION_F64 && precision != SDF_PRECISION_F32) return fail("sdf_eval_extern_points_host: bad precision");
    if (!t->n_extern) return fail("sdf_eval_extern_points_host: the tape has no user closures");
    if (n <= 0) return 0;
    sdf_ctx *c = t->ctx;
    HIPCHK(set_device(c->device));
    const size_t ext_bytes = (size_t)t->n_extern * (size_t)n * 24;
    if (c->scratch_in.ensure((size_t)n * dim * 8) || c->ext.ensure(ext_bytes)) return 1;
    HIPCHK(hipMemcpyAsync(c->scratch_in.p, h_pts, (size_t)n * dim * 8, hipMemcpyHostToDevice, c->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    LAUNCH_TAPE(k_eval_points_ext, dim3(grid), dim3(256), 0, t, precision, (const double *)c->scratch_in.p, (long long)n, dim,
                (double *)c->ext.p, 1, (double *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_ext_pts, c->ext.p, ext_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    return 0;
}

// phase 2: f(P) with the closures' values (n_extern x n float64, leaf-major) supplied by the host
int sdf_eval_points_extern_host(sdf_tape *t, const double *h_pts, int64_t n, int dim, const double *h_ext_vals, double *h_out,
                                int precision) {
    if (!t || !h_pts || !h_ext_vals || !h_out) return fail("sdf_eval_points_extern_host: NULL argument");
    if (dim != 2 && dim != 3) return fail("sdf_eval_points_extern_host: dim must be 2 or 3");
    if (precision != SDF_PRECISION_F64 && precision != SDF_PRECISION_F32) return fail("sdf_eval_points_extern_host: bad precision");
    if (!t->n_extern) return fail("sdf_eval_points_extern_host: the tape has no user closures");
    if (n <= 0) return 0;
    sdf_ctx *c = t->ctx;
    HIPCHK(set_device(c->device));
    const size_t ext_bytes = (size_t)t->n_extern * (size_t)n * 8;
    if (c->scratch_in.ensure((size_t)n * dim * 8) || c->scratch_out.ensure((size_t)n * 8) || c->ext.ensure(ext_bytes)) return 1;
    HIPCHK(hipMemcpyAsync(c->scratch_in.p, h_pts, (size_t)n * dim * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ext.p, h_ext_vals, ext_bytes, hipMemcpyHostToDevice, c->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    LAUNCH_TAPE(k_eval_points_ext, dim3(grid), dim3(256), 0, t, precision, (const double *)c->scratch_in.p, (long long)n, dim,
                (double *)c->ext.p, 0, (double *)c->scratch_out.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_out, c->scratch_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    return 0;
}

int sdf_marching_cubes(sdf_ctx *c, const void *d_volume, int n0, int n1, int n2, void *d_out, int64_t cap, int64_t *n_tris) {
    if (!c || !n_tris) return fail("sdf_marching_cubes: NULL argument");
    *n_tris = 0;
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;   // skimage: "Input array must be at least 2x2x2" -> empty batch
    if (!d_volume) return fail("sdf_marching_cubes: volume is NULL");
    HIPCHK(set_device(c->device));
    const long long nrows = (long long)(n0 - 1) * (n1 - 1);
    if (c->rows.ensure((size_t)nrows * 4) || c->rows_off.ensure((size_t)(nrows + 1) * 8)) return 1;
    unsigned long long *d_total = (unsigned long long *)c->rows_off.p + nrows;
    const unsigned grid = (unsigned)((nrows + 255) / 256);
    launch_k_mc_rows(dim3(grid), dim3(256), c->stream, (const McTables *)c->mc.p, (const float *)d_volume, n0, n1, n2, (unsigned *)c->rows.p);
    launch_k_scan_rows(dim3(1), dim3(1024), c->stream, (const unsigned *)c->rows.p, nrows,
                       (unsigned long long *)c->rows_off.p, d_total);
    HIPCHK(hipGetLastError());
    unsigned long long total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    *n_tris = (int64_t)total;
    if (total && d_out && cap > 0) {
        launch_k_mc_emit(dim3(grid), dim3(256), c->stream, (const McTables *)c->mc.p, (const float *)d_volume, n0, n1, n2,
                           (const unsigned long long *)c->rows_off.p, (float *)d_out, (unsigned long long)cap);
        HIPCHK(hipGetLastError());
        HIPCHK(stream_wait(c->stream));
    }
    return 0;
}

int sdf_marching_cubes_host(sdf_ctx *c, const float *h_vol, int n0, int n1, int n2, float *h_out, int64_t cap, int64_t *n_tris) {
    if (!c || !n_tris) return fail("sdf_marching_cubes_host: NULL argument");
    *n_tris = 0;
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;
    if (!h_vol) return fail("sdf_marching_cubes_host: volume is NULL");
    HIPCHK(set_device(c->device));
    const size_t n = (size_t)n0 * n1 * n2;
    if (c->scratch_in.ensure(n * 4)) return 1;
    if (cap > 0 && c->scratch_out.ensure((size_t)cap * 36)) return 1;
    HIPCHK(hipMemcpyAsync(c->scratch_in.p, h_vol, n * 4, hipMemcpyHostToDevice, c->stream));
    if (sdf_marching_cubes(c, c->scratch_in.p, n0, n1, n2, cap > 0 ? c->scratch_out.p : nullptr, cap, n_tris)) return 1;
    const int64_t ncopy = std::min<int64_t>(*n_tris, cap);
    if (ncopy > 0 && h_out) {
        HIPCHK(hipMemcpyAsync(h_out, c->scratch_out.p, (size_t)ncopy * 36, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(stream_wait(c->stream));
    }
    return 0;
}

}  // extern "C"

// k_mesh launch: the register-file variant is the smallest that holds the tape's slots, the
// shape (threads x samples per lane) a per-precision default found by measurement
static int launch_mesh(sdf_tape *t, const void *code, int precision, MeshArgs &a, int grid, int bs, hipStream_t st) {
    sdf_ctx *c = t->ctx;
    const size_t tile = (size_t)(bs + 1) * (bs + 1) * (bs + 1) * 4;
    const size_t bits_off = (MESH_LDS_VOL + tile + 15) & ~(size_t)15;
    const size_t nvox = (size_t)(bs + 1) * (bs + 1) * (bs + 1);
    const size_t list_off = (bits_off + ((nvox + 63) / 64 + 2) * 8 + 15) & ~(size_t)15;
    if (list_off + 4096 > c->lds_max) return fail("sdf_generate: device LDS too small for this batch size");
    a.bits_off = (int)bits_off;
    const size_t list_cap = std::min<size_t>((c->lds_max - list_off) / 4, 16384);
    a.list_off = (int)list_off; a.list_cap = (int)list_cap;
    const size_t lds = list_off + list_cap * 4;
    // two slots of sparse tiles share the dense tile's region (deferred emission, k_mesh); a slot has to hold its header,
    // some samples and the cell table of the per-cell counting -- else every tile stays dense
    // -- and between them and the sign bits the area through which a waiting batch's triangles are transposed (the sign bits
    // and the work area stay free: the next work item's record arrives there meanwhile)
    a.slot_bytes = 0; a.stage_off = 0;
    if (c->defer && !a.twopass && a.cull && bits_off > MESH_LDS_VOL + 16 * MESH_STAGE_BYTES) {
        const size_t slot = ((bits_off - MESH_LDS_VOL - 16 * MESH_STAGE_BYTES) / 2) & ~(size_t)15;
        if (slot >= MESH_SLOT_HDR + 2048 + 8192 && list_cap * 4 >= CULL_RECORD) {
            a.slot_bytes = (int)slot;
            a.stage_off = (int)(MESH_LDS_VOL + 2 * slot);
        }
    }
    // the first register-file variant that holds the tape's slots: 0 = (1,1), 1 = (2,2), 2 = (4,2), 3 = (2,4), 4 = (4,4), 5 = (8,8)
    static const uint32_t kFile[6][2] = {{1, 1}, {2, 2}, {4, 2}, {2, 4}, {4, 4}, {8, 8}};
    const uint32_t np = std::max(t->n_p, 1u), nd = std::max(t->n_d, 1u);
    int slots = 5;
    for (int k = 5; k >= 0; k--) if (np <= kFile[k][0] && nd <= kFile[k][1]) slots = k;
    if (c->mesh_slots >= 0 && c->mesh_slots <= 5 && np <= kFile[c->mesh_slots][0] && nd <= kFile[c->mesh_slots][1])
        slots = c->mesh_slots;                                                        // (tuning: another file that fits)
    // (one shape per register file and scheme: 1024 threads x 3 or 2 samples per lane, the 8-slot file 1024 x 1 -- sdf_mesh_inst.hip)
    if (precision != SDF_PRECISION_F64) return fail("k_mesh: float64 only");
    const int rc = t->full ? sdf_launch_mesh_f64_full(slots, 0, a.twopass, grid, lds, st, (const uint32_t *)code, t->d_c64, a)
                           : sdf_launch_mesh_f64(slots, 0, a.twopass, grid, lds, st, (const uint32_t *)code, t->d_c64, a);
    if (rc) return fail(std::string("k_mesh launch: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

// k_mesh2 (sdf_mesh2.h): the register file of the tape, or -1 when none of its variants holds it
static int mesh2_slots(const sdf_tape *t) {
    const uint32_t np = std::max(t->n_p, 1u), nd = std::max(t->n_d, 1u);
    return (np <= 1 && nd <= 1) ? 0 : ((np <= 2 && nd <= 2) ? 1 : ((np <= 2 && nd <= 4) ? 3 : -1));
}
// ... and its launch: two workgroups per compute unit, each with half of the CU's LDS -- the fixed areas and ONE region that the
// batch being sampled and the batch that waits share from its two ends
static int launch_mesh2(sdf_tape *t, const void *code, MeshArgs &a, int nb, hipStream_t st) {
    sdf_ctx *c = t->ctx;
    const size_t lds = (c->lds_max / 2) & ~(size_t)1023;
    if (lds < (size_t)M2_REGION + 16384) return fail("k_mesh2: device LDS too small");
    a.slot_bytes = (int)((lds - M2_REGION) & ~(size_t)15);
    a.bits_off = a.list_off = a.list_cap = a.stage_off = 0;      // (k_mesh's layout: not used)
    a.order = nullptr; a.tail = 0; a.park = nullptr; a.park_cap = 0;
    const int grid = std::min(nb, 2 * c->n_cu);
    const int rc = t->full ? sdf_launch_mesh2_f64_full(mesh2_slots(t), grid, lds, st, (const uint32_t *)code, t->d_c64, a)
                           : sdf_launch_mesh2_f64(mesh2_slots(t), grid, lds, st, (const uint32_t *)code, t->d_c64, a);
    if (rc) return fail(std::string("k_mesh2 launch: ") + (rc < 0 ? "no variant for this tape" : hipGetErrorString((hipError_t)rc)));
    return 0;
}

// the skip test (`_skip`, reference sdf/core.py:28-43) of batches [b0, b1) alone, enqueued on `st`: d_kinds[b] = 0 (skipped) or
// 255 (pending) for those batches; the axes are on the device already (X, then Y, then Z)
static int enqueue_skip(sdf_tape *t, const double *d_axes, int nx, int ny, int nz, int bs, int b0, int b1, int precision,
                        unsigned char *d_kinds, hipStream_t st) {
    if (b1 <= b0) return 0;
    GridDesc g = {};
    g.X = d_axes; g.Y = d_axes + nx; g.Z = d_axes + nx + ny;
    g.nx = nx; g.ny = ny; g.nz = nz; g.bs = bs;
    g.nbx = (nx + bs - 1) / bs; g.nby = (ny + bs - 1) / bs; g.nbz = (nz + bs - 1) / bs;
    PruneArgs pa = {};
    pa.first_block = 0x7fffffff;     // (no interval pass in this launch)
    const unsigned blocks = (unsigned)((b1 - b0 + SKIP_BATCHES_PER_BLOCK - 1) / SKIP_BATCHES_PER_BLOCK);
    LAUNCH_TAPE_ON(st, k_skip, dim3(blocks), dim3(256), 0, t, precision, g, b1, d_kinds, pa, (const double *)t->d_c64,
                   (const uint16_t *)t->d_rstart, (const uint16_t *)t->d_lstart, b0);
    HIPCHK(hipGetLastError());
    return 0;
}

// the per-call statistics from the counters the meshing pass left (end of sdf_generate / sdf_mesh_wait)
extern "C" int sdf_mesh_wait(sdf_mesh *m, int *emitted);

static void finish_stats(sdf_tape *t, sdf_mesh *m, const MeshCounters &h, int nb, bool pruning, uint32_t n_instr, unsigned long long key,
                         float ms_prepass, float ms_total) {
    m->work_begin = h.work_begin; m->work_end = h.work_end;
    m->st.n_skipped = nb - h.nwork;
    m->st.n_work_begin = m->work_begin; m->st.n_work_end = m->work_end;
    m->st.n_triangles = (int64_t)h.total;
    m->st.n_empty = h.n_empty; m->st.n_nonempty = h.n_nonempty;
    m->st.n_eval_voxels = (int64_t)h.n_eval; m->st.n_ambiguous_cells = (int64_t)h.n_ambiguous;
    m->st.n_pruned_instrs = pruning ? (int64_t)h.n_pruned : 0;
    m->st.n_sampled_voxels = (int64_t)h.n_sampled;
    // the kernel's own clock readings: 100 MHz ticks between the first workgroup's start and the last one's end
    m->st.ms_mesh_device = (h.t_first_inv && h.t_last > ~h.t_first_inv) ? (double)(h.t_last - ~h.t_first_inv) * 1e-5 : 0.0;
    m->st.sclk_mhz = h.clk_ticks ? (double)h.clk_cycles / (double)h.clk_ticks * 100.0 : 0.0;
    m->st.t_mesh_first_us = h.t_first_inv ? (double)(~h.t_first_inv) * 0.01 : 0.0;
    m->st.t_mesh_last_us = (double)h.t_last * 0.01;
    m->pruned = pruning;
    m->st.mesh_kernel = m->used_mesh2 ? 2 : 1;
    m->st.n_batch_instrs = (int64_t)(n_instr - 1) * (h.work_end - h.work_begin);
    t->hint_key = key; t->hint_total_tris = std::max<unsigned long long>(h.total, 1);
    {   // (per MODEL and grid, for sdf_generate_records; a handful of entries per job -- the map is emptied when it grows past 4096)
        auto &rh = t->ctx->rec_hints;
        if (rh.size() > 4096) rh.clear();
        sdf_ctx::RecHint &e = rh[std::make_pair(t->content_hash, key)];
        e.tris = std::max<unsigned long long>(h.total, 1); e.raw = h.n_raw;
    }
    if (!(t->mesh2_key == key && t->mesh2_state == 3)) { t->mesh2_key = key; t->mesh2_state = h.not_mesh2 ? 2 : 1; }
    m->st.ms_prepass = ms_prepass;
    m->st.ms_total = ms_total;
}

// what identifies "the same job on the same grid" for the capacity hints (sdf_tape::hint_key, sdf_ctx::rec_hints)
static unsigned long long grid_key(int nx, int ny, int nz, int bs, int sparse, int64_t shard_index, int64_t shard_count) {
    return ((unsigned long long)nx << 42) ^ ((unsigned long long)ny << 21) ^ (unsigned long long)nz ^
           ((unsigned long long)shard_index << 56) ^ ((unsigned long long)shard_count << 48) ^
           ((unsigned long long)bs << 36) ^ (sparse ? 1ull << 63 : 0ull);
}

static int generate_impl(sdf_tape *t, sdf_mesh *m, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                         int bs, int sparse, int64_t shard_index, int64_t shard_count, int precision, void *d_out, int64_t cap_out,
                         bool async_mode = false, int64_t slab_items = -1, hipStream_t lane_stream = nullptr,
                         const unsigned char *d_kinds_in = nullptr) {
    // slab_items >= 0: compact mode (sdf_generate_compact_async) -- d_out is a SLAB of capacity (slab_items, cap_out)
    // lane_stream: the stream the whole call is enqueued on (the exchange steps of sdf_comm run on lanes of their own)
    // d_kinds_in: the skip test's verdict for every batch is already on the device (0 skipped / 255 pending, n_batches
    // bytes: sdf_skip_kinds, possibly all-gathered from the ranks that each tested a share): k_skip is not run
    sdf_ctx *c = t->ctx;
    const bool compact = slab_items >= 0;
    // a free call slot; when all are held by calls in flight, the oldest of them is COLLECTED first (its counters
    // and event times live in the slot's pinned staging and events: reusing the slot before sdf_mesh_wait has
    // read them would hand that mesh this call's numbers)
    int slot = -1;
    for (int k = 0; k < SDF_CALL_SLOTS && slot < 0; k++)
        if (!c->slots[(c->slot_seq + (unsigned)k) % SDF_CALL_SLOTS].busy) slot = (int)((c->slot_seq + (unsigned)k) % SDF_CALL_SLOTS);
    if (slot < 0) {
        slot = (int)(c->slot_seq % SDF_CALL_SLOTS);
        CallSlot &held = c->slots[slot];
        if (held.owner && held.owner->pend.active) { if (sdf_mesh_wait(held.owner, nullptr)) return 1; }
        else { HIPCHK(event_wait(held.done)); }
        held.busy = false; held.owner = nullptr;
    }
    c->slot_seq = (unsigned)slot + 1u;
    CallSlot &cs = c->slots[slot];
    // Calls in flight (sdf_generate_to_device_async) each run on their slot's OWN stream: call i + 1's prepass then
    // fills the compute units that call i's k_mesh leaves idle in its tail (a persistent workgroup per CU, the last
    // batches finish at different times: 9 % of that kernel's CU-time) and the dispatch gaps of one call hide behind
    // the kernels of the other.  Everything a call touches is its own (per-mesh buffers, per-slot staging / events /
    // park slots), so the streams need no ordering among themselves.  An adopted caller stream is never left.
    const bool own_lane = async_mode && !compact && c->stream == c->own_stream && c->slot_streams;
    hipStream_t st = lane_stream ? lane_stream : (own_lane ? cs.stream : c->stream);
    m->stream = (own_lane || lane_stream) ? st : nullptr;
    char *stage = (char *)c->h_stage + (size_t)slot * SDF_STAGE_BYTES;
    GridDesc &g = m->g;
    g.nx = nx; g.ny = ny; g.nz = nz; g.bs = bs;
    g.nbx = (nx + bs - 1) / bs; g.nby = (ny + bs - 1) / bs; g.nbz = (nz + bs - 1) / bs;
    const long long nb64 = (long long)g.nbx * g.nby * g.nbz;
    if (nb64 > 0x7fffffffLL) return fail("sdf_generate: too many batches");
    const int nb = (int)nb64;
    m->st.n_batches = nb;
    m->st.n_grid_voxels = (int64_t)nx * ny * nz;
    if (nb == 0) {
        if (compact) HIPCHK(hipMemsetAsync(d_out, 0, sizeof(SlabHeader), st));   // an empty grid: an empty slab
        return 0;
    }

    if (m->axes.ensure((size_t)(nx + ny + nz) * 8) || m->kinds.ensure((size_t)nb) || m->worklist.ensure((size_t)nb * 4) ||
        m->status.ensure((size_t)nb * 8))
        return 1;
    if (!m->counters.p && !c->counter_pool.empty()) { m->counters = c->counter_pool.back(); c->counter_pool.pop_back(); }
    if (m->counters.ensure(sizeof(MeshCounters))) return 1;
    double *dX = (double *)m->axes.p, *dY = dX + nx, *dZ = dY + ny;
    g.X = dX; g.Y = dY; g.Z = dZ;
    HIPCHK(hipEventRecord(cs.e0, st));
    const size_t axis_bytes = (size_t)(nx + ny + nz) * 8;
    if (async_mode && axis_bytes > SDF_STAGE_BYTES - 256) return fail("sdf_generate_to_device_async: axes too long for the staging slot");
    if (axis_bytes <= SDF_STAGE_BYTES - 256) {   // one copy from pinned memory instead of three from pageable
        double *hs = (double *)stage;
        memcpy(hs, X, (size_t)nx * 8); memcpy(hs + nx, Y, (size_t)ny * 8); memcpy(hs + nx + ny, Z, (size_t)nz * 8);
        HIPCHK(hipMemcpyAsync(dX, hs, axis_bytes, hipMemcpyHostToDevice, st));
    } else {
        HIPCHK(hipMemcpyAsync(dX, X, (size_t)nx * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dY, Y, (size_t)ny * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dZ, Z, (size_t)nz * 8, hipMemcpyHostToDevice, st));
    }

    // ---- prepass: skip test for every batch, then the ordered work list (+ this shard's slice) ----
    // (three event records per call, not one per interval: each is a marker packet the queue has to drain to;
    // ms_prepass = ev[0] -> ev[2] includes the copy of the axes, ms_mesh = ev[2] -> ev[4], ms_total = ev[0] -> ev[4])
    // interval pass: per batch, which instructions never matter (float64 sampling only: the intervals
    // bound the float64 interpreter, not the float32 one).  The box of a batch is spanned by its first
    // and last coordinate per axis: monotone axes only.
    auto monotone = [](const double *a, int n) {
        bool up = true, down = true;
        for (int i = 1; i < n; i++) { up &= a[i - 1] <= a[i]; down &= a[i - 1] >= a[i]; }
        return up || down;
    };
    const uint32_t n_instr = t->n_words / 2;
    const bool intervals_ok = precision == SDF_PRECISION_F64 && monotone(X, nx) && monotone(Y, ny) && monotone(Z, nz);
    const bool pruning = c->prune && t->d_rstart && n_instr <= 256 && intervals_ok && t->n_consts < 0xFFFFF0u;
    // 64-bit words per batch tape: the instructions, one more END, the length; whole 64-byte lines
    const int tape_stride = (int)((n_instr + 2 + 7) & ~7u);
    PruneArgs pa = {};
    pa.first_block = 0x7fffffff;
    // many batches: the interval pass runs behind k_compact, for the surviving batches only (k_prune_list)
    // (a rank of a multi-GPU job prunes its own share of the work list only, whatever the grid's size)
    const bool prune_listed = pruning && sparse && (nb >= c->prune_list_min || shard_count > 1);
    unsigned skip_blocks = (sparse && !d_kinds_in) ? (unsigned)((nb + SKIP_BATCHES_PER_BLOCK - 1) / SKIP_BATCHES_PER_BLOCK) : 0u, prune_blocks = 0;
    size_t prune_lds = 0;
    if (pruning) {
        if (m->prune.ensure((size_t)nb * 64) || m->tapes.ensure((size_t)nb * tape_stride * 8)) return 1;
        pa.n_instr = (int)n_instr; pa.n_p = std::max(t->n_p, 1u); pa.n_d = std::max(t->n_d, 1u);
        pa.masks_out = (uint32_t *)m->prune.p; pa.tapes_out = (unsigned long long *)m->tapes.p; pa.tape_stride = tape_stride;
        pa.zero_off = t->n_consts;
        prune_lds = prune_lds_bytes(pa.n_p, pa.n_d);
        if (!prune_listed) {         // fused into k_skip's launch: every batch
            pa.first_block = (int)skip_blocks;
            prune_blocks = (unsigned)(((long long)nb * 8 + PRUNE_BLOCK - 1) / PRUNE_BLOCK);
        }
    }
    if (skip_blocks + prune_blocks) {
        if (prune_lds > 32768 && prune_blocks) {   // (more dynamic LDS than the default limit: tapes with many saved-point slots)
            const void *fn = t->ia_rare ? (t->full ? reinterpret_cast<const void *>(k_skip_rare<double, true>) : reinterpret_cast<const void *>(k_skip_rare<double, false>))
                                        : (t->full ? reinterpret_cast<const void *>(k_skip<double, true>) : reinterpret_cast<const void *>(k_skip<double, false>));
            HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prune_lds));
        }
        const size_t skip_lds = prune_blocks ? prune_lds : 0;
        if (t->ia_rare) LAUNCH_TAPE_ON(st, k_skip_rare, dim3(skip_blocks + prune_blocks), dim3(256), skip_lds, t, precision, g, nb, (unsigned char *)m->kinds.p, pa,
                                       (const double *)t->d_c64, (const uint16_t *)t->d_rstart, (const uint16_t *)t->d_lstart, 0);
        else LAUNCH_TAPE_ON(st, k_skip, dim3(skip_blocks + prune_blocks), dim3(256), skip_lds, t, precision, g, nb, (unsigned char *)m->kinds.p, pa,
                            (const double *)t->d_c64, (const uint16_t *)t->d_rstart, (const uint16_t *)t->d_lstart, 0);
    }
    if (!sparse) HIPCHK(hipMemsetAsync(m->kinds.p, 255, (size_t)nb, st));
    else if (d_kinds_in) HIPCHK(hipMemcpyAsync(m->kinds.p, d_kinds_in, (size_t)nb, hipMemcpyDeviceToDevice, st));   // (k_mesh writes its verdicts into the mesh's own copy)
    launch_k_compact(dim3(1), dim3(1024), st, (const unsigned char *)m->kinds.p, nb, (int *)m->worklist.p,
                       (MeshCounters *)m->counters.p, (unsigned long long *)m->status.p, (long long)shard_index,
                       (long long)shard_count);
    HIPCHK(hipGetLastError());
    if (prune_listed) {
        pa.worklist = (const int *)m->worklist.p; pa.ctr = (const MeshCounters *)m->counters.p;
        auto kp = t->full ? (t->ia_rare ? k_prune_list<true, true> : k_prune_list<true, false>) : (t->ia_rare ? k_prune_list<false, true> : k_prune_list<false, false>);
        if (prune_lds > 32768) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)prune_lds));
        hipLaunchKernelGGL(kp, dim3((unsigned)(((long long)nb * 8 + PRUNE_BLOCK - 1) / PRUNE_BLOCK)), dim3(PRUNE_BLOCK), prune_lds, st,
                           (const uint32_t *)t->d_code, g, nb, pa, (const double *)t->d_c64, (const uint16_t *)t->d_rstart,
                           (const uint16_t *)t->d_lstart);
        HIPCHK(hipGetLastError());
    }
    // second interval pass, per surviving batch: the sub-groups of 2^3 cells the surface cannot be in are not sampled
    // (k_mesh reads the batch's record into the list region of its LDS, which has to hold it: launch_mesh's layout)
    const size_t mesh_nvox = (size_t)(bs + 1) * (bs + 1) * (bs + 1);
    const size_t mesh_bits_off = (MESH_LDS_VOL + mesh_nvox * 4 + 15) & ~(size_t)15;
    const size_t mesh_list_off = (mesh_bits_off + ((mesh_nvox + 63) / 64 + 2) * 8 + 15) & ~(size_t)15;
    const bool culling = c->cull && intervals_ok && t->ia_complete && mesh_list_off + CULL_RECORD <= c->lds_max;
    // the tail of the work list is handed out by descending cost (MeshArgs::order, k_cull's estimates); fewer items
    // than k_mesh has workgroups
    const int tail_max = std::min<int>(MESH_TAIL_MAX, std::min(nb, c->n_cu) - 1);
    // (not for calls in flight next to others: the argument that a reordered tail cannot stall -- fewer tail items than
    // workgroups -- counts RESIDENT workgroups, and a k_mesh that shares the device with another call's k_mesh may have
    // fewer of them for a while; the neighbours fill the tail of such a call anyway)
    const bool tail_order = culling && c->tail_order && tail_max >= 2 && !async_mode;
    const unsigned long long key = grid_key(nx, ny, nz, bs, sparse, shard_index, shard_count);
    if (culling) {
        if (c->prof.p) HIPCHK(hipMemsetAsync((unsigned char *)c->prof.p + 128, 0, 384, st));
        if (m->cull.ensure((size_t)nb * CULL_RECORD) || (tail_order && m->order.ensure(MESH_TAIL_MAX * sizeof(int)))) return 1;
        const int ia_np = (int)std::max(t->n_p, 1u), ia_nd = (int)std::max(t->n_d, 1u);
        auto kc = t->full ? (t->ia_rare ? k_cull<true, true> : k_cull<true, false>) : (t->ia_rare ? k_cull<false, true> : k_cull_lean);
        int cull_block = CULL_BLOCK;
        if (c->cull_block == 128 && kc == k_cull_lean) { kc = k_cull_lean128; cull_block = 128; }
        // (the trig-capable variant with one or two waves per work item: the first level of the pass -- 64 boxes -- keeps
        // ONE wave of a workgroup busy whatever its size, so smaller workgroups mean more work items per compute unit)
        // measured (r03f, prepass of weave 2^33 / 2^27, gearlike 2^30, knurling 2^27, ms): 256 threads 8.95 / 1.52 / 0.325 /
        // 0.364; 128: 6.61 / 1.35 / 0.276 / 0.361; 64: 6.05 / 1.44 / 0.298 / 0.442 -- two waves are the default here
        if (t->full && !t->ia_rare && c->cull_block != 256) {
            cull_block = c->cull_block == 64 ? 64 : 128;
            kc = cull_block == 64 ? k_cull<true, false, 64> : k_cull<true, false, 128>;
        }
        // The third interval level (sub-groups of 2^3 cells) halves what k_mesh samples and costs 8 interval runs per
        // undecided group of 4^3 cells.  r04a, same box, prepass + k_mesh in ms, two levels -> three: example 2^27 0.068 + 0.280
        // -> 0.104 + 0.240, pawn 0.132 + 0.416 -> 0.276 + 0.289, blobby 2^30 0.229 + 0.874 -> 0.505 + 0.612 (level with or
        // ahead, and the prepass of the NEXT call hides behind k_mesh when calls are in flight); with trigonometry in the tape
        // the interval forms are dearer than the samples they save: gearlike 2^30 0.271 + 1.23 -> 0.64 + 0.99, knurling 2^27
        // 0.364 + 1.52 -> 1.58 + 1.25, weave 2^33 6.6 + 24.2 -> 19.9 + 15.4.  Hence three levels for the lean tapes, two for
        // the others -- which still list units of 2^3 samples instead of r03's cubes of 4^3.
        const int cull_levels = c->cull_levels ? c->cull_levels : (kc == k_cull_lean || kc == k_cull_lean128 ? 3 : 2);
        // (seven workgroups of 256 threads per CU instead of six -- 72 VGPRs, the interval state of 192 threads in LDS, so that
        // every work item of the 512^3 example is resident at once -- was measured in r04k: example prepass 0.093 vs 0.091 ms,
        // pawn 0.402 vs 0.266 ms: the levels then run in more passes.  Not the limit.)
        const size_t ia_bytes = std::min<size_t>((size_t)cull_block * (6 * ia_np + 2 * ia_nd) * 8, c->lds_max - 896 - CULL_SCRATCH);
        const size_t lds = 896 + CULL_SCRATCH + ia_bytes;
        if (lds > 32768) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kc, dim3(nb), dim3(cull_block), lds, st,
                           pruning ? (const uint32_t *)m->tapes.p : (const uint32_t *)t->d_code, (const double *)t->d_c64, g,
                           (const int *)m->worklist.p, (const MeshCounters *)m->counters.p, pruning ? tape_stride : 0, (int)n_instr,
                           ia_np, ia_nd, (int)ia_bytes, (unsigned char *)m->cull.p, (unsigned long long *)c->prof.p,
                           tail_order ? (int *)m->order.p : (int *)nullptr, tail_max, cull_levels);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(cs.e2, st));

    // ---- meshing.  The whole chain (prepass -> k_mesh) is enqueued without a host round trip: the
    // work-list length stays on the device and k_mesh writes the ordered float64 soup itself.  The
    // soup goes into the caller's device buffer when one was given (sdf_generate_to_device),
    // otherwise into a library buffer sized from the last call of this tape on the same grid (first
    // call: from the work-list length, which costs one synchronisation).  A soup that does not fit
    // is detected on the device (nothing is written past the capacity) and the pass is re-run into
    // a library buffer of the exact size. ----
    unsigned long long cap = 0;
    MeshCounters h;
    bool to_caller = compact || (d_out && cap_out > 0);
    bool quiet = true;     // nothing but k_mesh follows ev[2] on the stream, and the host did not stall in between
    if (!to_caller) {
        if (t->hint_key == key && t->hint_total_tris) {
            cap = t->hint_total_tris + t->hint_total_tris / 4 + 4096;
        } else {
            quiet = false;
            HIPCHK(hipMemcpyAsync(&h, m->counters.p, sizeof(h), hipMemcpyDeviceToHost, st));
            HIPCHK(stream_wait(st));
            const unsigned long long nshard0 = (unsigned long long)std::max(h.work_end - h.work_begin, 1);
            cap = std::max<unsigned long long>(4096ull * nshard0, 1ull << 16);
            // a guess, not a need (the overflow re-run finds the exact size): never more than half the free memory
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                cap = std::min<unsigned long long>(cap, std::max<unsigned long long>(free_b / 2 / (72 + 40), 1ull << 16));   // (+ 40 B per triangle: the two-pass arenas)
        }
    }
    float ms = 0;
    bool mesh2_failed = false;
    for (int attempt = 0;; attempt++) {
        MeshArgs a;
        a.compact = 0; a.xf = nullptr; a.xf_cap = 0; a.raw = nullptr; a.raw_cap = 0;
        a.twopass = 0; a.desc = nullptr; a.cells = nullptr; a.tlist = nullptr; a.cells_cap = a.tlist_cap = 0; a.block_item = nullptr;
        a.order = tail_order ? (const int *)m->order.p : nullptr; a.tail = tail_order ? tail_max : 0;
        if (compact) {
            const SlabLayout L(slab_items, cap_out);
            a.out = reinterpret_cast<double *>((unsigned char *)d_out + L.tris_off); a.out_cap = (unsigned long long)cap_out;
            a.compact = 1; a.xf = reinterpret_cast<double *>((unsigned char *)d_out + L.xf_off);
            a.raw = reinterpret_cast<float *>((unsigned char *)d_out + L.raw_off); a.raw_cap = L.raw_cap;
            a.xf_cap = (int)std::min<int64_t>(slab_items, 0x7fffffff);
        } else if (to_caller) {
            a.out = (double *)d_out; a.out_cap = (unsigned long long)cap_out;
        } else {
            if (!m->out.p && !c->arena_pool.empty()) { m->out = c->arena_pool.back(); c->arena_pool.pop_back(); }
            if (m->out.bytes < (size_t)cap * 72) quiet = false;     // (an allocation: the stream idles meanwhile)
            if (m->out.ensure((size_t)cap * 72)) {
                // a first-call guess that does not fit: shrink it and let the overflow re-run size the soup exactly
                bool ok = false;
                for (int k = 0; k < 6 && !ok && attempt == 0 && cap > (1ull << 16); k++) {
                    cap = std::max<unsigned long long>(cap / 4, 1ull << 16);
                    ok = m->out.ensure((size_t)cap * 72) == 0;
                }
                if (!ok) return 1;
            }
            a.out = (double *)m->out.p; a.out_cap = m->out.bytes / 72;
        }
        if (attempt) {   // (the first pass finds both cleared by k_compact)
            HIPCHK(hipMemsetAsync(m->counters.p, 0, MESH_COUNTERS_RESET_BYTES, st));
            HIPCHK(hipMemsetAsync(m->status.p, 0, (size_t)nb * 8, st));
        }
        a.g = g; a.worklist = (const int *)m->worklist.p;
        a.kinds = (unsigned char *)m->kinds.p; a.status = (unsigned long long *)m->status.p;
        a.ctr = (MeshCounters *)m->counters.p;
        a.mc = (const McTables *)c->mc.p;
        a.prof = (unsigned long long *)c->prof.p;
        // One pass or two?  (decided here: the two-pass scheme parks nothing -- its triangles are numbered by k_scan_items -- so a
        // call that takes it does not make its lane allocate park slots: 1.2 GB that the long jobs' lanes never touched, r04 advisor)
        const bool twopass = c->twopass >= 0 ? c->twopass != 0 : n_instr > 96;
        DevBuf &park = async_mode ? cs.park : c->park;   // (k_mesh kernels of calls in flight may overlap in time, whichever
                                                         // streams they run on: each call slot has its own staging slots)
        // k_mesh or k_mesh2?  (k_mesh2 holds sparse tiles only: culled batches of a tape with a variant, one pass; what it meets and
        // does not hold it flags, and the pass is repeated with k_mesh)
        const bool use_mesh2 = culling && !twopass && !mesh2_failed && c->defer && c->mesh2 != 0 && mesh2_slots(t) >= 0 &&
                               !(t->mesh2_key == key && t->mesh2_state >= 2) && (c->mesh2 > 0 || (t->mesh2_key == key && t->mesh2_state == 1));
        const bool parks = c->parking && !twopass && !use_mesh2;
        if (parks && !park.p) { quiet = false; if (park.ensure((size_t)c->n_cu * MESH_PARK_DEPTH * SDF_PARK_TRIS * 36)) return 1; }
        a.park = parks ? (float *)park.p : nullptr; a.park_cap = a.park ? SDF_PARK_TRIS : 0;
        a.park_spins = (unsigned)c->park_spins;
        a.cull = culling ? (const unsigned char *)m->cull.p : nullptr;
        a.tape_stride = pruning ? tape_stride : 0;
        a.n_instr = (int)n_instr;
        // One pass or two?  The one-pass kernel (look-back + parking inside the sampling kernel) is 7 - 20 % faster on
        // short tapes: it hides its triangle traffic behind other workgroups' arithmetic, which three kernels in a row
        // cannot.  On long tapes (weave at 2^33, 244 instructions) the two schemes tie -- 27.5 vs 28.0 ms -- and the
        // two-pass one moves a third of the bytes (9 GB against 25 GB per call: no parking, and the 4-slot sampling
        // kernel spills less without the emit phases): the tape's length decides (sdf_ctx_set_twopass / SDF_MESH_TWOPASS
        // override).
        if (twopass) {
            // the arenas of the two-pass scheme: a surface cell carries at least one triangle, so the soup's capacity
            // bounds both (a call whose arenas turn out too small is flagged and repeated like one whose soup is)
            const size_t cap_t = (size_t)a.out_cap;
            if (m->desc.bytes < (size_t)nb * sizeof(ItemDesc) || m->cellrecs.bytes < cap_t * 36 || m->trilist.bytes < cap_t * 4) quiet = false;
            if (m->desc.ensure((size_t)nb * sizeof(ItemDesc)) || m->cellrecs.ensure(cap_t * 36) || m->trilist.ensure(cap_t * 4) ||
                m->blockidx.ensure(((cap_t + 255) / 256 + 2) * sizeof(int)))
                return 1;
            a.twopass = 1; a.desc = (ItemDesc *)m->desc.p; a.cells = (unsigned *)m->cellrecs.p; a.tlist = (unsigned *)m->trilist.p;
            a.block_item = (const int *)m->blockidx.p;
            a.cells_cap = a.tlist_cap = (unsigned long long)cap_t;
        }
        if (a.prof) {   // (words 16.. are k_cull's: cleared before the prepass; behind byte 512: the workgroups' timelines)
            HIPCHK(hipMemsetAsync(a.prof, 0, 128, st));
            HIPCHK(hipMemsetAsync((unsigned char *)a.prof + 512, 0, 4096 * 32, st));
        }
        const int grid = std::min(nb, c->n_cu);   // persistent workgroups; surplus ones find the list empty
        const bool own_start = attempt > 0 || a.prof || !quiet;   // (something was enqueued, or the host waited, since ev[2])
        if (own_start) HIPCHK(hipEventRecord(cs.e3, st));
        if (use_mesh2 ? launch_mesh2(t, pruning ? m->tapes.p : (const void *)t->d_code, a, nb, st)
                      : launch_mesh(t, pruning ? m->tapes.p : (const void *)t->d_code, precision, a, grid, bs, st))
            return 1;
        m->used_mesh2 = use_mesh2;
        if (a.twopass) {
            const unsigned long long emit_blocks = (a.out_cap + 255ull) / 256ull;
            if (emit_blocks > 0x7fffffffull) return fail("sdf_generate: soup capacity too large for one k_emit2 launch");
            launch_k_scan_items(dim3(1), dim3(1024), st, (const ItemDesc *)m->desc.p, (MeshCounters *)m->counters.p,
                               (unsigned long long *)m->status.p, (int *)m->blockidx.p, emit_blocks + 1ull);
            launch_k_emit2(dim3((unsigned)std::max<unsigned long long>(emit_blocks, 1ull)), dim3(256), st, a);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(cs.e4, st));
        if (compact) {
            const unsigned pack_blocks = (unsigned)std::min<int64_t>(std::max<int64_t>((slab_items + 255) / 256, 1), 1024);
            HIPCHK((hipError_t)sdf_launch_pack_slab(pack_blocks, st, (const MeshCounters *)m->counters.p, (const unsigned long long *)m->status.p,
                                                    (unsigned char *)d_out, (long long)slab_items, (long long)cap_out));
        }
        MeshCounters *hp = (MeshCounters *)(stage + SDF_STAGE_BYTES - 256);   // pinned
        HIPCHK(hipMemcpyAsync(hp, m->counters.p, sizeof(h), hipMemcpyDeviceToHost, st));
        if (async_mode && attempt == 0) {   // the caller collects the result with sdf_mesh_wait
            HIPCHK(hipEventRecord(cs.done, st));
            cs.busy = true; cs.owner = m;
            sdf_mesh::Pending &pd = m->pend;
            pd.active = true; pd.tape = t; pd.slot = slot; pd.nb = nb; pd.bs = bs; pd.sparse = sparse; pd.precision = precision;
            pd.pruning = pruning; pd.own_start = own_start; pd.n_instr = n_instr; pd.key = key; pd.compact = compact;
            pd.d_out = d_out; pd.cap_out = cap_out; pd.shard_index = shard_index; pd.shard_count = shard_count;
            pd.nx = nx; pd.ny = ny; pd.nz = nz;
            pd.axes.assign(X, X + nx); pd.axes.insert(pd.axes.end(), Y, Y + ny); pd.axes.insert(pd.axes.end(), Z, Z + nz);
            return 0;
        }
        HIPCHK(stream_wait(st));
        h = *hp;
        HIPCHK(hipEventElapsedTime(&ms, own_start ? cs.e3 : cs.e2, cs.e4));
        m->st.ms_mesh = ms;
        if (c->prof.p) {
            unsigned long long pc[64];
            HIPCHK(hipMemcpy(pc, c->prof.p, 512, hipMemcpyDeviceToHost));
            {   // timeline of the workgroups: when each ran out of work and when it was done, relative to the first start
                const int grid = use_mesh2 ? std::min(nb, 2 * c->n_cu) : std::min(nb, c->n_cu);
                std::vector<unsigned long long> tl((size_t)4 * grid);
                HIPCHK(hipMemcpy(tl.data(), (unsigned char *)c->prof.p + 512, tl.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t_end = 0;
                double s_out = 0, s_done = 0, mn_out = 1e30, mx_out = 0;
                for (int i = 0; i < grid; i++) t0 = std::min(t0, tl[4 * i]);
                for (int i = 0; i < grid; i++) {
                    const double o = (double)(tl[4 * i + 1] - t0) * 0.01, d = (double)(tl[4 * i + 2] - t0) * 0.01;   // us
                    s_out += o; s_done += d; mn_out = std::min(mn_out, o); mx_out = std::max(mx_out, o); t_end = std::max(t_end, tl[4 * i + 2]);
                }
                double first_hi = 0;   // the latest start: workgroups that were not resident from the beginning start late
                for (int i = 0; i < grid; i++) first_hi = std::max(first_hi, (double)(tl[4 * i] - t0) * 0.01);
                fprintf(stderr, "[k_mesh prof] %d workgroups (kernel %d), last of them started after %.1f us; out of work after min %.1f avg %.1f max %.1f us; done after avg %.1f, last %.1f us\n",
                        grid, use_mesh2 ? 2 : 1, first_hi, mn_out, s_out / grid, mx_out, s_done / grid, (double)(t_end - t0) * 0.01);
            }
            fprintf(stderr, "[k_cull prof] work items by listed tasks (of 563; bins of 64, last: not culled): %llu %llu %llu %llu %llu %llu %llu %llu %llu | %llu\n",
                    pc[32], pc[33], pc[34], pc[35], pc[36], pc[37], pc[38], pc[39], pc[40], pc[41]);
            fprintf(stderr, "[k_cull prof] cycles of thread 0, summed over the workgroups: start %llu boxes %llu list %llu groups %llu (%llu passes, %llu groups) tasks %llu record %llu\n",
                    pc[16], pc[17], pc[18], pc[19], pc[23], pc[22], pc[20], pc[21]);
            fprintf(stderr, "[k_cull prof] task listing: which tasks %llu, scans %llu; start: range %llu batch %llu origin %llu axes %llu barrier %llu (rest: tape length)\n", pc[24], pc[25], pc[26], pc[27], pc[30], pc[28], pc[29]);
            fprintf(stderr, "[k_mesh prof] sampling: intervals %llu task list %llu interpreter %llu sign bits %llu\n", pc[8], pc[9], pc[10], pc[11]);
            fprintf(stderr, "[k_mesh prof] fine: atomic %llu barrier+rank %llu header %llu | rows %llu cells %llu | placing %llu look-back %llu | round end %llu\n",
                    pc[42], pc[43], pc[44], pc[45], pc[46], pc[47], pc[48], pc[49]);
            fprintf(stderr, "[k_mesh prof] %.3f ms; cycles/WG-sum: grab %llu sample %llu count %llu (of which placing the parked batch %llu) list %llu emit %llu tail %llu; %llu batches parked, %llu written one batch later from their slot\n",
                    ms, pc[0], pc[1], pc[2], pc[6], pc[3], pc[4], pc[5], pc[7], pc[12]);
        }
        m->st.n_retries = attempt;
        if (h.overflow & 2u) return fail("sdf_generate: ordered-allocation look-back timed out");
        if (h.overflow & (unsigned)MESH_OVERFLOW_NOT_MESH2) {   // k_mesh2 met a tile it does not hold: the same pass again, with k_mesh
            if (!use_mesh2 || attempt >= 3) return fail("sdf_generate: a tile was flagged as not k_mesh2's by a pass that did not run k_mesh2");
            if (getenv("SDF_MESH2_DEBUG")) fprintf(stderr, "[k_mesh2] flagged: overflow word 0x%x (32 dense tile, 64 tasks, 128 region, 256 cells, 512 list)\n", h.overflow);
            t->mesh2_key = key; t->mesh2_state = 3;
            mesh2_failed = true;
            continue;
        }
        if (compact) {   // (the synchronous records mode, sdf_generate_records: its caller sizes the slab again and repeats the call)
            const SlabLayout L(slab_items, cap_out);
            const bool raw_over = (long long)h.n_raw > L.raw_cap;
            m->n_raw = (long long)h.n_raw;
            m->rec_overflow = (h.overflow & 1u) != 0 || raw_over || (long long)(h.work_end - h.work_begin) > (long long)slab_items;
            m->rec_need_tris = std::max<long long>((long long)h.total, raw_over ? (long long)h.n_raw * SLAB_RAW_DIV : 0ll);
            m->emitted_to = nullptr;
            break;
        }
        if (h.overflow) {
            if (attempt >= 3) return fail("sdf_generate: soup buffer overflow persists");
            to_caller = false;                       // the exact need is known now: h.total
            cap = h.total + 1024;
            continue;
        }
        m->emitted_to = to_caller ? d_out : nullptr;
        break;
    }
    float ms_pre = 0, ms_tot = 0;
    HIPCHK(hipEventElapsedTime(&ms_pre, cs.e0, cs.e2));
    HIPCHK(hipEventElapsedTime(&ms_tot, cs.e0, cs.e4));
    finish_stats(t, m, h, nb, pruning, n_instr, key, ms_pre, ms_tot);
    return 0;
}

// `generate` for batch_size > 32 (reference sdf/core.py:87, 114-119 takes any batch size): the (batch_size + 1)^3 float32 tile
// of such a batch does not fit the LDS of a compute unit (33^3 = 144 KB of 160 KB does), so the fused kernels do not apply.  The
// batches go through device memory instead, a chunk of them per submission: k_eval_tiles samples the chunk's tiles into float32
// volumes (the interpreter, a lane per sample), k_field_rows / k_scan_rows / k_field_emit march them and write
// `points * scale + offset` into the ordered float64 soup -- the kernels behind sdf_generate_field, with the tape instead of a
// host callback.  The skip test is k_skip's, the work list k_compact's.  One host synchronisation per chunk: a chunk is
// >= 2.7e5 samples per batch, the launches are long.  Synchronous; the soup lives in library memory.
static int generate_big(sdf_tape *t, sdf_mesh *m, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                        int bs, int sparse, int64_t shard_index, int64_t shard_count, int precision) {
    sdf_ctx *c = t->ctx;
    hipStream_t st = c->stream;
    GridDesc &g = m->g;
    g.nx = nx; g.ny = ny; g.nz = nz; g.bs = bs;
    g.nbx = (nx + bs - 1) / bs; g.nby = (ny + bs - 1) / bs; g.nbz = (nz + bs - 1) / bs;
    const long long nb64 = (long long)g.nbx * g.nby * g.nbz;
    if (nb64 > 0x7fffffffLL) return fail("sdf_generate: too many batches");
    const int nb = (int)nb64;
    m->st.n_batches = nb;
    m->st.n_grid_voxels = (int64_t)nx * ny * nz;
    if (nb == 0) return 0;
    if (m->axes.ensure((size_t)(nx + ny + nz) * 8) || m->kinds.ensure((size_t)nb) || m->worklist.ensure((size_t)nb * 4) ||
        m->status.ensure((size_t)nb * 8) || m->counters.ensure(sizeof(MeshCounters)))
        return 1;
    double *dX = (double *)m->axes.p, *dY = dX + nx, *dZ = dY + ny;
    g.X = dX; g.Y = dY; g.Z = dZ;
    HIPCHK(hipEventRecord(c->ev[0], st));
    HIPCHK(hipMemcpyAsync(dX, X, (size_t)nx * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dY, Y, (size_t)ny * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dZ, Z, (size_t)nz * 8, hipMemcpyHostToDevice, st));
    if (sparse) { if (enqueue_skip(t, dX, nx, ny, nz, bs, 0, nb, precision, (unsigned char *)m->kinds.p, st)) return 1; }
    else HIPCHK(hipMemsetAsync(m->kinds.p, 255, (size_t)nb, st));
    launch_k_compact(dim3(1), dim3(1024), st, (const unsigned char *)m->kinds.p, nb, (int *)m->worklist.p, (MeshCounters *)m->counters.p,
                     (unsigned long long *)m->status.p, (long long)shard_index, (long long)shard_count);
    HIPCHK(hipGetLastError());
    MeshCounters h;
    HIPCHK(hipMemcpyAsync(&h, m->counters.p, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(stream_wait(st));
    std::vector<int> work((size_t)std::max(h.nwork, 1));
    if (h.nwork) HIPCHK(hipMemcpy(work.data(), m->worklist.p, (size_t)h.nwork * 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> kinds((size_t)nb);
    HIPCHK(hipMemcpy(kinds.data(), m->kinds.p, (size_t)nb, hipMemcpyDeviceToHost));
    HIPCHK(hipEventRecord(c->ev[1], st));
    m->work_begin = h.work_begin; m->work_end = h.work_end;
    m->st.n_skipped = nb - h.nwork;
    m->st.n_work_begin = h.work_begin; m->st.n_work_end = h.work_end;

    const size_t tile_max = (size_t)(bs + 1) * (bs + 1) * (bs + 1);
    const int slots = (bs * bs + 255) & ~255;                                          // row slots per tile
    const int CH = (int)std::max<size_t>(1, std::min<size_t>(32, ((size_t)256 << 20) / (tile_max * 4)));   // batches per submission: <= 256 MB of volumes
    std::vector<FieldTile> tiles((size_t)CH);
    std::vector<int> org((size_t)CH * 3);
    std::vector<unsigned long long> offs((size_t)CH * slots + 1);
    if (c->field_vol.ensure((size_t)CH * tile_max * 4) || c->field_tiles.ensure(sizeof(FieldTile) * CH + (size_t)CH * 12) ||
        c->rows.ensure((size_t)CH * slots * 4) || c->rows_off.ensure(((size_t)CH * slots + 1) * 8))
        return 1;
    int *d_org = reinterpret_cast<int *>((char *)c->field_tiles.p + sizeof(FieldTile) * CH);
    unsigned long long total = 0;
    // the look-back words of the fused path, written by the host here: per work item its inclusive triangle prefix, so that
    // sdf_mesh_batch_offsets serves these meshes too (the counts come back per chunk anyway)
    std::vector<unsigned long long> prefix((size_t)std::max(h.work_end - h.work_begin, 1));
    for (int w0 = h.work_begin; w0 < h.work_end; w0 += CH) {
        const int nt = std::min(CH, h.work_end - w0);
        size_t npts = 0, big = 0;
        for (int j = 0; j < nt; j++) {
            const int b = work[(size_t)(w0 + j)];
            const int ibz = b % g.nbz, iby = (b / g.nbz) % g.nby, ibx = b / (g.nbz * g.nby);       // (batch_origin, sdf_device.h)
            const int ox = ibx * bs, oy = iby * bs, oz = ibz * bs;
            const int lx = std::min(bs + 1, nx - ox), ly = std::min(bs + 1, ny - oy), lz = std::min(bs + 1, nz - oz);
            FieldTile &tl = tiles[(size_t)j];
            tl.vol_off = (long long)npts; tl.n0 = lx; tl.n1 = ly; tl.n2 = lz; tl.pad_ = 0;
            tl.of[0] = X[ox]; tl.of[1] = Y[oy]; tl.of[2] = Z[oz];
            tl.sc[0] = lx > 1 ? X[ox + 1] - X[ox] : 0.0; tl.sc[1] = ly > 1 ? Y[oy + 1] - Y[oy] : 0.0; tl.sc[2] = lz > 1 ? Z[oz + 1] - Z[oz] : 0.0;
            org[(size_t)3 * j] = ox; org[(size_t)3 * j + 1] = oy; org[(size_t)3 * j + 2] = oz;
            const size_t n = (size_t)lx * ly * lz;
            npts += n; big = std::max(big, n);
            m->st.n_eval_voxels += (int64_t)n;
        }
        const size_t nslots = (size_t)nt * slots;
        HIPCHK(hipMemcpyAsync(c->field_tiles.p, tiles.data(), sizeof(FieldTile) * (size_t)nt, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_org, org.data(), (size_t)nt * 12, hipMemcpyHostToDevice, st));
        LAUNCH_TAPE_ON(st, k_eval_tiles, dim3((unsigned)((big + 255) / 256), (unsigned)nt), dim3(256), 0, t, precision, (const double *)dX,
                       (const double *)dY, (const double *)dZ, (const FieldTile *)c->field_tiles.p, (const int *)d_org, (float *)c->field_vol.p);
        launch_k_field_rows(dim3((unsigned)(slots / 256), (unsigned)nt), dim3(256), st, (const McTables *)c->mc.p, (const float *)c->field_vol.p,
                            (const FieldTile *)c->field_tiles.p, (unsigned *)c->rows.p, slots);
        unsigned long long *d_total = (unsigned long long *)c->rows_off.p + nslots;
        launch_k_scan_rows(dim3(1), dim3(1024), st, (const unsigned *)c->rows.p, (long long)nslots, (unsigned long long *)c->rows_off.p, d_total);
        HIPCHK(hipGetLastError());
        // (per tile only its first slot's offset and the chunk's total are needed on the host)
        HIPCHK(hipMemcpy2DAsync(offs.data(), 8, c->rows_off.p, (size_t)slots * 8, 8, (size_t)nt, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&offs[(size_t)nt], d_total, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(stream_wait(st));
        const unsigned long long chunk_total = offs[(size_t)nt];
        for (int j = 0; j < nt; j++) {
            const unsigned long long cnt = offs[(size_t)j + 1] - offs[(size_t)j];
            kinds[(size_t)work[(size_t)(w0 + j)]] = cnt ? 2 : 1;
            if (cnt) m->st.n_nonempty++; else m->st.n_empty++;
            prefix[(size_t)(w0 + j - h.work_begin)] = MESH_FLAG_PFX | (total + offs[(size_t)j + 1]);
        }
        if (chunk_total) {
            if ((total + chunk_total) * 72 > m->out.bytes) {       // grow the soup (geometric), keeping what is there
                DevBuf bigger;
                if (bigger.ensure(std::max<size_t>((size_t)(total + chunk_total) * 72 * 2, (size_t)1 << 22))) return 1;
                if (total) HIPCHK(hipMemcpyAsync(bigger.p, m->out.p, (size_t)total * 72, hipMemcpyDeviceToDevice, st));
                HIPCHK(stream_wait(st));
                m->out.release();
                m->out = bigger;
            }
            launch_k_field_emit(dim3((unsigned)(slots / 256), (unsigned)nt), dim3(256), st, (const McTables *)c->mc.p, (const float *)c->field_vol.p,
                                (const FieldTile *)c->field_tiles.p, (const unsigned long long *)c->rows_off.p, (double *)m->out.p, total,
                                (unsigned long long)(m->out.bytes / 72), slots);
            HIPCHK(hipGetLastError());
            HIPCHK(stream_wait(st));   // (the chunk's buffers are refilled next)
            total += chunk_total;
        }
    }
    HIPCHK(hipEventRecord(c->ev[2], st));
    m->st.n_triangles = (int64_t)total;
    m->st.n_sampled_voxels = m->st.n_eval_voxels;
    m->st.n_batch_instrs = (int64_t)(t->n_words / 2 - 1) * (h.work_end - h.work_begin);
    // (work items of other shards stay 255 = "other shard", like the fused path)
    HIPCHK(hipMemcpyAsync(m->kinds.p, kinds.data(), (size_t)nb, hipMemcpyHostToDevice, st));
    if (h.work_end > h.work_begin)
        HIPCHK(hipMemcpyAsync((unsigned long long *)m->status.p + h.work_begin, prefix.data(), (size_t)(h.work_end - h.work_begin) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(stream_wait(st));
    float ms_pre = 0, ms_tot = 0;
    HIPCHK(hipEventElapsedTime(&ms_pre, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms_tot, c->ev[0], c->ev[2]));
    m->st.ms_prepass = ms_pre; m->st.ms_total = ms_tot; m->st.ms_mesh = ms_tot - ms_pre;
    m->emitted_to = nullptr;
    return 0;
}

extern "C" {

int sdf_mesh_destroy(sdf_mesh *m);

static int generate_entry(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                          int sparse, int64_t shard_index, int64_t shard_count, int precision, void *d_out, int64_t cap_out,
                          sdf_mesh **out, bool async_mode = false, int64_t slab_items = -1, const unsigned char *d_kinds_in = nullptr) {
    if (!t || !X || !Y || !Z || !out) return fail("sdf_generate: NULL argument");
    *out = nullptr;
    if (t->n_extern) return fail("sdf_generate: the tape reads user closures (L_EXTERN): mesh it with sdf_generate_field");
    if (bs < 1 || bs > SDF_BATCH_SIZE_MAX) return fail("sdf_generate: batch_size must be in 1..512");
    if (bs > 32 && slab_items >= 0) return fail("sdf_generate_compact: batch_size must be in 1..32 (batches of more than 33^3 samples are not part of the multi-GPU exchange)");
    if (bs > 32 && d_kinds_in) return fail("sdf_generate_from_kinds: batch_size must be in 1..32");
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail("sdf_generate: bad shard");
    if (precision == SDF_PRECISION_F32)
        return fail("sdf_generate: the meshing path samples in float64 (the reference's arithmetic); SDF_PRECISION_F32 was a diagnostic until round 4 -- "
                    "outside the 1e-5 tolerance at its maximum, slower than float64 behind the interval passes -- and was removed; sdf_eval_* and "
                    "sdf_estimate_bounds keep both precisions");
    if (precision != SDF_PRECISION_F64) return fail("sdf_generate: bad precision");
    if (nx < 0 || ny < 0 || nz < 0) return fail("sdf_generate: negative axis length");
    sdf_ctx *c = t->ctx;
    HIPCHK(set_device(c->device));
    sdf_mesh *m = new sdf_mesh();
    m->ctx = c;
    // (batch_size > 32: through device memory, synchronously, into library memory -- a caller buffer is reported as not filled,
    // like one that was too small: sdf_mesh_emit_device copies)
    if (bs > 32 ? generate_big(t, m, X, nx, Y, ny, Z, nz, bs, sparse, shard_index, shard_count, precision)
                : generate_impl(t, m, X, nx, Y, ny, Z, nz, bs, sparse, shard_index, shard_count, precision, d_out, cap_out, async_mode, slab_items, nullptr, d_kinds_in)) {
        const std::string keep = g_err;
        sdf_mesh_destroy(m);
        g_err = keep;
        return 1;
    }
    *out = m;
    return 0;
}

int sdf_generate(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                 int sparse, int64_t shard_index, int64_t shard_count, int precision, sdf_mesh **out) {
    return generate_entry(t, X, nx, Y, ny, Z, nz, bs, sparse, shard_index, shard_count, precision, nullptr, 0, out);
}

int sdf_generate_to_device(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                           int sparse, int64_t shard_index, int64_t shard_count, int precision, void *d_out,
                           int64_t cap_tris, int *emitted, sdf_mesh **out) {
    if (emitted) *emitted = 0;
    if (!d_out || cap_tris <= 0) return fail("sdf_generate_to_device: output buffer is NULL or empty");
    if (generate_entry(t, X, nx, Y, ny, Z, nz, bs, sparse, shard_index, shard_count, precision, d_out, cap_tris, out)) return 1;
    if (emitted) *emitted = ((*out)->emitted_to == d_out || (*out)->st.n_triangles == 0) ? 1 : 0;
    return 0;
}

int sdf_generate_from_kinds(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                            int64_t shard_index, int64_t shard_count, int precision, const void *d_kinds, sdf_mesh **out) {
    if (!d_kinds) return fail("sdf_generate_from_kinds: d_kinds is NULL");
    return generate_entry(t, X, nx, Y, ny, Z, nz, bs, 1, shard_index, shard_count, precision, nullptr, 0, out, false, -1, (const unsigned char *)d_kinds);
}

int sdf_generate_to_device_async(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                                 int sparse, int64_t shard_index, int64_t shard_count, int precision, void *d_out,
                                 int64_t cap_tris, sdf_mesh **out) {
    if (!d_out || cap_tris <= 0) return fail("sdf_generate_to_device_async: output buffer is NULL or empty");
    return generate_entry(t, X, nx, Y, ny, Z, nz, bs, sparse, shard_index, shard_count, precision, d_out, cap_tris, out, true);
}

// `generate` for a caller who wants the soup ON THE HOST (what the reference's `generate` returns, sdf/core.py:131-141): the triangles
// are written as 16-byte records into a slab of the library's (sdf_slab.h: local float32 coordinates + a transform per work item,
// the multi-GPU exchange unit) instead of as 72-byte float64 triangles, and sdf_mesh_emit_host_workers makes the float64 soup on host
// threads while the records are still arriving: 47 MB over PCIe instead of 212 MB at 512^3.  The slab is sized from what the last
// call of the same MODEL on the same grid needed (sdf_ctx::rec_hints, keyed by the tape's content: a fresh tape object of the same
// model finds it); without such a hint -- the first call -- the call is an ordinary sdf_generate, which leaves the hint.  A slab that
// turns out too small is sized again and the call repeated.  The mesh answers every reader: those that want the float64 soup on
// the device (STL records, weld, sdf_mesh_emit_device, ranges) get it from k_expand on demand.
int sdf_generate_records(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                         int sparse, int precision, sdf_mesh **out) {
    if (!t || !X || !Y || !Z || !out) return fail("sdf_generate_records: NULL argument");
    sdf_ctx *c = t->ctx;
    const unsigned long long key = grid_key(nx, ny, nz, bs, sparse, 0, 1);
    const auto it = c->rec_hints.find(std::make_pair(t->content_hash, key));
    const long long nb64 = (long long)((nx + std::max(bs, 1) - 1) / std::max(bs, 1)) * ((ny + std::max(bs, 1) - 1) / std::max(bs, 1)) * ((nz + std::max(bs, 1) - 1) / std::max(bs, 1));
    if (bs < 1 || bs > 32 || t->n_extern || it == c->rec_hints.end() || nb64 <= 0 || nb64 > 0x7fffffffLL || precision != SDF_PRECISION_F64)
        return generate_entry(t, X, nx, Y, ny, Z, nz, bs, sparse, 0, 1, precision, nullptr, 0, out);     // (every check and message of sdf_generate)
    HIPCHK(set_device(c->device));
    long long cap_tris = (long long)(it->second.tris + it->second.tris / 64 + 1024);
    if ((long long)it->second.raw > cap_tris / SLAB_RAW_DIV + SLAB_RAW_MIN) cap_tris = std::max<long long>(cap_tris, (long long)(it->second.raw + it->second.raw / 8) * SLAB_RAW_DIV);
    *out = nullptr;
    sdf_mesh *m = new sdf_mesh();
    m->ctx = c;
    m->records = true;
    int attempt = 0;
    for (;; attempt++) {
        m->slab_items = nb64; m->slab_tris = cap_tris;
        const SlabLayout L(m->slab_items, m->slab_tris);
        int rc = m->slab.ensure(L.bytes);
        if (!rc) rc = generate_impl(t, m, X, nx, Y, ny, Z, nz, bs, sparse, 0, 1, precision, m->slab.p, cap_tris, false, m->slab_items);
        if (!rc && m->rec_overflow && attempt >= 3) rc = fail("sdf_generate_records: slab overflow persists");
        if (rc) {
            const std::string keep = g_err;
            sdf_mesh_destroy(m);
            g_err = keep;
            return 1;
        }
        if (!m->rec_overflow) break;
        cap_tris = m->rec_need_tris + m->rec_need_tris / 64 + 1024;
    }
    m->st.n_retries = attempt;
    *out = m;
    return 0;
}

// The batch loop of `generate` (reference sdf/core.py:114-141) around a field that lives on the HOST: a user-written
// closure (reference README.md:258-295, sdf/d3.py:48-63), possibly calling device-resident sub-models itself.  The
// library does what the reference's `_skip` / `_worker` do around `sdf(P)`: it builds the points of the skip test and
// of every surviving batch (`_cartesian_product`, first axis slowest), hands them to the callback, and meshes the
// returned values on the device -- a chunk of batches per submission: float32 cast, marching cubes of all tiles,
// one scan for the order, `points * scale + offset` into the ordered float64 soup.
int sdf_generate_field(sdf_ctx *c, sdf_field_fn field, void *user, const double *X, int nx, const double *Y, int ny,
                       const double *Z, int nz, int bs, int sparse, int64_t shard_index, int64_t shard_count, sdf_mesh **out) {
    if (!c || !field || !X || !Y || !Z || !out) return fail("sdf_generate_field: NULL argument");
    *out = nullptr;
    if (bs < 1 || bs > SDF_BATCH_SIZE_MAX) return fail("sdf_generate_field: batch_size must be in 1..512");
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail("sdf_generate_field: bad shard");
    if (nx < 0 || ny < 0 || nz < 0) return fail("sdf_generate_field: negative axis length");
    HIPCHK(set_device(c->device));
    sdf_mesh *m = new sdf_mesh();
    m->ctx = c;
    void *h_pts = nullptr, *h_vals = nullptr;
    struct Guard {
        sdf_mesh *&m; void *&a; void *&b;
        ~Guard() { const std::string keep = g_err; if (a) sdf_host_free(a); if (b) sdf_host_free(b); if (m) sdf_mesh_destroy(m); g_err = keep; }
    } guard{m, h_pts, h_vals};
    GridDesc &g = m->g;
    g.nx = nx; g.ny = ny; g.nz = nz; g.bs = bs;
    g.nbx = (nx + bs - 1) / bs; g.nby = (ny + bs - 1) / bs; g.nbz = (nz + bs - 1) / bs;
    const long long nb64 = (long long)g.nbx * g.nby * g.nbz;
    if (nb64 > 0x7fffffffLL) return fail("sdf_generate_field: too many batches");
    const int nb = (int)nb64;
    m->st.n_batches = nb;
    m->st.n_grid_voxels = (int64_t)nx * ny * nz;
    if (nb == 0) { *out = m; m = nullptr; return 0; }
    auto origin = [&](int b, int &ox, int &oy, int &oz, int &lx, int &ly, int &lz) {   // (batch_origin, sdf_device.h)
        const int ibz = b % g.nbz, iby = (b / g.nbz) % g.nby, ibx = b / (g.nbz * g.nby);
        ox = ibx * bs; oy = iby * bs; oz = ibz * bs;
        lx = std::min(bs + 1, nx - ox); ly = std::min(bs + 1, ny - oy); lz = std::min(bs + 1, nz - oz);
    };
    const size_t tile_max = (size_t)(bs + 1) * (bs + 1) * (bs + 1);
    const int slots = bs <= 32 ? 1024 : ((bs * bs + 255) & ~255);                      // row slots per tile (k_field_rows)
    const int CH = (int)std::max<size_t>(1, std::min<size_t>(32, ((size_t)64 << 20) / tile_max));   // batches per submission: <= 64 M points = 2 GB of pinned points + values -- except that ONE tile is always taken whole: (512 + 1)^3 points = 4.3 GB at the largest batch size
    const size_t pts_cap = std::max<size_t>((size_t)CH * tile_max, (size_t)9 << 12);   // points per callback
    if (sdf_host_alloc(pts_cap * 24, &h_pts) || sdf_host_alloc(pts_cap * 8, &h_vals)) return 1;
    double *pts = (double *)h_pts, *vals = (double *)h_vals;
    std::vector<uint8_t> kinds((size_t)nb, 255);

    // ---- `_skip` (reference sdf/core.py:28-43): centre + the 8 corners of every batch through the field ----
    if (sparse) {
        const int per = (int)(pts_cap / 9);
        for (int b0 = 0; b0 < nb; b0 += per) {
            const int n = std::min(per, nb - b0);
            for (int j = 0; j < n; j++) {
                int ox, oy, oz, lx, ly, lz;
                origin(b0 + j, ox, oy, oz, lx, ly, lz);
                const double x0 = X[ox], x1 = X[ox + lx - 1], y0 = Y[oy], y1 = Y[oy + ly - 1], z0 = Z[oz], z1 = Z[oz + lz - 1];
                double *p = pts + (size_t)j * 27;
                p[0] = (x0 + x1) / 2; p[1] = (y0 + y1) / 2; p[2] = (z0 + z1) / 2;
                for (int k = 0; k < 8; k++) {             // itertools.product((x0, x1), (y0, y1), (z0, z1))
                    p[3 + 3 * k] = (k & 4) ? x1 : x0; p[4 + 3 * k] = (k & 2) ? y1 : y0; p[5 + 3 * k] = (k & 1) ? z1 : z0;
                }
            }
            if (field(user, pts, (int64_t)n * 9, vals)) return fail("sdf_generate_field: the field callback failed");
            for (int j = 0; j < n; j++) {
                int ox, oy, oz, lx, ly, lz;
                origin(b0 + j, ox, oy, oz, lx, ly, lz);
                const double x0 = X[ox], y0 = Y[oy], z0 = Z[oz];
                const double *p = pts + (size_t)j * 27, *v = vals + (size_t)j * 9;
                const double r = fabs(v[0]);
                const double d = sqrt(((p[0] - x0) * (p[0] - x0) + (p[1] - y0) * (p[1] - y0)) + (p[2] - z0) * (p[2] - z0));
                bool same = true;
                const bool pos = v[1] > 0.0;
                for (int k = 1; k <= 8; k++) same = same && (pos ? v[k] > 0.0 : v[k] < 0.0);
                kinds[(size_t)(b0 + j)] = (!(r <= d) && same) ? 0 : 255;
            }
        }
    }
    std::vector<int> work;
    for (int b = 0; b < nb; b++) if (kinds[(size_t)b]) work.push_back(b);
    const long long nwork = (long long)work.size();
    const int w_begin = (int)((nwork * shard_index) / shard_count), w_end = (int)((nwork * (shard_index + 1)) / shard_count);
    m->work_begin = w_begin; m->work_end = w_end;
    m->st.n_skipped = nb - (int64_t)nwork;
    m->st.n_work_begin = w_begin; m->st.n_work_end = w_end;

    // ---- `_worker` for the shard's batches, CH at a time ----
    std::vector<FieldTile> tiles((size_t)CH);
    std::vector<unsigned long long> offs((size_t)CH * slots + 1);
    unsigned long long total = 0;
    for (int w0 = w_begin; w0 < w_end; w0 += CH) {
        const int nt = std::min(CH, w_end - w0);
        size_t npts = 0;
        for (int j = 0; j < nt; j++) {
            int ox, oy, oz, lx, ly, lz;
            origin(work[(size_t)(w0 + j)], ox, oy, oz, lx, ly, lz);
            FieldTile &tl = tiles[(size_t)j];
            tl.vol_off = (long long)npts; tl.n0 = lx; tl.n1 = ly; tl.n2 = lz; tl.pad_ = 0;
            // scale = the batch's first axis step (reference sdf/core.py:58-59: `X[1] - X[0]` of the batch's slices);
            // a one-sample axis has none and the tile has no cells, so its value is never used
            tl.of[0] = X[ox]; tl.of[1] = Y[oy]; tl.of[2] = Z[oz];
            tl.sc[0] = lx > 1 ? X[ox + 1] - X[ox] : 0.0; tl.sc[1] = ly > 1 ? Y[oy + 1] - Y[oy] : 0.0; tl.sc[2] = lz > 1 ? Z[oz + 1] - Z[oz] : 0.0;
            double *p = pts + npts * 3;
            for (int ix = 0; ix < lx; ix++)
                for (int iy = 0; iy < ly; iy++)
                    for (int iz = 0; iz < lz; iz++, p += 3) { p[0] = X[ox + ix]; p[1] = Y[oy + iy]; p[2] = Z[oz + iz]; }
            npts += (size_t)lx * ly * lz;
            m->st.n_eval_voxels += (int64_t)lx * ly * lz;
        }
        if (field(user, pts, (int64_t)npts, vals)) return fail("sdf_generate_field: the field callback failed");
        const size_t nslots = (size_t)nt * slots;
        if (c->field_vals.ensure(npts * 8) || c->field_vol.ensure(npts * 4) || c->field_tiles.ensure(sizeof(FieldTile) * CH) ||
            c->rows.ensure(nslots * 4) || c->rows_off.ensure((nslots + 1) * 8))
            return 1;
        HIPCHK(hipMemcpyAsync(c->field_vals.p, vals, npts * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->field_tiles.p, tiles.data(), sizeof(FieldTile) * (size_t)nt, hipMemcpyHostToDevice, c->stream));
        launch_k_cast_f32(dim3((unsigned)((npts + 255) / 256)), dim3(256), c->stream, (const double *)c->field_vals.p,
                           (float *)c->field_vol.p, (long long)npts);
        launch_k_field_rows(dim3((unsigned)(slots / 256), (unsigned)nt), dim3(256), c->stream, (const McTables *)c->mc.p, (const float *)c->field_vol.p,
                           (const FieldTile *)c->field_tiles.p, (unsigned *)c->rows.p, slots);
        unsigned long long *d_total = (unsigned long long *)c->rows_off.p + nslots;
        launch_k_scan_rows(dim3(1), dim3(1024), c->stream, (const unsigned *)c->rows.p, (long long)nslots,
                           (unsigned long long *)c->rows_off.p, d_total);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(offs.data(), c->rows_off.p, (nslots + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(stream_wait(c->stream));
        const unsigned long long chunk_total = offs[nslots];
        for (int j = 0; j < nt; j++) {
            const unsigned long long cnt = offs[(size_t)(j + 1) * slots] - offs[(size_t)j * slots];
            kinds[(size_t)work[(size_t)(w0 + j)]] = cnt ? 2 : 1;
            if (cnt) m->st.n_nonempty++; else m->st.n_empty++;
        }
        if (chunk_total) {
            if ((total + chunk_total) * 72 > m->out.bytes) {       // grow the soup (geometric), keeping what is there
                DevBuf bigger;
                if (bigger.ensure(std::max<size_t>((size_t)(total + chunk_total) * 72 * 2, (size_t)1 << 22))) return 1;
                if (total) HIPCHK(hipMemcpyAsync(bigger.p, m->out.p, (size_t)total * 72, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(stream_wait(c->stream));
                m->out.release();
                m->out = bigger;
            }
            launch_k_field_emit(dim3((unsigned)(slots / 256), (unsigned)nt), dim3(256), c->stream, (const McTables *)c->mc.p, (const float *)c->field_vol.p,
                               (const FieldTile *)c->field_tiles.p, (const unsigned long long *)c->rows_off.p, (double *)m->out.p, total,
                               (unsigned long long)(m->out.bytes / 72), slots);
            HIPCHK(hipGetLastError());
            HIPCHK(stream_wait(c->stream));   // (the chunk's buffers are refilled next)
            total += chunk_total;
        }
    }
    m->st.n_triangles = (int64_t)total;
    m->st.n_sampled_voxels = m->st.n_eval_voxels;
    if (m->kinds.ensure((size_t)nb)) return 1;
    // (work items of other shards stay 255 = "other shard", like sdf_generate)
    HIPCHK(hipMemcpyAsync(m->kinds.p, kinds.data(), (size_t)nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(stream_wait(c->stream));
    *out = m;
    m = nullptr;
    return 0;
}

size_t sdf_slab_bytes(int64_t cap_items, int64_t cap_tris) {
    if (cap_items < 0 || cap_tris < 0) return 0;
    return SlabLayout(cap_items, cap_tris).bytes;
}

int sdf_generate_compact_async(sdf_tape *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int bs,
                               int sparse, int64_t shard_index, int64_t shard_count, int precision, void *d_slab,
                               int64_t cap_items, int64_t cap_tris, sdf_mesh **out) {
    if (!d_slab || cap_items < 0 || cap_tris < 0) return fail("sdf_generate_compact_async: slab is NULL or its capacities are negative");
    return generate_entry(t, X, nx, Y, ny, Z, nz, bs, sparse, shard_index, shard_count, precision, d_slab, cap_tris, out, true, cap_items);
}

int sdf_expand_slabs(sdf_ctx *c, const void *const *d_slabs, int n_slabs, int64_t cap_items, int64_t cap_tris, void *d_out, int64_t cap_out) {
    if (!c || !d_slabs || (!d_out && cap_out > 0)) return fail("sdf_expand_slabs: NULL argument");
    if (n_slabs < 1 || n_slabs > 64) return fail("sdf_expand_slabs: 1..64 slabs");
    if (cap_items < 0 || cap_tris < 0 || cap_out < 0) return fail("sdf_expand_slabs: negative capacity");
    if (cap_items == 0 || cap_out == 0) return 0;
    HIPCHK(set_device(c->device));
    SlabPtrs ptrs = {};
    for (int i = 0; i < n_slabs; i++) { if (!d_slabs[i]) return fail("sdf_expand_slabs: NULL slab"); ptrs.p[i] = (const unsigned char *)d_slabs[i]; }
    const unsigned long long blocks = ((unsigned long long)cap_out + 255ull) / 256ull;
    if (blocks > 0x7fffffffull) return fail("sdf_expand_slabs: soup capacity too large for one launch");
    HIPCHK((hipError_t)sdf_launch_expand(c->stream, ptrs, n_slabs, (long long)cap_items, (long long)cap_tris, (double *)d_out, (unsigned long long)cap_out));
    return 0;
}

int sdf_mesh_wait(sdf_mesh *m, int *emitted) {
    if (!m) return fail("sdf_mesh_wait: NULL argument");
    sdf_mesh::Pending &pd = m->pend;
    if (pd.active) {
        sdf_ctx *c = m->ctx;
        HIPCHK(set_device(c->device));
        CallSlot &cs = c->slots[pd.slot];
        HIPCHK(event_wait(cs.done));
        pd.active = false;
        const MeshCounters h = *(const MeshCounters *)((char *)c->h_stage + (size_t)pd.slot * SDF_STAGE_BYTES + SDF_STAGE_BYTES - 256);
        float ms = 0, ms_pre = 0, ms_tot = 0;
        HIPCHK(hipEventElapsedTime(&ms, pd.own_start ? cs.e3 : cs.e2, cs.e4));
        HIPCHK(hipEventElapsedTime(&ms_pre, cs.e0, cs.e2));
        HIPCHK(hipEventElapsedTime(&ms_tot, cs.e0, cs.e4));
        m->st.ms_mesh = ms;
        cs.busy = false; cs.owner = nullptr;      // (everything the slot held for this mesh has been read)
        if (h.overflow & 2u) return fail("sdf_generate: ordered-allocation look-back timed out");
        if (h.overflow & (unsigned)MESH_OVERFLOW_NOT_MESH2) { pd.tape->mesh2_key = pd.key; pd.tape->mesh2_state = 3; }   // (the repeat takes k_mesh)
        if ((h.overflow & (unsigned)MESH_OVERFLOW_NOT_MESH2) && !pd.compact) {
            // k_mesh2 met a tile it does not hold: the call is repeated synchronously, into the caller's buffer, with k_mesh
            const double *X = pd.axes.data(), *Y = X + pd.nx, *Z = Y + pd.ny;
            if (generate_impl(pd.tape, m, X, pd.nx, Y, pd.ny, Z, pd.nz, pd.bs, pd.sparse, pd.shard_index, pd.shard_count, pd.precision,
                              pd.d_out, pd.cap_out, false))
                return 1;
            m->st.n_retries += 1;
        } else if (h.overflow && pd.compact) {
            // a slab that was too small: the exchange protocol retries with larger slabs on EVERY rank (sdf_amd/dist.py)
            finish_stats(pd.tape, m, h, pd.nb, pd.pruning, pd.n_instr, pd.key, ms_pre, ms_tot);
            m->emitted_to = nullptr;
        } else if (h.overflow) {
            // the soup did not fit the caller's buffer: the call is repeated synchronously into library memory
            // (sized from the count just learned)
            pd.tape->hint_key = pd.key; pd.tape->hint_total_tris = std::max<unsigned long long>(h.total, 1);
            const double *X = pd.axes.data(), *Y = X + pd.nx, *Z = Y + pd.ny;
            if (generate_impl(pd.tape, m, X, pd.nx, Y, pd.ny, Z, pd.nz, pd.bs, pd.sparse, pd.shard_index, pd.shard_count, pd.precision,
                              nullptr, 0, false))
                return 1;
            m->st.n_retries += 1;
        } else {
            m->emitted_to = pd.compact ? nullptr : pd.d_out;
            m->st.n_retries = 0;
            finish_stats(pd.tape, m, h, pd.nb, pd.pruning, pd.n_instr, pd.key, ms_pre, ms_tot);
        }
        pd.axes.clear(); pd.axes.shrink_to_fit();
    }
    if (emitted) *emitted = (m->emitted_to != nullptr || m->st.n_triangles == 0) ? 1 : 0;
    return 0;
}

// every reader of a mesh first collects a call that is still in flight
#define MESH_READY(m) do { if ((m)->pend.active && sdf_mesh_wait((m), nullptr)) return 1; } while (0)

int sdf_mesh_stats(sdf_mesh *m, sdf_stats *out) {
    if (!m || !out) return fail("sdf_mesh_stats: NULL argument");
    MESH_READY(m);
    *out = m->st;
    return 0;
}

int64_t sdf_mesh_triangles(sdf_mesh *m) {
    if (!m) return 0;
    if (m->pend.active && sdf_mesh_wait(m, nullptr)) return -1;
    return m->st.n_triangles;
}

// where the soup of a mesh lives: the caller's buffer of sdf_generate_to_device, or the library's
static const void *mesh_soup(const sdf_mesh *m) { return m->emitted_to ? m->emitted_to : m->out.p; }

// a mesh of sdf_generate_records holds 16-byte records; a reader that wants the float64 soup on the device gets it from k_expand, once
static int ensure_soup(sdf_mesh *m) {
    if (!m->records || m->out.p || m->st.n_triangles == 0) return 0;
    sdf_ctx *c = m->ctx;
    HIPCHK(set_device(c->device));
    if (m->out.ensure((size_t)m->st.n_triangles * 72)) return 1;
    SlabPtrs ptrs = {};
    ptrs.p[0] = (const unsigned char *)m->slab.p;
    HIPCHK((hipError_t)sdf_launch_expand(c->stream, ptrs, 1, m->slab_items, m->slab_tris, (double *)m->out.p, (unsigned long long)m->st.n_triangles));
    return 0;
}
#define MESH_SOUP_READY(m) do { if (ensure_soup(m)) return 1; } while (0)

int sdf_mesh_emit_device(sdf_mesh *m, void *d_out) {
    if (!m || !d_out) return fail("sdf_mesh_emit_device: NULL argument");
    MESH_READY(m);
    MESH_SOUP_READY(m);
    sdf_ctx *c = m->ctx;
    if (m->st.n_triangles == 0 || d_out == mesh_soup(m)) return 0;
    HIPCHK(set_device(c->device));
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    HIPCHK(hipMemcpyAsync(d_out, mesh_soup(m), (size_t)m->st.n_triangles * 72, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev[4], c->stream));
    HIPCHK(stream_wait(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[3], c->ev[4]));
    m->st.ms_emit = ms;
    return 0;
}

// (Cutting a large device-to-host copy into pieces that travel on several streams at once was measured in r02: the
// 212 MB soup took 7.7 ms as one copy, 8.7 ms as two, 9.8 ms as four -- one copy already runs at the link's rate for
// pinned memory (28 GB/s on the test boxes).  A kernel that stores straight into the mapped pinned block, 32 to 2048
// workgroups: the same 7.5 ms.  One copy it stays.)
static int copy_to_host(sdf_ctx *c, void *h_dst, const void *d_src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    return 0;
}

// The soup on the host.  A mesh of sdf_generate_records sends its RECORDS (16 bytes per triangle + a transform per work item) and the
// float64 soup is made where it is wanted, by `workers` host threads (<= 0: as many as the machine has, at most 32 -- a caller's number: at most 64; the reference's
// `workers=` argument, sdf/core.py:87) -- block by block while the later records are still on the link: the pieces of the copy are
// followed by events, the calling thread publishes how far the records have arrived and takes blocks itself in between.  The
// arithmetic is k_expand's (`double(local) * scale + offset` on the same operands): the soup is the one the device would write.
// Any other mesh: one copy of the float64 soup, as before (`workers` is ignored).
int sdf_mesh_emit_host_workers(sdf_mesh *m, double *h_out, int workers) {
    if (!m || !h_out) return fail("sdf_mesh_emit_host: NULL argument");
    MESH_READY(m);
    if (m->st.n_triangles == 0) return 0;
    sdf_ctx *c = m->ctx;
    HIPCHK(set_device(c->device));
    if (!m->records || m->out.p) return copy_to_host(c, h_out, mesh_soup(m), (size_t)m->st.n_triangles * 72);
    const long long nt = m->st.n_triangles, ni = (long long)m->work_end - m->work_begin, nraw = std::min<long long>(m->n_raw, SlabLayout(m->slab_items, m->slab_tris).raw_cap);
    const SlabLayout L(m->slab_items, m->slab_tris);
    // pinned staging: [prefix ni x 8 | transforms ni x 48 | raw area nraw x 36 | records nt x 16]
    const size_t off_xf = (size_t)ni * 8, off_raw = off_xf + (size_t)ni * 48, off_rec = (off_raw + (size_t)nraw * 36 + 63) & ~(size_t)63;
    const size_t need = off_rec + (size_t)nt * 16;
    if (c->h_rec_bytes < need) {
        if (c->h_rec) (void)hipHostFree(c->h_rec);
        c->h_rec = nullptr; c->h_rec_bytes = 0;
        const size_t want = need + need / 8 + (1u << 20);
        if (host_malloc(&c->h_rec, want) != hipSuccess) { c->h_rec = nullptr; return fail("sdf_mesh_emit_host: pinned staging for the records"); }
        c->h_rec_bytes = want;
    }
    char *hs = (char *)c->h_rec;
    const char *slab = (const char *)m->slab.p;
    // the pieces: head (prefix, transforms, raw area) first, then the records in ~ 12 pieces of whole blocks
    sdfhost::ExpandJob job;
    static const long long rec_block = [] { const char *e = getenv("SDF_REC_BLOCK"); return e && atoll(e) >= 64 ? atoll(e) : 8192ll; }();     // (tuning)
    static const long long rec_pieces = [] { const char *e = getenv("SDF_REC_PIECES"); return e && atoll(e) >= 1 ? std::min(atoll(e), 64ll) : 12ll; }();
    job.block = rec_block;
    const long long nblk = (nt + job.block - 1) / job.block;
    const long long blk_per_piece = std::max<long long>(8, (nblk + rec_pieces - 1) / rec_pieces);
    const int npieces = (int)((nblk + blk_per_piece - 1) / blk_per_piece);
    while ((int)c->rec_ev.size() < npieces + 1) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->rec_ev.push_back(e);
    }
    hipStream_t st = c->stream;
    static const bool rec_trace = getenv("SDF_REC_TRACE") != nullptr;   // (diagnostics: when the pieces arrived, when the last block was written)
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); };
    // (r06j - r06l, 2 x 64 cores: 8 threads are bound by their own arithmetic (47 us per block of 8192 triangles, 2.2 ms), 32 by the memory
    // the block lies in (138 us per block, 1.8 ms), 64 are no faster and have outliers: the machine's count is capped at 32, a caller's at 64)
    int nthreads = workers > 0 ? std::min(workers, 64) : std::min((int)std::thread::hardware_concurrency(), 32);
    nthreads = std::max(1, nthreads);
    nthreads = (int)std::min<long long>(nthreads, std::max<long long>(nblk, 1));
    job.prefix = (const unsigned long long *)hs; job.xf = (const double *)(hs + off_xf);
    job.raw = (const float *)(hs + off_raw); job.raw_cap = std::max<long long>(nraw, 1);
    job.recs = (const Tri16 *)(hs + off_rec);
    job.n_items = ni; job.n_tris = nt; job.out = h_out;
    std::vector<float> blk_trace;
    if (rec_trace) { blk_trace.assign((size_t)2 * nblk, 0.0f); job.trace = blk_trace.data(); job.t_origin = tr0; }
    static std::mutex expand_mu;                         // (ONE expansion at a time per process: the pool serves one job)
    std::lock_guard<std::mutex> expand_lock(expand_mu);
    sdfhost::Pool &pool = sdfhost::Pool::get();
    // (from here on the helpers hold the job: an error lets them go before it returns)
#define RECCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { job.abort.store(1); pool.wait(job); return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    pool.start(job, nthreads - 1);                       // (the helpers wake up while the copies are enqueued; this thread is one of the workers too)
    RECCHK(hipMemcpyAsync(hs, slab + L.prefix_off, (size_t)ni * 8, hipMemcpyDeviceToHost, st));
    RECCHK(hipMemcpyAsync(hs + off_xf, slab + L.xf_off, (size_t)ni * 48, hipMemcpyDeviceToHost, st));
    if (nraw) RECCHK(hipMemcpyAsync(hs + off_raw, slab + L.raw_off, (size_t)nraw * 36, hipMemcpyDeviceToHost, st));
    RECCHK(hipEventRecord(c->rec_ev[0], st));
    for (int k = 0; k < npieces; k++) {
        const long long t0 = (long long)k * blk_per_piece * job.block, t1 = std::min(nt, t0 + blk_per_piece * job.block);
        RECCHK(hipMemcpyAsync(hs + off_rec + (size_t)t0 * 16, slab + L.tris_off + (size_t)t0 * 16, (size_t)(t1 - t0) * 16, hipMemcpyDeviceToHost, st));
        RECCHK(hipEventRecord(c->rec_ev[(size_t)k + 1], st));
    }
#undef RECCHK
    const double t_enq = tr_us();
    double t_piece[16] = {};
    hipError_t err = event_wait(c->rec_ev[0]);
    const double t_head = tr_us();
    for (int k = 0; k < npieces && err == hipSuccess; k++) {
        err = event_wait(c->rec_ev[(size_t)k + 1]);
        if (err == hipSuccess) job.avail.store(std::min(nt, (long long)(k + 1) * blk_per_piece * job.block), std::memory_order_release);
        if (k < 16) t_piece[k] = tr_us();
    }
    if (err != hipSuccess) job.abort.store(1);
    else sdfhost::expand_work(job);
    const double t_own = tr_us();
    pool.wait(job);
    if (rec_trace) {
        fprintf(stderr, "[records] %lld triangles, %d threads, %d pieces: enqueued %.0f us, head %.0f, pieces", nt, nthreads, npieces, t_enq, t_head);
        for (int k = 0; k < npieces && k < 16; k++) fprintf(stderr, " %.0f", t_piece[k]);
        fprintf(stderr, "; own share done %.0f, all done %.0f us\n", t_own, tr_us());
        double dur = 0, dmax = 0;
        for (long long b = 0; b < nblk; b++) { const double d = blk_trace[2 * b + 1] - blk_trace[2 * b]; dur += d; dmax = std::max(dmax, d); }
        fprintf(stderr, "[records] %lld blocks of %lld triangles: %.0f us each on average (max %.0f); block: started / written, every %lld-th:", nblk, job.block, dur / std::max<long long>(nblk, 1), dmax, std::max<long long>(nblk / 24, 1));
        for (long long b = 0; b < nblk; b += std::max<long long>(nblk / 24, 1)) fprintf(stderr, " %lld: %.0f / %.0f", b, blk_trace[2 * b], blk_trace[2 * b + 1]);
        fprintf(stderr, "\n");
    }
    if (err != hipSuccess) return fail(std::string("sdf_mesh_emit_host: copying the records: ") + hipGetErrorString(err));
    return 0;
}

int sdf_mesh_emit_host(sdf_mesh *m, double *h_out) { return sdf_mesh_emit_host_workers(m, h_out, 0); }

int sdf_mesh_emit_host_range(sdf_mesh *m, int64_t first_tri, int64_t n_tris, double *h_out) {
    if (!m || !h_out) return fail("sdf_mesh_emit_host_range: NULL argument");
    MESH_READY(m);
    if (first_tri < 0 || n_tris < 0 || first_tri + n_tris > m->st.n_triangles) return fail("sdf_mesh_emit_host_range: range outside the soup");
    if (n_tris == 0) return 0;
    MESH_SOUP_READY(m);
    HIPCHK(set_device(m->ctx->device));
    HIPCHK(hipMemcpyAsync(h_out, (const char *)mesh_soup(m) + (size_t)first_tri * 72, (size_t)n_tris * 72, hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(stream_wait(m->ctx->stream));
    return 0;
}

// Where each batch's triangles sit in this shard's soup: after k_mesh every work item's look-back word holds
// the inclusive prefix of the triangle counts up to and including it (ordered_base / publish_count).
int sdf_mesh_batch_offsets(sdf_mesh *m, int64_t *h_out) {
    if (!m || !h_out) return fail("sdf_mesh_batch_offsets: NULL argument");
    MESH_READY(m);
    const int64_t nb = m->st.n_batches;
    for (int64_t b = 0; b <= nb; b++) h_out[b] = 0;
    const int nw = m->work_end - m->work_begin;
    if (nb == 0 || nw <= 0) return 0;
    if (!m->status.p || !m->worklist.p) return fail("sdf_mesh_batch_offsets: this mesh was not produced by sdf_generate");
    HIPCHK(set_device(m->ctx->device));
    std::vector<int> wl((size_t)nw);
    std::vector<unsigned long long> stw((size_t)nw);
    HIPCHK(hipMemcpyAsync(wl.data(), (const int *)m->worklist.p + m->work_begin, (size_t)nw * 4, hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(hipMemcpyAsync(stw.data(), (const unsigned long long *)m->status.p + m->work_begin, (size_t)nw * 8, hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(stream_wait(m->ctx->stream));
    // h_out[b + 1] = triangles of batch b for now; the running sum follows
    unsigned long long prev = 0;
    for (int i = 0; i < nw; i++) {
        if ((stw[(size_t)i] >> 62) != 2ull) return fail("sdf_mesh_batch_offsets: a work item has no prefix (the meshing pass did not complete)");
        const unsigned long long incl = stw[(size_t)i] & MESH_VAL_MASK;
        if (incl < prev || wl[(size_t)i] < 0 || wl[(size_t)i] >= nb) return fail("sdf_mesh_batch_offsets: inconsistent look-back words");
        h_out[wl[(size_t)i] + 1] = (int64_t)(incl - prev);
        prev = incl;
    }
    for (int64_t b = 0; b < nb; b++) h_out[b + 1] += h_out[b];
    return 0;
}

int sdf_mesh_adopt_soup(sdf_ctx *c, const void *d_soup, int64_t n_tris, sdf_mesh **out) {
    if (!c || !out || n_tris < 0 || (n_tris > 0 && !d_soup)) return fail("sdf_mesh_adopt_soup: NULL argument or negative count");
    *out = nullptr;
    sdf_mesh *m = new sdf_mesh();
    m->ctx = c;
    m->emitted_to = const_cast<void *>(d_soup);
    m->st.n_triangles = n_tris;
    *out = m;
    return 0;
}

int sdf_mesh_emit_stl_host(sdf_mesh *m, void *h_out) {
    if (!m || !h_out) return fail("sdf_mesh_emit_stl_host: NULL argument");
    MESH_READY(m);
    const long long nt = m->st.n_triangles;
    if (nt == 0) return 0;
    MESH_SOUP_READY(m);
    sdf_ctx *c = m->ctx;
    HIPCHK(set_device(c->device));
    if (c->scratch_out.ensure((size_t)nt * 50)) return 1;
    launch_k_stl(dim3((unsigned)((nt + 255) / 256)), dim3(256), c->stream, (const double *)mesh_soup(m), nt,
                       (unsigned short *)c->scratch_out.p);
    HIPCHK(hipGetLastError());
    return copy_to_host(c, h_out, c->scratch_out.p, (size_t)nt * 50);
}

int sdf_mesh_weld(sdf_mesh *m, int64_t *n_unique) {
    if (!m || !n_unique) return fail("sdf_mesh_weld: NULL argument");
    MESH_READY(m);
    MESH_SOUP_READY(m);
    sdf_ctx *c = m->ctx;
    HIPCHK(set_device(c->device));
    if (m->weld_n < 0) {
        long long nu = 0;
        const int rc = sdfk::weld_device(c->stream, (const double *)mesh_soup(m), 3ll * (long long)m->st.n_triangles, &m->weld_pts, &m->weld_inv, &nu);
        if (rc) return fail(std::string("sdf_mesh_weld: ") + hipGetErrorString((hipError_t)rc));
        m->weld_n = nu;
    }
    *n_unique = (int64_t)m->weld_n;
    return 0;
}

int sdf_mesh_weld_fetch(sdf_mesh *m, double *h_points, int64_t *h_cells) {
    if (!m || !h_points || !h_cells) return fail("sdf_mesh_weld_fetch: NULL argument");
    if (m->weld_n < 0) return fail("sdf_mesh_weld_fetch: call sdf_mesh_weld first");
    if (m->weld_n == 0) return 0;
    sdf_ctx *c = m->ctx;
    HIPCHK(set_device(c->device));
    HIPCHK(hipMemcpyAsync(h_points, m->weld_pts, (size_t)m->weld_n * 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_cells, m->weld_inv, (size_t)m->st.n_triangles * 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
    return 0;
}

// ---- pinned host memory for results ----
// A device-to-host copy into fresh pageable memory runs at ~10 GB/s (page faults + the runtime's staging);
// into pinned memory it runs at the link rate.  Pinning is expensive (tens of ms for 200 MB), so the
// blocks are recycled: sdf_host_free hands a block back to a small free list, sdf_host_alloc takes the
// smallest block there that is large enough (and not more than twice the request) before pinning new
// memory.  The host side wraps a block as an ndarray whose owner frees it (sdf_amd/engine.py).
struct HostBlock { void *p; size_t bytes; };
static std::mutex g_host_mu;
static std::vector<HostBlock> g_host_free, g_host_live;
static size_t g_host_cached = 0;

int sdf_host_alloc(size_t bytes, void **out) {
    if (!out) return fail("sdf_host_alloc: NULL argument");
    *out = nullptr;
    const size_t want = std::max<size_t>((bytes + 4095) & ~(size_t)4095, 4096);
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        int best = -1;
        for (size_t i = 0; i < g_host_free.size(); i++)
            if (g_host_free[i].bytes >= want && g_host_free[i].bytes <= 2 * want &&
                (best < 0 || g_host_free[i].bytes < g_host_free[(size_t)best].bytes))
                best = (int)i;
        if (best >= 0) {
            const HostBlock b = g_host_free[(size_t)best];
            g_host_free.erase(g_host_free.begin() + best);
            g_host_cached -= b.bytes;
            g_host_live.push_back(b);
            *out = b.p;
            return 0;
        }
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {   // give the cached blocks back and retry once
        std::vector<HostBlock> drop;
        { std::lock_guard<std::mutex> g(g_host_mu); drop.swap(g_host_free); g_host_cached = 0; }
        for (auto &b : drop) (void)hipHostFree(b.p);
        e = hipHostMalloc(&p, want, hipHostMallocDefault);
    }
    if (e != hipSuccess) return fail(std::string("hipHostMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    std::lock_guard<std::mutex> g(g_host_mu);
    g_host_live.push_back({p, want});
    *out = p;
    return 0;
}

int sdf_host_free(void *p) {
    if (!p) return 0;
    HostBlock b{nullptr, 0};
    std::vector<HostBlock> drop;
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        for (size_t i = 0; i < g_host_live.size(); i++)
            if (g_host_live[i].p == p) { b = g_host_live[i]; g_host_live.erase(g_host_live.begin() + (long)i); break; }
        if (!b.p) return fail("sdf_host_free: not a block of sdf_host_alloc");
        g_host_free.push_back(b);
        g_host_cached += b.bytes;
        // keep at most 8 blocks / 2 GiB cached: the oldest go back to the system
        while (g_host_free.size() > 8 || g_host_cached > ((size_t)2 << 30)) {
            drop.push_back(g_host_free.front());
            g_host_cached -= g_host_free.front().bytes;
            g_host_free.erase(g_host_free.begin());
        }
    }
    for (auto &d : drop) (void)hipHostFree(d.p);
    return 0;
}

int sdf_mesh_kinds(sdf_mesh *m, uint8_t *h_out) {
    if (!m || !h_out) return fail("sdf_mesh_kinds: NULL argument");
    MESH_READY(m);
    if (m->st.n_batches == 0) return 0;
    HIPCHK(set_device(m->ctx->device));
    HIPCHK(hipMemcpyAsync(h_out, m->kinds.p, (size_t)m->st.n_batches, hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(stream_wait(m->ctx->stream));
    for (int64_t i = 0; i < m->st.n_batches; i++) if (h_out[i] == 255) h_out[i] = 3;
    return 0;
}

int sdf_mesh_prune_masks(sdf_mesh *m, uint32_t *h_out) {
    if (!m || !h_out) return fail("sdf_mesh_prune_masks: NULL argument");
    MESH_READY(m);
    if (!m->pruned) return fail("sdf_mesh_prune_masks: this mesh was generated without the interval prepass");
    const size_t n = (size_t)m->st.n_batches;
    if (n == 0) return 0;
    HIPCHK(set_device(m->ctx->device));
    HIPCHK(hipMemcpyAsync(h_out, m->prune.p, n * 64, hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(stream_wait(m->ctx->stream));
    return 0;
}

int sdf_mesh_destroy(sdf_mesh *m) {
    if (!m) return 0;
    sdf_ctx *c = m->ctx;
    (void)hipSetDevice(c->device);
    (void)stream_wait(c->stream);
    if (m->stream) (void)stream_wait(m->stream);        // (a call slot's lane)
    if (m->pend.active) { c->slots[m->pend.slot].busy = false; c->slots[m->pend.slot].owner = nullptr; m->pend.active = false; }   // (abandoned; the stream is idle now)
    if (m->out.p) {   // keep one soup buffer around for the next call
        if (c->arena_pool.empty()) c->arena_pool.push_back(m->out);
        else if (c->arena_pool.back().bytes < m->out.bytes) { c->arena_pool.back().release(); c->arena_pool.back() = m->out; }
        else m->out.release();
        m->out.p = nullptr; m->out.bytes = 0;
    }
    if (m->counters.p) { c->counter_pool.push_back(m->counters); m->counters.p = nullptr; m->counters.bytes = 0; }
    for (DevBuf *b : {&m->axes, &m->kinds, &m->worklist, &m->status, &m->prune, &m->tapes, &m->cull, &m->order, &m->desc, &m->cellrecs, &m->trilist, &m->blockidx, &m->slab}) b->release();
    (void)hipFree(m->weld_pts); (void)hipFree(m->weld_inv);
    delete m;
    return 0;
}

}  // extern "C"

#include "sdf_comm.inc"
