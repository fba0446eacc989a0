// sdf_mesh2.h -- the fused sample+march kernel as TWO workgroups of 512 threads per compute unit (round 6, gfx950).
//
//   k_mesh2  the same rounds as k_mesh (sdf_device.h: sample the units k_cull listed into a sparse tile in LDS, count and
//            list the triangles, publish the count, write the PREVIOUS batch's triangles where a decoupled look-back over
//            the status words puts them; reference `_worker`, sdf/core.py:45-60, `_marching_cubes` :16-18) -- but a
//            workgroup is 8 waves, holds <= 128 vector registers and <= 80 KB of LDS, and the launch has two of them per
//            compute unit: the hardware interleaves one workgroup's latency-bound phases (counting, look-back, emission,
//            the barriers between them) with the other's interpreter.  k_mesh's phases are serialised by the barriers of ONE
//            1024-thread workgroup per CU: 56 % of its wave cycles wait and two thirds of the SIMD cycles issue nothing
//            (profiles/r05ae_sync_pmc.json).
//
//            What makes two workgroups fit: there is no dense tile and no second copy of anything.  The two batches a
//            workgroup holds -- the one being sampled and the one waiting for its place in the soup -- share ONE region of LDS
//            from its two ends (even batches grow up from the bottom, odd ones down from the top: samples, then the triangle
//            list); the area the waiting batch's triangles are transposed through IS the sign-bit area (idle then); k_cull's
//            record lands in a work area sized for the tiles this kernel accepts.  A batch that does not fit next to the
//            waiting one makes the workgroup write the waiting one first (it then has the region to itself); what it cannot
//            hold at all -- tiles that are not culled, more than M2_NTL_MAX listed tasks, a list beyond the region -- is
//            FLAGGED (MESH_OVERFLOW_NOT_MESH2) and the host repeats the call with k_mesh, like a soup that did not fit.  There is no
//            parking: a waiting batch whose predecessors have not published waits (the CU's other workgroup has the SIMDs
//            meanwhile), and no cost-ordered tail.  Results are bit-identical to k_mesh's (same interpreter, same counting,
//            same vertex placement, same look-back words).
#pragma once
#include "sdf_device.h"

namespace sdfk {

enum { M2_BLOCK = 512, M2_NWAVE = 8, M2_NTL_MAX = MESH2_NTL_MAX, M2_UNIT_CAP = 8 * M2_NTL_MAX,
       // dynamic LDS: scan buffers | bcast | scan buffers 2 | ntri table | axes | two batch headers | triangle table |
       // two sets of column words | sign bits (= transposition area) | work area (k_cull's record) | the region
       M2_SUMS = 0, M2_BCAST = 64, M2_SUMS2 = 128, M2_NTRI = 192, M2_AXES = 448, M2_HDR = 1248, M2_TRI = 1376,
       M2_COL = 3936, M2_COL_BYTES = 1168, M2_BITS = M2_COL + 2 * M2_COL_BYTES, M2_BITS_BYTES = 4608,
       M2_WORK = M2_BITS + M2_BITS_BYTES, M2_WORK_SSTATE = CULL_ULIST + 2 * M2_UNIT_CAP, M2_WORK_COL = M2_WORK_SSTATE + 1024,
       M2_WORK_BYTES = M2_WORK_COL + 1160, M2_REGION = M2_WORK + M2_WORK_BYTES,
       M2_MIN_LIST = 4096,                 // bytes of list a batch must find behind its samples to be started next to a waiting one
       M2_STAGE_TRIS = 16 };               // triangles per wave and transposition step (8 waves x 16 x 36 B = the sign-bit area)
static_assert(M2_REGION % 16 == 0 && M2_WORK % 16 == 0 && M2_BITS % 16 == 0, "alignment of the LDS areas");
static_assert(M2_NWAVE * M2_STAGE_TRIS * 36 <= M2_BITS_BYTES && M2_NWAVE * M2_STAGE_TRIS * 36 <= M2_WORK_BYTES, "the transposition area");
static_assert((33 * 33 * 33 + 63) / 64 * 8 + 16 <= M2_BITS_BYTES, "sign bits of a full tile + the two words the row extraction reads ahead");
static_assert(M2_HDR + 128 <= M2_TRI && M2_TRI + 2560 <= M2_COL, "header layout");

// header of a batch that waits in the region: double xf[6] (offset[3], scale[3]) | int w, total, smp_off, list_off (bytes)
template <typename T, bool FULL, int NP, int ND, int NS>
__global__ __launch_bounds__(M2_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_mesh2(const uint32_t *__restrict__ code, const T *__restrict__ consts, MeshArgs a_byval) {
    typedef __attribute__((address_space(4))) const MeshArgs KArgs;
    static_assert(sizeof(const uint32_t *) + sizeof(const T *) == 16 && alignof(MeshArgs) <= 16, "MeshArgs sits at byte 16 of the kernel arguments");
    KArgs *ap = (KArgs *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + 16);
    (void)a_byval;
#define KA (*ap)
    typedef Vec<T, NS> V;
    constexpr int BLOCK = M2_BLOCK, NWAVE = M2_NWAVE, RPT = 1024 / BLOCK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *wave_sums = reinterpret_cast<int *>(smem + M2_SUMS);
    int *wave_sums_b = reinterpret_cast<int *>(smem + M2_SUMS2);
    int *bcast = reinterpret_cast<int *>(smem + M2_BCAST);
    unsigned char *ntri_lds = smem + M2_NTRI;
    double *axes = reinterpret_cast<double *>(smem + M2_AXES);
    unsigned short *tri_lds = reinterpret_cast<unsigned short *>(smem + M2_TRI);
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(smem + M2_BITS);
    unsigned *work = reinterpret_cast<unsigned *>(smem + M2_WORK);
    unsigned char *region = smem + M2_REGION;
    int tid = threadIdx.x;
#define SDF_FRESH() asm volatile("" : "+v"(tid))
    const GridDesc g = a_byval.g;
    const int R = KA.slot_bytes;                                   // bytes of the region (host: launch_mesh2)

    if (tid < 256) ntri_lds[tid] = (unsigned char)(KA.mc->ntri[tid] | (KA.mc->amb[tid] << 7));
    for (int i = tid; i < 256 * 5; i += BLOCK) {
        const int cfg = i / 5, j = i - 5 * cfg;
        const signed char *t3 = &KA.mc->tri[cfg][3 * j];
        tri_lds[i] = (unsigned short)(((unsigned)t3[0] & 15u) | (((unsigned)t3[1] & 15u) << 4) | (((unsigned)t3[2] & 15u) << 8));
    }
    for (int i = tid; i < M2_BITS_BYTES / 8; i += BLOCK) bits[i] = 0ull;   // (every round leaves them cleared for the next)
    const int work_begin = KA.ctr->work_begin, work_end = KA.ctr->work_end;
    // SDF_MESH_PROF=1: cycles of thread 0 per phase, summed over the workgroups (k_mesh's words: 0 taking + loading the item, 1 sampling,
    // 2 counting, 48 look-back, 4 emission, 5 the rest; 7 rounds counted twice, 12 emissions, 13 rounds that wrote the waiting batch first),
    // and the workgroup's timeline (start / out of work / done on the 100 MHz counter)
    if (KA.prof && tid == 0) KA.prof[64 + 4 * blockIdx.x] = wall_clock64();
    long long tprev = KA.prof ? clock64() : 0;
#define M2_PROF(K) do { if (KA.prof && tid == 0) { const long long tn = clock64(); atomicAdd(&KA.prof[K], (unsigned long long)(tn - tprev)); tprev = tn; } } while (0)
    if (tid == 0) {   // the kernel's start on the device's own clock (sdf_stats.ms_mesh_device, sclk_mhz)
        const unsigned long long tw = wall_clock64();
        atomicMax(&KA.ctr->t_first_inv, ~tw);
        if (blockIdx.x == 0) { KA.ctr->clk_cycles = (unsigned long long)clock64(); KA.ctr->clk_ticks = tw; }
    }
    auto settle = [&](int w_, unsigned long long excl, unsigned long long total_) {
        if (excl == ~0ull) atomicOr(&KA.ctr->overflow, 2u);           // look-back timed out (never expected)
        else if (excl + total_ > KA.out_cap) atomicOr(&KA.ctr->overflow, 1u);
        if (w_ == work_end - 1 && excl != ~0ull) KA.ctr->total = excl + total_;
    };
    auto hdr_xf = [&](int s_) { return reinterpret_cast<double *>(smem + M2_HDR + 64 * s_); };
    auto hdr_i = [&](int s_) { return reinterpret_cast<int *>(smem + M2_HDR + 64 * s_ + 48); };
    auto col_of = [&](int s_) { return reinterpret_cast<unsigned *>(smem + M2_COL + M2_COL_BYTES * s_); };

    int wait_side = -1, wait_bytes = 0;   // the batch that waits for its place: which end of the region, how much of it
    bool hold = false;        // the work item in `w` was taken in an earlier round (which wrote the waiting batch first)
    bool recount = false;     // ... and its samples and sign bits are in place already (its list did not fit next to the waiting batch)
    bool nx_valid = false;    // bcast[8] holds the counter value of the next item (drawn by the last wave during an emission)
    bool have = false;        // ... and its batch index / header are in bcast[9..10], record in the work area, axes in LDS
    int held_side = 0;        // recount: the end of the region the held batch's samples lie at
    int w = 0;
    for (;;) {
        SDF_FRESH();
        asm volatile("" : "+s"(ap));
        wait_side = uni(wait_side); wait_bytes = uni(wait_bytes); held_side = uni(held_side);
        if (!hold) {
            if (nx_valid) {
                if (tid == 0) bcast[0] = work_begin + bcast[8];
                have = uni(bcast[11]) != 0;
            } else {
                if (tid == 0) bcast[0] = work_begin + (int)atomicAdd(&KA.ctr->work_counter, 1u);
                have = false;
            }
            nx_valid = false;
            __syncthreads();
            w = bcast[0];
        }
        w = uni(w);
        const bool was_recount = recount;
        hold = false; recount = false;
        const bool finished = w >= work_end;
        if (finished && KA.prof && tid == 0 && KA.prof[64 + 4 * blockIdx.x + 1] == 0) KA.prof[64 + 4 * blockIdx.x + 1] = wall_clock64();
        if (finished && wait_side < 0) break;
        // ---- what kind of tile?  where does it go? ----
        bool flush_only = finished;       // this round only writes the waiting batch
        int b = 0, ntl = 0;
        bool degenerate = false;
        if (!finished) {
            int b_v;
            unsigned n0_v;
            if (have) { b_v = bcast[9]; n0_v = (unsigned)bcast[10]; }
            else {
                b_v = KA.worklist[w];
                n0_v = reinterpret_cast<const unsigned *>(KA.cull + (size_t)w * CULL_RECORD)[0];
            }
            b = uni(b_v);
            const unsigned n0 = (unsigned)uni((int)n0_v) & 0xFFFFu;
            ntl = (int)((n0 + 7u) >> 3);
            if (n0 == 0xFFFFu) { ntl = 0; degenerate = true; }   // (not culled: a tile without cells -- else not this kernel's, below)
        }
        int lx = 2, ly = 2, lz = 2;
        int ox = 0, oy = 0, oz = 0;
        if (!finished) batch_origin(g, b, ox, oy, oz, lx, ly, lz);
        const int smp_bytes = 256 * ntl;
        bool unsupported = false;
        unsigned why = 0;                 // (diagnostics: which limit a flagged tile met -- bits 5.. of the overflow word, SDF_MESH2_DEBUG)
        if (!finished) {
            if (degenerate && lx > 1 && ly > 1 && lz > 1) { unsupported = true; why = 32u; }              // a tile k_cull left dense
            if (ntl > M2_NTL_MAX || smp_bytes + M2_MIN_LIST > R) { unsupported = true; why = ntl > M2_NTL_MAX ? 64u : 128u; }
            if (!unsupported && !was_recount && wait_side >= 0 && wait_bytes + smp_bytes + M2_MIN_LIST > R) { flush_only = true; hold = true; }
        }
        const int side = was_recount ? held_side : (wait_side == 0 ? 1 : 0);
        unsigned char *cs = region + (side == 0 ? 0 : R - smp_bytes);                      // the batch's samples
        float *smp = reinterpret_cast<float *>(cs);
        const int list_free = R - (wait_side >= 0 ? wait_bytes : 0) - smp_bytes;           // bytes behind the samples
        int total = 0;
        bool counted = false;             // this batch has been counted and waits in the region from here on
        unsigned long long pre_dq = 0;
        if (!flush_only && (unsupported || degenerate)) {
            // ---- nothing to mesh here: an empty batch (or one the host will mesh with k_mesh: flagged) ----
            if (tid == 0) {
                if (unsupported) atomicOr(&KA.ctr->overflow, (unsigned)MESH_OVERFLOW_NOT_MESH2 | why);
                atomicAdd(&KA.ctr->n_empty, 1u);
                atomicAdd(&KA.ctr->n_eval, (unsigned long long)(lx * ly * lz));
                atomicAdd(&KA.ctr->n_sampled, (unsigned long long)(lx * ly * lz));
                KA.kinds[b] = 1;
                if (KA.tape_stride) {
                    const uint32_t *wc = code + (size_t)b * (size_t)KA.tape_stride * 2;
                    atomicAdd(&KA.ctr->n_pruned, (unsigned long long)KA.n_instr - reinterpret_cast<const unsigned long long *>(wc)[KA.tape_stride - 1]);
                }
                if (KA.compact && w - work_begin < KA.xf_cap) {
                    double *xf = KA.xf + (size_t)(w - work_begin) * 6;
                    const double x0 = g.X[ox], y0 = g.Y[oy], z0 = g.Z[oz];
                    xf[0] = x0; xf[1] = y0; xf[2] = z0;
                    xf[3] = lx > 1 ? g.X[ox + 1] - x0 : 0.0; xf[4] = ly > 1 ? g.Y[oy + 1] - y0 : 0.0; xf[5] = lz > 1 ? g.Z[oz + 1] - z0 : 0.0;
                }
            }
            if (tid < 64) {   // count 0, and its prefix (every item of a finished call carries one: sdf_mesh_batch_offsets, k_pack_slab)
                publish_count(KA.status, w, work_begin, 0ull);
                if (!unsupported) {
                    const unsigned long long excl = ordered_base(KA.status, w, work_begin, 0ull, MESH_SPIN_FOREVER, lookback_prefetch(KA.status, w, work_begin));
                    if (tid == 0) settle(w, excl, 0ull);
                }
            }
            flush_only = wait_side >= 0;
            if (!flush_only) { __syncthreads(); continue; }   // (uniform; bcast is rewritten at the top)
        } else if (!flush_only) {
            const TileTasks tt(lx, ly, lz);
            const int nvox = tt.nvox, lyz = tt.lyz;
            const int wave = tid >> 6, lane = tid & 63;
            const uint32_t *wcode = code + (size_t)__builtin_amdgcn_readfirstlane(b) * (size_t)KA.tape_stride * 2;
            if (!was_recount) {
                // k_cull's record of the batch: units and sub-group states into the work area, the column words to the batch's side
                if (have) {
                    for (int i = tid; i < 289; i += BLOCK) col_of(side)[i] = work[M2_WORK_COL / 4 + i];
                } else {
                    const unsigned *rec = reinterpret_cast<const unsigned *>(KA.cull + (size_t)w * CULL_RECORD);
                    const int nwords = (CULL_ULIST + 16 * ntl + 3) >> 2;
                    for (int i = tid; i < nwords; i += BLOCK) work[i] = rec[i];
                    for (int i = tid; i < 256; i += BLOCK) work[M2_WORK_SSTATE / 4 + i] = rec[CULL_SSTATE / 4 + i];
                    for (int i = tid; i < 289; i += BLOCK) col_of(side)[i] = rec[CULL_COLINFO / 4 + i];
                    if (tid < lx) axes[tid] = g.X[ox + tid];
                    else if (tid >= 64 && tid < 64 + ly) axes[33 + tid - 64] = g.Y[oy + tid - 64];
                    else if (tid >= 128 && tid < 128 + lz) axes[66 + tid - 128] = g.Z[oz + tid - 128];
                }
                __syncthreads();
                M2_PROF(0);
                if (KA.tape_stride && tid == 0)
                    atomicAdd(&KA.ctr->n_pruned, (unsigned long long)KA.n_instr - reinterpret_cast<const unsigned long long *>(wcode)[KA.tape_stride - 1]);
                // ---- 1a. sign bits of the decided sub-groups, straight from their two-bit states (k_mesh: cull-sign-fill) ----
                const unsigned short *units = reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(work) + CULL_ULIST);
                const unsigned *sstate = work + M2_WORK_SSTATE / 4;
                {
                    const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
                    const int hlast = (c2 - 1) >> 1;
                    for (int r = tid; r < lx * ly; r += BLOCK) {
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
                if (tid == 0) atomicAdd(&KA.ctr->n_sampled, (unsigned long long)ntl * 64ull);
                // ---- 1c. the listed tasks through the interpreter, NS per wave and pass; sample `lane` of task t at smp[64 t + lane] ----
                for (int t0 = wave * NS; t0 < ntl; t0 += NWAVE * NS) {
                    V px, py, pz;
                    SDF_UNROLL
                    for (int k = 0; k < NS; k++) {
                        const int tk = min(t0 + k, ntl - 1);
                        int ix, iy, iz;
                        cull_sample(units, tk, lane, lx, ly, lz, ix, iy, iz);
                        px.v[k] = (T)axes[ix]; py.v[k] = (T)axes[33 + iy]; pz.v[k] = (T)axes[66 + iz];
                    }
                    const V val = run_tape<T, FULL, NP, ND, NS>(wcode, consts, px, py, pz);
                    SDF_UNROLL
                    for (int k = 0; k < NS; k++) {
                        int ix, iy, iz;
                        const bool valid = t0 + k < ntl && cull_sample(units, t0 + k, lane, lx, ly, lz, ix, iy, iz);
                        if (valid) {
                            const int i = ix * lyz + iy * tt.lz + iz;
                            const float fv = (float)val.v[k];
                            smp[64 * (t0 + k) + lane] = fv;
                            if (fv > 0.0f) atomicOr(&bits[i >> 6], 1ull << (i & 63));
                        }
                    }
                }
                __syncthreads();
                SDF_FRESH();
                M2_PROF(1);
            }
            const TileView cvw{smp, col_of(side), lyz, lz, true};
            // ---- 2. count (wave 0 first asks for the waiting batch's predecessors: the answer arrives meanwhile) ----
            if (tid < 64 && wait_side >= 0) pre_dq = lookback_prefetch(KA.status, hdr_i(wait_side)[0], work_begin);
            const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
            const int nrows = c0 * c1;
            const float inv_c1 = 1.0f / (float)c1;
            auto row_signs = [&](int i0, int i1, unsigned long long *rb) -> unsigned {
                SDF_UNROLL
                for (int q = 0; q < 4; q++) {   // q = 2 * o0 + o1
                    const int o = (i0 + (q >> 1)) * lyz + (i1 + (q & 1)) * lz;
                    const unsigned long long w0 = bits[o >> 6], w1 = bits[(o >> 6) + 1];
                    const int sh = o & 63;
                    rb[q] = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
                }
                const unsigned long long any = rb[0] | rb[1] | rb[2] | rb[3];
                const unsigned long long all = rb[0] & rb[1] & rb[2] & rb[3];
                const unsigned long long ones = all & (all >> 1), zeros = ~any & ~(any >> 1);
                return (unsigned)(~(ones | zeros)) & (c2 >= 32 ? 0xFFFFFFFFu : ((1u << c2) - 1u));
            };
            int scan_ix = 0;
#define M2_SCAN(VAL, TOT) block_exclusive_scan1<BLOCK>((VAL), (scan_ix++ & 1) ? wave_sums_b : wave_sums, (TOT))
            int ncells = 0;
            unsigned row_mask[RPT];
            int row_cell0[RPT];
            SDF_UNROLL
            for (int k = 0; k < RPT; k++) {
                const int r = tid + k * BLOCK;
                unsigned mask = 0;
                if (r < nrows) {
                    unsigned long long rb[4];
                    const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                    mask = row_signs(i0, i1, rb);
                }
                row_mask[k] = mask;
                int tot;
                row_cell0[k] = ncells + M2_SCAN(__popc(mask), tot);
                ncells += tot;
            }
            ncells = uni(ncells);
            // The free area behind the samples, F entries, numbered AWAY from the samples (side 0: upwards from their end; side 1:
            // downwards from their start): the triangle list grows from entry 0, the cell table -- one entry per surface cell, needed
            // only until the cell's triangles are listed -- lies at its far end.  The cells go through in chunks of BLOCK (a thread per
            // cell); a chunk's triangles are listed right behind its scan, as long as the list stays below the table's unread part.
            unsigned *fa0 = reinterpret_cast<unsigned *>(side == 0 ? cs + smp_bytes : cs);
            const int fdir = side == 0 ? 1 : -1, foff = side == 0 ? 0 : -1;
#define M2_FA(D) fa0[fdir * (D) + foff]
            const int F = list_free >> 2;
            bool fits = ncells <= F;
            int my_amb = 0;
            if (fits) {   // (uniform)
                const int tab0 = F - ncells;
                SDF_UNROLL
                for (int k = 0; k < RPT; k++) {
                    const int r = tid + k * BLOCK;
                    unsigned m = row_mask[k];
                    int pos = tab0 + row_cell0[k];
                    while (m) {
                        const int i2 = __ffs((int)m) - 1;
                        m &= m - 1u;
                        M2_FA(pos) = (unsigned)r | ((unsigned)i2 << 10);
                        pos++;
                    }
                }
                __syncthreads();
                for (int s0 = 0; s0 < ncells; s0 += BLOCK) {   // (uniform)
                    const int sidx = s0 + tid;
                    int n = 0;
                    unsigned info = 0;
                    if (sidx < ncells) {
                        const unsigned ce = M2_FA(tab0 + sidx);
                        const int r = (int)(ce & 1023u), i2 = (int)((ce >> 10) & 31u), i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                        unsigned long long rb[4];
                        row_signs(i0, i1, rb);
                        const unsigned cfg = cell_config(rb, i2);
                        const unsigned e = ntri_lds[cfg];
                        if (e & 128u) {   // Lewiner's tests on the 8 corner samples pick the tiling (rare)
                            float c8[8];
                            double lv[8];
                            int off;
                            cvw.cell(i0, i1, i2, c8);
                            mc33_load_cell(c8, 4, 2, lv);
                            n = mc33_cell(lv, KA.mc->mc33, &off);
                            my_amb++;
                        } else n = (int)(e & 7u);
                        // entry: cell (15 bits) | ambiguous (1) | configuration (8) | triangle in cell (4)
                        info = ((unsigned)((i0 << 10) | (i1 << 5) | i2) << 13) | ((e & 128u) << 5) | (cfg << 4);
                    }
                    int tot;
                    const int off = total + M2_SCAN(n, tot);
                    const int t_new = uni(total + tot);
                    if (t_new > min(F, tab0 + s0 + BLOCK)) { fits = false; break; }   // (uniform) the list would run into the table's unread part
                    for (int j = 0; j < n; j++) M2_FA(off + j) = info | (unsigned)j;
                    total = t_new;
                }
            }
#undef M2_SCAN
            if (fits) {   // (uniform)
                // the sign bits are dead from here on: cleared NOW for the next tile's sign fill (behind this round's remaining barriers)
                for (int i = tid; i < M2_BITS_BYTES / 8; i += BLOCK) bits[i] = 0ull;
                if (tid < 64) publish_count(KA.status, w, work_begin, (unsigned long long)total);
                if (tid == 0) {
                    double *xf = hdr_xf(side);
                    xf[0] = axes[0]; xf[1] = axes[33]; xf[2] = axes[66];
                    xf[3] = axes[1] - axes[0]; xf[4] = axes[34] - axes[33]; xf[5] = axes[67] - axes[66];
                    int *m = hdr_i(side);
                    m[0] = w; m[1] = total; m[2] = (int)(cs - region); m[3] = (int)(reinterpret_cast<unsigned char *>(fa0) - region);
                    if (KA.compact && w - work_begin < KA.xf_cap) {   // the batch's transform travels with the compact soup
                        double *x2 = KA.xf + (size_t)(w - work_begin) * 6;
                        SDF_UNROLL for (int q = 0; q < 6; q++) x2[q] = xf[q];
                    }
                    atomicAdd(total ? &KA.ctr->n_nonempty : &KA.ctr->n_empty, 1u);
                    atomicAdd(&KA.ctr->n_eval, (unsigned long long)nvox);
                    KA.kinds[b] = total ? 2 : 1;
                }
                if (my_amb) atomicAdd(&KA.ctr->n_ambiguous, (unsigned long long)my_amb);
                counted = true;
            } else if (wait_side >= 0) {
                // the list does not fit next to the waiting batch: that one is written first, then this batch is counted again with
                // the region to itself (its samples and sign bits stay where they are)
                hold = true; recount = true; held_side = side; total = 0;
                if (KA.prof && tid == 0) atomicAdd(&KA.prof[7], 1ull);
            } else {
                // ... nor alone: not this kernel's (flagged; the host repeats the call with k_mesh)
                for (int i = tid; i < M2_BITS_BYTES / 8; i += BLOCK) bits[i] = 0ull;
                if (tid < 64) publish_count(KA.status, w, work_begin, 0ull);
                if (tid == 0) { atomicOr(&KA.ctr->overflow, (unsigned)MESH_OVERFLOW_NOT_MESH2 | (ncells > F ? 256u : 512u)); KA.kinds[b] = 1; }
                total = 0;
            }
        }
        SDF_FRESH();
        __syncthreads();   // (the header, the list)
        M2_PROF(2);
        if (KA.prof && tid == 0 && hold && !recount) atomicAdd(&KA.prof[13], 1ull);

        // ---- 3 + 4. the waiting batch's triangles, one round after it was counted ----
        if (wait_side >= 0) {   // (uniform)
            const int e_w = uni(hdr_i(wait_side)[0]), e_total = uni(hdr_i(wait_side)[1]);
            const float *e_smp = reinterpret_cast<const float *>(region + uni(hdr_i(wait_side)[2]));
            const unsigned *e_fa0 = reinterpret_cast<const unsigned *>(region + uni(hdr_i(wait_side)[3]));   // triangle t at e_fa0[e_dir * t + e_off]
            const int e_dir = wait_side == 0 ? 1 : -1, e_off = wait_side == 0 ? 0 : -1;
            const TileView vw{e_smp, col_of(wait_side), 0, 0, true};
            const double *exf = hdr_xf(wait_side);
            const double of0 = exf[0], of1 = exf[1], of2 = exf[2], sc0 = exf[3], sc1 = exf[4], sc2 = exf[5];
            const bool asked = !flush_only && !unsupported && !degenerate;   // (pre_dq was asked for during this round's counting)
            if (tid < 64) {
                const unsigned long long pre = asked ? pre_dq : lookback_prefetch(KA.status, e_w, work_begin);
                const unsigned long long excl = ordered_base(KA.status, e_w, work_begin, (unsigned long long)e_total, MESH_SPIN_FOREVER, pre);
                if (tid == 0) {
                    settle(e_w, excl, (unsigned long long)e_total);
                    reinterpret_cast<unsigned long long *>(bcast + 2)[0] = excl;
                    bcast[12] = 0;   // (the triangles are handed to the waves in chunks of 64: below)
                }
            }
            __syncthreads();
            // ---- the last wave takes the next work item while the others start on the triangles: item, record, axes into LDS
            // (the work area and `axes` are idle; not while an item is held, whose record may still be needed) ----
            const bool prefetch = !finished && !hold;   // (uniform)
            if (prefetch && tid >= BLOCK - 64) {
                const int ln = tid & 63;
                int idx = 0;
                if (ln == 0) idx = (int)atomicAdd(&KA.ctr->work_counter, 1u);
                idx = uni(idx);
                const int nw_ = work_begin + idx;
                int loaded = 0, nb_ = 0;
                unsigned nn0 = 0xFFFFu;
                if (nw_ < work_end) {   // (wave-uniform)
                    const unsigned *rec = reinterpret_cast<const unsigned *>(KA.cull + (size_t)nw_ * CULL_RECORD);
                    const int nbv = KA.worklist[nw_];
                    const unsigned n0v = rec[0];
                    nb_ = uni(nbv);
                    nn0 = (unsigned)uni((int)n0v) & 0xFFFFu;
                    int nox, noy, noz, nlx, nly, nlz;
                    batch_origin(g, nb_, nox, noy, noz, nlx, nly, nlz);
                    if (ln < nlx) axes[ln] = g.X[nox + ln];
                    if (ln < nly) axes[33 + ln] = g.Y[noy + ln];
                    if (ln < nlz) axes[66 + ln] = g.Z[noz + ln];
                    if (nn0 != 0xFFFFu && nn0 <= (unsigned)M2_UNIT_CAP) {
                        const int nwords = (int)((CULL_ULIST + 2u * ((nn0 + 7u) & ~7u) + 3u) >> 2);
                        for (int i = ln; i < nwords; i += 64) work[i] = rec[i];
                        for (int i = ln; i < 256; i += 64) work[M2_WORK_SSTATE / 4 + i] = rec[CULL_SSTATE / 4 + i];
                        for (int i = ln; i < 290; i += 64) work[M2_WORK_COL / 4 + i] = rec[CULL_COLINFO / 4 + i];
                    }
                    loaded = 1;
                }
                if (ln == 0) { bcast[8] = idx; bcast[9] = nb_; bcast[10] = (int)nn0; bcast[11] = loaded; }
            }
            if (prefetch) nx_valid = true;
            const unsigned long long base = uni64(reinterpret_cast<unsigned long long *>(bcast + 2)[0]);
            M2_PROF(48);
            if (KA.prof && tid == 0) atomicAdd(&KA.prof[12], 1ull);
            const bool fits = base != ~0ull && base + (unsigned long long)e_total <= KA.out_cap;
            // the transposition area: the sign bits' (cleared above, cleared again below) -- or, while a held batch's sign bits must
            // survive (recount), the work area (no item is taken then, and the held batch's record has served)
            float *stg = reinterpret_cast<float *>(recount ? smem + M2_WORK : smem + M2_BITS) + (tid >> 6) * (M2_STAGE_TRIS * 9);
            double *dst0 = KA.out + base * 9ull;
            for (; fits;) {
                int ch = 0;
                if ((tid & 63) == 0) ch = atomicAdd(&bcast[12], 1);
                const int t0 = 64 * uni(ch);
                if (t0 >= e_total) break;
                const int t = t0 + (tid & 63);
                const bool live = t < e_total;
                const unsigned e = e_fa0[e_dir * (live ? t : e_total - 1) + e_off];
                const int j = (int)(e & 15u), cfg = (int)((e >> 4) & 255u), cell = (int)(e >> 13);
                const int i0 = cell >> 10, i1 = (cell >> 5) & 31, i2 = cell & 31;
                float o[9];
                if (e & 4096u) {
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
                if (KA.compact) {   // (uniform) the exchange's 16-byte record, straight from the registers
                    if (live) store_tri16(Tri16Sink{KA.out, KA.raw, KA.raw_cap, &KA.ctr->n_raw}, base + (unsigned long long)t, o);
                } else {
                    // through LDS: a wave's 64 triangles in four steps of 16 (144 coordinates: lane l stores coordinates l, 64 + l,
                    // 128 + l of the step -- consecutive lanes, consecutive addresses).  Coordinate c belongs to axis c % 3 and
                    // 64 % 3 == 1: the axis of a lane's k-th coordinate is (l + k) % 3.
                    const int ln = tid & 63;
                    const int a0 = ln % 3;
                    const double s_[3] = {a0 == 0 ? sc0 : (a0 == 1 ? sc1 : sc2), a0 == 0 ? sc1 : (a0 == 1 ? sc2 : sc0), a0 == 0 ? sc2 : (a0 == 1 ? sc0 : sc1)};
                    const double o_[3] = {a0 == 0 ? of0 : (a0 == 1 ? of1 : of2), a0 == 0 ? of1 : (a0 == 1 ? of2 : of0), a0 == 0 ? of2 : (a0 == 1 ? of0 : of1)};
                    SDF_UNROLL
                    for (int h = 0; h < 64 / M2_STAGE_TRIS; h++) {
                        if (live && (ln / M2_STAGE_TRIS) == h) { SDF_UNROLL for (int q = 0; q < 9; q++) stg[(ln % M2_STAGE_TRIS) * 9 + q] = o[q]; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int nval = min(M2_STAGE_TRIS, e_total - t0 - M2_STAGE_TRIS * h) * 9;   // (<= 0: nothing)
                        double *dstw = dst0 + (size_t)(t0 + M2_STAGE_TRIS * h) * 9;
                        SDF_UNROLL for (int k = 0; k < 3; k++) { const int c = 64 * k + ln; if (c < nval) SDF_SOUP_STORE(dstw + c, (double)stg[c] * s_[k % 3] + o_[k % 3]); }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();   // (the area is rewritten by the next step)
                    }
                }
            }
            __syncthreads();   // (the region, bcast, the transposition area are reused)
            M2_PROF(4);
            if (!recount && !KA.compact) {   // the sign bits' area served as the transposition area: cleared for the next sign fill
                for (int i = tid; i < M2_BITS_BYTES / 8; i += BLOCK) bits[i] = 0ull;
                __syncthreads();
            }
        }
        if (counted) { wait_side = side; wait_bytes = smp_bytes + ((4 * total + 15) & ~15); }
        else { wait_side = -1; wait_bytes = 0; }
        M2_PROF(5);
        if (finished) break;
    }
    if (KA.prof && tid == 0) KA.prof[64 + 4 * blockIdx.x + 2] = wall_clock64();
    if (tid == 0) {
        const unsigned long long tw = wall_clock64();
        atomicMax(&KA.ctr->t_last, tw);
        if (blockIdx.x == 0) { KA.ctr->clk_cycles = (unsigned long long)clock64() - KA.ctr->clk_cycles; KA.ctr->clk_ticks = tw - KA.ctr->clk_ticks; }
    }
#undef M2_FA
#undef M2_PROF
#undef SDF_FRESH
#undef KA
}

// host-side launcher of one FULL family (sdf_mesh2_inst.hip); slots as for k_mesh: 0 = (1,1), 1 = (2,2), 3 = (2,4) register files (the others: -1, not built)
#define SDF_DECLARE_MESH2_LAUNCH(NAME, T) \
    int NAME(int slots, int grid, size_t lds, hipStream_t stream, const uint32_t *code, const T *consts, const MeshArgs &a)
SDF_DECLARE_MESH2_LAUNCH(sdf_launch_mesh2_f64, double);
SDF_DECLARE_MESH2_LAUNCH(sdf_launch_mesh2_f64_full, double);

}  // namespace sdfk
