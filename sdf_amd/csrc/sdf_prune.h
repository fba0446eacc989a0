// sdf_prune.h -- the interval prepass kernel body: per batch, the tape without the instructions
// that cannot matter inside that batch (the interval machinery itself is in sdf_interval.h).
#pragma once
#include "sdf_device.h"
#include "sdf_interval.h"

namespace sdfk {

// (the tape's float64 constant pool and rstart / lstart -- per instruction: first instruction of the right operand /
// of the left chain -- are kernel parameters of their own: as `__restrict__` arguments their loads are scalar
// loads; as members of this struct they were vector loads the arithmetic waited for)
struct PruneArgs {
    int n_instr;                     // instructions of the tape, END included
    int n_p, n_d;                    // saved-point / saved-distance slots the tape uses (>= 1 each)
    uint32_t *masks_out;             // [batch][16]: skip / forced bits (diagnostics)
    unsigned long long *tapes_out;   // [batch][tape_stride]: the batch's own tape; word tape_stride-1 = its length
    int tape_stride;
    int first_block;                 // workgroups >= first_block of the prepass kernel run the interval pass
    uint32_t zero_off;               // consts[zero_off + 1] == +0.0 (appended by sdf_tape_create)
    const int *worklist;             // k_prune_list: the batches to prune are this shard's slice of the work list
    const MeshCounters *ctr;         //               (NULL: every batch, by index)
};

// The interval pass of one workgroup of the prepass kernel (k_skip, sdf_hip.hip): 8 lanes per batch
// (lane = octant of the batch's box of sample coordinates), PRUNE_BLOCK / 8 batches per workgroup.
// It runs for EVERY batch, next to the skip test rather than after it: the work list is not known
// yet, and a launch of its own behind k_compact would sit on the critical path.
template <bool FULL, bool RARE>
__device__ __forceinline__ void prune_block(const uint32_t *__restrict__ code, const double *__restrict__ c64,
                                            const uint16_t *__restrict__ rstart, const uint16_t *__restrict__ lstart,
                                            const PruneArgs &pa, const GridDesc &g, int nbatches, int block, double *lds) {
    const int gid = block * PRUNE_BLOCK + threadIdx.x;
    const int oct = gid & 7;
    int b = gid >> 3;
    bool live = b < nbatches;
    if (pa.worklist) {               // (work items of this shard instead of all batches)
        const int w = pa.ctr->work_begin + (gid >> 3);
        live = w < pa.ctr->work_end;
        b = live ? pa.worklist[w] : 0;
    }
    Ival bx = ia::pt(0.0), by = ia::pt(0.0), bz = ia::pt(0.0);
    if (live) {
        int ox, oy, oz, lx, ly, lz;
        batch_origin(g, b, ox, oy, oz, lx, ly, lz);
        const int hx = (lx - 1) / 2, hy = (ly - 1) / 2, hz = (lz - 1) / 2;      // the octants share the middle sample
        bx = (oct & 4) ? Ival{g.X[ox + hx], g.X[ox + lx - 1]} : Ival{g.X[ox], g.X[ox + hx]};
        by = (oct & 2) ? Ival{g.Y[oy + hy], g.Y[oy + ly - 1]} : Ival{g.Y[oy], g.Y[oy + hy]};
        bz = (oct & 1) ? Ival{g.Z[oz + hz], g.Z[oz + lz - 1]} : Ival{g.Z[oz], g.Z[oz + hz]};
        if (bx.lo > bx.hi) { const double t = bx.lo; bx.lo = bx.hi; bx.hi = t; }     // (a descending axis)
        if (by.lo > by.hi) { const double t = by.lo; by.lo = by.hi; by.hi = t; }
        if (bz.lo > bz.hi) { const double t = bz.lo; bz.lo = bz.hi; bz.hi = t; }
        if (ia::bad(bx) || ia::bad(by) || ia::bad(bz)) { bx = by = bz = ia::top(); }  // (NaN coordinates)
    }
    uint32_t masks[16];
    for (int k = 0; k < 16; k++) masks[k] = 0;
    ia_run_tape<true, FULL, RARE>(code, c64, rstart, lstart, pa.n_instr, bx, by, bz, live, IaShared{lds, pa.n_p, PRUNE_BLOCK}, pa.n_d, masks);
    const bool store = live && oct == 0;
    unsigned long long *out = pa.tapes_out + (size_t)(live ? b : 0) * pa.tape_stride;
    const int n = compact_tape(reinterpret_cast<const unsigned long long *>(code), pa.n_instr, masks, out, store, pa.zero_off);
    if (store) {
        out[n] = out[n - 1];                                    // the interpreter looks one instruction ahead
        out[pa.tape_stride - 1] = (unsigned long long)n;
        for (int k = 0; k < 16; k++) pa.masks_out[(size_t)b * 16 + k] = masks[k];
    }
}

}  // namespace sdfk
