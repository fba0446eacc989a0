// sdf_plain.hip -- every kernel of the library that is NOT a tape interpreter: the ordered compaction of the work list,
// marching cubes of caller-supplied volumes and tiles, the two-pass meshing's scan and emission, the multi-GPU exchange
// unit's kernels (k_pack_slab / k_expand / k_collect_headers), the STL records.
//
// A translation unit of its own because it is built WITHOUT -structurizecfg-skip-uniform-regions (build.sh).  The tape
// interpreters (sdf_hip.hip, sdf_mesh_inst.hip) need that option -- their dispatch is a scalar jump through a table
// (asm goto + s_setpc, sdf_interp.h) that the structurizer must leave alone -- but it is not safe for kernels whose
// lanes diverge: r03's k_expand, written here first in the interpreters' unit, came out wrong for every workgroup that
// straddled into the last slab (a divergent search loop next to a uniform one), and right without the option.
// tests/test_gpu.py::test_expand_synthetic_slabs holds that case.
#include <algorithm>

#include "sdf_device.h"
#include "sdf_plain.h"
#include "sdf_slab.h"
#include "sdf_split.h"

using namespace sdfk;

// ordered compaction of the pending batches into the work list (single workgroup)
// (also clears the look-back words and the counters of the meshing pass that follows, so the
// common path needs no memset launches)
__global__ __launch_bounds__(1024) void k_compact(const unsigned char *__restrict__ kinds, int nbatches,
                                                  int *__restrict__ worklist, MeshCounters *__restrict__ ctr,
                                                  unsigned long long *__restrict__ status,
                                                  long long shard_index, long long shard_count) {
    __shared__ int wave_sums[16];
    int base = 0;
    for (int start = 0; start < nbatches; start += 1024) {
        const int b = start + threadIdx.x;
        if (b < nbatches) status[b] = 0ull;
        const int f = (b < nbatches && kinds[b] != 0) ? 1 : 0;
        int tot;
        const int pos = block_exclusive_scan<1024>(f, wave_sums, tot);
        if (f) worklist[base + pos] = b;
        base += tot;
    }
    if (threadIdx.x == 0) {   // contiguous chunk of the work list for this shard (same formula as sdf_amd/dist.py)
        MeshCounters z = {};
        *ctr = z;
        ctr->nwork = base;
        ctr->work_begin = (int)(((long long)base * shard_index) / shard_count);
        ctr->work_end = (int)(((long long)base * (shard_index + 1)) / shard_count);
    }
}

// ---- marching cubes of a caller-supplied volume -------------------------------------------
__global__ __launch_bounds__(256) void k_mc_rows(const McTables *__restrict__ mc, const float *__restrict__ vol, int n0, int n1, int n2,
                                                 unsigned int *__restrict__ row_count) {
    const int c1 = n1 - 1, c2 = n2 - 1;
    const long long nrows = (long long)(n0 - 1) * c1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows) return;
    const int i0 = (int)(t / c1), i1 = (int)(t - (long long)i0 * c1);
    const int s0 = n1 * n2, s1 = n2;
    const float *row = vol + (long long)i0 * s0 + (long long)i1 * s1;
    unsigned prev = plane_bits(row, s0, s1);
    unsigned cnt = 0;
    for (int i2 = 0; i2 < c2; i2++) {
        const unsigned next = plane_bits(row + i2 + 1, s0, s1);
        const unsigned cfg = spread4(prev) | (spread4(next) << 1);
        if (mc->amb[cfg]) {
            double lv[8];
            int off;
            mc33_load_cell(row + i2, s0, s1, lv);
            cnt += (unsigned)mc33_cell(lv, mc->mc33, &off);
        } else {
            cnt += mc->ntri[cfg];
        }
        prev = next;
    }
    row_count[t] = cnt;
}

__global__ __launch_bounds__(1024) void k_scan_rows(const unsigned int *__restrict__ cnt, long long n,
                                                    unsigned long long *__restrict__ off, unsigned long long *total) {
    __shared__ int wave_sums[16];
    unsigned long long base = 0;
    for (long long start = 0; start < n; start += 1024) {
        const long long i = start + threadIdx.x;
        const int v = i < n ? (int)cnt[i] : 0;
        int tot;
        const int pos = block_exclusive_scan<1024>(v, wave_sums, tot);
        if (i < n) off[i] = base + (unsigned long long)pos;
        base += (unsigned long long)tot;
    }
    if (threadIdx.x == 0) *total = base;
}

__global__ __launch_bounds__(256) void k_mc_emit(const McTables *__restrict__ mc, const float *__restrict__ vol, int n0, int n1, int n2,
                                                 const unsigned long long *__restrict__ row_off, float *__restrict__ out,
                                                 unsigned long long cap) {
    const int c1 = n1 - 1, c2 = n2 - 1;
    const long long nrows = (long long)(n0 - 1) * c1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows) return;
    const int i0 = (int)(t / c1), i1 = (int)(t - (long long)i0 * c1);
    const int s0 = n1 * n2, s1 = n2;
    const float *row = vol + (long long)i0 * s0 + (long long)i1 * s1;
    unsigned long long k = row_off[t];
    unsigned prev = plane_bits(row, s0, s1);
    for (int i2 = 0; i2 < c2; i2++) {
        const unsigned next = plane_bits(row + i2 + 1, s0, s1);
        const unsigned cfg = spread4(prev) | (spread4(next) << 1);
        prev = next;
        if (mc->amb[cfg]) {
            double lv[8];
            int off;
            mc33_load_cell(row + i2, s0, s1, lv);
            const int nt = mc33_cell(lv, mc->mc33, &off);
            for (int j = 0; j < nt; j++, k++) {
                if (k >= cap) return;
                mc33_triangle(row + i2, s0, s1, i0, i1, i2, mc->mc33, j, out + k * 9ull);
            }
            continue;
        }
        const int nt = mc->ntri[cfg];
        for (int j = 0; j < nt; j++, k++) {
            if (k >= cap) return;
            for (int q = 0; q < 3; q++) mc_vertex(row + i2, s0, s1, i0, i1, i2, mc->tri[cfg][3 * j + q], out + k * 9ull + q * 3);
        }
    }
}

// ---- marching cubes of many caller-supplied tiles in one submission (FieldTile: sdf_plain.h) ----

__global__ __launch_bounds__(256) void k_cast_f32(const double *__restrict__ in, float *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];     // volume.astype(float32) inside skimage
}

__device__ __forceinline__ unsigned mc_row_count(const McTables *__restrict__ mc, const float *__restrict__ row, int s0, int s1, int c2) {
    unsigned prev = plane_bits(row, s0, s1);
    unsigned cnt = 0;
    for (int i2 = 0; i2 < c2; i2++) {
        const unsigned next = plane_bits(row + i2 + 1, s0, s1);
        const unsigned cfg = spread4(prev) | (spread4(next) << 1);
        if (mc->amb[cfg]) {
            double lv[8];
            int off;
            mc33_load_cell(row + i2, s0, s1, lv);
            cnt += (unsigned)mc33_cell(lv, mc->mc33, &off);
        } else {
            cnt += mc->ntri[cfg];
        }
        prev = next;
    }
    return cnt;
}

__global__ __launch_bounds__(256) void k_field_rows(const McTables *__restrict__ mc, const float *__restrict__ vol,
                                                    const FieldTile *__restrict__ tiles, unsigned int *__restrict__ row_count, int slots) {
    const FieldTile tl = tiles[blockIdx.y];
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);        // row slot 0 .. slots - 1 (1024 for tiles of <= 33^3 samples)
    const int c0 = tl.n0 - 1, c1 = tl.n1 - 1, c2 = tl.n2 - 1;
    unsigned cnt = 0;
    if (c0 > 0 && c1 > 0 && c2 > 0 && t < c0 * c1) {
        const int i0 = t / c1, i1 = t - i0 * c1;
        const int s0 = tl.n1 * tl.n2, s1 = tl.n2;
        cnt = mc_row_count(mc, vol + tl.vol_off + (long long)i0 * s0 + (long long)i1 * s1, s0, s1, c2);
    }
    row_count[(size_t)blockIdx.y * (size_t)slots + t] = cnt;
}

__global__ __launch_bounds__(256) void k_field_emit(const McTables *__restrict__ mc, const float *__restrict__ vol,
                                                    const FieldTile *__restrict__ tiles, const unsigned long long *__restrict__ row_off,
                                                    double *__restrict__ out, unsigned long long base, unsigned long long cap, int slots) {
    const FieldTile tl = tiles[blockIdx.y];
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int c0 = tl.n0 - 1, c1 = tl.n1 - 1, c2 = tl.n2 - 1;
    if (c0 <= 0 || c1 <= 0 || c2 <= 0 || t >= c0 * c1) return;
    const int i0 = t / c1, i1 = t - i0 * c1;
    const int s0 = tl.n1 * tl.n2, s1 = tl.n2;
    const float *row = vol + tl.vol_off + (long long)i0 * s0 + (long long)i1 * s1;
    unsigned long long k = base + row_off[(size_t)blockIdx.y * (size_t)slots + t];
    unsigned prev = plane_bits(row, s0, s1);
    for (int i2 = 0; i2 < c2; i2++) {
        const unsigned next = plane_bits(row + i2 + 1, s0, s1);
        const unsigned cfg = spread4(prev) | (spread4(next) << 1);
        prev = next;
        int nt = mc->ntri[cfg];
        const bool amb = mc->amb[cfg] != 0;
        if (amb) {
            double lv[8];
            int off;
            mc33_load_cell(row + i2, s0, s1, lv);
            nt = mc33_cell(lv, mc->mc33, &off);
        }
        for (int j = 0; j < nt; j++, k++) {
            if (k >= cap) return;
            float o[9];
            if (amb) mc33_triangle(row + i2, s0, s1, i0, i1, i2, mc->mc33, j, o);
            else for (int q = 0; q < 3; q++) mc_vertex(row + i2, s0, s1, i0, i1, i2, mc->tri[cfg][3 * j + q], o + q * 3);
            double *d = out + k * 9ull;
            for (int q = 0; q < 9; q += 3) {
                d[q] = (double)o[q] * tl.sc[0] + tl.of[0];
                d[q + 1] = (double)o[q + 1] * tl.sc[1] + tl.of[1];
                d[q + 2] = (double)o[q + 2] * tl.sc[2] + tl.of[2];
            }
        }
    }
}

// ---- two-pass meshing: the second and third kernel (the first is k_mesh with MeshArgs.twopass) ----
// k_scan_items: the work items' triangle counts -> their inclusive prefix in work-list (= reference) order, written as
// the same look-back words the one-pass kernel leaves (sdf_mesh_batch_offsets, k_pack_slab read them), and the total.
__global__ __launch_bounds__(1024) void k_scan_items(const ItemDesc *__restrict__ desc, MeshCounters *__restrict__ ctr,
                                                     unsigned long long *__restrict__ status, int *__restrict__ block_item,
                                                     unsigned long long n_blocks) {
    __shared__ int wave_sums[16];
    const int w_begin = ctr->work_begin, w_end = ctr->work_end;
    unsigned long long base = 0;
    for (int start = w_begin; start < w_end; start += 1024) {
        const int w = start + (int)threadIdx.x;
        const int v = w < w_end ? (int)desc[w].ntri : 0;
        int tot;
        const int pos = block_exclusive_scan<1024>(v, wave_sums, tot);
        if (w < w_end) {
            const unsigned long long first = base + (unsigned long long)pos, end = first + (unsigned long long)v;
            status[w] = MESH_FLAG_PFX | end;
            // block_item[b] = the work item that owns triangle 256 b (k_emit2 starts its search there instead of at the
            // ends of the list: fifteen dependent loads less per workgroup at weave 2^33)
            for (unsigned long long b = (first + 255ull) >> 8; (b << 8) < end && b < n_blocks; b++) block_item[b] = w;
        }
        base += (unsigned long long)tot;
    }
    if (threadIdx.x == 0) ctr->total = base;
}

// k_emit2: one lane per TRIANGLE of the whole soup (256 consecutive triangles per workgroup, whatever work items they
// belong to): find the triangle's work item in the prefix (binary search over the look-back words, L1-resident), fetch
// its entry and its cell's record, run the three edge interpolations on the record's 8 corner samples (mc_vertex /
// mc33_triangle -- the very functions the one-pass kernel runs on its LDS tile, with the strides of a 2 x 2 x 2 volume),
// pass the 9 local coordinates through LDS so that consecutive lanes store consecutive coordinates, and write
// `points * scale + offset` (reference sdf/core.py:58-60) of the triangle's own work item -- or, for the multi-GPU
// exchange, the 16-byte record of its local float32 form (store_tri16) -- straight to the final place.  The grid covers the soup's CAPACITY (the host does
// not know the count); workgroups beyond the total leave at once.
__global__ __launch_bounds__(256) void k_emit2(MeshArgs a) {
    __shared__ float tri[256 * 9];
    __shared__ unsigned recs[256 * 9];
    __shared__ int item_of[256];
    const unsigned long long total = a.ctr->total;
    const unsigned long long T0 = (unsigned long long)blockIdx.x * 256ull;
    if (T0 >= total) return;
    if (total > a.out_cap || (a.ctr->overflow & 1u)) {       // the soup or the arenas were too small: flagged, the call is repeated
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&a.ctr->overflow, 1u);
        return;
    }
    __shared__ int w_range[2];
    const int tid = threadIdx.x;
    const int nt = (int)min(256ull, total - T0);
    const int w_begin = a.ctr->work_begin, w_end = a.ctr->work_end;
    // the work item of a triangle T: the smallest w whose inclusive prefix exceeds T.  The workgroup's 256 consecutive
    // triangles span one or two items as a rule: two lanes search the whole prefix (for the first and the last
    // triangle), everybody else only between their answers
    if (tid < 2) {
        const unsigned long long T = tid == 0 ? T0 : T0 + (unsigned long long)(nt - 1);
        // (k_scan_items' index: the owner of this block's first triangle, and of the next block's -- which is the
        // last item this block can touch -- bracket the search)
        int lo = a.block_item ? a.block_item[blockIdx.x] : w_begin, hi = w_end - 1;
        if (a.block_item && T0 + 256ull < total) hi = a.block_item[blockIdx.x + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((a.status[mid] & MESH_VAL_MASK) > T) hi = mid; else lo = mid + 1;
        }
        w_range[tid] = lo;
    }
    __syncthreads();
    if (tid < nt) {
        const unsigned long long T = T0 + (unsigned long long)tid;
        int lo = w_range[0], hi = w_range[1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((a.status[mid] & MESH_VAL_MASK) > T) hi = mid; else lo = mid + 1;
        }
        const int w = lo;
        const ItemDesc *d = a.desc + w;
        const unsigned ntri = d->ntri;
        const unsigned long long t = T - ((a.status[w] & MESH_VAL_MASK) - ntri);
        const unsigned e = a.tlist[d->list_off + t];
        const unsigned *src = a.cells + (d->cell_off + (unsigned long long)(e >> 4)) * 9ull;
        unsigned *rec = recs + tid * 9;
        for (int q = 0; q < 9; q++) rec[q] = src[q];
        const unsigned info = rec[0];
        const int j = (int)(e & 15u), cfg = (int)((info >> 4) & 255u), cell = (int)(info >> 13);
        const int i0 = cell >> 10, i1 = (cell >> 5) & 31, i2 = cell & 31;
        const float *corner = reinterpret_cast<const float *>(rec + 1);          // 2 x 2 x 2 samples: strides 4, 2, 1
        float *o = tri + tid * 9;
        if (info & 4096u) {
            float tmp[9];
            mc33_triangle(corner, 4, 2, i0, i1, i2, a.mc->mc33, j, tmp);
            for (int q = 0; q < 9; q++) o[q] = tmp[q];
        } else {
            const signed char *tt = &a.mc->tri[0][0] + cfg * 16 + 3 * j;
            float v[3];
            for (int q = 0; q < 3; q++) { mc_vertex(corner, 4, 2, i0, i1, i2, tt[q], v); o[3 * q] = v[0]; o[3 * q + 1] = v[1]; o[3 * q + 2] = v[2]; }
        }
        item_of[tid] = w;
    }
    __syncthreads();
    const int n9 = nt * 9;
    const unsigned long long at = T0 * 9ull;                  // (a multiple of 9: coordinate e of the workgroup belongs to axis e % 3)
    if (a.compact) {   // the exchange's 16-byte record (sdf_slab.h)
        if (tid < nt) store_tri16(a, T0 + (unsigned long long)tid, tri + tid * 9);
    } else {
        double *dst = a.out + at;
        for (int e = tid; e < n9; e += 256) {
            const double *xf = a.desc[item_of[e / 9]].xf;
            const int ax = e % 3;
            dst[e] = (double)tri[e] * xf[3 + ax] + xf[ax];
        }
    }
}

// ---- STL records (reference sdf/stl.py:4-24): float32 vertices, normal = normalised cross ----
__global__ __launch_bounds__(256) void k_stl(const double *__restrict__ pts, long long ntri, unsigned short *__restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntri) return;
    float p[9];
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = (float)pts[t * 9 + i];
    const float ax = p[3] - p[0], ay = p[4] - p[1], az = p[5] - p[2];
    const float bx = p[6] - p[0], by = p[7] - p[1], bz = p[8] - p[2];
    // np.cross / np.linalg.norm in float32: separate, individually rounded products, sums and the
    // quotient (the translation unit is built with -ffp-contract=off; sqrtf and '/' are the
    // correctly rounded forms, the __f*_rn intrinsics map to native approximations here)
    float nx = ay * bz - az * by;
    float ny = az * bx - ax * bz;
    float nz = ax * by - ay * bx;
    const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    nx = nx / len; ny = ny / len; nz = nz / len;
    float rec[12] = {nx, ny, nz, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]};
    unsigned short *o = out + t * 25;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const unsigned u = __float_as_uint(rec[i]);
        o[2 * i] = (unsigned short)(u & 0xFFFFu);
        o[2 * i + 1] = (unsigned short)(u >> 16);
    }
    o[24] = 0;
}

// ---- the multi-GPU exchange unit (layout: sdf_slab.h) ----
// header + the shard's look-back words (inclusive triangle prefix per work item) into the slab, behind k_mesh
__global__ __launch_bounds__(256) void k_pack_slab(const MeshCounters *__restrict__ ctr, const unsigned long long *__restrict__ status,
                                                   unsigned char *__restrict__ slab, long long cap_items, long long cap_tris) {
    const SlabLayout L(cap_items, cap_tris);
    const long long n_items = (long long)ctr->work_end - ctr->work_begin;
    unsigned long long *prefix = reinterpret_cast<unsigned long long *>(slab + L.prefix_off);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_items && i < cap_items; i += (long long)gridDim.x * blockDim.x)
        prefix[i] = status[ctr->work_begin + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SlabHeader h = {};
        h.n_tris = (long long)ctr->total; h.n_items = n_items;
        h.n_raw = (long long)ctr->n_raw;
        const bool raw_over = h.n_raw > L.raw_cap;
        h.need_tris = raw_over ? (h.n_raw > h.n_tris / SLAB_RAW_DIV ? h.n_raw * SLAB_RAW_DIV : h.n_tris) : h.n_tris;
        if (h.need_tris < h.n_tris) h.need_tris = h.n_tris;
        h.overflow = (long long)ctr->overflow | (n_items > cap_items ? 4 : 0) | ((long long)ctr->total > cap_tris ? 1 : 0) | (raw_over ? 8 : 0);
        h.n_empty = ctr->n_empty; h.n_nonempty = ctr->n_nonempty; h.n_eval = (long long)ctr->n_eval;
        h.n_ambiguous = (long long)ctr->n_ambiguous; h.n_sampled = (long long)ctr->n_sampled; h.n_pruned = (long long)ctr->n_pruned;
        h.n_work_total = ctr->nwork;
        *reinterpret_cast<SlabHeader *>(slab) = h;
    }
}

// gathered slabs (in final order) -> the ordered float64 soup: `points * scale + offset` (reference sdf/core.py:58-60)
// per work item.  ONE LANE PER OUTPUT TRIANGLE, 256 consecutive triangles of the soup per workgroup whatever slabs and
// work items they come from: a lane finds its slab in the running totals of the gathered headers (LDS) and its work item
// in the slab's prefix words (the first and the last lane search the whole prefix, the others only between their answers),
// brings the item's transform and its nine local float32 into LDS, and then consecutive lanes write consecutive
// coordinates.  (A workgroup per (slab, work item) -- round 2 -- spent its time on the short items' start-up: 106 MB in,
// 212 MB out took 0.107 ms = 3.0 TB/s.)  The grid covers the soup's capacity; workgroups beyond the total leave at once.
__global__ __launch_bounds__(256) void k_expand(SlabPtrs slabs, int n_slabs, long long cap_items, long long cap_tris,
                                                double *__restrict__ out, unsigned long long cap_out) {
    __shared__ unsigned long long sbase[65];
    __shared__ float tri[256 * 9];
    __shared__ double xfs[256 * 6];
    __shared__ int range_[4];
    __shared__ const unsigned char *sp[64];      // (the slab pointers, indexed per lane below: out of LDS, not out of the kernel argument)
    const SlabLayout L(cap_items, cap_tris);
    const int tid = threadIdx.x;
    if (tid < 64) {
        long long n = 0;
        sp[tid] = slabs.p[tid < n_slabs ? tid : 0];
        if (tid < n_slabs) { n = reinterpret_cast<const SlabHeader *>(slabs.p[tid])->n_tris; n = n < 0 ? 0 : (n > cap_tris ? cap_tris : n); }
        // exclusive running totals over the (<= 64) slabs: one wave, shuffles
        unsigned long long v = (unsigned long long)n, incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned lo = __shfl_up((unsigned)incl, d, 64), hi = __shfl_up((unsigned)(incl >> 32), d, 64);
            if (tid >= d) incl += ((unsigned long long)hi << 32) | lo;
        }
        sbase[tid + 1] = incl;
        if (tid == 0) sbase[0] = 0ull;
    }
    __syncthreads();
    unsigned long long total = sbase[n_slabs];
    if (total > cap_out) total = cap_out;
    const unsigned long long T0 = (unsigned long long)blockIdx.x * 256ull;
    if (T0 >= total) return;
    const int nt = (int)(total - T0 < 256ull ? total - T0 : 256ull);
    // slab and work item of a soup triangle T: the slab by a walk over the totals, the item = the smallest i whose
    // inclusive prefix exceeds the triangle's index within the slab
    auto slab_of = [&](unsigned long long T) { int s = 0; while (s + 1 < n_slabs && sbase[s + 1] <= T) s++; return s; };
    auto item_of = [&](int s, unsigned long long t, int lo, int hi) {
        const unsigned long long *prefix = reinterpret_cast<const unsigned long long *>(sp[s] + L.prefix_off);
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((prefix[mid] & MESH_VAL_MASK) > t) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    auto items_in = [&](int s) {
        const long long n = reinterpret_cast<const SlabHeader *>(sp[s])->n_items;
        return (int)(n < 1 ? 1 : (n > cap_items ? cap_items : n));
    };
    if (tid < 2) {
        const unsigned long long T = tid == 0 ? T0 : T0 + (unsigned long long)(nt - 1);
        const int s = slab_of(T);
        range_[tid] = s;
        range_[2 + tid] = item_of(s, T - sbase[s], 0, items_in(s) - 1);
    }
    __syncthreads();
    if (tid < nt) {
        const unsigned long long T = T0 + (unsigned long long)tid;
        int s = range_[0], lo = range_[2], hi = range_[3];
        if (range_[1] != s) { s = slab_of(T); lo = s == range_[0] ? range_[2] : 0; hi = s == range_[1] ? range_[3] : items_in(s) - 1; }   // (the workgroup straddles slabs)
        const unsigned long long t = T - sbase[s];
        const int i = item_of(s, t, lo, hi);
        const double *xf = reinterpret_cast<const double *>(sp[s] + L.xf_off) + (size_t)i * 6;
        const Tri16 rec = reinterpret_cast<const Tri16 *>(sp[s] + L.tris_off)[t];
        SDF_UNROLL for (int q = 0; q < 6; q++) xfs[tid * 6 + q] = xf[q];
        float o9[9];
        if (rec.code & TRI16_RAW) {   // (rare: a vertex inside a cell) the nine floats wait in the slab's raw area
            const unsigned long long ri = (unsigned long long)__float_as_uint(rec.f[0]);
            const float *src = reinterpret_cast<const float *>(sp[s] + L.raw_off) + (ri < (unsigned long long)L.raw_cap ? ri : 0ull) * 9ull;
            SDF_UNROLL for (int q = 0; q < 9; q++) o9[q] = src[q];
        } else {
            slab_decode16(rec, o9);
        }
        SDF_UNROLL for (int q = 0; q < 9; q++) tri[tid * 9 + q] = o9[q];
    }
    __syncthreads();
    double *dst = out + T0 * 9ull;
    for (int e = tid; e < nt * 9; e += 256) {
        const int tr = e / 9, ax = (e - 9 * tr) % 3;
        dst[e] = (double)tri[e] * xfs[tr * 6 + 3 + ax] + xfs[tr * 6 + ax];
    }
}

// gathered slabs' headers into one contiguous block (one small copy to the host instead of one per slab)
__global__ __launch_bounds__(64) void k_collect_headers(SlabPtrs slabs, int n_slabs, long long *__restrict__ out) {
    const int s = blockIdx.x, i = threadIdx.x;
    if (s < n_slabs && i < 16) out[s * 16 + i] = reinterpret_cast<const long long *>(slabs.p[s])[i];
}

int sdf_launch_pack_slab(unsigned blocks, hipStream_t stream, const MeshCounters *ctr, const unsigned long long *status,
                         unsigned char *slab, long long cap_items, long long cap_tris) {
    hipLaunchKernelGGL(k_pack_slab, dim3(blocks), dim3(256), 0, stream, ctr, status, slab, cap_items, cap_tris);
    return (int)hipGetLastError();
}

int sdf_launch_expand(hipStream_t stream, const SlabPtrs &slabs, int n_slabs, long long cap_items, long long cap_tris, double *out,
                      unsigned long long cap_out) {
    const unsigned long long blocks = (cap_out + 255ull) / 256ull;
    if (blocks == 0) return 0;
    if (blocks > 0x7fffffffull) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_expand, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, n_slabs, cap_items, cap_tris, out, cap_out);
    return (int)hipGetLastError();
}

int sdf_launch_collect_headers(hipStream_t stream, const SlabPtrs &slabs, int n_slabs, long long *out) {
    hipLaunchKernelGGL(k_collect_headers, dim3((unsigned)n_slabs), dim3(64), 0, stream, slabs, n_slabs, out);
    return (int)hipGetLastError();
}

// ---- launchers ----
void launch_k_compact(dim3 grid, dim3 block, hipStream_t stream, const unsigned char *kinds, int nbatches, int *worklist, MeshCounters *ctr,
                      unsigned long long *status, long long shard_index, long long shard_count) {
    hipLaunchKernelGGL(k_compact, grid, block, 0, stream, kinds, nbatches, worklist, ctr, status, shard_index, shard_count);
}
void launch_k_mc_rows(dim3 grid, dim3 block, hipStream_t stream, const McTables *mc, const float *vol, int n0, int n1, int n2, unsigned int *row_count) {
    hipLaunchKernelGGL(k_mc_rows, grid, block, 0, stream, mc, vol, n0, n1, n2, row_count);
}
void launch_k_scan_rows(dim3 grid, dim3 block, hipStream_t stream, const unsigned int *cnt, long long n, unsigned long long *off, unsigned long long *total) {
    hipLaunchKernelGGL(k_scan_rows, grid, block, 0, stream, cnt, n, off, total);
}
void launch_k_mc_emit(dim3 grid, dim3 block, hipStream_t stream, const McTables *mc, const float *vol, int n0, int n1, int n2,
                      const unsigned long long *row_off, float *out, unsigned long long cap) {
    hipLaunchKernelGGL(k_mc_emit, grid, block, 0, stream, mc, vol, n0, n1, n2, row_off, out, cap);
}
void launch_k_cast_f32(dim3 grid, dim3 block, hipStream_t stream, const double *in, float *out, long long n) {
    hipLaunchKernelGGL(k_cast_f32, grid, block, 0, stream, in, out, n);
}
void launch_k_field_rows(dim3 grid, dim3 block, hipStream_t stream, const McTables *mc, const float *vol, const FieldTile *tiles, unsigned int *row_count,
                         int slots) {
    hipLaunchKernelGGL(k_field_rows, grid, block, 0, stream, mc, vol, tiles, row_count, slots);
}
void launch_k_field_emit(dim3 grid, dim3 block, hipStream_t stream, const McTables *mc, const float *vol, const FieldTile *tiles,
                         const unsigned long long *row_off, double *out, unsigned long long base, unsigned long long cap, int slots) {
    hipLaunchKernelGGL(k_field_emit, grid, block, 0, stream, mc, vol, tiles, row_off, out, base, cap, slots);
}
void launch_k_scan_items(dim3 grid, dim3 block, hipStream_t stream, const ItemDesc *desc, MeshCounters *ctr, unsigned long long *status,
                         int *block_item, unsigned long long n_blocks) {
    hipLaunchKernelGGL(k_scan_items, grid, block, 0, stream, desc, ctr, status, block_item, n_blocks);
}
void launch_k_emit2(dim3 grid, dim3 block, hipStream_t stream, const MeshArgs &a) {
    hipLaunchKernelGGL(k_emit2, grid, block, 0, stream, a);
}
void launch_k_stl(dim3 grid, dim3 block, hipStream_t stream, const double *pts, long long ntri, unsigned short *out) {
    hipLaunchKernelGGL(k_stl, grid, block, 0, stream, pts, ntri, out);
}

// ---- split meshing, second kernel (sdf_split.h): marching cubes + ordered emission of the tiles k_sample left in the arena ----
// Workgroups of 256 threads, several per compute unit; each takes work items IN ORDER from `march_counter` (what makes the
// look-back safe: every predecessor of an item is held by a running workgroup) and does for one tile what k_mesh's phases 2 - 4
// do (sdf_device.h), on the sign bits and column words in LDS and the samples in the arena:
//   rows    a thread owns FOUR CONSECUTIVE (i0, i1) rows of cells: their surface-cell masks from the sign strings, one block
//           scan -> every surface cell's running index in soup order (i0, i1, i2 ascending: skimage's emission order)
//   cells   chunks of up to 2048 surface cells: the rows' threads scatter (row, column) into a table; a thread then owns a RUN
//           of consecutive cells of the chunk (ceil(cells / 256) each), looks up their triangle counts (ambiguous
//           configurations: Lewiner's tests on the 8 corner samples) -- one block scan -> every cell's first triangle -- and
//           writes the per-triangle list (cell | configuration | triangle of the cell) over the table
//   place   the tile's count is published, wave 0 walks back over its predecessors' words (ordered_base, blocking: the other
//           workgroups of the compute unit fill the wait)
//   emit    a lane per triangle: three edge interpolations on samples fetched from the arena (TileView), transposed through LDS
//           so that consecutive lanes store consecutive coordinates, `points * scale + offset` in float64 (reference
//           sdf/core.py:58-60)
// The common tile -- at most 2048 surface cells and MARCH_LCAP triangles -- is classified ONCE; any other one classifies every
// chunk for the count and again per window of MARCH_LCAP triangles for the emission (rows and cells are recomputed rather
// than kept in registers across the phases: nothing but the loop state lives through the emission).

// triangles of an ambiguous cell: Lewiner's tests on its 8 corner samples pick the tiling (rare; out of line: its registers are
// the call's, not the kernel's)
static __device__ __attribute__((noinline)) int march_amb_count(const float *smp, const unsigned *colinfo, int lyz, int lz, bool sparse,
                                                                int i0, int i1, int i2, const signed char *tab) {
    const TileView vw{smp, colinfo, lyz, lz, sparse};
    float c8[8];
    double lv[8];
    int off;
    vw.cell(i0, i1, i2, c8);
    mc33_load_cell(c8, 4, 2, lv);
    return mc33_cell(lv, tab, &off);
}
// triangle j of an ambiguous cell, its nine local coordinates straight into the wave's transposition area
static __device__ __attribute__((noinline)) void march_amb_triangle(const float *smp, const unsigned *colinfo, int lyz, int lz, bool sparse,
                                                                    int i0, int i1, int i2, const signed char *tab, int j, float *dst) {
    const TileView vw{smp, colinfo, lyz, lz, sparse};
    float c8[8], oa[9];
    vw.cell(i0, i1, i2, c8);
    mc33_triangle(c8, 4, 2, i0, i1, i2, tab, j, oa);
    for (int q = 0; q < 9; q++) dst[q] = oa[q];
}

// MODE 0: count, place and emit in one go (the look-back WAITS for predecessors that are still counting: a convoy behind the
// slowest tile in flight); 1: count only -- sign bits into the arena, the count published; 2: emit only, behind a MODE 1 launch:
// every count is known, nothing waits, the cells are classified again (cheaper than keeping them)
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k_march(MeshArgs a) {
    constexpr int RPT = 1024 / BLOCK, CPT = MARCH_CELLS / BLOCK;
    __shared__ int wave_sums[16];
    __shared__ int bcast[8];
    __shared__ unsigned char ntri_lds[256];                                 // ntri | ambiguous << 7
    __shared__ unsigned short tri_lds[256 * 5];                             // e0 | e1 << 4 | e2 << 8 of triangle j of configuration cfg at [5 cfg + j]
    __shared__ unsigned colinfo[292];                                       // k_cull's column words (TileView, sparse form)
    __shared__ unsigned long long bits[((33 * 33 * 33 + 63) >> 6) + 2];
    __shared__ double xf_lds[6];                                            // offset[3], scale[3] of the tile in hand
    __shared__ float stage[BLOCK / 64][64 * 9];                             // per wave: 64 triangles, transposed on their way out
    __shared__ __attribute__((aligned(16))) unsigned lst[MARCH_LCAP];       // the chunk's cell table, then the triangle list
    static_assert(MARCH_LCAP >= MARCH_CELLS, "the cell table lives in the list");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GridDesc g = a.g;
    // (what the phases below need of the argument block, as values of their own: a by-value kernel argument whose address a
    // lambda captures is copied to private memory as a whole)
    const signed char *const mc33 = a.mc->mc33;
    double *const soup = a.out;
    const bool compact = a.compact != 0;
    const Tri16Sink sink{a.out, a.raw, a.raw_cap, &a.ctr->n_raw};
    if (tid < 256) ntri_lds[tid] = (unsigned char)(a.mc->ntri[tid] | (a.mc->amb[tid] << 7));
    for (int i = tid; i < 256 * 5; i += BLOCK) {
        const int cfg = i / 5, j = i - 5 * cfg;
        const signed char *t3 = &a.mc->tri[cfg][3 * j];
        tri_lds[i] = (unsigned short)(((unsigned)t3[0] & 15u) | (((unsigned)t3[1] & 15u) << 4) | (((unsigned)t3[2] & 15u) << 8));
    }
    const int work_begin = a.ctr->work_begin, work_end = a.ctr->work_end;
    const bool have_arena = a.ctr->tile_cursor <= a.tiles_cap256;           // (else: flagged by k_sample, nothing was sampled, the call is repeated)
    bool first_item = true;
    for (;;) {
        // (where nothing ever waits -- MODE 1 / 2 -- a workgroup's FIRST item is its own index: a thousand workgroups drawing from one
        // counter in the same microsecond queue up at its address; MODE 0 must hand every item to a workgroup that is running)
        if (tid == 0) {
            int idx;
            if (MODE != 0 && first_item) idx = (int)blockIdx.x;
            else idx = (MODE != 0 ? (int)gridDim.x : 0) + (int)atomicAdd(MODE == 2 ? &a.ctr->emit_counter : &a.ctr->march_counter, 1u);
            bcast[0] = work_begin + idx;
        }
        first_item = false;
        __syncthreads();
        const int w = __builtin_amdgcn_readfirstlane(bcast[0]);
        if (w >= work_end) break;
        // wave 0 asks for the predecessors' status words now: the answer arrives while the cells are counted
        unsigned long long pre = 0;
        if (tid < 64 && MODE != 1) pre = lookback_prefetch(a.status, w, work_begin);
        const unsigned *rec = reinterpret_cast<const unsigned *>(a.cull + (size_t)w * CULL_RECORD);
        const int b = __builtin_amdgcn_readfirstlane(a.worklist[w]);
        const unsigned n0 = (unsigned)__builtin_amdgcn_readfirstlane((int)rec[0]) & 0xFFFFu;
        const unsigned long long off256 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)rec[1]);
        const bool culled = n0 != 0xFFFFu;
        const int ntl = culled ? (int)((n0 + 7u) >> 3) : -1;
        int ox, oy, oz, lx, ly, lz;
        batch_origin(g, b, ox, oy, oz, lx, ly, lz);
        const int lyz = ly * lz, nvox = lx * lyz, nwords = (nvox + 63) >> 6;
        const bool have_tile = have_arena;
        const float *tile = reinterpret_cast<const float *>(a.tiles + off256 * 256ull);
        // ---- the tile's sign-bit volume (value > 0), one word per 64 consecutive samples: the marching phases classify cells from
        // these bits.  Culled tile: the samples of DECIDED sub-groups get their bits straight from the sub-group states (a row of
        // lz samples along z = `pos | pos << 1` of its 16 two-bit states, "positive" = 01: k_mesh's sign fill), the evaluated
        // samples -- 64 per listed task, cull_sample's order -- OR theirs in.  Dense tile: a ballot per word. ----
        unsigned long long *tile_bits = reinterpret_cast<unsigned long long *>(a.tiles + off256 * 256ull + tile_data_bytes(ntl, nvox));
        if (MODE == 2) {   // (the counting launch left them in the arena)
            if (have_tile) {
                for (int i = tid; i < nwords + 2; i += BLOCK) bits[i] = tile_bits[i];
                if (culled) for (int i = tid; i < 289; i += BLOCK) colinfo[i] = rec[CULL_COLINFO / 4 + i];
            }
        } else if (have_tile && culled) {
            for (int i = tid; i < nwords + 2; i += BLOCK) bits[i] = 0ull;
            for (int i = tid; i < 289; i += BLOCK) colinfo[i] = rec[CULL_COLINFO / 4 + i];
            __syncthreads();
            const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
            const int hlast = (c2 - 1) >> 1;
            const unsigned *sstate = rec + CULL_SSTATE / 4;
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
            const unsigned short *units = reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(rec) + CULL_ULIST);
            for (int s0 = tid; s0 < 64 * ntl; s0 += BLOCK) {
                int ix, iy, iz;
                if (cull_sample(units, s0 >> 6, s0 & 63, lx, ly, lz, ix, iy, iz) && tile[s0] > 0.0f) {
                    const int i = ix * lyz + iy * lz + iz;
                    atomicOr(&bits[i >> 6], 1ull << (i & 63));
                }
            }
        } else if (have_tile) {
            for (int i0 = tid; i0 < 64 * nwords; i0 += BLOCK) {               // (whole waves: a wave owns a word)
                const unsigned long long mword = __ballot(i0 < nvox && tile[min(i0, nvox - 1)] > 0.0f);
                if (lane == 0) bits[i0 >> 6] = mword;
            }
            if (tid < 2) bits[nwords + tid] = 0ull;                          // the row extraction reads one word ahead
        }
        if (MODE == 1 && have_tile) {
            __syncthreads();
            for (int i = tid; i < nwords + 2; i += BLOCK) tile_bits[i] = bits[i];
        }
        if (tid == 0 && have_tile && MODE != 2) {                             // (statistics k_mesh keeps while it samples)
            if (a.tape_stride) {
                const unsigned long long *wc = reinterpret_cast<const unsigned long long *>(a.code_for_stats) + (size_t)b * (size_t)a.tape_stride;
                atomicAdd(&a.ctr->n_pruned, (unsigned long long)a.n_instr - wc[a.tape_stride - 1]);
            }
            atomicAdd(&a.ctr->n_sampled, culled ? (unsigned long long)ntl * 64ull : (unsigned long long)nvox);
        }
        // points * scale + offset (reference sdf/core.py:58-60): offset = the batch's first sample, scale = its first step, per axis
        if (tid < 3) {
            const double *ax = tid == 0 ? g.X + ox : (tid == 1 ? g.Y + oy : g.Z + oz);
            const double o_ = ax[0];
            xf_lds[tid] = o_;
            xf_lds[3 + tid] = ax[(tid == 0 ? lx : (tid == 1 ? ly : lz)) > 1 ? 1 : 0] - o_;
        }
        __syncthreads();
        const int c0 = lx - 1, c1 = ly - 1, c2 = lz - 1;
        const int nrows = (have_tile && c0 > 0 && c1 > 0 && c2 > 0) ? c0 * c1 : 0;
        const float inv_c1 = 1.0f / (float)max(c1, 1);
        auto row_signs = [&](int i0, int i1, unsigned long long *rb) __attribute__((always_inline)) -> unsigned {
#pragma unroll
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
        // ---- rows: the surface-cell masks of this thread's four consecutive rows, the cells' running index; returns the
        // tile's surface cells ----
        auto rows = [&](unsigned *row_mask, int *row_cell0) __attribute__((always_inline)) -> int {
            int mine = 0;
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int r = RPT * tid + k;
                unsigned mask = 0;
                if (r < nrows) {
                    unsigned long long rb[4];
                    const int i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                    mask = row_signs(i0, i1, rb);
                }
                row_mask[k] = mask;
                row_cell0[k] = mine;
                mine += __popc(mask);
            }
            int n;
            const int excl = block_exclusive_scan<BLOCK>(mine, wave_sums, n);
#pragma unroll
            for (int k = 0; k < RPT; k++) row_cell0[k] += excl;
            return __builtin_amdgcn_readfirstlane(n);
        };
        // ---- cells: chunk `ch` of the surface cells -> this thread's run of them: entry | triangles << 28 each (a cell has at
        // most 12 triangles, the entry ends at bit 27); returns the thread's triangle sum ----
        int my_amb = 0;
        auto classify = [&](int ch, int ncells, const unsigned *row_mask, const int *row_cell0, unsigned *cinfo, bool count_amb) __attribute__((always_inline)) -> int {
            const int cbase = ch * MARCH_CELLS, ccount = min(MARCH_CELLS, ncells - cbase);
            const int cpt = (ccount + BLOCK - 1) / BLOCK;                    // cells per thread (uniform, <= CPT)
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int r = RPT * tid + k;
                unsigned m = row_mask[k];
                int pos = row_cell0[k] - cbase;
                while (m) {
                    const int i2 = __ffs((int)m) - 1;
                    m &= m - 1u;
                    if (pos >= 0 && pos < MARCH_CELLS) lst[pos] = (unsigned)r | ((unsigned)i2 << 10);
                    pos++;
                }
            }
            __syncthreads();
            int sum = 0;
#pragma unroll
            for (int q = 0; q < CPT; q++) {
                const int s = cpt * tid + q;
                int n = 0;
                unsigned info = 0;
                if (q < cpt && s < ccount) {
                    const unsigned ce = lst[s];
                    const int r = (int)(ce & 1023u), i2 = (int)((ce >> 10) & 31u), i0 = fast_div(r, inv_c1), i1 = r - i0 * c1;
                    unsigned long long rb[4];
                    row_signs(i0, i1, rb);
                    const unsigned cfg = cell_config(rb, i2);
                    const unsigned e = ntri_lds[cfg];
                    if (e & 128u) {
                        n = march_amb_count(tile, colinfo, lyz, lz, culled, i0, i1, i2, mc33);
                        if (count_amb) my_amb++;
                    } else n = (int)(e & 7u);
                    // entry: cell (15 bits) | ambiguous (1) | configuration (8) | triangle in cell (4)
                    info = ((unsigned)((i0 << 10) | (i1 << 5) | i2) << 13) | ((e & 128u) << 5) | (cfg << 4);
                }
                cinfo[q] = info | ((unsigned)n << 28);
                sum += n;
            }
            return sum;
        };
        // the triangles [lo, lo + MARCH_LCAP) of a chunk into the list (over the table: every thread has read its entries
        // before the scan's barriers); `excl` = this thread's first triangle in the chunk
        auto write_list = [&](const unsigned *cinfo, int excl, int lo) __attribute__((always_inline)) {
            int pos = excl - lo;
#pragma unroll
            for (int q = 0; q < CPT; q++) {
                const int n = (int)(cinfo[q] >> 28);
                const unsigned ent = cinfo[q] & 0x0FFFFFFFu;
                for (int j = 0; j < n; j++, pos++)
                    if (pos >= 0 && pos < MARCH_LCAP) lst[pos] = ent | (unsigned)j;
            }
        };
        int total = 0, ncells, nchunks;
        bool listed = false;            // the common tile: its whole list is in LDS after the count
        {
            unsigned row_mask[RPT], cinfo[CPT];
            int row_cell0[RPT];
            ncells = rows(row_mask, row_cell0);
            nchunks = (ncells + MARCH_CELLS - 1) / MARCH_CELLS;
            for (int ch = 0; ch < nchunks; ch++) {                          // (uniform)
                const int sum = classify(ch, ncells, row_mask, row_cell0, cinfo, MODE != 2);
                int tot;
                const int excl = block_exclusive_scan<BLOCK>(sum, wave_sums, tot);
                total += tot;
                if (MODE != 1 && nchunks == 1 && tot <= MARCH_LCAP) { write_list(cinfo, excl, 0); listed = true; }   // (made visible by the allocation's barrier)
            }
        }
        total = __builtin_amdgcn_readfirstlane(total);
        if (MODE != 2) {
            // ---- the tile's count is public from here on; bookkeeping that needs no position ----
            if (tid < 64) publish_count(a.status, w, work_begin, (unsigned long long)total);
            if (a.compact && tid == 0 && w - work_begin < a.xf_cap) {   // the batch's transform travels with the compact soup
                double *xf = a.xf + (size_t)(w - work_begin) * 6;
                for (int q = 0; q < 6; q++) xf[q] = xf_lds[q];
            }
            if (tid == 0) {
                atomicAdd(total ? &a.ctr->n_nonempty : &a.ctr->n_empty, 1u);
                atomicAdd(&a.ctr->n_eval, (unsigned long long)nvox);
                a.kinds[b] = total ? 2 : 1;
            }
            if (my_amb) atomicAdd(&a.ctr->n_ambiguous, (unsigned long long)my_amb);
        }
        if (MODE == 1) continue;                                              // (the loop's top barrier orders this item's LDS reads before the next item's writes)
        // ---- place: the exclusive prefix of the triangle counts over the work list (wave 0; blocking) ----
        if (tid < 64) {
            const unsigned long long excl = ordered_base(a.status, w, work_begin, (unsigned long long)total, MESH_SPIN_FOREVER, pre);
            if (tid == 0) {
                if (excl == ~0ull) atomicOr(&a.ctr->overflow, 2u);           // look-back timed out (never expected)
                else if (excl + (unsigned long long)total > a.out_cap) atomicOr(&a.ctr->overflow, 1u);
                if (w == work_end - 1 && excl != ~0ull) a.ctr->total = excl + (unsigned long long)total;
                reinterpret_cast<unsigned long long *>(bcast + 2)[0] = excl;
            }
        }
        __syncthreads();
        const unsigned long long base = uni64(reinterpret_cast<unsigned long long *>(bcast + 2)[0]);
        const bool fits = base != ~0ull && base + (unsigned long long)total <= a.out_cap;
        // ---- emit: the triangles [0, ecn) of the list in LDS go to soup positions pos0 .. ----
        auto emit = [&](int ecn, unsigned long long pos0) __attribute__((always_inline)) {
            double *dst0 = soup + pos0 * 9ull;
            float *stg = stage[wave];
            const TileView vw{tile, colinfo, lyz, lz, culled};
            for (int t0 = wave * 64; t0 < ecn; t0 += BLOCK) {               // (whole waves: the transposition below is wave-wide)
                const int t = t0 + lane;
                const bool live = t < ecn;
                const unsigned e = lst[live ? t : ecn - 1];
                const int j = (int)(e & 15u), cfg = (int)((e >> 4) & 255u), cell = (int)(e >> 13);
                const int i0 = cell >> 10, i1 = (cell >> 5) & 31, i2 = cell & 31;
                if (compact) {   // (uniform) the exchange's 16-byte record, straight from the registers
                    float o[9];
                    if (e & 4096u) {
                        march_amb_triangle(tile, colinfo, lyz, lz, culled, i0, i1, i2, mc33, j, stg + lane * 9);
#pragma unroll
                        for (int q = 0; q < 9; q++) o[q] = stg[lane * 9 + q];
                    } else {
                        const unsigned tt3 = tri_lds[5 * cfg + min(j, 4)];
                        mc_vertex_view(vw, i0, i1, i2, (int)(tt3 & 15u), o);
                        mc_vertex_view(vw, i0, i1, i2, (int)((tt3 >> 4) & 15u), o + 3);
                        mc_vertex_view(vw, i0, i1, i2, (int)(tt3 >> 8), o + 6);
                    }
                    if (live) store_tri16(sink, pos0 + (unsigned long long)t, o);
                    continue;
                }
                // through LDS: lane l leaves triangle t0 + l (9 floats) at stg[9 l ..]; afterwards lane l stores coordinates
                // 64 k + l, k = 0 .. 8, of the wave's 576: consecutive lanes, consecutive addresses.  Coordinate c belongs to axis
                // c % 3 and 64 % 3 == 1: the axis of a lane's k-th coordinate is (l + k) % 3.
                if (e & 4096u) march_amb_triangle(tile, colinfo, lyz, lz, culled, i0, i1, i2, mc33, j, stg + lane * 9);
                else {
                    const unsigned tt3 = tri_lds[5 * cfg + min(j, 4)];
#pragma unroll
                    for (int v = 0; v < 3; v++) {
                        float o[3];
                        mc_vertex_view(vw, i0, i1, i2, (int)((tt3 >> (4 * v)) & 15u), o);
                        stg[lane * 9 + 3 * v] = o[0]; stg[lane * 9 + 3 * v + 1] = o[1]; stg[lane * 9 + 3 * v + 2] = o[2];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int nval = min(64, ecn - t0) * 9;
                double *dstw = dst0 + (size_t)t0 * 9;
                int ax = lane % 3;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int c = 64 * k + lane;
                    if (c < nval) dstw[c] = (double)stg[c] * xf_lds[3 + ax] + xf_lds[ax];
                    ax = ax == 2 ? 0 : ax + 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();   // (the area is rewritten by the wave's next 64 triangles)
            }
        };
        if (fits && total > 0) {                                             // (uniform)
            if (listed) {
                emit(total, base);
                __syncthreads();   // (the list is rewritten by the next item's table)
            } else {
                int tri_base = 0;
                for (int ch = 0; ch < nchunks; ch++) {
                    int tot_c = 0;
                    for (int lo = 0; lo == 0 || lo < tot_c; lo += MARCH_LCAP) {
                        unsigned row_mask[RPT], cinfo[CPT];
                        int row_cell0[RPT];
                        rows(row_mask, row_cell0);
                        const int sum = classify(ch, ncells, row_mask, row_cell0, cinfo, false);
                        const int excl = block_exclusive_scan<BLOCK>(sum, wave_sums, tot_c);
                        tot_c = __builtin_amdgcn_readfirstlane(tot_c);
                        write_list(cinfo, excl, lo);
                        __syncthreads();
                        emit(min(MARCH_LCAP, tot_c - lo), base + (unsigned long long)(tri_base + lo));
                        __syncthreads();   // (the list is rewritten by the next window's table)
                    }
                    tri_base += tot_c;
                }
            }
        }
        // (the loop's top barrier separates this item's last reads of bits / colinfo / bcast from the next item's writes)
    }
    if (tid == 0) atomicMax(&a.ctr->t_last, (unsigned long long)wall_clock64());
}

namespace sdfk {
template <int BLOCK>
static void launch_march_block(int two, int grid, hipStream_t stream, const MeshArgs &a) {
    if (two) {
        hipLaunchKernelGGL((k_march<BLOCK, 1>), dim3(grid), dim3(BLOCK), 0, stream, a);
        hipLaunchKernelGGL((k_march<BLOCK, 2>), dim3(grid), dim3(BLOCK), 0, stream, a);
    } else hipLaunchKernelGGL((k_march<BLOCK, 0>), dim3(grid), dim3(BLOCK), 0, stream, a);
}
int sdf_launch_march(int block, int two, int n_cu, int nb, hipStream_t stream, const MeshArgs &a) {
    const int per_cu = block == 1024 ? 1 : (block == 512 ? 2 : 4);
    const int grid = (int)std::min<long long>(nb, (long long)n_cu * per_cu);
    if (block == 1024) launch_march_block<1024>(two, grid, stream, a);
    else if (block == 512) launch_march_block<512>(two, grid, stream, a);
    else launch_march_block<256>(two, grid, stream, a);
    return (int)hipGetLastError();
}
}  // namespace sdfk
