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
#include "sdf_device.h"
#include "sdf_plain.h"
#include "sdf_slab.h"

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
