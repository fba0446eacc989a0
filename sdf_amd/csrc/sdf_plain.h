// sdf_plain.h -- launchers of the kernels that are NOT tape interpreters (sdf_plain.hip, built without the interpreters'
// structurizer option; build.sh).  Each launches on `stream` with the given grid and block; the caller checks
// hipGetLastError() as it did when the kernels shared its translation unit.
#pragma once
#include "sdf_device.h"

// ---- marching cubes of MANY caller-supplied tiles in one submission (sdf_generate_field: the volumes of a
// chunk of batches sampled by a host callback; generate_big: batch_size > 32, sampled by k_eval_tiles).  A tile
// has at most `slots` rows of cells (1024 for batch_size <= 32; a multiple of 256); row slot tile * slots + t
// carries the row's triangle count (0 beyond the tile's rows), so one scan over the slots numbers the triangles
// of the whole chunk in reference order. ----
struct FieldTile {
    long long vol_off;      // first sample of the tile in the chunk's value buffer
    int n0, n1, n2, pad_;
    double of[3], sc[3];    // points * scale + offset (reference sdf/core.py:58-60)
};

void launch_k_compact(dim3 grid, dim3 block, hipStream_t stream, const unsigned char *kinds, int nbatches, int *worklist, sdfk::MeshCounters *ctr,
                      unsigned long long *status, long long shard_index, long long shard_count);
void launch_k_mc_rows(dim3 grid, dim3 block, hipStream_t stream, const sdfk::McTables *mc, const float *vol, int n0, int n1, int n2, unsigned int *row_count);
void launch_k_scan_rows(dim3 grid, dim3 block, hipStream_t stream, const unsigned int *cnt, long long n, unsigned long long *off, unsigned long long *total);
void launch_k_mc_emit(dim3 grid, dim3 block, hipStream_t stream, const sdfk::McTables *mc, const float *vol, int n0, int n1, int n2,
                      const unsigned long long *row_off, float *out, unsigned long long cap);
void launch_k_cast_f32(dim3 grid, dim3 block, hipStream_t stream, const double *in, float *out, long long n);
void launch_k_field_rows(dim3 grid, dim3 block, hipStream_t stream, const sdfk::McTables *mc, const float *vol, const FieldTile *tiles, unsigned int *row_count,
                         int slots);
void launch_k_field_emit(dim3 grid, dim3 block, hipStream_t stream, const sdfk::McTables *mc, const float *vol, const FieldTile *tiles,
                         const unsigned long long *row_off, double *out, unsigned long long base, unsigned long long cap, int slots);
void launch_k_scan_items(dim3 grid, dim3 block, hipStream_t stream, const sdfk::ItemDesc *desc, sdfk::MeshCounters *ctr, unsigned long long *status,
                         int *block_item, unsigned long long n_blocks);
void launch_k_emit2(dim3 grid, dim3 block, hipStream_t stream, const sdfk::MeshArgs &a);
void launch_k_stl(dim3 grid, dim3 block, hipStream_t stream, const double *pts, long long ntri, unsigned short *out);
