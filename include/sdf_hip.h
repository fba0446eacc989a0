/*
 * sdf_hip.h -- C ABI of libsdf_hip.so, the MI355X (gfx950) sampling + meshing engine.
 *
 * The reference (fogleman/sdf) is a single Python process with no FFI boundary; the seams
 * this ABI replaces are the module-level functions of reference sdf/core.py (cited per entry
 * point).  The host side that binds these symbols is sdf_amd/engine.py (ctypes); the
 * maintainer-side binding is shown in INTEGRATION.md.
 *
 * Conventions: plain C types only; every function returns 0 on success and a non-zero code on
 * failure, with a thread-local message available from sdf_last_error(); no C++ exception crosses
 * the boundary.  Handles are opaque.  A context owns one HIP stream (or adopts the caller's) and
 * all device scratch; calls on one context must not overlap in time (one caller per context);
 * different contexts (e.g. one per GPU / per process) are independent.
 * "host" pointers are ordinary pageable memory, "device" pointers are HIP device memory on the
 * context's GPU.
 */
#ifndef SDF_HIP_H
#define SDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDF_ABI_VERSION 9

#define SDF_PRECISION_F64 0 /* float64 evaluation like the reference's NumPy path: what every sdf_generate* entry point samples in */
#define SDF_PRECISION_F32 1 /* float32 evaluation: sdf_eval_* and sdf_estimate_bounds only (the meshing path refuses it since round 5) */

typedef struct sdf_ctx sdf_ctx;   /* one HIP device + stream + scratch arenas            */
typedef struct sdf_tape sdf_tape; /* a lowered model (op tape + constants) on the device */
typedef struct sdf_mesh sdf_mesh; /* the result of one sdf_generate call                 */

/* per-call statistics; the first four are what reference sdf/core.py:144-145 prints */
typedef struct sdf_stats {
    int64_t n_batches;
    int64_t n_skipped;
    int64_t n_empty;            /* within this call's shard */
    int64_t n_nonempty;         /* within this call's shard */
    int64_t n_triangles;        /* within this call's shard */
    int64_t n_grid_voxels;      /* len(X)*len(Y)*len(Z)                                    */
    int64_t n_eval_voxels;      /* samples evaluated by the meshing kernel (this shard)    */
    int64_t n_ambiguous_cells;  /* surface cells with an ambiguous sign configuration      */
    int64_t n_work_begin;       /* this shard's range in the surviving-batch work list     */
    int64_t n_work_end;
    int64_t n_retries;          /* meshing re-runs after a triangle-arena overflow         */
    double ms_prepass;          /* HIP-event times on the context's stream                 */
    double ms_mesh;             /* the fused sample+march kernel alone (last run)          */
    double ms_emit;             /* ordered gather / f64 transform kernel (last emit)       */
    double ms_total;            /* first launch to last kernel of sdf_generate             */
    int64_t n_pruned_instrs;    /* tape instructions the interval prepass removed, summed over the shard's batches */
    int64_t n_batch_instrs;     /* (instructions per tape) x (batches of the shard): the total they come out of    */
    int64_t n_sampled_voxels;   /* of n_eval_voxels, the samples that went through the interpreter (the rest lie in
                                 * cell groups whose interval excludes the surface; in lots of 64)                 */
    double ms_mesh_device;      /* the sample+march kernel by the device's own constant-rate counter: first workgroup's
                                 * start to last workgroup's end (no HIP event, no host in the measurement)        */
    double sclk_mhz;            /* shader clock that kernel ran at (its cycle counter against the constant one)    */
    double t_mesh_first_us;     /* when that kernel's first workgroup started / its last one ended, microseconds on the */
    double t_mesh_last_us;      /* device's constant-rate counter: calls in flight can be laid on ONE time axis          */
    int64_t mesh_kernel;        /* which fused kernel meshed the call (ABI 8): 1 = k_mesh, one workgroup of 1024 threads per
                                 * compute unit; 2 = k_mesh2, two of 512 (sdf_ctx_set_mesh2); 0 = neither (batch_size > 32,
                                 * closures, adopted soups) */
} sdf_stats;

int sdf_abi_version(void);
/* what built this library: the compiler's version lines and the interpreters' flags, as csrc/build.sh recorded them (ABI 7) */
const char *sdf_build_info(void);
const char *sdf_last_error(void);
int sdf_device_count(void); /* <= 0 when no HIP device is usable */

/* free / total device memory (hipMemGetInfo) */
int sdf_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
/* TEST HOOK: the nth device / pinned-host allocation the library makes from now on fails once (0: off).  The tests
 * walk it through sdf_ctx_create / sdf_tape_create / sdf_generate and check that every error path returns what it
 * had taken (tests/test_gpu.py::test_failed_allocations_leak_nothing). */
int sdf_test_fail_alloc(int nth);

int sdf_ctx_create(int device, sdf_ctx **out);
int sdf_ctx_destroy(sdf_ctx *ctx);
/* adopt a caller-owned hipStream_t (e.g. torch's current stream); NULL returns to the own stream */
int sdf_ctx_set_stream(sdf_ctx *ctx, void *hip_stream);
/* the interval prepass of sdf_generate (see sdf_tape_set_prune_info) is on by default; 0 switches it
 * off for this context (results are identical either way; the environment variable SDF_PRUNE=0
 * sets the initial state) */
int sdf_ctx_set_prune(sdf_ctx *ctx, int enabled);
/* the same for the second interval pass of sdf_generate: inside a batch, groups of 4^3 cells whose
 * interval excludes the surface are not sampled (SDF_CULL=0 sets the initial state; results are
 * identical either way) */
int sdf_ctx_set_cull(sdf_ctx *ctx, int enabled);
/* how sdf_generate meshes: 0 = one kernel (ordered look-back + parking inside the sampling kernel), 1 = three kernels
 * (sample + classify / number the triangles / emit), -1 = the library's choice by the tape's length (default; the
 * environment variable SDF_MESH_TWOPASS sets the initial state).  Results are identical either way. */
int sdf_ctx_set_twopass(sdf_ctx *ctx, int mode);
/* Inside the one-kernel scheme: 1 (default; SDF_DEFER sets the initial state) = a culled batch keeps only the samples the
 * interval pass listed (a sparse tile) and STAYS in the CU's LDS while the workgroup samples its next batch, so that its
 * triangles are written once, in their final place, when the earlier batches' counts are known; 0 = dense tiles, a
 * batch whose predecessors are not counted yet is parked in device memory and moved later (the r03 scheme).  Results are
 * identical either way. */
int sdf_ctx_set_defer(sdf_ctx *ctx, int on);
/* Which fused kernel meshes a call (ABI 8): k_mesh -- one persistent workgroup of 1024 threads per compute unit -- or k_mesh2 --
 * two of 512 threads, each with half the CU's LDS, so that one workgroup's counting / look-back / emission overlaps the other's
 * interpreter; it holds sparse tiles only.  0 (default; SDF_MESH2 sets the initial state) = never: measured in round 6, k_mesh2
 * is bit-identical and 4 - 6 % slower at 512^3 (profiles/r06e_two_wg.json); -1 = k_mesh2 when the previous call of the same tape
 * on the same grid found every tile to be its (the first call takes k_mesh), 1 = k_mesh2 whenever the tape has a variant (a tile
 * it does not hold is flagged on the device and the pass repeated with k_mesh).  Results are identical either way;
 * sdf_stats.mesh_kernel says which one ran. */
int sdf_ctx_set_mesh2(sdf_ctx *ctx, int mode);
/* interval levels of the second interval pass: 2 = boxes of 8^3 and groups of 4^3 cells, 3 = + sub-groups of 2^3 cells,
 * 0 = the library's choice by the tape (default; SDF_CULL_LEVELS sets the initial state).  Results are identical. */
int sdf_ctx_set_cull_levels(sdf_ctx *ctx, int levels);
/* Scheduling of the meshing pass: 1 (default; SDF_TAIL_ORDER sets the initial state) = the last few hundred surviving
 * batches are handed to the workgroups by descending cost estimate instead of by position (shorter tail of the
 * kernel), 0 = strictly in list order.  Results are identical either way. */
int sdf_ctx_set_tail_order(sdf_ctx *ctx, int on);
int sdf_ctx_synchronize(sdf_ctx *ctx);
/* give the device memory the library caches for reuse back to the driver (between jobs of very different sizes) */
int sdf_ctx_trim(sdf_ctx *ctx);

/* Upload an op tape produced by sdf_amd/tape.py (2 x uint32 per instruction, float64 constants).
 * Plays the role of the reference's closure tree (reference sdf/d3.py:48-63). */
int sdf_tape_create(sdf_ctx *ctx, const uint32_t *code, uint32_t n_words, const double *consts,
                    uint32_t n_consts, uint32_t n_pslots, uint32_t n_dslots, sdf_tape **out);
/* Optional: per instruction, where the right operand and the left operand chain of a hard
 * min / max combine start (0xFFFF: not a combine) -- produced by sdf_amd/tape.py next to the tape.
 * With it, sdf_generate runs an interval prepass per surviving batch and skips the instructions
 * that provably cannot influence any sample of that batch (results are unchanged bit for bit). */
int sdf_tape_set_prune_info(sdf_tape *tape, const uint16_t *rstart, const uint16_t *lstart, uint32_t n_instr);
int sdf_tape_destroy(sdf_tape *tape);

/* f(P) for N points of dimension dim (2 or 3): replaces SDF3.__call__ / SDF2.__call__
 * (reference sdf/d3.py:24-25, sdf/d2.py:23-24).  Output is float64 in both precisions. */
int sdf_eval_points(sdf_tape *tape, const void *d_points, int64_t n, int dim, void *d_out, int precision);
int sdf_eval_points_host(sdf_tape *tape, const double *h_points, int64_t n, int dim, double *h_out,
                         int precision);
/* f on the cartesian product X x Y x Z, first axis slowest: replaces `_cartesian_product` +
 * `sdf(P)` (reference sdf/core.py:20-26, :50-52, :78, :232).  h_out has nx*ny*nz elements. */
int sdf_eval_grid_host(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z,
                       int nz, double *h_out, int precision);

/* Models with user-written closures (the reference's documented extension point: a function decorated with
 * @sdf3 / @op3 returns `f(p)`, NumPy code -- reference README.md:258-295, sdf/d3.py:48-63).  The closure runs on
 * the host, in the user's interpreter; the tape reads its values through L_EXTERN leaves.  f(P) then takes two
 * device passes with the closures in between:
 *   sdf_eval_extern_points_host   h_ext_points[k][i][3] = the point leaf k sees for sample i (n_extern x n x 3)
 *   sdf_eval_points_extern_host   f(P) given h_ext_values[k][i] = closure k at that point      (n_extern x n)
 * The plain entry points refuse such a tape. */
int sdf_tape_extern_count(sdf_tape *tape);
int sdf_eval_extern_points_host(sdf_tape *tape, const double *h_points, int64_t n, int dim, double *h_ext_points,
                                int precision);
int sdf_eval_points_extern_host(sdf_tape *tape, const double *h_points, int64_t n, int dim, const double *h_ext_values,
                                double *h_out, int precision);
/* The batch loop of `generate` (reference sdf/core.py:114-141) around a field evaluated by a HOST callback:
 * `field(user, points (n x 3 float64, host), n, values (n float64, host))` returns 0, or non-zero to abort.  The
 * library builds the points of the skip test (`_skip`, core.py:28-43) and of every surviving batch
 * (`_cartesian_product`, core.py:20-26), calls the field, and meshes the values on the device (float32 cast,
 * marching cubes, `points * scale + offset`); the result is an ordinary sdf_mesh. */
typedef int (*sdf_field_fn)(void *user, const double *points, int64_t n, double *values);
int sdf_generate_field(sdf_ctx *ctx, sdf_field_fn field, void *user, const double *X, int nx, const double *Y, int ny,
                       const double *Z, int nz, int batch_size, int sparse, int64_t shard_index, int64_t shard_count,
                       sdf_mesh **out);

/* `_estimate_bounds` (reference sdf/core.py:62-82: up to 32 rounds of a 16^3 probe grid shrinking from +-1e9) in one
 * launch: h_out6 = x0, y0, z0, x1, y1, z1.  Fails -- with NumPy's message -- where the reference raises (a round in
 * which no probe lies within half a cell diagonal of the surface). */
int sdf_estimate_bounds(sdf_tape *tape, double *h_out6, int precision);

/* Marching cubes of a C-order float32 volume (n0,n1,n2) at level 0: replaces `_marching_cubes`
 * (reference sdf/core.py:16-18 -> skimage.measure.marching_cubes(volume, 0)).  Writes up to
 * cap_tris triangles (9 float32 each: 3 vertices in volume index coordinates, reference soup
 * order) and always reports the full count in *n_tris (call again with a larger buffer if it
 * exceeds cap_tris).  A volume with no surface yields *n_tris == 0 (the reference turns the
 * corresponding skimage exceptions into an empty batch, sdf/core.py:53-56). */
int sdf_marching_cubes(sdf_ctx *ctx, const void *d_volume, int n0, int n1, int n2, void *d_out_tris,
                       int64_t cap_tris, int64_t *n_tris);
int sdf_marching_cubes_host(sdf_ctx *ctx, const float *h_volume, int n0, int n1, int n2, float *h_out_tris,
                            int64_t cap_tris, int64_t *n_tris);

/* The batch loop of `generate` (reference sdf/core.py:114-141) with `_worker` (:45-60) and
 * `_skip` (:28-43) inside: X, Y, Z are the float64 `np.arange` axes (host), batch_size the
 * reference's BATCH_SIZE (1 .. 512; the reference takes any, core.py:87, 114-119), sparse its
 * `sparse=` flag.  shard_index/shard_count split the surviving-batch work list into contiguous
 * chunks (1 GPU: 0, 1).  The triangles stay on the device in the returned mesh until emitted.
 * batch_size <= 32 (the default: 32) runs the fused LDS-resident kernels; a larger batch's
 * (batch_size + 1)^3 float32 tile does not fit a compute unit's LDS and goes through device
 * memory, a chunk of batches per submission, synchronously, into library memory: the variants
 * below that take a caller buffer then report it as not filled (*emitted = 0), exactly as for a
 * buffer that is too small; sdf_generate_compact_async / sdf_generate_from_kinds (the multi-GPU
 * exchange) refuse batch_size > 32. */
int sdf_generate(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                 int batch_size, int sparse, int64_t shard_index, int64_t shard_count, int precision,
                 sdf_mesh **out);
/* The same, writing the ordered soup (`points * scale + offset`, reference sdf/core.py:58-60, :141)
 * straight into caller-owned DEVICE memory of 9*cap_tris doubles.  *emitted = 1 when the soup
 * (sdf_mesh_triangles() triangles) is in d_out; 0 when it did not fit: nothing is written past
 * cap_tris, the contents of d_out are then unspecified, and the mesh holds the complete soup in
 * library memory (the meshing pass was repeated) -- fetch it with sdf_mesh_emit_device/_host. */
int sdf_generate_to_device(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                           int batch_size, int sparse, int64_t shard_index, int64_t shard_count, int precision,
                           void *d_out, int64_t cap_tris, int *emitted, sdf_mesh **out);
/* The same without the host synchronisation: the call is enqueued on the context's stream and the mesh
 * is returned "in flight"; sdf_mesh_wait (or any function that reads the mesh) collects it.  Up to 8
 * calls of one context may be in flight (a ninth waits for the oldest); give each its own d_out.  This
 * is how a caller that meshes many jobs back to back (bench.py) keeps the device busy while the host
 * prepares the next submission; the reference has no counterpart (its generate() is synchronous).
 * sdf_mesh_wait: *emitted as for sdf_generate_to_device (0: the soup did not fit d_out, the call was
 * repeated into library memory). */
int sdf_generate_to_device_async(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                                 int batch_size, int sparse, int64_t shard_index, int64_t shard_count, int precision,
                                 void *d_out, int64_t cap_tris, sdf_mesh **out);
int sdf_mesh_wait(sdf_mesh *mesh, int *emitted);
/* `generate` for a caller who wants the soup ON THE HOST -- the list of points the reference's generate() returns
 * (reference sdf/core.py:131-141): like sdf_generate (one device, the whole work list), but the triangles are written as
 * 16-byte records into a slab of the library's (the exchange unit below: marching cubes' local float32 coordinates + one
 * transform per work item) instead of as 72-byte float64 triangles, and sdf_mesh_emit_host_workers makes the float64 soup
 * on host threads while the records arrive: 47 MB over PCIe instead of 212 MB at 512^3, the same bits (it performs
 * k_expand's `double(local) * scale + offset` on the same operands).  The slab is sized from the last call of the same
 * model on the same grid; the first such call, batch_size > 32 and tapes with user closures are served by sdf_generate.
 * Every reader of the mesh works: those that need the float64 soup on the device get it from k_expand on demand. */
int sdf_generate_records(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                         int batch_size, int sparse, int precision, sdf_mesh **out);
/* Multi-GPU exchange (north_star: "batches shard naturally over the 8 GPUs of one node with an RCCL all-gather of
 * triangle buffers"; the reference itself has no distributed path).  A rank meshes its shard of the surviving-batch
 * work list into a SLAB of fixed capacity in caller-owned device memory -- header (counts, statistics, overflow
 * flag), per-work-item triangle prefix and transform, and the triangles as 16-BYTE records (the three along-edge float32
 * of marching cubes' local coordinates bit for bit + one word for the cell and the edges; a triangle with a vertex inside
 * a cell keeps its nine floats in a small raw area: sdf_amd/csrc/sdf_slab.h, restated in sdf_amd/slabcodec.py) instead of
 * the 72 bytes of the float64 soup -- the caller all-gathers the equal-sized slabs of all
 * ranks in ONE collective, and sdf_expand_slabs writes the ordered float64 soup from the gathered slabs (given in
 * final order) on every rank.  Everything is only ENQUEUED on the context's stream; the caller reads the 128-byte
 * headers (int64[16]: n_tris, n_items, overflow, n_empty, n_nonempty, n_eval, n_ambiguous, n_sampled, n_pruned,
 * n_work_total, n_raw, need_tris, ...) once at the end; overflow != 0 anywhere: repeat with capacities >= the largest
 * n_items and max(n_tris, need_tris). */
size_t sdf_slab_bytes(int64_t cap_items, int64_t cap_tris);
int sdf_generate_compact_async(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                               int batch_size, int sparse, int64_t shard_index, int64_t shard_count, int precision,
                               void *d_slab, int64_t cap_items, int64_t cap_tris, sdf_mesh **out);
int sdf_expand_slabs(sdf_ctx *ctx, const void *const *d_slabs, int n_slabs, int64_t cap_items, int64_t cap_tris,
                     void *d_out, int64_t cap_out_tris);
/* The multi-GPU step inside the library: one process per GPU, RCCL (dlopen'ed: librccl.so.1) over xGMI.
 *   sdf_comm_available   1 if librccl could be loaded in this process, else 0 (sdf_last_error says why).  Local and cheap:
 *                        the ranks agree on it BEFORE anything collective is started, so that a rank without the
 *                        library does not leave the others inside ncclCommInitRank
 *   sdf_comm_unique_id   rank 0 draws one 128-byte id per lane (ncclGetUniqueId) and hands them to the other ranks out
 *                        of band (sdf_amd/dist.py: through the process group that launched the ranks)
 *   sdf_comm_create      collective: every rank, same ids, its own rank.  A communicator has 1 or 2 LANES -- a lane is
 *                        an RCCL communicator + streams + persistent slab / gathered-slab / soup buffers -- so that two
 *                        steps can be in flight: step i + 1 meshes while step i's all-gather is on the links.
 *   sdf_generate_sharded_async   enqueue one step on a lane: mesh this rank's share of the surviving-batch work list
 *                        (contiguous, `chunks` shards per rank; from SDF_SKIP_SHARD_MIN = 32768 batches on the skip
 *                        test itself is shared out too and its one-byte verdicts all-gathered) into slabs, ONE
 *                        ncclAllGather per shard, sdf_expand_slabs' kernel, the gathered headers to pinned memory.
 *                        Nothing waits on the host.  Every rank must submit the same steps in the same order.
 *   sdf_exchange_wait    the step's ONE host synchronisation: reads the gathered headers; slabs that were too small
 *                        are flagged there -- every rank sees the same headers and repeats the step with the
 *                        capacities the headers ask for (first call of a job: a guess, later calls: what the last one
 *                        needed).  *d_soup = the complete ordered float64 soup (9 doubles per triangle, reference
 *                        order: sdf/core.py:141) in library memory on THIS rank's GPU, valid until the next step is
 *                        submitted on the same lane; identical on every rank and to the single-GPU soup. */
#define SDF_COMM_ID_BYTES 128
typedef struct sdf_comm sdf_comm;
typedef struct sdf_exchange sdf_exchange;
typedef struct sdf_exchange_stats {
    int64_t n_batches, n_skipped, n_empty, n_nonempty, n_triangles, n_grid_voxels, n_eval_voxels, n_ambiguous_cells,
        n_sampled_voxels, n_pruned_instrs;      /* whole job (sums over the ranks' slabs)                             */
    int64_t n_retries, chunks, world, slab_bytes;
    double ms_mesh, ms_exchange, ms_expand, ms_total;   /* this rank, HIP events on the lane: prepass + meshing of its
                                                 * shard(s) / until the last all-gather has landed / k_expand / all   */
    int64_t per_rank_triangles[64];
} sdf_exchange_stats;
int sdf_comm_available(void);
int sdf_comm_unique_id(void *out_id128);
int sdf_comm_create(sdf_ctx *ctx, const void *ids, int n_lanes, int rank, int world, sdf_comm **out);
int sdf_comm_destroy(sdf_comm *comm);
int sdf_generate_sharded_async(sdf_comm *comm, sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z,
                               int nz, int batch_size, int sparse, int precision, int chunks, int lane, sdf_exchange **out);
int sdf_exchange_wait(sdf_exchange *x, void **d_soup, int64_t *n_tris);
int sdf_exchange_stats_get(sdf_exchange *x, sdf_exchange_stats *out);
int sdf_exchange_destroy(sdf_exchange *x);
/* `_skip` (reference sdf/core.py:28-43) alone, for batches [b_begin, b_end) of the grid: d_kinds (device memory, one
 * byte per batch of the WHOLE grid) receives 0 (skipped) / 255 (to be meshed) at those positions.  And `generate` for a
 * grid whose verdicts are already there (every byte of d_kinds set): what the sharded step does between its ranks. */
int sdf_skip_kinds(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz, int batch_size,
                   int64_t b_begin, int64_t b_end, int precision, void *d_kinds);
int sdf_generate_from_kinds(sdf_tape *tape, const double *X, int nx, const double *Y, int ny, const double *Z, int nz,
                            int batch_size, int64_t shard_index, int64_t shard_count, int precision, const void *d_kinds,
                            sdf_mesh **out);
/* device -> host copy on the context's stream (for soups in library memory: sdf_exchange_wait) */
int sdf_memcpy_to_host(sdf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int sdf_mesh_stats(sdf_mesh *mesh, sdf_stats *out);
int64_t sdf_mesh_triangles(sdf_mesh *mesh);
/* write the (3T,3) float64 world-space soup (reference order; `points * scale + offset`,
 * reference sdf/core.py:58-60) into caller-owned device / host memory of 9*T doubles */
int sdf_mesh_emit_device(sdf_mesh *mesh, void *d_out);
int sdf_mesh_emit_host(sdf_mesh *mesh, double *h_out);
/* the same with the reference's `workers=` (sdf/core.py:87, 131): the number of host threads that expand the records of a
 * mesh of sdf_generate_records into the float64 soup (<= 0: the machine's, at most 32; a caller's number: at most 64); ignored by any other mesh, whose
 * soup is copied as it is.  sdf_mesh_emit_host = workers 0. */
int sdf_mesh_emit_host_workers(sdf_mesh *mesh, double *h_out, int workers);
/* triangles [first_tri, first_tri + n_tris) of the soup only (9 doubles each) */
int sdf_mesh_emit_host_range(sdf_mesh *mesh, int64_t first_tri, int64_t n_tris, double *h_out);
/* n_batches + 1 entries: h_out[b] = index (within this shard's soup) of the first triangle of batch b in
 * reference batch order, h_out[n_batches] = sdf_mesh_triangles(); batches that were skipped, are empty
 * or belong to another shard have h_out[b] == h_out[b + 1].  This is the bookkeeping the reference keeps
 * implicitly by extending its point list batch after batch (reference sdf/core.py:137-141). */
int sdf_mesh_batch_offsets(sdf_mesh *mesh, int64_t *h_out);
/* T binary-STL records of 50 bytes (f32 normal, 3 x f32 vertex, u16 0), i.e. the body that
 * `write_binary_stl` writes after the 84-byte header (reference sdf/stl.py:4-24) */
int sdf_mesh_emit_stl_host(sdf_mesh *mesh, void *h_out);
/* Vertex weld on the device: what `np.unique(points, axis=0, return_inverse=True)` computes for the
 * reference's non-STL export (reference sdf/core.py:160-164, `_mesh`).  sdf_mesh_weld sorts and
 * deduplicates the soup rows in library memory and reports the number of unique rows;
 * sdf_mesh_weld_fetch copies them out: h_points = n_unique x 3 float64 in lexicographic order,
 * h_cells = T x 3 int64, the unique-row index of every soup row. */
int sdf_mesh_weld(sdf_mesh *mesh, int64_t *n_unique);
/* A mesh handle over a float64 soup that ALREADY sits in device memory (n_tris x 9 doubles, e.g. the gathered soup of a
 * multi-GPU step): sdf_mesh_emit_stl_host / sdf_mesh_weld / sdf_mesh_emit_host* then work on it like on a soup the
 * library generated.  The memory stays the caller's (alive and complete until the handle is destroyed). */
int sdf_mesh_adopt_soup(sdf_ctx *ctx, const void *d_soup, int64_t n_tris, sdf_mesh **out);
int sdf_mesh_weld_fetch(sdf_mesh *mesh, double *h_points, int64_t *h_cells);
/* Pinned host memory for the results above: copies into it run at the link rate (fresh pageable memory:
 * ~10 GB/s).  Blocks are recycled through a small free list inside the library (pinning is slow), so
 * free what you allocate.  Any "host" pointer of this API may point into such a block. */
int sdf_host_alloc(size_t bytes, void **out);
int sdf_host_free(void *p);
/* per-batch classification, n_batches bytes: 0 skipped, 1 empty, 2 nonempty, 3 other shard */
int sdf_mesh_kinds(sdf_mesh *mesh, uint8_t *h_out);
/* diagnostics: what the interval prepass decided, 16 words per batch in batch order like
 * sdf_mesh_kinds (words 0..7: bit i set = instruction i skipped; 8..15: combine i takes its right
 * operand; decided for every batch, used by the ones that were meshed).  Fails when the mesh was
 * generated without the prepass. */
int sdf_mesh_prune_masks(sdf_mesh *mesh, uint32_t *h_out);
int sdf_mesh_destroy(sdf_mesh *mesh);

#ifdef __cplusplus
}
#endif
#endif /* SDF_HIP_H */
