"""Multi-GPU sharding of one `generate` call: one process per GPU, `torch.distributed`
with the "nccl" backend (= RCCL over xGMI on ROCm).

The reference has no distributed path; its only parallelism is an ordered thread-pool map
over independent batches (reference sdf/core.py:131-133).  Here the *surviving* batches
(post sparse-skip work list, in the reference's X-major batch order) are split into
`world_size` contiguous chunks, every rank meshes its chunk on its own GPU, and ONE
exchange step all-gathers the triangle buffers so that every rank ends with the complete
soup; concatenating in rank order reproduces the single-GPU (= reference) triangle order
(SURVEY.md section 8e).  The skip prepass is recomputed on every rank (it is ~1e-4 of the
work) so the work list needs no communication.

The exchange unit is a SLAB of fixed capacity per rank (include/sdf_hip.h): a 128-byte header
(triangle count, work-item count, overflow flag, statistics), and the payload.  RCCL has no
all-gather-v, and a separate all-gather of the counts would put a host round trip between two
collectives on a path whose whole device time is a few hundred microseconds -- so the counts
travel INSIDE the one all-gather of equal-sized slabs, the capacities come from the previous
call of the same job (first call: an upper bound), and a slab that turns out too small is
flagged in its header: every rank sees the same gathered headers, takes the same decision and
repeats the step with larger slabs.  The host synchronises ONCE per step, to read the headers.

On GPUs the payload is marching cubes' own output, 9 float32 per triangle in the batch's local
voxel coordinates (36 bytes instead of the 72 of the float64 soup), plus a 56-byte record per
batch (triangle prefix + `points * scale + offset` transform, reference sdf/core.py:58-60);
`k_mesh` writes that form directly into the slab and `k_expand` produces the ordered float64
soup from the gathered slabs on every rank (csrc/sdf_hip.hip).  Engines that return host soups
(models with user closures, the CPU stand-in of tests/test_dist.py) ship the float64 soup itself
through the same protocol.

With `chunks` > 1 a rank cuts its share into that many consecutive shards and meshes shard k + 1
while the all-gather of shard k is in flight (the collective runs on RCCL's own stream).
"""
import os
import weakref

import numpy as np

HEADER_WORDS = 16          # int64 words at the head of a slab (include/sdf_hip.h)
H_TRIS, H_ITEMS, H_OVERFLOW, H_EMPTY, H_NONEMPTY, H_EVAL, H_AMBIGUOUS, H_SAMPLED, H_PRUNED, H_WORK = range(10)

_HINTS = weakref.WeakKeyDictionary()      # tape object -> {job key: (cap_items, cap_tris, total_tris)}
_HINTS_BY_ID = {}                         # the same for objects that cannot be weakly referenced
_STREAMS = {}                             # device index -> the torch stream the exchange steps run on


def _dist():
    try:
        import torch.distributed as td
    except Exception:
        return None
    if td.is_available() and td.is_initialized():
        return td
    return None


def world_size():
    td = _dist()
    return td.get_world_size() if td is not None else 1


def rank():
    td = _dist()
    return td.get_rank() if td is not None else 0


def shard_bounds(n_work, r, world):
    """contiguous chunk [lo, hi) of the work list for rank r (same formula as
    csrc/sdf_hip.hip `k_compact`)"""
    return (n_work * r) // world, (n_work * (r + 1)) // world


def _hints_for(tape):
    try:
        return _HINTS.setdefault(tape, {})
    except TypeError:
        return _HINTS_BY_ID.setdefault(id(tape), {})


class HostCodec:
    """slabs whose payload is the float64 soup itself (72 bytes per triangle): for engines whose meshes come
    back as host arrays -- models with user closures (sdf_generate_field) and CPU stand-ins"""

    def __init__(self, eng):
        self.eng = eng

    def slab_bytes(self, cap_items, cap_tris):
        return 8 * HEADER_WORDS + 72 * int(cap_tris)

    def pack(self, tape, X, Y, Z, batch_size, sparse, shard, slab, cap_items, cap_tris):
        import torch
        mesh = self.eng.generate(tape, X, Y, Z, batch_size, sparse, shard=shard)
        try:
            st = mesh.stats()
            pts = np.ascontiguousarray(mesh.points(), dtype=np.float64).reshape(-1)
        finally:
            mesh.close()
        t = len(pts) // 9
        h = np.zeros(HEADER_WORDS, np.int64)
        h[H_TRIS] = t
        h[H_ITEMS] = int(st.get('n_work_end', 0)) - int(st.get('n_work_begin', 0))
        h[H_OVERFLOW] = 1 if t > cap_tris else 0
        h[H_EMPTY], h[H_NONEMPTY] = st['empty'], st['nonempty']
        h[H_EVAL], h[H_AMBIGUOUS] = st['n_eval_voxels'], st['n_ambiguous_cells']
        h[H_SAMPLED] = st.get('n_sampled_voxels', st['n_eval_voxels'])
        h[H_PRUNED] = st.get('n_pruned_instrs', 0)
        h[H_WORK] = int(st['batches']) - int(st['skipped'])
        slab[:8 * HEADER_WORDS].copy_(torch.from_numpy(h.view(np.uint8)))
        n = min(t, int(cap_tris))
        if n:
            slab[8 * HEADER_WORDS:8 * HEADER_WORDS + 72 * n].copy_(torch.from_numpy(pts[:9 * n].view(np.uint8)))
        return None

    def expand(self, slabs, headers_hint, cap_items, cap_tris, out, out_cap):
        # (host payloads: the headers are needed to place the pieces, which costs this codec an extra read)
        import torch
        base = 0
        for s in slabs:
            t = int(s[:8].cpu().numpy().view(np.int64)[0])
            n = max(0, min(t, int(cap_tris), int(out_cap) - base))
            if n:
                out[9 * base:9 * (base + n)].copy_(s[8 * HEADER_WORDS:8 * HEADER_WORDS + 72 * n].view(torch.float64))
            base += max(0, min(t, int(cap_tris)))


class DeviceCodec:
    """slabs written and expanded by the HIP library (compact float32 payload, see the module docstring)"""

    def __init__(self, eng):
        self.eng = eng

    def slab_bytes(self, cap_items, cap_tris):
        return self.eng.slab_bytes(cap_items, cap_tris)

    def pack(self, tape, X, Y, Z, batch_size, sparse, shard, slab, cap_items, cap_tris):
        return self.eng.generate_compact(tape, X, Y, Z, batch_size, sparse, shard, slab.data_ptr(), cap_items, cap_tris)

    def expand(self, slabs, headers_hint, cap_items, cap_tris, out, out_cap):
        self.eng.expand_slabs([s.data_ptr() for s in slabs], cap_items, cap_tris, out.data_ptr(), out_cap)


def generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None, chunks=None):
    """every rank returns (soup: flat float64 torch tensor of 9*T values on `device`, in
    reference order; merged stats dict).  On GPUs the soup never leaves the device."""
    import torch
    td = _dist()
    if td is None:
        raise RuntimeError('torch.distributed is not initialised')
    world, r = td.get_world_size(group), td.get_rank(group)
    if device is None:
        backend = td.get_backend(group)
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    on_gpu = device.type == 'cuda'
    externs = getattr(getattr(tape, 'tape', None), 'externs', None)
    codec = DeviceCodec(eng) if (on_gpu and hasattr(eng, 'generate_compact') and not externs) else HostCodec(eng)
    if chunks is None:
        chunks = int(os.environ.get('SDF_DIST_CHUNKS', '1'))
    C = max(1, min(int(chunks), 64 // max(world, 1)))

    s = int(batch_size)
    nb = (-(-len(X) // s)) * (-(-len(Y) // s)) * (-(-len(Z) // s))
    hints = _hints_for(tape)
    key = (len(X), len(Y), len(Z), s, bool(sparse), world, C, type(codec).__name__)
    if key in hints:
        cap_items, cap_tris, total_hint = hints[key]
    else:                       # first call: a shard is a contiguous piece of the work list, at most 1/(world*C) of ALL batches
        cap_items = -(-nb // (world * C)) + 1
        # (triangles: a guess, 4096 per batch -- 2.4 x what the BASELINE models produce -- within 8 GB of gathered
        # slabs; a slab that is too small is flagged in its header and the step repeated)
        cap_tris = max(min(4096 * cap_items, (8 << 30) // (72 * world * C)), 1 << 16)
        total_hint = 0

    # The step runs on a stream of its own that the engine adopts: its kernels, torch's allocations and the collective
    # are then ordered among themselves without a host round trip.  (torch's DEFAULT stream has the null handle, which
    # the engine's sdf_ctx_set_stream reads as "back to your own stream" -- adopting it would silently unorder the
    # meshing kernels and the all-gather.)  The caller's stream waits for the step's at the end.
    adopted = False
    outer = step_stream = None
    if on_gpu:
        outer = torch.cuda.current_stream(device)
        step_stream = _STREAMS.get(device.index)
        if step_stream is None:
            step_stream = _STREAMS[device.index] = torch.cuda.Stream(device)
        step_stream.wait_stream(outer)
        if hasattr(eng, 'set_stream'):
            eng.set_stream(step_stream.cuda_stream)
            adopted = True
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if on_gpu else None
    import contextlib
    try:
        with (torch.cuda.stream(step_stream) if on_gpu else contextlib.nullcontext()):
            for attempt in range(6):
                sb = codec.slab_bytes(cap_items, cap_tris)
                if ev:
                    ev[0].record()
                gathered, works, meshes = [], [], []
                for j in range(C):
                    mine = torch.empty(sb, dtype=torch.uint8, device=device)
                    meshes.append(codec.pack(tape, X, Y, Z, batch_size, sparse, (r * C + j, world * C), mine, cap_items, cap_tris))
                    g = torch.empty(world * sb, dtype=torch.uint8, device=device)
                    if ev and j == C - 1:
                        ev[1].record()
                    # one collective per shard; with several shards the gather of shard j overlaps the meshing of shard j + 1
                    works.append(td.all_gather_into_tensor(g, mine, group=group, async_op=C > 1))
                    gathered.append((g, mine))
                for w in works:
                    if w is not None and C > 1:
                        w.wait()
                if ev:
                    ev[2].record()
                slabs = [gathered[j][0][rr * sb:(rr + 1) * sb] for rr in range(world) for j in range(C)]      # final order
                out_cap = max(total_hint + total_hint // 8 + 4096, 1) if total_hint else world * C * cap_tris
                out = torch.empty(out_cap * 9, dtype=torch.float64, device=device)
                codec.expand(slabs, None, cap_items, cap_tris, out, out_cap)
                if ev:
                    ev[3].record()
                # the ONE host synchronisation of the step: the gathered headers
                heads = torch.stack([sl[:8 * HEADER_WORDS] for sl in slabs]).cpu().numpy().view(np.int64).reshape(len(slabs), HEADER_WORDS)
                for m in meshes:
                    if m is not None:
                        m.close()
                total = int(heads[:, H_TRIS].sum())
                need_items, need_tris = int(heads[:, H_ITEMS].max()), int(heads[:, H_TRIS].max())
                if (heads[:, H_OVERFLOW] & 2).any():
                    raise RuntimeError('sdf_amd.dist: a rank reported a look-back timeout')
                ok = not heads[:, H_OVERFLOW].any() and need_items <= cap_items and need_tris <= cap_tris and total <= out_cap
                total_hint = total
                if ok:
                    break
                # every rank sees the same headers, so every rank repeats the step with the same larger capacities
                cap_items = max(cap_items, need_items + need_items // 8 + 16)
                cap_tris = max(cap_tris, need_tris + need_tris // 8 + 1024)
            else:
                raise RuntimeError('sdf_amd.dist: slab capacities did not converge')
            # capacities for the next call of this job: what this one needed, with some slack
            hints[key] = (need_items + need_items // 8 + 16, need_tris + need_tris // 8 + 1024, total)
            soup = out[:total * 9]
    finally:
        if adopted:
            eng.set_stream(0)
        if on_gpu:
            outer.wait_stream(step_stream)        # whatever the caller enqueues next sees the soup
    if on_gpu:
        soup.record_stream(outer)                 # (allocated on the step's stream, used on the caller's)

    per_rank = heads[:, H_TRIS].reshape(world, C).sum(axis=1)
    merged = {
        'batches': nb, 'n_batches': nb,
        'skipped': nb - int(heads[0, H_WORK]), 'n_skipped': nb - int(heads[0, H_WORK]),
        'empty': int(heads[:, H_EMPTY].sum()), 'nonempty': int(heads[:, H_NONEMPTY].sum()),
        'n_eval_voxels': int(heads[:, H_EVAL].sum()), 'n_ambiguous_cells': int(heads[:, H_AMBIGUOUS].sum()),
        'n_sampled_voxels': int(heads[:, H_SAMPLED].sum()), 'n_pruned_instrs': int(heads[:, H_PRUNED].sum()),
        'triangles': total, 'n_triangles': total, 'per_rank_triangles': [int(c) for c in per_rank],
        'n_grid_voxels': len(X) * len(Y) * len(Z), 'n_retries': attempt, 'chunks': C,
        'slab_bytes': sb, 'payload': 'f32 local + per-batch transform' if isinstance(codec, DeviceCodec) else 'f64 soup',
    }
    merged['n_empty'], merged['n_nonempty'] = merged['empty'], merged['nonempty']
    if ev:      # (the headers' copy has synchronised the stream: the events are complete)
        merged['ms_mesh'] = ev[0].elapsed_time(ev[1])
        merged['ms_exchange'] = ev[1].elapsed_time(ev[2])
        merged['ms_expand'] = ev[2].elapsed_time(ev[3])
    else:
        merged['ms_mesh'] = merged['ms_exchange'] = merged['ms_expand'] = 0.0
    return soup, merged


def generate_sharded(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None):
    """every rank returns (points (3T,3) float64 ndarray, merged stats dict)"""
    soup, merged = generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse, device, group)
    return soup.cpu().numpy().reshape(-1, 3), merged
