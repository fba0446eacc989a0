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
_STREAMS = {}                             # device index -> {lane: the torch stream exchange steps of that lane run on}


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


class ShardedStep:
    """one exchange step in flight (submit_sharded): everything is enqueued, nothing has been read back"""
    __slots__ = ('eng', 'tape', 'args', 'device', 'group', 'world', 'C', 'nb', 'codec', 'key', 'caps', 'out_cap', 'sb',
                 'slabs', 'keep', 'out', 'meshes', 'events', 'stream', 'outer', 'lane', 'attempt')


def _job(eng, tape, X, Y, Z, batch_size, sparse, device, group, chunks):
    import torch
    td = _dist()
    if td is None:
        raise RuntimeError('torch.distributed is not initialised')
    world = td.get_world_size(group)
    if device is None:
        backend = td.get_backend(group)
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    externs = getattr(getattr(tape, 'tape', None), 'externs', None)
    codec = DeviceCodec(eng) if (device.type == 'cuda' and hasattr(eng, 'generate_compact') and not externs) else HostCodec(eng)
    if chunks is None:
        chunks = int(os.environ.get('SDF_DIST_CHUNKS', '1'))
    C = max(1, min(int(chunks), 64 // max(world, 1)))
    s = int(batch_size)
    nb = (-(-len(X) // s)) * (-(-len(Y) // s)) * (-(-len(Z) // s))
    key = (len(X), len(Y), len(Z), s, bool(sparse), world, C, type(codec).__name__)
    return device, world, C, nb, codec, key


def submit_sharded(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None, chunks=None, lane=0, _caps=None, _attempt=0):
    """enqueue one exchange step -- mesh this rank's shard(s) into slabs, all-gather, expand -- and return without
    waiting for any of it; `collect_sharded` finishes the step.  Steps submitted on different `lane`s (0 / 1) run on
    streams of their own, so step i + 1's meshing overlaps step i's collective."""
    import contextlib
    import torch
    td = _dist()
    device, world, C, nb, codec, key = _job(eng, tape, X, Y, Z, batch_size, sparse, device, group, chunks)
    r = td.get_rank(group)
    on_gpu = device.type == 'cuda'
    hints = _hints_for(tape)
    if _caps is not None:
        cap_items, cap_tris, total_hint = _caps
    elif key in hints:
        cap_items, cap_tris, total_hint = hints[key]
    else:                       # first call: a shard is a contiguous piece of the work list, at most 1/(world*C) of ALL batches
        cap_items = -(-nb // (world * C)) + 1
        # (triangles: a guess, 4096 per batch -- 2.4 x what the BASELINE models produce -- within 8 GB of gathered
        # slabs; a slab that is too small is flagged in its header and the step repeated)
        cap_tris = max(min(4096 * cap_items, (8 << 30) // (72 * world * C)), 1 << 16)
        total_hint = 0

    st = ShardedStep()
    st.eng, st.tape, st.args, st.device, st.group = eng, tape, (X, Y, Z, batch_size, sparse, chunks), device, group
    st.world, st.C, st.nb, st.codec, st.key, st.caps, st.lane, st.attempt = world, C, nb, codec, key, (cap_items, cap_tris, total_hint), lane, _attempt
    # The step runs on a stream of its own that the engine adopts: its kernels, torch's allocations and the collective
    # are then ordered among themselves without a host round trip.  (torch's DEFAULT stream has the null handle, which
    # the engine's sdf_ctx_set_stream reads as "back to your own stream" -- adopting it would silently unorder the
    # meshing kernels and the all-gather.)  The caller's stream waits for the step's when the step is collected.
    adopted = False
    st.outer = st.stream = None
    if on_gpu:
        st.outer = torch.cuda.current_stream(device)
        lanes = _STREAMS.setdefault(device.index, {})
        st.stream = lanes.get(lane)
        if st.stream is None:
            st.stream = lanes[lane] = torch.cuda.Stream(device)
        st.stream.wait_stream(st.outer)
        if hasattr(eng, 'set_stream'):
            eng.set_stream(st.stream.cuda_stream)
            adopted = True
    st.events = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if on_gpu else None
    ev = st.events
    try:
        with (torch.cuda.stream(st.stream) if on_gpu else contextlib.nullcontext()):
            sb = st.sb = codec.slab_bytes(cap_items, cap_tris)
            if ev:
                ev[0].record()
            gathered, works, st.meshes = [], [], []
            for j in range(C):
                mine = torch.empty(sb, dtype=torch.uint8, device=device)
                st.meshes.append(codec.pack(tape, X, Y, Z, batch_size, sparse, (r * C + j, world * C), mine, cap_items, cap_tris))
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
            st.slabs = [gathered[j][0][rr * sb:(rr + 1) * sb] for rr in range(world) for j in range(C)]      # final order
            st.keep = gathered
            st.out_cap = max(total_hint + total_hint // 8 + 4096, 1) if total_hint else world * C * cap_tris
            st.out = torch.empty(st.out_cap * 9, dtype=torch.float64, device=device)
            codec.expand(st.slabs, None, cap_items, cap_tris, st.out, st.out_cap)
            if ev:
                ev[3].record()
    finally:
        if adopted:
            eng.set_stream(0)
    return st


def collect_sharded(st):
    """finish a step: the ONE host synchronisation (the gathered headers), the verdict on the capacities -- every rank
    sees the same headers, so every rank repeats an undersized step with the same larger slabs -- and the result:
    (soup: flat float64 torch tensor of 9*T values in reference order, merged stats dict)"""
    import contextlib
    import torch
    on_gpu = st.device.type == 'cuda'
    cap_items, cap_tris, _ = st.caps
    with (torch.cuda.stream(st.stream) if on_gpu else contextlib.nullcontext()):
        heads = torch.stack([sl[:8 * HEADER_WORDS] for sl in st.slabs]).cpu().numpy().view(np.int64).reshape(len(st.slabs), HEADER_WORDS)
    for m in st.meshes:
        if m is not None:
            m.close()
    st.meshes = []
    total = int(heads[:, H_TRIS].sum())
    need_items, need_tris = int(heads[:, H_ITEMS].max()), int(heads[:, H_TRIS].max())
    if (heads[:, H_OVERFLOW] & 2).any():
        raise RuntimeError('sdf_amd.dist: a rank reported a look-back timeout')
    ok = not heads[:, H_OVERFLOW].any() and need_items <= cap_items and need_tris <= cap_tris and total <= st.out_cap
    if not ok:
        if st.attempt >= 5:
            raise RuntimeError('sdf_amd.dist: slab capacities did not converge')
        X, Y, Z, batch_size, sparse, chunks = st.args
        caps = (max(cap_items, need_items + need_items // 8 + 16), max(cap_tris, need_tris + need_tris // 8 + 1024), total)
        if on_gpu:
            st.outer.wait_stream(st.stream)
        return collect_sharded(submit_sharded(st.eng, st.tape, X, Y, Z, batch_size, sparse, st.device, st.group, chunks, st.lane,
                                              _caps=caps, _attempt=st.attempt + 1))
    # capacities for the next call of this job: what this one needed, with some slack
    _hints_for(st.tape)[st.key] = (need_items + need_items // 8 + 16, need_tris + need_tris // 8 + 1024, total)
    soup = st.out[:total * 9]
    if on_gpu:
        st.outer.wait_stream(st.stream)           # whatever the caller enqueues next sees the soup
        soup.record_stream(st.outer)              # (allocated on the step's stream, used on the caller's)
    world, C, nb = st.world, st.C, st.nb
    per_rank = heads[:, H_TRIS].reshape(world, C).sum(axis=1)
    merged = {
        'batches': nb, 'n_batches': nb,
        'skipped': nb - int(heads[0, H_WORK]), 'n_skipped': nb - int(heads[0, H_WORK]),
        'empty': int(heads[:, H_EMPTY].sum()), 'nonempty': int(heads[:, H_NONEMPTY].sum()),
        'n_eval_voxels': int(heads[:, H_EVAL].sum()), 'n_ambiguous_cells': int(heads[:, H_AMBIGUOUS].sum()),
        'n_sampled_voxels': int(heads[:, H_SAMPLED].sum()), 'n_pruned_instrs': int(heads[:, H_PRUNED].sum()),
        'triangles': total, 'n_triangles': total, 'per_rank_triangles': [int(c) for c in per_rank],
        'n_grid_voxels': len(st.args[0]) * len(st.args[1]) * len(st.args[2]), 'n_retries': st.attempt, 'chunks': C,
        'slab_bytes': st.sb, 'payload': 'f32 local + per-batch transform' if isinstance(st.codec, DeviceCodec) else 'f64 soup',
    }
    merged['n_empty'], merged['n_nonempty'] = merged['empty'], merged['nonempty']
    if st.events:      # (the headers' copy has synchronised the step's stream: the events are complete)
        ev = st.events
        merged['ms_mesh'] = ev[0].elapsed_time(ev[1])
        merged['ms_exchange'] = ev[1].elapsed_time(ev[2])
        merged['ms_expand'] = ev[2].elapsed_time(ev[3])
    else:
        merged['ms_mesh'] = merged['ms_exchange'] = merged['ms_expand'] = 0.0
    st.slabs = st.keep = None
    return soup, merged


def generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None, chunks=None):
    """every rank returns (soup: flat float64 torch tensor of 9*T values on `device`, in
    reference order; merged stats dict).  On GPUs the soup never leaves the device."""
    return collect_sharded(submit_sharded(eng, tape, X, Y, Z, batch_size, sparse, device, group, chunks))


def generate_sharded(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None):
    """every rank returns (points (3T,3) float64 ndarray, merged stats dict)"""
    soup, merged = generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse, device, group)
    return soup.cpu().numpy().reshape(-1, 3), merged
