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

On GPUs the payload is marching cubes' own output in the batch's local voxel coordinates as
16-byte records (the three along-edge float32 bit for bit + one word for the cell and the edges;
a triangle with a vertex inside a cell travels raw: csrc/sdf_slab.h, restated in slabcodec.py)
instead of the 72 bytes of the float64 soup, plus a 56-byte record per batch (triangle prefix +
`points * scale + offset` transform, reference sdf/core.py:58-60); `k_mesh` writes that form
directly into the slab and `k_expand` produces the ordered float64 soup from the gathered slabs
on every rank (csrc/sdf_plain.hip).  Engines that return host soups
(models with user closures, the CPU stand-in of tests/test_dist.py) ship the float64 soup itself
through the same protocol.

With `chunks` > 1 a rank cuts its share into that many consecutive shards and meshes shard k + 1
while the all-gather of shard k is in flight (the collective runs on RCCL's own stream).
"""
import os
import weakref

import numpy as np

HEADER_WORDS = 16          # int64 words at the head of a slab (include/sdf_hip.h)
MAX_SLABS = 64             # slabs one sdf_expand_slabs call takes
H_TRIS, H_ITEMS, H_OVERFLOW, H_EMPTY, H_NONEMPTY, H_EVAL, H_AMBIGUOUS, H_SAMPLED, H_PRUNED, H_WORK, H_RAW, H_NEED_TRIS = range(12)

_HINTS = weakref.WeakKeyDictionary()      # tape object -> {job key: (cap_items, cap_tris, total_tris)}
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
    """capacity hints of the jobs run with this tape object; they live exactly as long as the object (a model that
    cannot be weakly referenced gets none: ids are recycled, a table keyed by id() would hand a new model the
    capacities of a dead one)"""
    try:
        return _HINTS.setdefault(tape, {})
    except TypeError:
        return {}


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
    """slabs written and expanded by the HIP library (16-byte triangle records, see the module docstring)"""

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
    # (an SDF / Tape with closures handed over directly has to be lowered before its externs can be seen)
    lowered = eng.tape_for(tape) if hasattr(eng, 'tape_for') else tape
    externs = getattr(getattr(lowered, 'tape', None), 'externs', None)
    if chunks is None:
        chunks = int(os.environ.get('SDF_DIST_CHUNKS', '1'))
    C = max(1, min(int(chunks), MAX_SLABS // max(world, 1)))
    # sdf_expand_slabs takes at most MAX_SLABS slabs per call: a larger world ships the float64 soup through host memory
    # (batch_size > 32 goes through device memory in chunks, csrc generate_big: its soup comes back like a closure model's)
    on_device = device.type == 'cuda' and hasattr(eng, 'generate_compact') and not externs and world * C <= MAX_SLABS and int(batch_size) <= 32
    codec = DeviceCodec(eng) if on_device else HostCodec(eng)
    s = int(batch_size)
    nb = (-(-len(X) // s)) * (-(-len(Y) // s)) * (-(-len(Z) // s))
    key = (len(X), len(Y), len(Z), s, bool(sparse), world, C, type(codec).__name__)
    return device, world, C, nb, codec, key


def _all_gather(td, g, mine, group, world, sb, async_op):
    """ONE collective: the equal-sized slabs of all ranks into g, in rank order.  (RCCL and gloo on host tensors take
    the single-buffer form; a backend that refuses it for these tensors -- gloo with device tensors in some builds --
    gets the list form over views of the same buffer.)"""
    try:
        return td.all_gather_into_tensor(g, mine, group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):
        return td.all_gather([g[r * sb:(r + 1) * sb] for r in range(world)], mine, group=group, async_op=async_op)


# ---- the native path: RCCL called from inside the library (csrc/sdf_comm.inc) ----
_COMMS = {}      # (engine id, group id) -> engine.Comm
_NATIVE_BROKEN = []   # why the native path was given up in this process (empty: it was not)


class NativeStep:
    """one exchange step in flight inside the library (engine.Exchange); same life cycle as ShardedStep"""
    __slots__ = ('comm', 'xch', 'lane', 'device', 'result')


def _native_comm(eng, td, group):
    """the library's communicator for this process group (created once, collectively: rank 0 draws the RCCL ids,
    the group that launched the ranks carries them to the others)"""
    key = (id(eng), id(group) if group is not None else 0)
    comm = _COMMS.get(key)
    if comm is None:
        from . import engine as _engine
        r, world = td.get_rank(group), td.get_world_size(group)
        # every rank says whether librccl loads HERE before anything collective inside the library is started: a rank
        # that raised on its own would leave the others in a broadcast (or, later, in ncclCommInitRank) for ever.
        # All ranks see the same list and take the same branch.
        ok, why = _engine.Comm.available(eng.lib)
        flags = [None] * world
        td.all_gather_object(flags, (bool(ok), why), group=group)
        bad = ['rank %d: %s' % (i, f[1]) for i, f in enumerate(flags) if not f[0]]
        if bad:
            raise RuntimeError('; '.join(bad))
        ids = [None]
        if r == 0:
            try:
                ids[0] = _engine.Comm.unique_ids(eng.lib, 2)
            except Exception as e:           # (the others must hear about it, not wait for it)
                ids[0] = e
        src = td.get_global_rank(group, 0) if group is not None else 0
        td.broadcast_object_list(ids, src=src, group=group)
        if isinstance(ids[0], Exception):
            raise RuntimeError('rank 0 could not draw the communicator ids: %r' % (ids[0],))
        # The creation is collective inside the library (ncclCommInitRank).  A rank whose creation FAILED (a stream, an
        # event, pinned memory, the RCCL call itself) must not go on alone on the torch path while the others use the
        # native one: the ranks compare notes and, if any of them failed, all close what they have and fall back together.
        comm, err = None, None
        try:
            comm = _engine.Comm(eng, ids[0], r, world)
        except Exception as e:
            err = e
        made = [None] * world
        td.all_gather_object(made, (err is None, repr(err)), group=group)
        bad = ['rank %d: %s' % (i, f[1]) for i, f in enumerate(made) if not f[0]]
        if bad:
            if comm is not None:
                comm.close()
            raise RuntimeError('communicator not created on every rank: ' + '; '.join(bad))
        _COMMS[key] = comm
        comm._lane_owner = {}
    return comm


def shutdown_native():
    """destroy the library's communicators of this process (collective, like their creation)"""
    for comm in list(_COMMS.values()):
        comm.close()
    _COMMS.clear()


def _native_ok(eng, td, tape, device, group):
    mode = os.environ.get('SDF_DIST_NATIVE', '1')
    if mode == '0' or device.type != 'cuda' or not hasattr(eng, 'lib'):
        return False
    # (SDF_DIST_NATIVE=force: the library's exchange under ANY torch backend -- torch then only carries the communicator
    # ids; what the tests use to run it between processes that share one GPU, with a stand-in for librccl)
    if (td.get_backend(group) != 'nccl' and mode != 'force') or td.get_world_size(group) > MAX_SLABS:
        return False
    lowered = eng.tape_for(tape)
    return not getattr(lowered.tape, 'externs', None)


def _native_finish(step):
    import torch
    if step.result is None:
        soup, st = step.xch.wait()
        t = torch.as_tensor(soup, device=step.device) if soup.n_triangles else torch.empty(0, dtype=torch.float64, device=step.device)
        step.result = (t, st, soup)
        if step.comm._lane_owner.get(step.lane) is step:
            del step.comm._lane_owner[step.lane]
    return step.result


def submit_sharded(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None, chunks=None, lane=0, _caps=None, _attempt=0):
    """enqueue one exchange step -- mesh this rank's shard(s) into slabs, all-gather, expand -- and return without
    waiting for any of it; `collect_sharded` finishes the step.  Steps submitted on different `lane`s (0 / 1) run on
    streams of their own, so step i + 1's meshing overlaps step i's collective.

    On GPUs under the "nccl" backend the step runs INSIDE the library (csrc/sdf_comm.inc: ncclAllGather called from
    there, persistent buffers, no interpreter between submit and collect; SDF_DIST_NATIVE=0 keeps the torch.distributed
    path below, which is also what every other backend / engine uses).  The soup of a native step lives in library
    memory and stays valid until the next step is submitted on the same lane."""
    import contextlib
    import torch
    td = _dist()
    if td is None:
        raise RuntimeError('torch.distributed is not initialised')
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if td.get_backend(group) == 'nccl' else torch.device('cpu')
    comm = None
    # (batch_size > 32 is not part of the native exchange -- sdf_generate_compact refuses it: the torch.distributed path below
    # ships those soups as float64 through the host codec; a property of the call, the same on every rank)
    if int(batch_size) <= 32 and _native_ok(eng, td, tape, device, group) and not _NATIVE_BROKEN:
        try:
            comm = _native_comm(eng, td, group)
        except Exception as e:       # (librccl not loadable, communicator refused: every rank fails alike and takes the torch path)
            import sys
            sys.stderr.write('sdf_amd.dist: native RCCL exchange unavailable (%s); using torch.distributed\n' % (e,))
            _NATIVE_BROKEN.append(repr(e))
    if comm is not None:
        if chunks is None:
            chunks = int(os.environ.get('SDF_DIST_CHUNKS', '1'))
        held = comm._lane_owner.get(lane)
        if held is not None:                 # the lane's buffers are about to be reused: finish that step, keep its soup
            t, st, _ = _native_finish(held)
            held.result = (t.clone(), st, None)
            # (the copy is enqueued on torch's current stream, the next step rewrites the lane's soup on the lane's own
            # stream: nothing else orders the two)
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
        step = NativeStep()
        step.comm, step.lane, step.device, step.result = comm, lane, device, None
        step.xch = comm.submit(tape, X, Y, Z, batch_size, sparse, chunks=chunks, lane=lane)
        comm._lane_owner[lane] = step
        return step
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
        # (triangles: a guess, 4096 per batch of the shard -- 2.4 x what the surviving batches of the BASELINE models
        # produce, and most batches do not survive -- within 8 GB for gathered slabs plus the expanded soup (72 B)
        # together (a triangle of a slab is a 16-byte record plus 1/128 of a 36-byte raw entry: 89 B with the soup's 72); a slab that is too small is flagged in its header, the headers carry the exact need, and the step is
        # repeated once with that.  The capacities size the collective: nothing rank-local -- free memory, say -- may
        # enter the formula, every rank must arrive at the same numbers)
        budget = 8 << 30
        cap_tris = max(min(4096 * cap_items, budget // (89 * world * C)), 1 << 16)
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
                works.append(_all_gather(td, g, mine, group, world, sb, C > 1))
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
    if isinstance(st, NativeStep):
        soup, merged, _ = _native_finish(st)
        return soup, merged
    on_gpu = st.device.type == 'cuda'
    cap_items, cap_tris, _ = st.caps
    with (torch.cuda.stream(st.stream) if on_gpu else contextlib.nullcontext()):
        heads = torch.stack([sl[:8 * HEADER_WORDS] for sl in st.slabs]).cpu().numpy().view(np.int64).reshape(len(st.slabs), HEADER_WORDS)
    for m in st.meshes:
        if m is not None:
            m.close()
    st.meshes = []
    total = int(heads[:, H_TRIS].sum())
    # (device slabs: a raw area that was too small asks for the triangle capacity that comes with a larger one, csrc/sdf_slab.h)
    need_items, need_tris = int(heads[:, H_ITEMS].max()), int(max(heads[:, H_TRIS].max(), heads[:, H_NEED_TRIS].max()))
    if (heads[:, H_OVERFLOW] & 2).any():
        raise RuntimeError('sdf_amd.dist: a rank reported a look-back timeout')
    ok = not heads[:, H_OVERFLOW].any() and need_items <= cap_items and need_tris <= cap_tris and total <= st.out_cap
    if not ok:
        if st.attempt >= 5:
            raise RuntimeError('sdf_amd.dist: slab capacities did not converge')
        X, Y, Z, batch_size, sparse, chunks = st.args
        caps = (max(cap_items, need_items + need_items // 8 + 16), max(cap_tris, need_tris + need_tris // 64 + 1024), total)
        if on_gpu:
            st.outer.wait_stream(st.stream)
        return collect_sharded(submit_sharded(st.eng, st.tape, X, Y, Z, batch_size, sparse, st.device, st.group, chunks, st.lane,
                                              _caps=caps, _attempt=st.attempt + 1))
    # capacities for the next call of this job: what this one needed, with some slack
    _hints_for(st.tape)[st.key] = (need_items + need_items // 8 + 16, need_tris + need_tris // 64 + 1024, total)
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
        'slab_bytes': st.sb, 'payload': '16-byte triangle records (local coordinates) + per-batch transform' if isinstance(st.codec, DeviceCodec) else 'f64 soup',
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
