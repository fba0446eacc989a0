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

RCCL has no all-gather-v: counts are all-gathered first, buffers are padded to the
largest count and exchanged with one `all_gather_into_tensor`, then compacted.
"""
import os

import numpy as np


_PAD_HINT = {}     # (tape, grid, shard) -> padded triangle count of the previous exchange
_UNEVEN_BROKEN = []   # non-empty once an all_gather of unequal sizes was refused: the padded exchange is used from then on


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
    csrc/sdf_hip.cpp `shard_range`)"""
    return (n_work * r) // world, (n_work * (r + 1)) // world


def _local_tensor(mesh, t_pad, device):
    import torch
    buf = torch.zeros(t_pad * 9, dtype=torch.float64, device=device)
    t = mesh.n_triangles
    if t:
        if device.type == 'cuda':
            mesh.emit_device(buf.data_ptr())     # ordered gather kernel writes straight into it
        else:
            buf[:t * 9] = torch.from_numpy(np.ascontiguousarray(mesh.points()).reshape(-1))
    return buf


def generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None):
    """every rank returns (soup: flat float64 torch tensor of 9*T values on `device`, in
    reference order; merged stats dict).  The soup never leaves the device."""
    import torch
    td = _dist()
    if td is None:
        raise RuntimeError('torch.distributed is not initialised')
    world, r = td.get_world_size(group), td.get_rank(group)
    if device is None:
        backend = td.get_backend(group)
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')

    # torch allocations / fills and the collectives run on torch's current stream: the engine
    # adopts it for this call so its kernels are ordered with them (its own stream otherwise races
    # with e.g. the zero-fill of the padded exchange buffer)
    adopted = False
    if device.type == 'cuda' and hasattr(eng, 'set_stream'):
        eng.set_stream(torch.cuda.current_stream(device).cuda_stream)
        adopted = True
    # On a GPU the shard's soup is written by the meshing kernel straight into the exchange buffer
    # (sized from the previous call; no zero fill, no device-to-device copy).
    local = None
    key = (id(tape), len(X), len(Y), len(Z), batch_size, bool(sparse), r, world)
    if device.type == 'cuda' and hasattr(eng, 'lib'):
        cap = _PAD_HINT.get(key, 0)
        cap = cap + cap // 8 + 4096 if cap else 1 << 20
        local = torch.empty(cap * 9, dtype=torch.float64, device=device)
        mesh = eng.generate(tape, X, Y, Z, batch_size, sparse, shard=(r, world), out_ptr=local.data_ptr(), out_cap=cap)
    else:
        mesh = eng.generate(tape, X, Y, Z, batch_size, sparse, shard=(r, world))
    try:
        st = mesh.stats()
        t_local = mesh.n_triangles

        # 1) counts (and the additive statistics) in one small all-gather
        mine = torch.tensor([t_local, st['empty'], st['nonempty'], st['n_eval_voxels'],
                             st['n_ambiguous_cells']], dtype=torch.int64, device=device)
        allc = torch.empty(world * mine.numel(), dtype=torch.int64, device=device)
        td.all_gather_into_tensor(allc, mine, group=group)
        allc = allc.view(world, -1).cpu().numpy()
        counts = allc[:, 0]
        t_pad = int(counts.max())

        # 2) the exchange step: one padded all-gather of the triangle buffers, then compaction.
        # SDF_DIST_UNEVEN=1 (opt-in: not measurable on the one-GPU boxes this was built on) lets RCCL gather
        # the unequal shards straight into views of the final soup instead (torch runs that as grouped
        # broadcasts): no padding, no compaction copy.
        _PAD_HINT[key] = t_pad
        if t_pad:
            soup = None
            if (device.type == 'cuda' and td.get_backend(group) == 'nccl' and local is not None
                    and getattr(mesh, 'emitted', False) and int(counts.min()) > 0 and not _UNEVEN_BROKEN
                    and os.environ.get('SDF_DIST_UNEVEN') == '1'):
                try:
                    total = int(counts.sum())
                    soup = torch.empty(total * 9, dtype=torch.float64, device=device)
                    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64) * 9
                    outs = [soup[int(offs[i]):int(offs[i + 1])] for i in range(world)]
                    td.all_gather(outs, local[:t_local * 9], group=group)
                except Exception:              # (raised by every rank alike: sizes are checked before anything is sent)
                    _UNEVEN_BROKEN.append(True)
                    soup = None
            if soup is None:
                if local is not None and getattr(mesh, 'emitted', False) and local.numel() >= t_pad * 9:
                    local = local[:t_pad * 9]          # (the tail beyond this rank's count is padding)
                else:
                    local = _local_tensor(mesh, t_pad, device)
                gathered = torch.empty(world * t_pad * 9, dtype=torch.float64, device=device)
                td.all_gather_into_tensor(gathered, local, group=group)
                gathered = gathered.view(world, t_pad * 9)
                if all(int(c) == t_pad for c in counts):
                    soup = gathered.reshape(-1)
                else:
                    parts = [gathered[i, :int(counts[i]) * 9] for i in range(world) if counts[i]]
                    soup = torch.cat(parts) if parts else gathered[0, :0]
        else:
            soup = torch.empty(0, dtype=torch.float64, device=device)
    finally:
        mesh.close()
        if adopted:
            eng.set_stream(0)

    merged = dict(st)
    merged['empty'] = merged['n_empty'] = int(allc[:, 1].sum())
    merged['nonempty'] = merged['n_nonempty'] = int(allc[:, 2].sum())
    merged['n_eval_voxels'] = int(allc[:, 3].sum())
    merged['n_ambiguous_cells'] = int(allc[:, 4].sum())
    merged['triangles'] = merged['n_triangles'] = int(counts.sum())
    merged['per_rank_triangles'] = [int(c) for c in counts]
    return soup, merged


def generate_sharded(eng, tape, X, Y, Z, batch_size, sparse, device=None, group=None):
    """every rank returns (points (3T,3) float64 ndarray, merged stats dict)"""
    soup, merged = generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse, device, group)
    return soup.cpu().numpy().reshape(-1, 3), merged
