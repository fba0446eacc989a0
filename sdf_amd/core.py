"""Sampling + meshing driver: the drop-in for reference sdf/core.py.

Same entry points and keyword arguments as the reference (``generate`` sdf/core.py:84-150,
``save`` :152-158, ``sample_slice`` :202-232, ``show_slice`` :234-244), but the batch
loop, the sparse skip test, SDF sampling and marching cubes all run in HIP kernels behind
the C ABI of include/sdf_hip.h (see sdf_amd/engine.py).  What stays on the host is what
the reference also does once per call in scalar Python: bounds -> step -> ``np.arange``
axes (kept in NumPy float64 so the grid is bit-identical to the reference's).

Differences a caller can observe:
* ``generate`` returns ONE float64 ndarray of shape (3*T, 3) instead of a Python list of
  3*T tiny arrays (same ``len``, indexing, ``write_binary_stl`` and ``np.unique``
  behaviour).
* ``workers`` does not drive the sampling (the device runs every batch concurrently); it is the
  number of host threads that turn the device's 16-byte triangle records into the float64 rows
  of the returned array (``sdf_mesh_emit_host_workers``; at most 32 -- more were measured to be no faster).
* when ``torch.distributed`` is initialised with world_size > 1 the surviving batches are
  sharded over the ranks and the triangle buffers all-gathered (sdf_amd/dist.py), so
  every rank still returns the complete soup in reference order.
"""
import multiprocessing
import time

import numpy as np

from . import stl

WORKERS = multiprocessing.cpu_count()
SAMPLES = 2 ** 22
BATCH_SIZE = 32


def _marching_cubes(volume, level=0):
    """skimage's Lewiner marching cubes of a host volume, as the triangle soup (3T, 3) float32 in index
    coordinates (reference sdf/core.py:16-18: `verts[faces].reshape((-1, 3))`), on the device
    (`sdf_marching_cubes_host`: k_mc_rows / k_scan_rows / k_mc_emit).  Raises what skimage raises: ValueError
    for a volume that is not 3-D, smaller than 2 x 2 x 2 or whose range does not contain `level`, RuntimeError
    when no surface is found -- `_worker` turns any of them into an empty batch, like the reference's.
    The reference only ever passes level 0; another level is subtracted in float64 on the float32-cast
    volume (skimage's own order) and the result cast back to float32 for the device."""
    from . import engine
    vol = np.asarray(volume)
    if vol.ndim != 3:
        raise ValueError('Input volume should be a 3D numpy array.')
    if min(vol.shape) < 2:
        raise ValueError('Input array must be at least 2x2x2.')
    level = float(level)
    if level < vol.min() or level > vol.max():
        raise ValueError('Surface level must be within volume data range.')
    v32 = vol.astype(np.float32)
    if level != 0.0:
        v32 = (v32.astype(np.float64) - level).astype(np.float32)
    soup = engine.get_engine().marching_cubes(v32)
    if len(soup) == 0:
        raise RuntimeError('No surface found at the given iso value.')
    return soup


def _skip(sdf, job):
    """the sparse skip test of one batch (reference sdf/core.py:28-43): True when the batch cannot hold
    surface.  The nine probes -- centre and the 8 corners of the batch's box -- are evaluated in one call of
    `sdf` (on the device for the library's own nodes); the comparisons are the reference's.  `generate` itself
    runs the same test for every batch of a grid in one launch (`k_skip`)."""
    X, Y, Z = job
    lo = np.array([X[0], Y[0], Z[0]])
    hi = np.array([X[-1], Y[-1], Z[-1]])
    mid = (lo + hi) / 2
    sel = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=bool)   # itertools.product order
    probes = np.vstack([mid[None, :], np.where(sel, hi, lo)])
    values = np.asarray(sdf(probes)).reshape(-1)
    if abs(values[0]) <= np.linalg.norm(mid - lo):
        return False
    corners = values[1:]
    return bool(np.all(corners > 0) if corners[0] > 0 else np.all(corners < 0))


def _worker(sdf, job, step, sparse):
    """one batch of `generate` (reference sdf/core.py:45-60): None when the skip test drops it, [] when it
    has no surface, else its triangles `points * scale + offset` as a (3T, 3) float64 array.  The job is meshed
    as a grid of its own whose first batch is the job (batch_size = its longest axis - 1; what lies behind it
    are slivers one sample thick, which have no cells): the same fused kernels `generate` uses."""
    from . import engine
    X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in job)
    if min(len(X), len(Y), len(Z)) < 2:
        return None if (sparse and len(X) and len(Y) and len(Z) and _skip(sdf, job)) else []
    bs = max(len(X), len(Y), len(Z)) - 1
    mesh = engine.get_engine().generate(sdf, X, Y, Z, bs, sparse)
    try:
        if sparse and mesh.kinds()[0] == 0:
            return None
        points = mesh.points()
    finally:
        mesh.close()
    return points if len(points) else []


def _cartesian_product(*arrays):
    """(N, d) points, first axis slowest (reference sdf/core.py:20-26); host helper kept
    for API compatibility -- the device generates grid points from the axes itself"""
    grids = np.meshgrid(*arrays, indexing='ij')
    return np.stack([g.reshape(-1) for g in grids], axis=-1)


def _estimate_bounds(sdf):
    """iterative 16^3 shrink from +-1e9 (reference sdf/core.py:62-82), on the device"""
    from . import engine
    eng = engine.get_engine()
    tape = eng.tape_for(sdf)
    if eng.precision == engine.PRECISION_F64:        # the whole loop in one launch (k_estimate_bounds)
        b = eng.estimate_bounds(tape)
        if b is not None:
            return b
    # (float32 sampling, and models with user closures: the reference's loop, the probes evaluated by eval_grid)
    s = 16
    x0 = y0 = z0 = -1e9
    x1 = y1 = z1 = 1e9
    prev = None
    for i in range(32):
        X = np.linspace(x0, x1, s)
        Y = np.linspace(y0, y1, s)
        Z = np.linspace(z0, z1, s)
        d = np.array([X[1] - X[0], Y[1] - Y[0], Z[1] - Z[0]])
        threshold = np.linalg.norm(d) / 2
        if threshold == prev:
            break
        prev = threshold
        volume = eng.eval_grid(tape, X, Y, Z)
        where = np.argwhere(np.abs(volume) <= threshold)
        x1, y1, z1 = (x0, y0, z0) + where.max(axis=0) * d + d / 2
        x0, y0, z0 = (x0, y0, z0) + where.min(axis=0) * d - d / 2
    return ((x0, y0, z0), (x1, y1, z1))


def grid_axes(bounds, step=None, samples=SAMPLES):
    """bounds/step/samples -> (X, Y, Z, (dx, dy, dz)) exactly as reference sdf/core.py:94-112"""
    (x0, y0, z0), (x1, y1, z1) = bounds
    if step is None and samples is not None:
        volume = (x1 - x0) * (y1 - y0) * (z1 - z0)
        step = (volume / samples) ** (1 / 3)
    try:
        dx, dy, dz = step
    except TypeError:
        dx = dy = dz = step
    X = np.arange(x0, x1, dx)
    Y = np.arange(y0, y1, dy)
    Z = np.arange(z0, z1, dz)
    return X, Y, Z, (dx, dy, dz)


def generate(
        sdf,
        step=None, bounds=None, samples=SAMPLES,
        workers=WORKERS, batch_size=BATCH_SIZE,
        verbose=True, sparse=True, _stl=False, _weld=False):
    """reference sdf/core.py:84-150.  `batch_size` up to 512 (the reference takes any: a larger one is refused with a message; up
    to 32 runs the fused kernels, above that the batches go through device memory -- a model with user closures then hands its
    callback one whole tile at a time, (batch_size + 1)^3 points: 4.3 GB of pinned host memory at 512).  (`_stl=True` is what `save` uses for .stl files: the soup
    stays on the device and the 50-byte STL records come back instead of the points; `_weld=True` is
    what `save` uses for every other format: the soup is welded on the device and the indexed mesh
    (unique points, cells) comes back.)"""

    from . import engine, dist
    start = time.time()
    eng = engine.get_engine()
    tape = eng.tape_for(sdf)

    if bounds is None:
        bounds = _estimate_bounds(tape)      # (the tape lowered above: lowering the model a second time is 0.08 ms of a 2.5 ms call)
    (x0, y0, z0), (x1, y1, z1) = bounds
    X, Y, Z, (dx, dy, dz) = grid_axes(bounds, step, samples)

    if verbose:
        print('min %g, %g, %g' % (x0, y0, z0))
        print('max %g, %g, %g' % (x1, y1, z1))
        print('step %g, %g, %g' % (dx, dy, dz))

    s = batch_size
    num_batches = (-(-len(X) // s)) * (-(-len(Y) // s)) * (-(-len(Z) // s))
    if verbose:
        def overlapped(n):       # samples counted with the 1-sample batch overlap (core.py:121-122)
            return sum(min(s + 1, n - i) for i in range(0, n, s))
        num_samples = overlapped(len(X)) * overlapped(len(Y)) * overlapped(len(Z))
        print('%d samples in %d batches with %d workers' % (num_samples, num_batches, workers))

    records = welded = None
    if dist.world_size() > 1:
        soup, stats = dist.generate_sharded_device(eng, tape, X, Y, Z, batch_size, sparse)
        if (_stl or _weld) and getattr(soup, 'is_cuda', False) and hasattr(eng, 'adopt_soup'):
            # the gathered soup stays on the device: STL records / the weld are made there, as for one GPU
            import torch
            torch.cuda.current_stream(soup.device).synchronize()
            mesh = eng.adopt_soup(soup.data_ptr(), soup.numel() // 9)
            try:
                if _stl:
                    records = mesh.stl_records()
                else:
                    welded = mesh.weld()
                points = np.empty((3 * mesh.n_triangles, 0))     # only its length is used below
            finally:
                mesh.close()
            del soup
        else:
            points = soup.cpu().numpy().reshape(-1, 3)
    else:
        # (the soup is wanted on the host: it travels as 16-byte records and `workers` host threads make the float64
        # rows from them -- the one place the reference's `workers=` still means something here)
        mesh = eng.generate(tape, X, Y, Z, batch_size, sparse, records=not (_stl or _weld))
        try:
            stats = mesh.stats()
            if _stl:
                records = mesh.stl_records()
                points = np.empty((3 * mesh.n_triangles, 0))     # only its length is used below
            elif _weld:
                welded = mesh.weld()
                points = np.empty((3 * mesh.n_triangles, 0))
            else:
                points = mesh.points(min(int(workers), 32) if workers else 0)
        finally:
            mesh.close()

    if verbose:
        print('%d skipped, %d empty, %d nonempty' % (stats['skipped'], stats['empty'], stats['nonempty']))
        triangles = len(points) // 3
        seconds = time.time() - start
        print('%d triangles in %g seconds' % (triangles, seconds))

    generate.last_stats = stats
    if _stl:
        return records if records is not None else stl.stl_records(points).view(np.uint8).reshape(-1)
    if _weld:
        if welded is None:      # (multi-process: the gathered soup is welded like the reference does it)
            pts, cells = np.unique(points, axis=0, return_inverse=True)
            welded = (pts, np.asarray(cells).reshape((-1, 3)))
        return welded
    return points


generate.last_stats = None


def save(path, *args, **kwargs):
    """reference sdf/core.py:152-158"""
    if path.lower().endswith('.stl'):
        # normals and the 50-byte records are made on the device (k_stl); byte-identical to
        # stl.write_binary_stl(path, points) -- tests/test_gpu.py
        records = generate(*args, _stl=True, **kwargs)
        stl.write_stl_records(path, records)
    else:
        # the vertex weld (np.unique over 3T rows in the reference) runs on the device: sdf_mesh_weld
        points, cells = generate(*args, _weld=True, **kwargs)
        import meshio
        meshio.Mesh(points, [('triangle', cells)]).write(path)


def _mesh(points):
    """vertex weld of a host-side soup for non-STL formats (reference sdf/core.py:160-164; needs
    meshio).  `save` does not come through here: it welds on the device (`generate(_weld=True)`)."""
    import meshio
    points, cells = np.unique(points, axis=0, return_inverse=True)
    cells = [('triangle', np.asarray(cells).reshape((-1, 3)))]
    return meshio.Mesh(points, cells)


def sample_slice(
        sdf, w=1024, h=1024,
        x=None, y=None, z=None, bounds=None):
    """a w x h image of the field on an axis-aligned plane (reference sdf/core.py:202-232)"""
    from . import engine
    eng = engine.get_engine()
    tape = eng.tape_for(sdf)

    if bounds is None:
        bounds = _estimate_bounds(sdf)
    (x0, y0, z0), (x1, y1, z1) = bounds

    if x is not None:
        X = np.array([x])
        Y = np.linspace(y0, y1, w)
        Z = np.linspace(z0, z1, h)
        extent = (Z[0], Z[-1], Y[0], Y[-1])
        axes = 'ZY'
    elif y is not None:
        Y = np.array([y])
        X = np.linspace(x0, x1, w)
        Z = np.linspace(z0, z1, h)
        extent = (Z[0], Z[-1], X[0], X[-1])
        axes = 'ZX'
    elif z is not None:
        Z = np.array([z])
        X = np.linspace(x0, x1, w)
        Y = np.linspace(y0, y1, h)
        extent = (Y[0], Y[-1], X[0], X[-1])
        axes = 'YX'
    else:
        raise Exception('x, y, or z position must be specified')

    return eng.eval_grid(tape, X, Y, Z).reshape((w, h)), extent, axes


def show_slice(*args, **kwargs):
    """matplotlib viewer around sample_slice (reference sdf/core.py:234-244)"""
    import matplotlib.pyplot as plt
    show_abs = kwargs.pop('abs', False)
    a, extent, axes = sample_slice(*args, **kwargs)
    if show_abs:
        a = np.abs(a)
    im = plt.imshow(a, extent=extent, origin='lower')
    plt.xlabel(axes[0])
    plt.ylabel(axes[1])
    plt.colorbar(im)
    plt.show()
