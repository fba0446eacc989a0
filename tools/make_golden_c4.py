#!/opt/conda/bin/python3.9
"""BASELINE config 4 (weave at samples=2**33: 4097 x 4097 x 512 samples, 266 256 batches) by the UNMODIFIED reference
functions, at full size (build container only: needs /root/reference and the conda interpreter with scikit-image 0.18.3):

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden_c4.py [processes]

`sdf.core.generate` itself cannot be run at this size: it keeps the soup as a Python list of 1.6 x 10^8 three-element
ndarrays (reference sdf/core.py:131-139: > 20 GB) and its thread pool runs the NumPy closures of 36 leaf evaluations per
point mostly under the GIL (~10 h).  What is run here instead is the reference's own per-batch function -- `sdf.core._worker`
(reference sdf/core.py:45-60: `_skip`, `_cartesian_product`, the model's closure, `_marching_cubes` = skimage 0.18.3,
`points * scale + offset`), imported, not restated -- on exactly the jobs `generate` builds (reference sdf/core.py:103-117:
`np.arange` axes on the bounds `_estimate_bounds` returned, 33-sample overlapping slices, `itertools.product(Xs, Ys, Zs)`),
in PROCESSES, and the results are reduced in batch order the way `generate`'s loop consumes them (sdf/core.py:131-139):

    kinds        0 skipped / 1 empty / 2 nonempty per batch, batch order           (uint8[266256])
    tris         triangles per batch                                               (uint16[266256])
    ntri         their sum
    sha256       of the float64 soup in `generate`'s order (for the record: the device's libm differs in the last bits)
    sample_tris  every STRIDE-th triangle of the soup (STRIDE = 9973, a prime: the sample walks through all batches)

-> tests/golden/full_c4_weave_s33.npz.  The driver lines above are the only part of `generate` that is restated; that they
build the same jobs is checked against `generate` itself at 2**19 (tests/golden/gen_weave_s19.npz: same kinds, same hash)
before the long run starts.
"""
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference: sdf, core, fixtures)

core = mg.core
STRIDE = 9973
CHUNK = 64          # batches per task handed to a process

_f = None
_jobs = None


def jobs_of(bounds, samples, batch_size=32):
    """the batches `generate` builds (reference sdf/core.py:92-117)"""
    (x0, y0, z0), (x1, y1, z1) = bounds
    step = ((x1 - x0) * (y1 - y0) * (z1 - z0) / samples) ** (1 / 3)
    X = np.arange(x0, x1, step)
    Y = np.arange(y0, y1, step)
    Z = np.arange(z0, z1, step)
    s = batch_size
    Xs = [X[i:i + s + 1] for i in range(0, len(X), s)]
    Ys = [Y[i:i + s + 1] for i in range(0, len(Y), s)]
    Zs = [Z[i:i + s + 1] for i in range(0, len(Z), s)]
    return step, (len(X), len(Y), len(Z)), Xs, Ys, Zs


def _init(bounds, samples):
    global _f, _jobs
    _f = mg.fixtures.build('ex_weave', mg.NS)
    step, _, Xs, Ys, Zs = jobs_of(bounds, samples)
    _jobs = (Xs, Ys, Zs, step)


def _chunk(c):
    """batches [c * CHUNK, (c + 1) * CHUNK) through the reference's `_worker`"""
    Xs, Ys, Zs, step = _jobs
    ny, nz = len(Ys), len(Zs)
    n = len(Xs) * ny * nz
    out = []
    with np.errstate(all='ignore'):
        for b in range(c * CHUNK, min(n, (c + 1) * CHUNK)):
            ix, r = divmod(b, ny * nz)
            iy, iz = divmod(r, nz)
            res = core._worker(_f, (Xs[ix], Ys[iy], Zs[iz]), (step, step, step), True)
            if res is None:
                out.append((0, None))
            elif len(res) == 0:
                out.append((1, None))
            else:
                out.append((2, np.ascontiguousarray(res, dtype=np.float64)))
    return c, out


def reduce_run(bounds, samples, procs, progress=True):
    step, shape, Xs, Ys, Zs = jobs_of(bounds, samples)
    n = len(Xs) * len(Ys) * len(Zs)
    kinds = np.zeros(n, np.uint8)
    tris = np.zeros(n, np.uint32)
    h = hashlib.sha256()
    sample = []
    t_seen = 0
    t0 = time.time()
    nchunks = (n + CHUNK - 1) // CHUNK
    with mp.Pool(procs, initializer=_init, initargs=(bounds, samples)) as pool:
        for c, out in pool.imap(_chunk, range(nchunks), chunksize=1):
            for k, (kind, pts) in enumerate(out):
                b = c * CHUNK + k
                kinds[b] = kind
                if pts is not None:
                    t = pts.reshape(-1, 3, 3)
                    tris[b] = len(t)
                    h.update(pts.tobytes())
                    first = (-t_seen) % STRIDE          # soup index t_seen + first is the next multiple of STRIDE
                    if first < len(t):
                        sample.append(t[first::STRIDE].copy())
                    t_seen += len(t)
            if progress and (c % 64 == 0 or c == nchunks - 1):
                el = time.time() - t0
                print('  %6d / %d chunks, %9d triangles, %.0f s (eta %.0f s)' % (c + 1, nchunks, t_seen, el, el / (c + 1) * (nchunks - c - 1)),
                      flush=True)
    return {
        'bounds': np.array(bounds, np.float64), 'step': np.array(step, np.float64), 'shape': np.array(shape),
        'kinds': kinds, 'tris': tris.astype(np.uint16) if tris.max() < 65536 else tris, 'ntri': np.array(t_seen),
        'sha256': np.frombuffer(h.digest(), np.uint8), 'sample_stride': np.array(STRIDE),
        'sample_tris': np.concatenate(sample) if sample else np.zeros((0, 3, 3)),
        'seconds': np.array(time.time() - t0), 'processes': np.array(procs),
    }


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 2)
    b = np.load(os.path.join(mg.OUT, 'bounds.npz'))['ex_weave']
    bounds = (tuple(float(x) for x in b[0]), tuple(float(x) for x in b[1]))
    # the restated driver lines against `generate` itself (2**19: seconds)
    g = np.load(os.path.join(mg.OUT, 'gen_weave_s19.npz'))
    gb = (tuple(float(x) for x in g['bounds'][0]), tuple(float(x) for x in g['bounds'][1]))
    small = reduce_run(gb, 2 ** 19, procs, progress=False)
    assert np.array_equal(small['kinds'], g['kinds']) and int(small['ntri']) == int(g['ntri'])
    assert small['sha256'].tobytes() == g['sha256'].tobytes(), 'the restated driver lines do not reproduce generate()'
    print('driver lines == generate() at 2**19 (%d triangles, same sha256)' % int(small['ntri']), flush=True)
    rec = reduce_run(bounds, 2 ** 33, procs)
    rec['fixture'] = np.array('ex_weave')
    rec['samples'] = np.array(2 ** 33)
    np.savez_compressed(os.path.join(mg.OUT, 'full_c4_weave_s33.npz'), **rec)
    k = rec['kinds']
    print('c4_weave_s33: %d batches (s/e/n %d/%d/%d), %d triangles, sha256 %s, %.0f s with %d processes' % (
        len(k), (k == 0).sum(), (k == 1).sum(), (k == 2).sum(), int(rec['ntri']), bytes(rec['sha256']).hex(), float(rec['seconds']), procs),
        flush=True)


if __name__ == '__main__':
    main()
