#!/opt/conda/bin/python3.9
"""Time the UNMODIFIED reference's CPU path on BASELINE config 2 (example, samples=2**27 -> 512^3):

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/time_reference.py [--out FILE] [--repeat N]

`sdf.core.generate` (reference sdf/core.py:84-150: NumPy over a thread pool + skimage marching cubes),
`workers=1` and `workers=os.cpu_count()`, wall clock around generate() only, best of N.  Prints one JSON
object; bench.py runs this script itself when the reference and this interpreter exist on the box it runs
on (kind "reference"), and otherwise reports the committed output of a run in the build container
(profiles/reference_cpu.json) with that provenance.
"""
import argparse
import hashlib
import json
import os
import platform
import sys
import time

import numpy as np

sys.path.insert(0, '/root/reference')
import sdf  # noqa: E402  (the reference)
from sdf import core  # noqa: E402

# the reference's own _estimate_bounds for examples/example.py (tests/golden/bounds.npz)
BOUNDS = ((-0.8454300600008358, -0.8454300600008358, -0.8454300600008358),
          (0.8454307895539046, 0.8454307895539046, 0.8454307895539046))


def model():
    f = sdf.sphere(1) & sdf.box(1.5)
    c = sdf.cylinder(0.5)
    f -= c.orient(sdf.X) | c.orient(sdf.Y) | c.orient(sdf.Z)
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples-log2', type=int, default=27)
    ap.add_argument('--repeat', type=int, default=1)
    ap.add_argument('--workers', default='1,all')
    ap.add_argument('--out')
    args = ap.parse_args()
    f = model()
    samples = 2 ** args.samples_log2
    (x0, y0, z0), (x1, y1, z1) = BOUNDS
    step = ((x1 - x0) * (y1 - y0) * (z1 - z0) / samples) ** (1 / 3)
    n = [len(np.arange(a, b, step)) for a, b in ((x0, x1), (y0, y1), (z0, z1))]
    voxels = n[0] * n[1] * n[2]
    runs = []
    for w in args.workers.split(','):
        workers = os.cpu_count() if w == 'all' else int(w)
        best, tris, sha = None, 0, ''
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            pts = core.generate(f, step=step, bounds=BOUNDS, workers=workers, verbose=False)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            tris = len(pts) // 3
            if not sha:
                sha = hashlib.sha256(np.array(pts, dtype=np.float64).tobytes()).hexdigest()
            del pts
        runs.append({'workers': workers, 'seconds': round(best, 3), 'voxels_per_sec': round(voxels / best, 1),
                     'triangles_per_sec': round(tris / best, 1), 'triangles': tris, 'soup_sha256': sha})
    out = {'workload': 'example @ samples=2**%d -> %dx%dx%d grid, sparse=True, batch_size=32' % (args.samples_log2, *n),
           'path': 'reference sdf/core.py:84-150 generate() unmodified, numpy %s, scikit-image, python %s'
                   % (np.__version__, platform.python_version()),
           'host_cores': os.cpu_count(), 'grid_voxels': voxels, 'runs': runs}
    text = json.dumps(out)
    print(text)
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(json.dumps(out, indent=1) + '\n')


if __name__ == '__main__':
    main()
