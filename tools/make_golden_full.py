#!/opt/conda/bin/python3.9
"""Reference goldens at the sizes BASELINE.json names (build container only: needs /root/reference
and the conda interpreter with scikit-image 0.18.3):

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden_full.py [tag ...]

Runs the UNMODIFIED reference `generate` (reference sdf/core.py:84-150, workers=1) on the bounds
its own `_estimate_bounds` returned (tests/golden/bounds.npz) and records, per configuration:
bounds, step, the per-batch classification in batch order, the triangle count, the sha256 of the
float64 soup and every `stride`-th triangle of it (for the models whose arithmetic goes through
libm, where the device agrees to a tolerance and not bit for bit).  -> tests/golden/full_<tag>.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)

JOBS = [
    # tag, fixture, samples (BASELINE.json configs; C4 = weave 2**33 at full size: tools/make_golden_c4.py)
    ('c2_example_s27', 'ex_example', 2 ** 27),
    ('c5_blobby_s30', 'ex_blobby', 2 ** 30),
    ('c3_gearlike_s30', 'ex_gearlike', 2 ** 30),
    ('weave_s24', 'ex_weave', 2 ** 24),
    ('knurling_s27', 'ex_knurling', 2 ** 27),
    ('pawn_s27', 'ex_pawn', 2 ** 27),
]
STRIDE = 997       # triangles kept: every STRIDE-th (a prime, so the sample walks through all batches)


def main():
    want = sys.argv[1:]
    bounds = np.load(os.path.join(mg.OUT, 'bounds.npz'))
    for tag, name, samples in JOBS:
        if want and tag not in want:
            continue
        b = bounds[name]
        bnd = (tuple(float(x) for x in b[0]), tuple(float(x) for x in b[1]))
        rec = mg.run_generate(name, True, bounds=bnd, samples=samples)
        pts = rec.pop('points').reshape(-1, 3, 3)
        rec['sample_stride'] = np.array(STRIDE)
        rec['sample_tris'] = pts[::STRIDE].copy()
        rec['fixture'] = np.array(name)
        rec['samples'] = np.array(samples)
        np.savez_compressed(os.path.join(mg.OUT, 'full_%s.npz' % tag), **rec)
        print(tag, 'sha256', bytes(rec['sha256']).hex(), flush=True)


if __name__ == '__main__':
    main()
