#!/usr/bin/env python3
"""The float32 mode's distance from the float64 (= reference) soup at the sizes of BASELINE configs 2, 3 and 5, and at the
small sizes tests/test_gpu.py::test_float32_envelope asserts on (GPU box):
    python tools/f32envelope.py [model:log2samples ...]        -> one JSON object per job
bench.py prints the same object for its headline job as the SECONDARY field `f32_envelope`."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.set_device(0)      # (before the engine: the order the tests and bench.py use)
import bench
from sdf_amd import core, engine

eng = engine.get_engine(0)
jobs = sys.argv[1:] or ['example:22', 'blobby:21', 'gearlike:21', 'example:27', 'gearlike:30', 'blobby:30']
for job in jobs:
    model, k = job.split(':')
    f, _ = bench.build_model(model)
    bounds = bench.EXAMPLE_BOUNDS if model == 'example' else core._estimate_bounds(f)
    X, Y, Z, _ = core.grid_axes(bounds, samples=2 ** int(k))
    eng.trim()
    r = bench.f32_envelope(eng, eng.tape_for(f), X, Y, Z, calls=10)
    r.pop('what')
    print(json.dumps({'job': job, 'grid': [len(X), len(Y), len(Z)], **r}), flush=True)
