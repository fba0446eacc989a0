#!/usr/bin/env python3
"""end-to-end wall time of the drop-in entry point f.save('out.stl') (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdf import *   # the alias package: exactly the reference's import line

f = sphere(1) & box(1.5)
c = cylinder(0.5)
f -= c.orient(X) | c.orient(Y) | c.orient(Z)
for samples in (2 ** 22, 2 ** 27):
    for rep in range(2):
        t0 = time.perf_counter()
        f.save('/tmp/out.stl', samples=samples, verbose=False)
        dt = time.perf_counter() - t0
    print('f.save(samples=2**%d): %.3f s, %d bytes' % (samples.bit_length() - 1, dt, os.path.getsize('/tmp/out.stl')))
t0 = time.perf_counter(); pts = f.generate(samples=2 ** 27, verbose=False); print('f.generate(2**27) -> ndarray %s: %.3f s' % (pts.shape, time.perf_counter() - t0))
