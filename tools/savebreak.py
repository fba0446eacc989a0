#!/usr/bin/env python3
"""where the wall time of f.save('out.stl', samples=2**27) goes (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdf import *
from sdf_amd import core, engine, stl

f = sphere(1) & box(1.5)
c = cylinder(0.5)
f -= c.orient(X) | c.orient(Y) | c.orient(Z)
eng = engine.get_engine()
for rep in range(3):
    t = [time.perf_counter()]
    bounds = core._estimate_bounds(f); t.append(time.perf_counter())
    Xa, Ya, Za, _ = core.grid_axes(bounds, samples=2 ** 27); t.append(time.perf_counter())
    m = eng.generate(f, Xa, Ya, Za); t.append(time.perf_counter())
    rec = m.stl_records(); t.append(time.perf_counter())
    m.close(); t.append(time.perf_counter())
    stl.write_stl_records('/tmp/out.stl', rec); t.append(time.perf_counter())
    del rec
    names = ['bounds', 'axes', 'generate', 'stl_records (k_stl + D2H 147 MB)', 'close', 'file write']
    print('rep %d: ' % rep + ', '.join('%s %.1f ms' % (n, 1e3 * (b - a)) for n, a, b in zip(names, t, t[1:])) + '; total %.1f ms' % (1e3 * (t[-1] - t[0])))
