#!/usr/bin/env python3
"""What would a THIRD culling level buy?  (CPU only: the host build of the interval interpreter, no GPU.)
    python tools/cull3study.py [model:log2samples ...]
k_cull decides groups of 4^3 cells by interval arithmetic; a sample is evaluated iff it belongs to an undecided group
(sdf_device.h cull_tasks).  This script redoes that decision on the host for every surviving batch of a job -- the
reference's kinds from tests/golden/full_*.npz -- and then splits every undecided group into 8 sub-groups of 2^3 cells
(and those into single cells): the share of samples each level leaves to the interpreter, and the number of interval
runs it costs.  (Whole tape, not the per-batch pruned one: the device's intervals are the same or tighter.)"""
import ctypes, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sdf_amd, fixtures
from sdf_amd import core, tape as tape_mod
import test_interval_host as tih

ns = {k: getattr(sdf_amd, k) for k in dir(sdf_amd) if not k.startswith('_')}
so = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'libia_tape_cull3.so')
subprocess.check_call([tih.HIPCC, '--offload-host-only', '-O2', '-std=c++17', '-ffp-contract=off', '-w', '-fPIC', '-shared', '-I',
                       os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', so, os.path.join(ROOT, 'tests', 'native', 'interval_tape_host.hip')])
lib = ctypes.CDLL(so)
lib.ia_tape_boxes.restype = ctypes.c_int
lib.ia_tape_boxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
GOLD = {'example:27': 'full_c2_example_s27.npz', 'blobby:30': 'full_c5_blobby_s30.npz', 'gearlike:30': 'full_c3_gearlike_s30.npz',
        'weave:24': 'full_weave_s24.npz', 'knurling:27': 'full_knurling_s27.npz'}


def intervals(t, boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    out = np.empty((len(boxes), 2))
    code = np.ascontiguousarray(t.code, dtype=np.uint32)
    consts = np.ascontiguousarray(np.concatenate([t.consts, [0.0]]))
    assert lib.ia_tape_boxes(code.ctypes.data, consts.ctypes.data, t.n_instr, t.n_pslots, t.n_dslots, boxes.ctypes.data, len(boxes), out.ctypes.data) == 0
    return out


def undecided(iv):
    return ~((iv[:, 0] > 1e-30) | (iv[:, 1] < -1e-30))


for job in (sys.argv[1:] or ['example:27']):
    model, k = job.split(':')
    g = np.load(os.path.join(ROOT, 'tests', 'golden', GOLD[job]))
    f = fixtures.build('ex_' + model, ns)
    X, Y, Z, _ = core.grid_axes(tuple(map(tuple, g['bounds'])), samples=2 ** int(k))
    t = tape_mod.lower(f)
    kinds = g['kinds']
    nbx, nby, nbz = (-(-len(a) // 32) for a in (X, Y, Z))
    work = np.flatnonzero(kinds != 0)
    if len(work) > 4000:                                   # a sample of the surviving batches is enough for shares
        work = work[np.random.default_rng(1).choice(len(work), 4000, replace=False)]
    t0 = time.time()
    tot = dict(samples=0, l2=0, l3=0, l4=0, runs2=0, runs3=0, runs4=0, und2=0, und3=0)
    for b in work:
        bx, r = divmod(int(b), nby * nbz); by, bz = divmod(r, nbz)
        ax = [a[32 * o: 32 * o + 33] for a, o in ((X, bx), (Y, by), (Z, bz))]
        n = [len(a) for a in ax]
        if min(n) < 2:
            continue
        c = [m - 1 for m in n]                               # cells per axis
        def level(size, parents=None):
            """boxes of size^3 cells; parents: boolean array over the coarser level's groups (None: all)"""
            ng = [-(-ci // size) for ci in c]
            idx = np.stack(np.meshgrid(*[np.arange(m) for m in ng], indexing='ij'), -1).reshape(-1, 3)
            if parents is not None:
                pshape = parents.shape
                keep = parents[np.minimum(idx[:, 0] // 2, pshape[0] - 1), np.minimum(idx[:, 1] // 2, pshape[1] - 1), np.minimum(idx[:, 2] // 2, pshape[2] - 1)]
                idx = idx[keep]
            lo = [ax[d][np.minimum(idx[:, d] * size, c[d])] for d in range(3)]
            hi = [ax[d][np.minimum(idx[:, d] * size + size, c[d])] for d in range(3)]
            boxes = np.stack([np.minimum(lo[0], hi[0]), np.maximum(lo[0], hi[0]), np.minimum(lo[1], hi[1]), np.maximum(lo[1], hi[1]),
                              np.minimum(lo[2], hi[2]), np.maximum(lo[2], hi[2])], 1)
            und = np.zeros(ng, bool)
            if len(idx):
                u = undecided(intervals(t, boxes))
                und[idx[u, 0], idx[u, 1], idx[u, 2]] = True
            return und, len(idx)
        def evaluated(und, size):
            """samples that belong to an undecided group (a sample belongs to the groups of the cells around it)"""
            m = np.zeros(n, bool)
            for gi in np.argwhere(und):
                sl = tuple(slice(int(gi[d]) * size, min(int(gi[d]) * size + size, c[d]) + 1) for d in range(3))
                m[sl] = True
            return int(m.sum())
        u2, r2 = level(4)
        # today's TASKS (cull_tasks): cubes of 4^3 samples [4a, 4a + 3] per axis, listed if any group among {a - 1, a}^3 is
        # undecided; the samples with index 32 on an axis in face tasks by the same rule (counted per sample here)
        def unknown_in(lo, hi):                 # any undecided group among the cells [lo, hi] (clipped), per axis
            sl = tuple(slice(max(lo[d], 0) >> 2, (min(hi[d], c[d] - 1) >> 2) + 1) for d in range(3))
            return bool(u2[sl].any())
        if n == [33, 33, 33]:
            near = np.zeros((9, 9, 9), bool)     # near[a]: any of the groups {a - 1, a}^3 undecided (a = 8: the samples with index 32)
            for d0 in (0, 1):
                for d1 in (0, 1):
                    for d2 in (0, 1):
                        near |= np.pad(u2, ((d0, 1 - d0), (d1, 1 - d1), (d2, 1 - d2)))
            w = np.array([4] * 8 + [1])
            today = int((near * w[:, None, None] * w[None, :, None] * w[None, None, :]).sum())
            cubes = today
        else:
            # a ragged tile: tasks are runs of 64 consecutive samples (linear index), their box the rows / planes they touch
            lyz, nvox = n[1] * n[2], n[0] * n[1] * n[2]
            today = 0
            for task in range((nvox + 63) >> 6):
                i0, i1 = task * 64, min(task * 64 + 63, nvox - 1)
                x0, x1 = i0 // lyz, i1 // lyz
                y0, y1, z0, z1 = 0, n[1] - 1, 0, n[2] - 1
                if x0 == x1:
                    y0, y1 = (i0 - x0 * lyz) // n[2], (i1 - x0 * lyz) // n[2]
                    if y0 == y1:
                        z0, z1 = i0 - x0 * lyz - y0 * n[2], i1 - x0 * lyz - y0 * n[2]
                if unknown_in((x0 - 1, y0 - 1, z0 - 1), (x1, y1, z1)):
                    today += i1 - i0 + 1
            # ... and if it had cubes of 4^3 samples like the regular tile (clipped to the tile)
            cubes = 0
            for a0 in range(-(-n[0] // 4)):
                for a1 in range(-(-n[1] // 4)):
                    for a2 in range(-(-n[2] // 4)):
                        a = (a0, a1, a2)
                        if unknown_in(tuple(4 * q - 1 for q in a), tuple(4 * q + 3 for q in a)):
                            cubes += int(np.prod([min(4, n[d] - 4 * a[d]) for d in range(3)]))
            tot['ragged'] = tot.get('ragged', 0) + 1
            tot['ragged_samples'] = tot.get('ragged_samples', 0) + nvox
            tot['ragged_today'] = tot.get('ragged_today', 0) + today
            tot['ragged_cubes'] = tot.get('ragged_cubes', 0) + cubes
        tot['today_tasks'] = tot.get('today_tasks', 0) + today
        tot['cube_tasks'] = tot.get('cube_tasks', 0) + cubes
        u3, r3 = level(2, u2)
        u4, r4 = level(1, u3)
        # the third level as TASKS: units of 2^3 samples [2u, 2u + 1] per axis (8 units to a task of 64 lanes), listed if any
        # sub-group among {u - 1, u}^3 is undecided; the last sample of a 33-sample axis as a unit of its own
        nu = [-(-m // 2) for m in n]
        sub = np.zeros(nu, bool)
        for d0 in (0, 1):
            for d1 in (0, 1):
                for d2 in (0, 1):
                    padded = np.pad(u3, ((d0, 0), (d1, 0), (d2, 0)))[:nu[0], :nu[1], :nu[2]]
                    full = np.zeros(nu, bool)
                    full[:padded.shape[0], :padded.shape[1], :padded.shape[2]] = padded
                    sub |= full
        wts = [np.minimum(2, np.array(n[d]) - 2 * np.arange(nu[d])) for d in range(3)]
        tot['units3'] = tot.get('units3', 0) + int((sub * wts[0][:, None, None] * wts[1][None, :, None] * wts[2][None, None, :]).sum())
        tot['samples'] += n[0] * n[1] * n[2]
        tot['l2'] += evaluated(u2, 4); tot['l3'] += evaluated(u3, 2); tot['l4'] += evaluated(u4, 1)
        tot['runs2'] += r2; tot['runs3'] += r3; tot['runs4'] += r4
        tot['und2'] += int(u2.sum()); tot['und3'] += int(u3.sum())
    s = tot['samples']
    print('%s: %d surviving batches studied, %.0f s' % (job, len(work), time.time() - t0))
    print('  samples of those batches            %12d' % s)
    print('  evaluated with 4^3 groups (today)   %12d  %5.1f %%   interval runs per batch %6.0f (undecided groups %5.1f)' % (tot['l2'], 100.0 * tot['l2'] / s, tot['runs2'] / len(work), tot['und2'] / len(work)))
    if 'today_tasks' in tot:
        print('  ... in the device\'s TASKS of 64       %12d  %5.1f %%   (bench: n_sampled / n_eval)' % (tot['today_tasks'], 100.0 * tot['today_tasks'] / s))
        if tot.get('ragged'):
            print('      of which the %d ragged tiles (an axis with fewer than 33 samples; tasks = runs of 64 samples): %.1f %% of their %d samples;'
                  ' with cubes of 4^3 samples like the regular tile: %.1f %%' % (tot['ragged'], 100.0 * tot['ragged_today'] / tot['ragged_samples'], tot['ragged_samples'],
                                                                               100.0 * tot['ragged_cubes'] / tot['ragged_samples']))
            print('  ... with cube tasks on every tile      %12d  %5.1f %%' % (tot['cube_tasks'], 100.0 * tot['cube_tasks'] / s))
    print('  + 2^3 sub-groups of the undecided   %12d  %5.1f %%   + interval runs per batch %6.0f (undecided sub-groups %5.1f)' % (tot['l3'], 100.0 * tot['l3'] / s, tot['runs3'] / len(work), tot['und3'] / len(work)))
    print('  ... in units of 2^3 samples, 8 to a task %11d  %5.1f %%' % (tot['units3'], 100.0 * tot['units3'] / s))
    print('  + single cells of those             %12d  %5.1f %%   + interval runs per batch %6.0f' % (tot['l4'], 100.0 * tot['l4'] / s, tot['runs4'] / len(work)))
