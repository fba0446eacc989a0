#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sampling + meshing hot path.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; started plainly (`python bench.py
--gpus N`) the script launches those N ranks itself (same launcher, 127.0.0.1) and rank 0's line comes out of it.

Workload (BASELINE.json configs[1]): the canonical CSG example (reference examples/example.py:
sphere & box - 3 cylinders) on the 512^3 grid the reference builds for samples=2**27 over its own
estimated bounds, sparse=True, batch_size=32.  One "step" = one complete pass of the hot path:
skip prepass -> work list -> interval passes -> fused sample+march kernel writing the ordered float64 (3T,3)
soup in HBM (N > 1: every rank meshes its share of the work list into a slab, ONE RCCL all-gather, expansion into
the same ordered soup on every rank).  The axes are the only input (3 x 512 float64); the output stays in HBM
inside the timed region (the PCIe-inclusive rate is reported separately as `value_incl_d2h`).  float64 sampling =
the reference's NumPy precision.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (k_mesh), `cpu_baseline` = the CPU path
TIMED ON THIS BOX IN THIS RUN: the reference's own (NumPy thread pool + skimage; kind "reference") where it is installed, else the C
port of its algorithm that serves as checker, on one host core (kind "port": the GPU boxes have no reference);
`cpu_reference_recorded` = the reference's numbers from the build container where it could not be timed here (another machine: said
so in its `kind`), `cpu_port` = the C oracle timed live on one host core (always), `isolated_calls` = what ONE
synchronous call costs (min / median / max over back-to-back calls with nothing else in flight: what a drop-in
caller of f.generate() sees), `generate_e2e` = `f.generate(samples=2**27)` on a fresh model end to end (bounds estimate, tape,
meshing, the copy of the soup to the host: like for like with `cpu_baseline`), `sustained` = the headline job over 2000 steps
with the shader clock the kernels measured, `clocks`, `other_configs` = BASELINE configs 3 - 5 at their real sizes (a few steps
each; `whole_soup_vs_oracle`: every coordinate of their soups against the CPU checker meshing the same grid on the host's
cores), `parity_per_rank` (N > 1: every rank hashes the soup it holds).
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# bounds the reference's _estimate_bounds returns for the example (tests/golden/bounds.npz['ex_example'])
EXAMPLE_BOUNDS = ((-0.8454300600008358, -0.8454300600008358, -0.8454300600008358),
                  (0.8454307895539046, 0.8454307895539046, 0.8454307895539046))
# sha256 of the float64 soup the UNMODIFIED reference produces on that grid at samples=2**27
# (tools/make_golden_full.py -> tests/golden/full_c2_example_s27.npz; 2 945 152 triangles)
EXAMPLE_S27_SHA256 = '51db77f1a24d68538de3fdfb99fdc379fad9917d3076090ed7086060bb89b88a'
REFERENCE_DIR = '/root/reference'
REFERENCE_PYTHON = '/opt/conda/bin/python3.9'

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6

# BASELINE.json configs[2..4] at their real sizes: (model, log2 samples, triangles the reference / the golden runs give)
# (+ timed steps: enough of them that filling and draining the four-deep pipeline does not dominate a millisecond job)
# (weave LAST: its host-side comparison ends with up to 128 checker processes being killed, each holding a piece of a 3.9 GB soup -- the
# millisecond job measured right behind that, blobby, came out at 1.28 - 1.52 ms per step with four calls in flight in two of five runs
# (r05y, r05af) against 0.88 everywhere else)
OTHER_CONFIGS = [('gearlike', 30, 10204096, 24), ('blobby', 30, 4048520, 24), ('weave', 33, 53943912, 6)]
# (diagnostics: SDF_BENCH_SKIP=incl,e2e,sustained leaves optional single-GPU sections out; SDF_BENCH_OTHER_ORDER=blobby,gearlike reorders / selects the other configs)
SKIP = set(filter(None, os.environ.get('SDF_BENCH_SKIP', '').split(',')))
if os.environ.get('SDF_BENCH_OTHER_ORDER'):
    OTHER_CONFIGS = [c for n in os.environ['SDF_BENCH_OTHER_ORDER'].split(',') for c in OTHER_CONFIGS if c[0] == n]


# DESIGN.md section 6, arithmetic for 8 GPUs (one GPU's measured stage times / 8 + fixed costs + 16 B per triangle (sdf_slab.h records) over one xGMI
# link at ~76 GB/s + the expansion every rank repeats); printed next to the measured stage times of an N > 1 run
EXPECTED_SCALING = {
    'example': 'C2 512^3: break-even by construction (~1.0 - 1.3 x at 8 GPUs): the work that divides is ~0.03 ms of a ~0.3 ms step; fixed: skip test, '
               'all-gather of 6 MB slabs (16 B per triangle: 47 MB over seven links, ~0.08 ms on the wire), k_expand of the whole soup on every rank (~0.08 ms)',
    'gearlike': 'C3 2^30: ~2.5 x at 8 GPUs (1.6 ms -> ~0.6 ms: 21 MB slabs (16 B per triangle) ~0.3 ms on the wire overlap the next step\'s meshing with two lanes)',
    'weave': 'C4 2^33: ~4 x at 8 GPUs (24 ms -> ~6 ms: meshing / 8 ~3 ms, 115 MB slab (16 B per triangle) ~1.5 ms on the wire, k_expand ~1.6 ms)',
    'blobby': 'C5 2^30: ~2 x at 4 GPUs (1.2 ms -> ~0.6 ms)',
}


def build_model(name):
    import sdf_amd as s
    if name == 'example':
        f = s.sphere(1) & s.box(1.5)
        c = s.cylinder(0.5)
        f -= c.orient(s.X) | c.orient(s.Y) | c.orient(s.Z)
        return f, None
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fixtures
    ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
    return fixtures.build('ex_' + name, ns), None


def reference_cpu_baseline(samples_log2):
    """the `cpu_baseline` object from the reference's own generate(): live (subprocess of the reference's
    interpreter, tools/time_reference.py) or, where the reference is not installed, the committed numbers"""
    script = os.path.join(ROOT, 'tools', 'time_reference.py')
    rec, kind = None, None
    if os.path.isdir(REFERENCE_DIR) and os.path.exists(REFERENCE_PYTHON) and not os.environ.get('SDF_BENCH_NO_LIVE_REFERENCE'):
        try:
            env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
            out = subprocess.run([REFERENCE_PYTHON, '-W', 'ignore', script, '--samples-log2', str(samples_log2)],
                                 env=env, capture_output=True, text=True, timeout=600, check=True).stdout
            rec, kind = json.loads(out.strip().splitlines()[-1]), 'reference'
        except Exception as e:           # fall back to the committed run
            sys.stderr.write('live reference timing failed: %r\n' % (e,))
    if rec is None:
        path = os.path.join(ROOT, 'profiles', 'reference_cpu.json')
        if not os.path.exists(path):
            return None
        rec = json.load(open(path))
        kind = 'reference (measured in the build container, %d vCPU; %s absent on this box)' % (rec['host_cores'], REFERENCE_DIR)
    best = max(rec['runs'], key=lambda r: r['voxels_per_sec'])
    return {'value': best['voxels_per_sec'], 'unit': 'voxels/s', 'cores': best['workers'], 'kind': kind,
            'sample': '%s; %s; whole workload, %.1f s with workers=%d; all runs: %s'
                      % (rec['path'], rec['workload'], best['seconds'], best['workers'],
                         ', '.join('workers=%d %.1f s' % (r['workers'], r['seconds']) for r in rec['runs'])),
            'triangles_per_sec': best['triangles_per_sec'], 'host_cores': rec['host_cores'],
            'soup_sha256': best.get('soup_sha256')}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks (one per GPU) under
    torch.distributed.run on this node and hand their output through; rank 0 prints the JSON line"""
    import socket
    import torch
    if not os.environ.get('SDF_BENCH_ONE_DEVICE'):
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:
            raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible on this node' % (args.gpus, n))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % args.gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # (dmabuf IPC: RCCL across processes needs it on this driver)
    raise SystemExit(subprocess.call(cmd, env=env))


def read_clocks():
    """current shader / memory / fabric clocks in MHz from `rocm-smi --showclocks` (best effort; None where the tool is
    missing).  The clock the kernels actually ran at is measured by the kernels themselves (`sclk_mhz_in_kernel`)."""
    import re
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        pick = {}
        for name, mhz in re.findall(r'GPU\[0\]\s*:\s*(\w+) clock level:[^(]*\((\d+)Mhz\)', out):
            pick[name + '_mhz'] = int(mhz)
        return pick or None
    except Exception:
        return None


def trace(msg):
    """progress markers on stderr (SDF_BENCH_TRACE=1), tagged with the rank"""
    if os.environ.get('SDF_BENCH_TRACE'):
        sys.stderr.write('[bench rank %s %.4f s] %s\n' % (os.environ.get('RANK', '0'), time.perf_counter() % 1000.0, msg))
        sys.stderr.flush()


def watchdog(out):
    """The optional `other_configs` section must not take the headline line down with it.  If it has not finished after
    SDF_BENCH_OTHER_TIMEOUT_S seconds (default 600 for N > 1 -- a rank stuck in a collective its peers never entered --, 300 on one
    GPU -- r05ae: a default run sat in this section until its caller's 600 s limit, no line printed; the section's host side runs
    the CPU checker in up to 128 processes) rank 0 prints the line it already has (with the reason in `other_configs`) and every
    rank exits."""
    import threading
    world = int(os.environ.get('WORLD_SIZE', '1'))
    limit = float(os.environ.get('SDF_BENCH_OTHER_TIMEOUT_S', '600' if world > 1 else '300'))

    def fire():
        if out is not None:
            out['other_configs'] = [{'error': 'other_configs did not finish within %.0f s; skipped' % limit}]
            out['stalled'] = ['other_configs']      # (the OPTIONAL section stalled; the headline above it was measured)
            print(json.dumps(out), flush=True)
        # exit code 0 on purpose: the headline measurement is complete and valid; the stall is in the line (`stalled`), and
        # the headline's own watchdog (below) exits with 3 when the measurement itself hangs
        sys.stdout.flush()
        os._exit(0)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def pipelined_overlap(spans, dev_ms):
    """With several calls in flight two k_mesh launches -- lanes of their own -- may be resident at once: a k_mesh workgroup
    holds a whole CU, so the later launch gets the CUs the earlier one leaves, and BOTH launches' first-workgroup-to-last
    spans stretch (up to 2 x) while the steps complete at the same rate.  That is the spread of `kernel_ms_pipelined`
    (VERDICT r03, weak 4), not a slow step: per timed step, its span on the device's constant-rate counter and how much
    of it lies inside the spans of the steps before and after it."""
    if not spans or not any(b > a for a, b in spans):
        return None
    t0 = min(a for a, b in spans if b > a)
    rows = []
    for i, (a, b) in enumerate(spans):
        ov = 0.0
        for j, (c, d) in enumerate(spans):
            if j != i and d > c:
                ov += max(0.0, min(b, d) - max(a, c))
        rows.append({'start_us': round(a - t0, 1), 'span_ms': round((b - a) * 1e-3, 4), 'shared_with_neighbours_ms': round(ov * 1e-3, 4)})
    alone = [r['span_ms'] for r in rows if r['shared_with_neighbours_ms'] < 0.02 * r['span_ms']]
    starts = sorted(r['start_us'] for r in rows)
    gaps = [b - a for a, b in zip(starts, starts[1:])]
    return {'steps': rows if len(rows) <= 40 else rows[:40],
            'span_ms_of_steps_that_ran_alone': stats3(alone) if alone else None,
            'start_to_start_ms': stats3([g * 1e-3 for g in gaps]) if gaps else None,
            'note': 'span = first workgroup start to last workgroup end of k_mesh on the device clock; a span overlapped by a neighbour '
                    'is a launch that shared the CUs, not a slower kernel: the start-to-start interval is the step time'}


def valu_issue(pmc_kernel, k_ms, iso, world, sampled_vox, plain, special, st):
    """share of the kernel's SIMD cycles spent issuing vector-ALU instructions, from the committed counters of this source"""
    if not (pmc_kernel.get('SQ_INSTS_VALU') and world == 1 and iso and k_ms > 0 and iso['sclk_mhz_in_kernel']['median']):
        return None
    n = float(pmc_kernel['SQ_INSTS_VALU']['mean'])
    cyc = 1024.0 * k_ms * 1e-3 * float(iso['sclk_mhz_in_kernel']['median']) * 1e6        # SIMD cycles of the launch
    kept = 1.0 - (st.get('n_pruned_instrs', 0) / st['n_batch_instrs'] if st.get('n_batch_instrs') else 0.0)
    f64 = min(n, sampled_vox / 64.0 * (plain + 20.0 * special) * kept)
    return {'upper_all_4_cycles': round(4.0 * n / cyc, 3), 'lower_all_2_cycles': round(2.0 * n / cyc, 3),
            'weighted_f64_4_rest_2': round((4.0 * f64 + 2.0 * (n - f64)) / cyc, 3), 'f64_wave_instructions_est': round(f64)}


def stats3(v):
    v = np.asarray(v, dtype=np.float64)
    return {'min': round(float(v.min()), 4), 'median': round(float(np.median(v)), 4), 'max': round(float(v.max()), 4), 'n': int(len(v))}


def _oracle_worker(jobs, order, outdir, wid):
    """worker of whole_soup_vs_oracle (its own process, no GPU): the checker's soup of the pieces `order` names, each left as a file of
    its own (written under another name, then renamed: the parent only ever sees whole files)"""
    import oracle
    from sdf_amd import core
    f = X = None
    for k in order:
        model, bounds, log2, b0, b1 = jobs[k]
        if f is None:
            f, _ = build_model(model)
            X, Y, Z, _ = core.grid_axes(bounds, samples=2 ** log2)
        pts = oracle.generate(f, X, Y, Z, 32, True, batch_range=(b0, b1)).points
        tmp = os.path.join(outdir, 'w%d.tmp.npy' % wid)
        np.save(tmp, pts)
        os.replace(tmp, os.path.join(outdir, 'piece%d.npy' % k))


def soup_verdict(compared, expected, differ, worst_over_extent):
    """the fields every in-bench comparison of soup coordinates reports: `coverage` = the share of the soup's coordinates that WERE
    compared, and a verdict only where something was (a comparison of nothing says null, not true -- VERDICT r05 item 5)"""
    cov = (compared / expected) if expected else 1.0
    some = compared > 0 or expected == 0
    return {'coordinates': int(compared), 'coordinates_expected': int(expected), 'coverage': round(cov, 6),
            'coordinates_that_differ': int(differ) if some else None,
            'share_bit_equal': round(1.0 - differ / max(compared, 1), 9) if some else None,
            'max_abs_diff_over_extent': worst_over_extent if some else None,
            'within_1e-5': bool(worst_over_extent <= 1e-5) if some else None,
            'whole_soup': bool(compared == expected)}


def whole_soup_vs_oracle(model, bounds, log2, soup_host, offsets, budget_cores=128, budget_s=240.0):
    """Coordinates of a soup against the CPU checker's (oracle/sdf_oracle.c, the reference's algorithm restated): the grid's batches
    are cut into pieces, host processes mesh WHOLE pieces with the checker in a seeded random order (so that whatever the budget
    reaches is spread over the grid), and every finished piece is compared with the rows of the soup the device attributes to its
    batches (`offsets` = Mesh.batch_offsets(): where each batch's triangles start).  Returns counts and `coverage` -- the share of the
    soup that was compared; models that go through libm (sin / cos / atan2: gearlike, weave) are pinned by tolerance.  (Plain processes
    that leave their pieces as files, polled against a deadline and killed at the end: nothing here can wait for a queue, a lock or a
    pool's shutdown -- r05ae: a default run of this file never came back from its optional sections.)"""
    import multiprocessing as mp
    import shutil
    import tempfile
    n_batches = len(offsets) - 1
    cores = max(1, min(os.cpu_count() or 1, budget_cores))
    pieces = max(cores * 8, 64)
    cuts = [n_batches * i // pieces for i in range(pieces + 1)]
    jobs = [(model, bounds, log2, cuts[i], cuts[i + 1]) for i in range(pieces) if cuts[i + 1] > cuts[i]]
    # (pieces without a triangle on the device side are still meshed: a checker that finds triangles there is a difference)
    order = np.random.RandomState(12345).permutation(len(jobs))
    extent = float(np.ptp(np.asarray(bounds), axis=0).max())
    compared = differ = done = 0
    worst = 0.0
    t0 = time.perf_counter()
    deadline = t0 + max(budget_s, 5.0)
    outdir = tempfile.mkdtemp(prefix='sdf_soup_', dir='/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None)
    ctx = mp.get_context('spawn')
    nproc = min(cores, len(jobs))
    procs = [ctx.Process(target=_oracle_worker, args=(jobs, [int(k) for k in order[wid::nproc]], outdir, wid), daemon=True) for wid in range(nproc)]
    err = None
    try:
        for p in procs:
            p.start()
        while done < len(jobs) and err is None:
            names = [n for n in os.listdir(outdir) if n.startswith('piece')]
            if not names:
                if time.perf_counter() > deadline or not any(p.is_alive() for p in procs):
                    break
                time.sleep(0.005)
                continue
            for name in names:
                k = int(name[5:-4])
                path = os.path.join(outdir, name)
                pts = np.load(path)
                os.remove(path)
                b0, b1 = jobs[k][3], jobs[k][4]
                r0, r1 = 3 * int(offsets[b0]), 3 * int(offsets[b1])
                done += 1
                if len(pts) != r1 - r0:
                    err = {'error': 'the checker has %d vertices in batches [%d, %d), the soup %d' % (len(pts), b0, b1, r1 - r0)}
                    break
                mine = soup_host[r0:r1]
                ne = mine != pts
                differ += int(ne.sum())
                if ne.any():
                    worst = max(worst, float(np.abs(mine - pts)[ne].max()))
                compared += 3 * len(pts)
            if time.perf_counter() > deadline:
                break
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
        for p in procs:
            p.join(timeout=2.0)
        shutil.rmtree(outdir, ignore_errors=True)
    if err is not None:
        return err
    out = soup_verdict(compared, 3 * len(soup_host), differ, worst / extent)
    out.update({'pieces_compared': done, 'pieces': len(jobs), 'checker_seconds': round(time.perf_counter() - t0, 1), 'host_processes': nproc,
                'what': 'coordinates of the soup against oracle/sdf_oracle.c meshing whole pieces of the grid on the host (seeded random order of pieces; '
                        'coverage = share of the soup compared within the time budget)'})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--model', default='example', help='example | gearlike | blobby | weave | knurling')
    ap.add_argument('--samples-log2', type=int, default=27, help='grid = samples=2**k through the reference step rule')
    ap.add_argument('--precision', default='f64', choices=['f64'], help='the meshing path is float64 (the reference\'s arithmetic); float32 sampling was removed in round 5')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the BASELINE configs 3 - 5 section')
    ap.add_argument('--no-f32-envelope', action='store_true', help='(accepted and ignored: the float32 section was removed with the mode, round 5)')
    ap.add_argument('--sync', action='store_true', help='one step in flight: every call synchronises before the next is submitted')
    ap.add_argument('--inflight', type=int, default=6, help='steps in flight on a single GPU (each on a lane of its own; at most 8)')
    ap.add_argument('--chunks', type=int, default=None, help='N > 1: shards per rank and step (default 1: one all-gather per step)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1 and 'RANK' not in os.environ:
        self_launch(args)
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    import torch
    # debugging aids for boxes with ONE GPU: SDF_BENCH_ONE_DEVICE=1 puts every rank on device 0, SDF_BENCH_BACKEND=gloo
    # exchanges through gloo (RCCL refuses two ranks on one device), SDF_BENCH_COMM_DEVICE=cuda keeps the slabs on the
    # device all the same (the device side of the protocol between two real processes)
    backend = os.environ.get('SDF_BENCH_BACKEND', 'nccl')
    if os.environ.get('SDF_BENCH_ONE_DEVICE'):
        local_rank = 0
        os.environ['SDF_AMD_DEVICE'] = '0'      # (what engine.get_engine() without an argument -- core._estimate_bounds -- resolves to)
    torch.cuda.set_device(local_rank)
    td = None
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            td.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            td.init_process_group(backend, rank=rank, world_size=world)

    from sdf_amd import core, engine, dist
    eng = engine.get_engine(local_rank)
    eng.precision = engine.PRECISION_F64
    dev = torch.device('cuda', local_rank)
    comm_dev = dev if (backend == 'nccl' or os.environ.get('SDF_BENCH_COMM_DEVICE') == 'cuda') else torch.device('cpu')
    stat_dev = dev if backend == 'nccl' else torch.device('cpu')      # (the few scalars the ranks exchange about the run itself)
    clocks_idle = read_clocks() if rank == 0 else None

    def soup_sha(soup, tris):
        """sha256 of the first `tris` triangles of a soup on the device (torch tensor), copied out in pieces"""
        import hashlib
        h = hashlib.sha256()
        piece = 1 << 22
        for t0 in range(0, tris, piece):
            h.update(soup[9 * t0:9 * min(tris, t0 + piece)].cpu().numpy().tobytes())
        return h.hexdigest()

    def measure(model, samples_log2, steps, warmup, depth, bounds=None):
        """W untimed + K timed steps of one job; a step is complete when its counters (N > 1: the gathered slab
        headers) are back on the host.  Returns timings, per-step kernel times, the last step's soup + statistics."""
        trace('measure %s 2^%d: %d steps, %d in flight' % (model, samples_log2, steps, depth))
        eng.synchronize()
        eng.trim()      # (blocks cached for the previous job's sizes would otherwise be evicted -- hipFree -- inside this job's timed steps)
        f, _ = build_model(model)
        tape = eng.tape_for(f)
        if bounds is None:
            bounds = EXAMPLE_BOUNDS if model == 'example' else core._estimate_bounds(f)
        X, Y, Z, _ = core.grid_axes(bounds, samples=2 ** samples_log2)
        trace('bounds and axes ready: %dx%dx%d' % (len(X), len(Y), len(Z)))
        state = {'n': 0}
        inflight, mesh_ms, exch_ms, dev_ms, sclk, spans = [], [], [], [], [], []

        def collect():
            mesh, buf = inflight.pop(0)
            mesh.wait()
            t = mesh.n_triangles
            trace('collected a call: %d triangles, in the caller buffer: %s' % (t, mesh.emitted))
            if not mesh.emitted:               # (the soup did not fit: it was meshed again into library memory)
                big = torch.empty(max(t + t // 8, 1) * 9, dtype=torch.float64, device=dev)
                mesh.emit_device(big.data_ptr())
                state['bufs'] = [big if b is buf else b for b in state['bufs']]      # (identity, not tensor ==)
                buf = big
            state['soup'] = buf
            st = mesh.stats()
            state['stats'], state['tris'] = st, t
            mesh_ms.append(st['ms_mesh']); dev_ms.append(st['ms_mesh_device']); sclk.append(st['sclk_mhz'])
            spans.append((st.get('t_mesh_first_us', 0.0), st.get('t_mesh_last_us', 0.0)))
            mesh.close()

        def collect_dist():
            soup, st = dist.collect_sharded(inflight.pop(0))
            trace('collected a step: %d triangles, %d retries' % (st['triangles'], st.get('n_retries', 0)))
            state['soup'], state['stats'], state['tris'] = soup, st, st['triangles']
            mesh_ms.append(st['ms_mesh'])            # this rank: prepass + k_mesh of its shard, into the slab
            exch_ms.append((st['ms_exchange'], st['ms_expand']))

        def one_step():
            if world == 1:
                # every step writes its ordered float64 soup into a device buffer of its own (`depth` of them
                # alternate), sized from the previous steps (first: a guess)
                bufs = state.setdefault('bufs', [torch.empty(9 * (1 << 22), dtype=torch.float64, device=dev) for _ in range(depth)])
                buf = bufs[state['n'] % depth]
                state['n'] += 1
                while len(inflight) >= depth:
                    collect()
                buf = state['bufs'][(state['n'] - 1) % depth]      # (collect may have replaced a buffer that was too small)
                inflight.append((eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel() // 9, wait=False), buf))
                trace('submitted call %d' % state['n'])
            else:
                # N > 1: steps in flight run on lanes of their own: step i + 1's meshing runs under step i's all-gather
                while len(inflight) >= depth:
                    collect_dist()
                state['n'] += 1
                trace('submit step %d' % state['n'])
                inflight.append(dist.submit_sharded(eng, tape, X, Y, Z, 32, True, device=comm_dev, lane=state['n'] % 2, chunks=args.chunks))
                trace('submitted step %d' % state['n'])

        def sync():
            while inflight:
                collect() if world == 1 else collect_dist()
            eng.synchronize()
            torch.cuda.synchronize()
            if td is not None:
                td.barrier()
                torch.cuda.synchronize()

        # set-up, not a step (every rank alike): each of the context's eight call slots allocates its k_mesh park slots on
        # first use (1.2 GB, ~35 - 50 ms each; the slots rotate, so the EIGHTH step of a process would still pay that inside
        # a timed region that starts after three warm-up steps), output buffers that are too small are replaced (N = 1),
        # the lanes' slabs and soup shrink from the first step's upper bounds to what the job needs (N > 1)
        for _ in range(max(depth + 1, 9)):      # (eight call slots rotate: every one of them has been used once before anything is timed)
            one_step()
        sync()
        # (like timeit: no cyclic garbage collection inside the timed region.  With torch imported a full collection takes
        # 30 - 50 ms on these hosts -- a hundred steps of this job -- and when it falls is a matter of how many containers the
        # process has allocated so far: tools/disttime.py showed it as a "slow mode" of whichever configuration it hit)
        gc.collect()
        gc.disable()
        try:
            for _ in range(warmup):
                one_step()
            sync()
            del mesh_ms[:], exch_ms[:], dev_ms[:], sclk[:], spans[:]
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            sync()                                 # every one of the K steps is complete (collected) here
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        trace('measure %s done: %.3f ms per step' % (model, 1e3 * dt / steps))
        assert len(mesh_ms) == steps
        if td is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=stat_dev)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            dt = float(tt.item())
        return {'f': f, 'tape': tape, 'X': X, 'Y': Y, 'Z': Z, 'dt': dt, 'mesh_ms': list(mesh_ms), 'exch_ms': list(exch_ms),
                'dev_ms': list(dev_ms), 'sclk': list(sclk), 'spans': list(spans), 'state': state, 'grid_voxels': len(X) * len(Y) * len(Z)}

    DEPTH = 1 if args.sync else (max(1, min(args.inflight, 8)) if world == 1 else 2)
    # N > 1: a rank that dies or stalls leaves the others inside a collective for ever.  The headline measurement of a healthy
    # job takes seconds; if it has not come back after SDF_BENCH_HEADLINE_TIMEOUT_S (default 900) every rank says so and
    # exits instead of holding its GPU until somebody kills the job.
    headline_timer = None
    if world > 1:
        import threading
        limit = float(os.environ.get('SDF_BENCH_HEADLINE_TIMEOUT_S', '900'))

        def give_up():
            sys.stderr.write('bench.py rank %d: the headline measurement did not finish within %.0f s (a rank stuck in a collective?); giving up\n' % (rank, limit))
            sys.stderr.flush()
            os._exit(3)
        headline_timer = threading.Timer(limit, give_up)
        headline_timer.daemon = True
        headline_timer.start()
    res = measure(args.model, args.samples_log2, args.steps, args.warmup, DEPTH)
    if headline_timer is not None:
        headline_timer.cancel()
    clocks_busy = read_clocks() if rank == 0 else None       # (right behind the timed region)
    f, tape, X, Y, Z, dt, state = res['f'], res['tape'], res['X'], res['Y'], res['Z'], res['dt'], res['state']
    grid_voxels = res['grid_voxels']
    mesh_ms, exch_ms = res['mesh_ms'], res['exch_ms']
    st = state['stats']
    tris = int(state['tris'])
    ms_per_step = 1e3 * dt / args.steps
    value = grid_voxels * args.steps / dt
    per_rank = None
    # ---- the headline is measured: from here on nothing may keep its line from being printed (one GPU; N > 1: the headline's own
    # watchdog above and `watchdog` below).  If the sections that follow -- isolated calls, end to end, sustained run, CPU baselines,
    # the other configurations -- have not led to the full line after SDF_BENCH_TOTAL_TIMEOUT_S seconds (default 480), the line is
    # printed with what is known by then and the process exits (code 0: the measurement is complete and valid; `stalled` says so) ----
    guard = None
    if world == 1:
        import threading

        def last_resort():
            print(json.dumps({'metric': 'grid voxels/sec (sampled + meshed), canonical CSG example', 'value': round(value, 1), 'unit': 'voxels/s',
                              'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
                              'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
                              'config': {'workload': '%s @ samples=2**%d -> %dx%dx%d grid, sparse=True, batch_size=32'
                                                     % (args.model, args.samples_log2, len(X), len(Y), len(Z)), 'triangles': tris},
                              'roofline': None, 'cpu_baseline': None,
                              'stalled': ['a section behind the headline measurement did not finish in time: only the headline is reported']}), flush=True)
            sys.stdout.flush()
            os._exit(0)
        guard = threading.Timer(float(os.environ.get('SDF_BENCH_TOTAL_TIMEOUT_S', '480')), last_resort)
        guard.daemon = True
        guard.start()
    if td is not None:
        # per-rank device times of the three stages of a step (means over the timed steps)
        mine = torch.tensor([float(np.mean(mesh_ms)), float(np.mean([e[0] for e in exch_ms])), float(np.mean([e[1] for e in exch_ms]))],
                            dtype=torch.float64, device=stat_dev)
        allr = torch.empty(3 * world, dtype=torch.float64, device=stat_dev)
        td.all_gather_into_tensor(allr, mine)
        per_rank = allr.cpu().numpy().reshape(world, 3)

    # parity check inside the bench run: the sha256 of the soup the LAST TIMED STEP left on the device
    # (copied out after the timed region) against the hash of the reference's own soup on this grid
    # N > 1: EVERY rank hashes the soup IT holds after the exchange (rank-local: a transport bug that corrupts one rank's copy shows
    # up as a parity failure of that rank, not as a hang); the verdicts -- one number per rank -- travel in one small all-gather
    # behind the timed region
    check = soup_hash = parity_per_rank = None
    if not args.no_check:
        last = state.get('soup')
        if last is not None and tris * 9 <= last.numel() and (rank == 0 or world > 1):
            soup_hash = soup_sha(last, tris)
        headline_job = args.model == 'example' and args.samples_log2 == 27 and args.precision == 'f64'
        if headline_job:
            check = bool(soup_hash == EXAMPLE_S27_SHA256 and
                         (st['batches'], st['skipped'], st['empty'], st['nonempty'], tris) == (4096, 2352, 120, 1624, 2945152))
        if td is not None:
            mine_ok = torch.tensor([1.0 if (check if headline_job else soup_hash is not None) else 0.0,
                                    float(int(soup_hash[:12], 16)) if soup_hash else -1.0], dtype=torch.float64, device=stat_dev)
            allv = torch.empty(2 * world, dtype=torch.float64, device=stat_dev)
            td.all_gather_into_tensor(allv, mine_ok)
            allv = allv.cpu().numpy().reshape(world, 2)
            # (every rank's verdict against the reference's hash, and whether all ranks hold the SAME soup: the first 48 bits of their hashes)
            parity_per_rank = {'ok': [bool(v) for v in allv[:, 0]], 'all_ranks_hold_the_same_soup': bool(len(set(allv[:, 1].tolist())) == 1)}
            if headline_job and rank == 0:
                check = bool(check and all(parity_per_rank['ok']) and parity_per_rank['all_ranks_hold_the_same_soup'])
        if rank != 0:
            check = None

    # ---- ONE call at a time (single GPU): what a drop-in caller of generate() sees.  A warm-up burst, then >= 20
    # synchronous calls back to back, nothing else in flight; per call: wall time submit -> counters on the host,
    # k_mesh by HIP events and by the kernel's own clock, the prepass ----
    iso = None
    if world == 1:
        buf = state['bufs'][0]
        n_lat = max(20, min(args.steps, 50))
        wall, k_ev, k_dev, pre, sclk = [], [], [], [], []
        for i in range(5 + n_lat):
            t1 = time.perf_counter()
            mesh = eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel() // 9)
            w = time.perf_counter() - t1
            s1 = mesh.stats()
            mesh.close()
            if i >= 5:
                wall.append(1e3 * w); k_ev.append(s1['ms_mesh']); k_dev.append(s1['ms_mesh_device']); pre.append(s1['ms_prepass']); sclk.append(s1['sclk_mhz'])
        iso = {'wall_ms': stats3(wall), 'k_mesh_ms_hip_events': stats3(k_ev), 'k_mesh_ms_device_clock': stats3(k_dev),
               'prepass_ms': stats3(pre), 'sclk_mhz_in_kernel': stats3(sclk),
               'what': 'synchronous sdf_generate_to_device calls back to back after 5 warm-up calls, nothing else in flight'}
    clocks_iso = read_clocks() if rank == 0 else None

    # PCIe-inclusive variant (single GPU): same step + D2H of the soup into the ndarray `generate` returns (a recycled pinned block)
    incl = None
    if world == 1 and 'incl' not in SKIP:
        n_incl = max(1, min(args.steps, 5))
        for i in range(2 + n_incl):          # (two untimed passes: the result blocks are pinned once, then recycled)
            if i == 2:
                t1 = time.perf_counter()
            mesh = eng.generate(tape, X, Y, Z, 32, True)
            mesh.points()
            mesh.close()
        incl = grid_voxels * n_incl / (time.perf_counter() - t1)

    # ---- what a DROP-IN caller waits for: `f.generate(samples=2**27, verbose=False)` through sdf_amd.core on a FRESH model every
    # time -- bounds estimate (k_estimate_bounds), lowering to a tape + upload, grid, meshing, D2H of the soup into the ndarray the
    # reference's signature returns -- next to its parts; comparable like for like with the reference's generate() of
    # cpu_baseline (which includes `_estimate_bounds` too).  Never `value`. ----
    e2e = None
    if world == 1 and args.model == 'example' and args.precision == 'f64' and 'e2e' not in SKIP:
        trace('generate end to end')
        tt, tb, tl, tg, tp = [], [], [], [], []
        for i in range(12):
            f2, _ = build_model(args.model)
            t1 = time.perf_counter()
            pts = f2.generate(samples=2 ** args.samples_log2, verbose=False)
            tt.append(1e3 * (time.perf_counter() - t1))
            n_e2e = len(pts) // 3
            del pts
            # (its two device-side parts on their own: meshing into 16-byte records, and records -> float64 rows on the host threads)
            t1 = time.perf_counter(); mesh = eng.generate(tape, X, Y, Z, 32, True, records=True); tg.append(1e3 * (time.perf_counter() - t1))
            t1 = time.perf_counter(); pts = mesh.points(min(core.WORKERS, 32)); tp.append(1e3 * (time.perf_counter() - t1))
            mesh.close()
            del pts
            f3, _ = build_model(args.model)
            t1 = time.perf_counter(); core._estimate_bounds(f3); tb.append(1e3 * (time.perf_counter() - t1))
            f4, _ = build_model(args.model)
            t1 = time.perf_counter(); eng.tape_for(f4); tl.append(1e3 * (time.perf_counter() - t1))
        e2e = {'wall_ms': stats3(tt[2:]), 'triangles': n_e2e, 'voxels_per_sec_median': round(grid_voxels / (1e-3 * float(np.median(tt[2:]))), 1),
               'of_which_ms': {'estimate_bounds': stats3(tb[2:]), 'lower_and_upload_tape': stats3(tl[2:]),
                               'meshing_into_records': stats3(tg[2:]), 'records_to_float64_rows_on_host_threads': stats3(tp[2:])},
               'host_threads': min(core.WORKERS, 32),
               'what': 'f.generate(samples=2**%d, verbose=False) on a fresh model object, 10 calls after 2 warm-up calls: bounds + tape + grid + '
                       'meshing into 16-byte triangle records (sdf_generate_records) + D2H of the records (47 MB instead of the 212 MB of the '
                       'float64 soup) while `workers` host threads write the (n, 3) float64 ndarray from them (pinned blocks); '
                       'until round 5: D2H of the float64 soup, 4.4 ms' % args.samples_log2}

    # ---- a SUSTAINED run of the headline job: >= 2000 steps (>= 0.4 s of kernels back to back), same steps in flight, with the
    # shader clock the kernels measured themselves -- the 20-step headline is a 5 ms burst that the clocks could flatter ----
    sustained = None
    if world == 1 and args.model == 'example' and not args.sync and 'sustained' not in SKIP:
        trace('sustained run')
        rs = measure(args.model, args.samples_log2, 2000, 0, DEPTH)
        sustained = {'steps': 2000, 'seconds': round(rs['dt'], 4), 'ms_per_step': round(1e3 * rs['dt'] / 2000, 4),
                     'value': round(rs['grid_voxels'] * 2000 / rs['dt'], 1), 'sclk_mhz_in_kernel': stats3(rs['sclk']) if rs['sclk'] else None,
                     'kernel_ms': stats3(rs['mesh_ms']), 'clocks_after': read_clocks()}
        del rs

    # ---- BASELINE configs 3 - 5 at their real sizes (every rank takes part; a few steps each).  This section comes LAST
    # and, for N > 1, under a watchdog: it is the one place where a rank-local failure (an allocation that fails on one
    # rank only) would leave the other ranks inside a collective for ever, and the headline line must not depend on it ----
    def run_other_configs():
        others = []
        for model, log2, want_tris, K_OTHER in OTHER_CONFIGS:
            trace('other config %s 2^%d' % (model, log2))
            try:
                # (the headline's method: same steps in flight -- and, on one GPU, also one call at a time: the long two-pass
                # jobs lose by overlapping, the short ones win; the better of the two is the line's value, both are printed)
                # (at most four in flight here: a 2^33 job keeps 6 GB per call on the device, and the long two-pass jobs gain
                # nothing from deeper queues)
                depth_o = min(DEPTH, 4) if world == 1 else 2
                # (eight untimed steps in front of the timed ones -- every call slot's lane once: with two, the first configuration of this
                # section had ONE submission of ~ 7 ms inside its 24 timed steps on some runs, r06v / r06w: + 0.2 - 0.3 ms per step,
                # "four in flight slower than one"; nothing in the kernels -- tools/sessions/gpu_r06p.sh)
                W_OTHER = 8 if K_OTHER > 8 else 2
                r = measure(model, log2, K_OTHER, W_OTHER, depth_o)
                r0_spans, r0_dev = list(r['spans']), list(r['dev_ms'])
                used, by_depth = depth_o, None
                iso_mesh = iso_pre = None        # the kernels' own durations: from the run with ONE call in flight (launches that share the CUs stretch)
                if world == 1 and depth_o > 1:
                    one = measure(model, log2, K_OTHER, 1, 1)
                    iso_mesh, iso_pre = float(np.median(one['mesh_ms'])), float(one['state']['stats']['ms_prepass'])
                    by_depth = {'steps_in_flight_%d' % depth_o: round(1e3 * r['dt'] / K_OTHER, 4), 'steps_in_flight_1': round(1e3 * one['dt'] / K_OTHER, 4)}
                    if one['dt'] < r['dt']:
                        r, used = one, 1
                    del one
                # (how the timed steps of the run with calls in flight lay on the device's clock: start-to-start gaps and each k_mesh span)
                po = pipelined_overlap(r0_spans, r0_dev) if world == 1 else None
                s2, t2 = r['state']['stats'], int(r['state']['tris'])
                o = {'workload': '%s @ samples=2**%d -> %dx%dx%d grid' % (model, log2, len(r['X']), len(r['Y']), len(r['Z'])),
                     'n_gpus': world, 'steps': K_OTHER,
                     'steps_in_flight': used, 'ms_per_step_by_depth': by_depth,
                     'in_flight_run_on_the_device_clock': ({k: v for k, v in po.items() if k != 'steps'} if po else None),
                     'ms_per_step': round(1e3 * r['dt'] / K_OTHER, 4),
                     'value': round(r['grid_voxels'] * K_OTHER / r['dt'], 1), 'unit': 'voxels/s', 'triangles': t2,
                     'triangles_per_sec': round(t2 * K_OTHER / r['dt'], 1), 'batches': int(s2['batches']), 'skipped': int(s2['skipped']),
                     'triangles_match_reference': bool(t2 == want_tris),
                     'device_ms': ({'prepass': round(iso_pre if iso_pre is not None else float(s2['ms_prepass']), 4),
                                    'mesh': round(iso_mesh if iso_mesh is not None else float(np.median(r['mesh_ms'])), 4),
                                    'what': 'HIP events of one call at a time (prepass; k_mesh [+ k_scan_items + k_emit2])'} if world == 1 else
                                   {'mesh': round(float(np.mean(r['mesh_ms'])), 4), 'exchange': round(float(np.mean([e[0] for e in r['exch_ms']])), 4),
                                    'expand': round(float(np.mean([e[1] for e in r['exch_ms']])), 4), 'slab_bytes': s2.get('slab_bytes')})}
                if world == 1:      # the config's own roofline line: k_mesh (+ k_emit2 for two-pass jobs) against the soup's bytes
                    km = iso_mesh if iso_mesh is not None else float(np.median(r['mesh_ms']))
                    o['roofline'] = {'kernel': 'k_mesh (+ k_scan_items + k_emit2 where the tape takes the two-pass scheme)', 'kernel_ms': round(km, 4),
                                     'algorithmic_bytes_72B': 72 * t2, 'achieved_GBps_72B': round(72e-6 * t2 / km, 1) if km > 0 else None,
                                     'frac_72B': round(72e-6 * t2 / km / HBM_PEAK_GBS, 5) if km > 0 else None,
                                     'frac_36B': round(36e-6 * t2 / km / HBM_PEAK_GBS, 5) if km > 0 else None,
                                     'interpreted_voxel_share': round(float(s2.get('n_sampled_voxels', 0)) / max(float(s2['n_eval_voxels']), 1.0), 4)}
                else:               # what DESIGN.md section 6 expects of this config on N GPUs (arithmetic, next to the measured stage times)
                    o['expected_scaling_note'] = EXPECTED_SCALING.get(model)
                # every 997th triangle of the REFERENCE's soup at this size (tests/golden/full_*.npz, tools/make_golden_full.py; weave at
                # 2**33: every 9973rd, by the reference's per-batch function over all 266,256 batches, tools/make_golden_c4.py)
                # against the same triangles of the soup this run left on the device: positions, not only the count
                gold = {'gearlike': 'full_c3_gearlike_s30.npz', 'blobby': 'full_c5_blobby_s30.npz', 'weave': 'full_c4_weave_s33.npz'}.get(model)
                gold = os.path.join(ROOT, 'tests', 'golden', gold) if gold else None
                if gold and rank == 0 and os.path.exists(gold) and not args.no_check and r['state'].get('soup') is not None:
                    gd = np.load(gold)
                    stride, ref_tris = int(gd['sample_stride']), gd['sample_tris']
                    if 'seconds' in gd.files:   # the reference's own time for this configuration, as recorded when the fixture was made (build container, 8 vCPU)
                        o['reference_cpu'] = {'seconds': round(float(gd['seconds']), 1), 'workers': int(gd['processes']) if 'processes' in gd.files else 1,
                                              'what': ('sdf.core._worker over generate\'s job list in processes (tools/make_golden_c4.py)' if 'processes' in gd.files
                                                       else 'sdf.core.generate, workers=1 (tools/make_golden_full.py)')}
                    # the whole soup's sha256 against the reference's (blobby has no libm call: equal by construction; gearlike and
                    # weave go through sin / cos / atan2: a tolerance pin by contract -- whether this run happens to be bit-equal is printed)
                    o['soup_sha256_equals_reference'] = bool(int(gd['ntri']) == t2 and soup_sha(r['state']['soup'], t2) == bytes(gd['sha256']).hex())
                    if int(gd['ntri']) == t2:
                        mine = r['state']['soup'][:9 * t2].view(t2, 3, 3)[::stride].cpu().numpy()
                        extent = float(np.ptp(np.asarray(gd['bounds']), axis=0).max())
                        dev = float(np.abs(mine - ref_tris).max()) / extent if mine.shape == ref_tris.shape else None
                        o['reference_sampled_triangles'] = {'n': int(len(ref_tris)), 'stride': stride, 'max_abs_dev_over_extent': dev,
                                                            'coverage': round(float(len(ref_tris)) / max(t2, 1), 6) if dev is not None else 0.0,
                                                            'bit_equal_share': round(float((mine == ref_tris).mean()), 6) if dev is not None else None,
                                                            'within_1e-5': bool(dev <= 1e-5) if dev is not None else None}
                # ... and the WHOLE soup, coordinate by coordinate, against the CPU checker meshing the same grid on the host's cores
                # (single GPU, rank 0; weave at 2**33 is ~ 4400 core-seconds of checker: only where the host has the cores)
                if world == 1 and rank == 0 and not args.no_check and not args.no_cpu_baseline and r['state'].get('soup') is not None \
                        and (model != 'weave' or (os.cpu_count() or 1) >= 64):
                    trace('whole soup of %s against the checker' % model)
                    host = r['state']['soup'][:9 * t2].cpu().numpy().reshape(-1, 3)
                    # (weave at 2**33 is 166 s of checker on 128 cores -- measured, r05j: all 485,495,208 coordinates bit-equal; the default
                    # line compares the whole pieces of the grid that 40 s of checker finish -- a seeded random subset, `coverage` says how much of the
                    # soup that was -- SDF_BENCH_WHOLE_SOUP_S=600 the whole of it)
                    # (which rows belong to which batch: one more call, outside every timed region)
                    mo = eng.generate(r['f'], r['X'], r['Y'], r['Z'], 32, True)
                    offs = mo.batch_offsets()
                    mo.close()
                    o['whole_soup_vs_oracle'] = whole_soup_vs_oracle(model, core._estimate_bounds(r['f']), log2, host, offs,
                                                                     budget_s=float(os.environ.get('SDF_BENCH_WHOLE_SOUP_S', '40')))
                    del host, offs
                others.append(o)
                del r
            except Exception as e:          # (reported, never fatal for the headline line)
                others.append({'workload': '%s @ samples=2**%d' % (model, log2), 'error': repr(e)[:300]})
            if td is not None:
                td.barrier()
        return others

    def leave():
        """every rank tears the process group down at the same point (a rank that closes its connections while another is
        still alive has been seen to abort the survivor under gloo)"""
        if td is not None:
            td.barrier()
            td.destroy_process_group()

    want_others = not args.no_other_configs and args.model == 'example' and args.samples_log2 == 27 and args.precision == 'f64'
    if rank != 0:
        if want_others:
            watchdog(None)
            run_other_configs()
        leave()
        return

    # ---- roofline of the dominant kernel (k_mesh), from HIP events on the stream the kernel runs on ----
    # single GPU: the MEDIAN over the isolated calls above (the kernel alone on the device); with several calls in flight
    # the events around k_mesh also cover its wait for compute units the neighbouring calls hold: `pipelined`
    k_ms = iso['k_mesh_ms_hip_events']['median'] if world == 1 else float(np.mean(mesh_ms))
    shard_tris = int(st.get('n_triangles', tris)) if world == 1 else int(max(st.get('per_rank_triangles', [tris])))
    # fused design: the kernel's only HBM product is the ordered float64 soup, 9 doubles = 72 B per
    # triangle (SURVEY 8d counts 36 B for a float32 soup; the reference's soup is float64); the ranks of an
    # N-GPU job write the 16-byte slab records instead (csrc/sdf_slab.h)
    alg_bytes = (72.0 if world == 1 else 16.0) * shard_tris
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    plain, special = tape.tape.flop_estimate()
    eval_vox = int(st['n_eval_voxels']) if world == 1 else int(st['n_eval_voxels'] // world)
    # of those, the samples that went through the interpreter (the rest were decided by the interval
    # passes: k_cull); the flop estimate below is for the model's whole tape, before per-batch pruning
    sampled_vox = int(st.get('n_sampled_voxels', st['n_eval_voxels']))
    sampled_vox = sampled_vox if world == 1 else sampled_vox // world
    # HBM traffic of k_mesh per launch from the PMC passes of tools/profile.sh (separate rocprofv3
    # --pmc runs of this same command; FETCH_SIZE/WRITE_SIZE corrected as MI355X_MICROARCH.md says,
    # see tools/summarize_prof.py); the newest committed summary is used
    traffic = traffic_src = stale_src = None
    pmc_kernel = {}
    if args.model == 'example' and args.samples_log2 == 27 and world == 1 and args.precision == 'f64':
        import glob
        for prof in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')), reverse=True):
            try:                          # (the newest summary of THIS workload: same algorithmic bytes per launch)
                rec = json.load(open(prof))
                if abs(float(rec.get('algorithmic_bytes_per_launch') or 0) - alg_bytes) <= 1e-3 * alg_bytes:
                    # (only a summary taken on THIS source: the kernels' sources are hashed into it, tools/summarize_prof.py)
                    if rec.get('source_id') == engine.source_id():
                        traffic = rec.get('hbm_bytes_per_launch')
                        traffic_src = os.path.basename(prof)
                        pmc_kernel = (rec.get('kernels') or {}).get(rec.get('dominant_kernel') or '', {})
                    elif stale_src is None:
                        stale_src = '%s (source_id %s, this build %s)' % (os.path.basename(prof), rec.get('source_id'), engine.source_id())
            except Exception:
                traffic = None
            if traffic:
                break
    roofline = {
        'kernel': 'k_mesh<%s>' % ('double' if args.precision == 'f64' else 'float'),
        'bound': 'hbm', 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic,
        # the two ways of counting the soup: 72 B per triangle (float64, what the reference's generate() returns and this
        # kernel writes: `frac` above) and SURVEY.md 8(d)'s 36 B per triangle (a float32 soup)
        'frac_72B': round(achieved / HBM_PEAK_GBS, 6), 'frac_36B': round(0.5 * achieved / HBM_PEAK_GBS, 6),
        'traffic_over_algorithmic': round(traffic / alg_bytes, 3) if traffic else None,
        'source_id': engine.source_id(),
        'traffic_source': ('profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this source_id, committed; '
                           'not re-measured in this run)' % traffic_src) if traffic_src
                          else ('none for this source: the newest committed summary is stale -- %s' % stale_src if stale_src else None),
        'kernel_ms_source': ('median of the HIP-event times around k_mesh over the isolated synchronous calls of this run '
                             '(`isolated_calls`: min / median / max, and the same kernel by its own device clock)') if world == 1
                            else 'HIP events on the exchange lane: prepass + k_mesh of this rank\'s shard (mean over the timed steps)',
        'algorithmic_bytes_per_launch': alg_bytes, 'kernel_ms': round(k_ms, 4),
        'kernel_ms_pipelined': stats3(mesh_ms) if world == 1 else None,
        'pipelined': pipelined_overlap(res['spans'], res['dev_ms']) if world == 1 else None,
        'kernel_ms_device_clock_pipelined': stats3(res['dev_ms']) if world == 1 and res['dev_ms'] else None,
        'note': 'path is VALU/latency bound by construction (SURVEY 8d): see valu',
        'valu': {'eval_voxels_per_launch': eval_vox, 'interpreted_voxels_per_launch': sampled_vox,
                 'pruned_instr_fraction': round(st.get('n_pruned_instrs', 0) / st['n_batch_instrs'], 4) if st.get('n_batch_instrs') else None,
                 'flops_per_voxel_est': plain + special,
                 'achieved_tflops_est': round((plain + special) * sampled_vox / (k_ms * 1e-3) / 1e12, 3) if k_ms > 0 else 0,
                 'peak_tflops': FP64_VECTOR_PEAK_TFLOPS if args.precision == 'f64' else 157.3,
                 # from the committed counters of this source (same summary as `traffic`): vector-ALU wave-instructions per
                 # launch, and the share of the kernel's SIMD cycles they occupy at 4 cycles each (1024 SIMDs, the shader
                 # clock the kernel measured itself): how much of k_mesh is instruction issue
                 'valu_wave_instructions_per_launch': (pmc_kernel.get('SQ_INSTS_VALU') or {}).get('mean'),
                 # upper bound: every VALU wave-instruction charged 4 cycles (true for float64 arithmetic only: a SIMD is 16 lanes
                 # wide for it); lower bound: every one 2 cycles (MI355X_MICROARCH.md: a VALU instruction issues over 2 cycles);
                 # weighted: 4 cycles for the interpreter's float64 instructions (interpreted voxels x the tape's operations per
                 # voxel after pruning, sqrt / division sequences at ~ 20), 2 for everything else (marching, bookkeeping: int / f32)
                 'valu_issue_fraction': valu_issue(pmc_kernel, k_ms, iso, world, sampled_vox, plain, special, st)},
    }

    # ---- CPU baseline 1: the reference's own path (reference sdf/core.py:84-150, NumPy thread pool + skimage) ----
    # timed live when the reference and its interpreter exist on this box (the build container); the GPU boxes
    # have neither, there the committed numbers of the build-container run are reported with that provenance
    cpu_ref = None
    # (both CPU baselines: on rank 0 of the single-GPU run only)
    if not args.no_cpu_baseline and world == 1 and args.model == 'example' and args.samples_log2 == 27:
        cpu_ref = reference_cpu_baseline(args.samples_log2)

    # ---- CPU baseline 2: the C oracle (a port of the reference path), one core, same workload, always live ----
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle
        nb = st['batches']
        # bounded sample: ~10-30 s of single-core work
        frac = 1.0 if (args.model == 'example' and args.samples_log2 <= 27) else 0.1
        b1 = max(1, int(nb * frac))
        t2 = time.perf_counter()
        r = oracle.generate(f, X, Y, Z, 32, True, batch_range=(0, b1))
        cdt = time.perf_counter() - t2
        cpu = {'value': round(grid_voxels * (b1 / nb) / cdt, 1), 'unit': 'voxels/s', 'cores': 1, 'kind': 'port',
               'sample': 'oracle/sdf_oracle.c generate() on batches [0,%d) of %d of the same %dx%dx%d grid, '
                         '%.1f s on one host core of %d; triangles %d' % (b1, nb, len(X), len(Y), len(Z), cdt,
                                                                         os.cpu_count(), len(r.points) // 3),
               'triangles_per_sec': round(len(r.points) // 3 / cdt, 1)}

    out = {
        'metric': 'grid voxels/sec (sampled + meshed), canonical CSG example',
        'value': round(value, 1), 'unit': 'voxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': '%s @ samples=2**%d -> %dx%dx%d grid, sparse=True, batch_size=32'
                               % (args.model, args.samples_log2, len(X), len(Y), len(Z)),
                   'batches': int(st['batches']), 'skipped': int(st['skipped']), 'empty': int(st['empty']),
                   'nonempty': int(st['nonempty']), 'triangles': tris,
                   'parallelism': 'work-list shards x%d + one RCCL all-gather of the slabs per step' % world if world > 1 else 'single GPU'},
        'triangles_per_sec': round(tris * args.steps / dt, 1),
        'eval_voxels_per_sec': round(int(st['n_eval_voxels']) * args.steps / dt, 1),
        # (`value` counts every voxel of the grid -- the reference's unit of work; of those the skip test and the interval passes decide
        # all but these, which are the ones that go through the tape interpreter)
        'interpreted_voxels_per_sec': round(int(st.get('n_sampled_voxels', 0)) * args.steps / dt, 1),
        'interpreted_share_of_grid': round(int(st.get('n_sampled_voxels', 0)) / max(grid_voxels, 1), 5),
        'value_incl_d2h': round(incl, 1) if incl else None,
        'generate_e2e': e2e,
        'sustained': sustained,
        'steps_in_flight': DEPTH,
        'latency_ms_per_call': iso['wall_ms']['median'] if iso else None,
        'isolated_calls': iso,
        'clocks': {'idle_before': clocks_idle, 'after_timed_region': clocks_busy, 'after_isolated_calls': clocks_iso,
                   'sclk_mhz_in_kernel_pipelined': stats3(res['sclk']) if res['sclk'] else None,
                   'source': 'rocm-smi --showclocks; sclk_mhz_in_kernel: k_mesh workgroup 0, shader cycle counter / 100 MHz counter'},
        'device_ms': ({'prepass': iso['prepass_ms']['median'], 'mesh': round(k_ms, 4), 'mesh_pipelined': round(float(np.median(mesh_ms)), 4),
                       'emit': round(st.get('ms_emit', 0.0), 4)} if world == 1 else
                      {'per_rank_mesh': [round(float(v), 4) for v in per_rank[:, 0]],          # prepass + k_mesh of the rank's shard
                       'per_rank_exchange': [round(float(v), 4) for v in per_rank[:, 1]],      # the all-gather of the slabs
                       'per_rank_expand': [round(float(v), 4) for v in per_rank[:, 2]]}),      # slabs -> float64 soup
        'exchange_ms': None if world == 1 else round(float(per_rank[:, 1].max()), 4),
        'exchange': None if world == 1 else {'payload': st.get('payload'), 'slab_bytes': st.get('slab_bytes'), 'chunks': st.get('chunks'),
                                             'collectives_per_step': st.get('chunks'), 'host_syncs_per_step': 1,
                                             'driver': st.get('exchange', 'torch.distributed (%s)' % backend)},
        'parity_check': check,
        'parity_per_rank': parity_per_rank,
        'parity': {'soup_sha256': soup_hash, 'reference_sha256': EXAMPLE_S27_SHA256 if check is not None else None,
                   'what': 'sha256 of the float64 soup of the last timed step (copied from its device buffer after the '
                           'timed region) vs the unmodified reference on the same grid (tests/golden/full_c2_example_s27.npz)'},
        'roofline': roofline,
        # `cpu_baseline` = what was TIMED ON THIS BOX IN THIS RUN: the unmodified reference where it is installed (the build container:
        # kind 'reference'), else the C port of its algorithm used as checker (kind 'port', one core) -- the GPU boxes have no reference.
        # The reference's own numbers from the build container travel next to it (`cpu_reference_recorded`: another machine, said so).
        'cpu_baseline': cpu_ref if (cpu_ref is not None and cpu_ref.get('kind') == 'reference') else (cpu if cpu is not None else cpu_ref),
        'cpu_reference_recorded': cpu_ref if (cpu_ref is not None and cpu_ref.get('kind') != 'reference') else None,
        'cpu_port': cpu,
        'cpu_baseline_note': "kind 'port' on a GPU box is by the task's rules, not an omission: the reference is Python and may not travel in any form "
                             "(source, bytecode or otherwise), so oracle/_ref cannot hold it; where /root/reference exists (the build container) this object is the live reference",
        'other_configs': None,
    }
    if want_others:
        timer = watchdog(out)
        out['other_configs'] = run_other_configs()
        if timer is not None:
            timer.cancel()
    if guard is not None:
        guard.cancel()
    print(json.dumps(out), flush=True)
    leave()


if __name__ == '__main__':
    main()
