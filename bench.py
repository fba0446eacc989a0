#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sampling + meshing hot path.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): the canonical CSG example (reference examples/example.py:
sphere & box - 3 cylinders) on the 512^3 grid the reference builds for samples=2**27 over its own
estimated bounds, sparse=True, batch_size=32.  One "step" = one complete pass of the hot path:
skip prepass -> work list -> fused sample+march kernel -> ordered gather to the float64 (3T,3)
soup in HBM (and, for N>1, the RCCL all-gather of the rank soups).  The axes are the only input
(3 x 512 float64); the output stays in HBM inside the timed region (the PCIe-inclusive rate is
reported separately as `value_incl_d2h`).  float64 sampling = the reference's NumPy precision.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (k_mesh),
`cpu_baseline` = the reference's own CPU path (NumPy thread pool + skimage; live where the reference is
installed, else the committed build-container run, see `kind`), `cpu_port` = the C oracle timed live on one host core.
"""
import argparse
import json
import os
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
    import subprocess
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--model', default='example', help='example | gearlike | blobby | weave | knurling')
    ap.add_argument('--samples-log2', type=int, default=27, help='grid = samples=2**k through the reference step rule')
    ap.add_argument('--precision', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--sync', action='store_true', help='one step in flight: every call synchronises before the next is submitted')
    ap.add_argument('--inflight', type=int, default=4, help='steps in flight on a single GPU (each on a lane of its own; at most 4)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)')

    import torch
    # debugging aids for boxes with ONE GPU: SDF_BENCH_ONE_DEVICE=1 puts every rank on device 0,
    # SDF_BENCH_BACKEND=gloo exchanges through host memory (RCCL refuses two ranks on one device)
    backend = os.environ.get('SDF_BENCH_BACKEND', 'nccl')
    if os.environ.get('SDF_BENCH_ONE_DEVICE'):
        local_rank = 0
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
    eng.precision = engine.PRECISION_F64 if args.precision == 'f64' else engine.PRECISION_F32
    f, _ = build_model(args.model)
    tape = eng.tape_for(f)
    if args.model == 'example':
        bounds = EXAMPLE_BOUNDS
    else:
        bounds = core._estimate_bounds(f)
    X, Y, Z, step = core.grid_axes(bounds, samples=2 ** args.samples_log2)
    grid_voxels = len(X) * len(Y) * len(Z)
    dev = torch.device('cuda', local_rank)
    comm_dev = dev if backend == 'nccl' else torch.device('cpu')

    state = {}

    DEPTH = 1 if args.sync else max(1, min(args.inflight, 4))   # steps in flight (single GPU): step i+1 is submitted before step i is collected
    inflight = []

    def collect():
        mesh, buf = inflight.pop(0)
        mesh.wait()
        t = mesh.n_triangles
        if not mesh.emitted:               # (the soup did not fit: it was meshed again into library memory)
            big = torch.empty(max(t + t // 8, 1) * 9, dtype=torch.float64, device=dev)
            mesh.emit_device(big.data_ptr())
            state['bufs'] = [big if b is buf else b for b in state['bufs']]      # (identity, not tensor ==)
            buf = big
        state['last_buf'] = buf
        st = mesh.stats()
        state['stats'] = st
        state['tris'] = t
        mesh_ms.append(st['ms_mesh'])
        mesh.close()

    def collect_dist():
        soup, st = dist.collect_sharded(inflight.pop(0))
        state['buf'] = soup
        state['stats'] = st
        state['tris'] = st['triangles']
        mesh_ms.append(st['ms_mesh'])            # this rank: prepass + k_mesh of its shard, into the slab
        exch_ms.append((st['ms_exchange'], st['ms_expand']))

    def one_step():
        if world == 1:
            # every step writes its ordered float64 soup into a device buffer of its own (DEPTH of them
            # alternate), sized from the previous steps (first: a guess); the step is complete when its
            # counters are back on the host (collect)
            bufs = state.setdefault('bufs', [torch.empty(9 * (1 << 22), dtype=torch.float64, device=dev) for _ in range(DEPTH)])
            buf = bufs[state.get('n', 0) % DEPTH]
            state['n'] = state.get('n', 0) + 1
            while len(inflight) >= DEPTH:
                collect()
            inflight.append((eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel() // 9, wait=False), buf))
        else:
            # N > 1: two exchange steps in flight on lanes of their own: step i + 1's meshing runs under step i's
            # all-gather; a step is complete when its gathered headers are back on the host (collect_dist)
            while len(inflight) >= (1 if args.sync else 2):
                collect_dist()
            state['n'] = state.get('n', 0) + 1
            inflight.append(dist.submit_sharded(eng, tape, X, Y, Z, 32, True, device=comm_dev, lane=state['n'] % 2))

    def sync():
        while inflight:
            collect() if world == 1 else collect_dist()
        eng.synchronize()
        torch.cuda.synchronize()
        if td is not None:
            td.barrier()
            torch.cuda.synchronize()

    mesh_ms, exch_ms = [], []
    if world == 1:          # set-up, not a step: every call lane allocates its staging on first use (hundreds of MB each)
        for _ in range(DEPTH + 1):
            one_step()
        sync()
    for _ in range(args.warmup):
        one_step()
    sync()
    del mesh_ms[:], exch_ms[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()                                 # every one of the K steps is complete (collected) here
    dt = time.perf_counter() - t0
    assert len(mesh_ms) == args.steps
    per_rank = None
    if td is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank device times of the three stages of a step (means over the timed steps)
        mine = torch.tensor([float(np.mean(mesh_ms)), float(np.mean([e[0] for e in exch_ms])), float(np.mean([e[1] for e in exch_ms]))],
                            dtype=torch.float64, device=comm_dev)
        allr = torch.empty(3 * world, dtype=torch.float64, device=comm_dev)
        td.all_gather_into_tensor(allr, mine)
        per_rank = allr.cpu().numpy().reshape(world, 3)

    st = state['stats']
    tris = int(state['tris'])
    ms_per_step = 1e3 * dt / args.steps
    value = grid_voxels * args.steps / dt

    # latency of ONE call (submit -> counters back on the host), nothing else in flight
    latency_ms = None
    if world == 1:
        sync()
        n_lat = max(1, min(args.steps, 20))
        buf = state['bufs'][0]
        t1 = time.perf_counter()
        lat_mesh_ms, lat_pre_ms = [], []
        for _ in range(n_lat):
            mesh = eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel() // 9)
            lat_mesh_ms.append(mesh.stats()['ms_mesh'])
            lat_pre_ms.append(mesh.stats()['ms_prepass'])
            mesh.close()
        latency_ms = 1e3 * (time.perf_counter() - t1) / n_lat

    # PCIe-inclusive variant (single GPU): same step + D2H of the soup into the ndarray `generate` returns (a recycled pinned block)
    incl = None
    if world == 1:
        sync()
        n_incl = max(1, min(args.steps, 5))
        pts = None
        for i in range(2 + n_incl):          # (two untimed passes: the result blocks are pinned once, then recycled)
            if i == 2:
                t1 = time.perf_counter()
            mesh = eng.generate(tape, X, Y, Z, 32, True)
            pts = mesh.points()
            mesh.close()
        incl = grid_voxels * n_incl / (time.perf_counter() - t1)

    # parity check inside the bench run: the sha256 of the soup the LAST TIMED STEP left in its device buffer
    # (copied out after the timed region) against the hash of the reference's own soup on this grid
    check = None
    soup_sha = None
    if not args.no_check and rank == 0:
        import hashlib
        last = state.get('last_buf') if world == 1 else state.get('buf')   # the soup the last timed step left on the device
        if last is not None and tris * 9 <= last.numel():
            soup_sha = hashlib.sha256(last[:tris * 9].cpu().numpy().tobytes()).hexdigest()
        if args.model == 'example' and args.samples_log2 == 27 and args.precision == 'f64':
            check = bool(soup_sha == EXAMPLE_S27_SHA256 and
                         (st['batches'], st['skipped'], st['empty'], st['nonempty'], tris) == (4096, 2352, 120, 1624, 2945152))

    if rank != 0:
        if td is not None:
            td.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (k_mesh), from HIP events on the library's stream ----
    # (single GPU: the kernel's duration is taken from the calls that ran ALONE -- the latency loop above; in the timed
    # region two calls are in flight on lanes of their own and the events around k_mesh include its wait for the
    # compute units the previous call's k_mesh still holds; that figure is reported as `mesh_pipelined`)
    k_ms = float(np.mean(lat_mesh_ms)) if world == 1 else float(np.mean(mesh_ms))
    shard_tris = int(st.get('n_triangles', tris)) if world == 1 else int(max(st.get('per_rank_triangles', [tris])))
    # fused design: the kernel's only HBM product is the ordered float64 soup, 9 doubles = 72 B per
    # triangle (SURVEY 8d counts 36 B for a float32 soup; the reference's soup is float64)
    alg_bytes = 72.0 * shard_tris
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
    traffic = traffic_src = None
    if args.model == 'example' and args.samples_log2 == 27 and world == 1 and args.precision == 'f64':
        import glob
        for prof in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')), reverse=True):
            try:                          # (the newest summary of THIS workload: same algorithmic bytes per launch)
                rec = json.load(open(prof))
                if abs(float(rec.get('algorithmic_bytes_per_launch') or 0) - alg_bytes) <= 1e-3 * alg_bytes:
                    traffic = rec.get('hbm_bytes_per_launch')
                    traffic_src = os.path.basename(prof)
            except Exception:
                traffic = None
            if traffic:
                break
    roofline = {
        'kernel': 'k_mesh<%s>' % ('double' if args.precision == 'f64' else 'float'),
        'bound': 'hbm', 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic,
        'traffic_source': ('profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not re-measured '
                           'in this run)' % traffic_src) if traffic_src else None,
        'kernel_ms_source': 'HIP events around k_mesh in calls that ran alone (the latency loop of this run); with several '
                            'calls in flight the same events read %.4f ms because the kernel then shares the compute units '
                            'with the neighbouring calls\' prepass' % float(np.mean(mesh_ms)) if world == 1 else 'HIP events',
        'algorithmic_bytes_per_launch': alg_bytes, 'kernel_ms': round(k_ms, 4),
        'note': 'path is VALU/latency bound by construction (SURVEY 8d): see valu',
        'valu': {'eval_voxels_per_launch': eval_vox, 'interpreted_voxels_per_launch': sampled_vox,
                 'pruned_instr_fraction': round(st.get('n_pruned_instrs', 0) / st['n_batch_instrs'], 4) if st.get('n_batch_instrs') else None,
                 'flops_per_voxel_est': plain + special,
                 'achieved_tflops_est': round((plain + special) * sampled_vox / (k_ms * 1e-3) / 1e12, 3) if k_ms > 0 else 0,
                 'peak_tflops': FP64_VECTOR_PEAK_TFLOPS if args.precision == 'f64' else 157.3},
    }

    # ---- CPU baseline 1: the reference's own path (reference sdf/core.py:84-150, NumPy thread pool + skimage) ----
    # timed live when the reference and its interpreter exist on this box (the build container); the GPU boxes
    # have neither, there the committed numbers of the build-container run are reported with that provenance
    cpu_ref = None
    if not args.no_cpu_baseline and args.model == 'example' and args.samples_log2 == 27:
        cpu_ref = reference_cpu_baseline(args.samples_log2)

    # ---- CPU baseline 2: the C oracle (a port of the reference path), one core, same workload, always live ----
    cpu = None
    if not args.no_cpu_baseline:
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
                   'parallelism': 'work-list shards x%d + RCCL all-gather' % world if world > 1 else 'single GPU'},
        'triangles_per_sec': round(tris * args.steps / dt, 1),
        'eval_voxels_per_sec': round(int(st['n_eval_voxels']) * args.steps / dt, 1),
        'value_incl_d2h': round(incl, 1) if incl else None,
        'steps_in_flight': DEPTH if world == 1 else (1 if args.sync else 2),
        'latency_ms_per_call': round(latency_ms, 4) if latency_ms else None,
        'device_ms': ({'prepass': round(float(np.mean(lat_pre_ms)), 4), 'mesh': round(k_ms, 4), 'mesh_pipelined': round(float(np.mean(mesh_ms)), 4),
                       'emit': round(st.get('ms_emit', 0.0), 4)} if world == 1 else
                      {'per_rank_mesh': [round(float(v), 4) for v in per_rank[:, 0]],          # prepass + k_mesh of the rank's shard
                       'per_rank_exchange': [round(float(v), 4) for v in per_rank[:, 1]],      # the all-gather of the slabs
                       'per_rank_expand': [round(float(v), 4) for v in per_rank[:, 2]]}),      # slabs -> float64 soup
        'exchange_ms': None if world == 1 else round(float(per_rank[:, 1].max()), 4),
        'exchange': None if world == 1 else {'payload': st.get('payload'), 'slab_bytes': st.get('slab_bytes'), 'chunks': st.get('chunks'),
                                             'collectives_per_step': st.get('chunks'), 'host_syncs_per_step': 1},
        'parity_check': check,
        'parity': {'soup_sha256': soup_sha, 'reference_sha256': EXAMPLE_S27_SHA256 if check is not None else None,
                   'what': 'sha256 of the float64 soup of the last timed step (copied from its device buffer after the '
                           'timed region) vs the unmodified reference on the same grid (tests/golden/full_c2_example_s27.npz)'},
        'roofline': roofline,
        'cpu_baseline': cpu_ref if cpu_ref is not None else cpu,
        'cpu_port': cpu,
    }
    print(json.dumps(out))
    if td is not None:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
