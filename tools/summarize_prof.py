#!/usr/bin/env python3
"""Condense the raw rocprofv3 CSVs of tools/profile.sh into the committed summaries:

    profiles/<tag>_kernel_stats.md     per-kernel calls / total / average / share (--kernel-trace --stats)
    profiles/<tag>_pmc.json            per-kernel PMC averages per launch + HBM traffic of the
                                       dominant kernel, corrected as MI355X_MICROARCH.md prescribes
                                       (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
                                       128-B requests at 64 B -> x2; WRITE_SIZE is calibrated against
                                       k_gather, whose written byte count is known exactly: 72 B/triangle)

usage: tools/summarize_prof.py <tag> <raw dir>
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'\s*\[clone.*', '', name)
    m = re.match(r'(?:void\s+)?([\w:]+(?:<[^(]*>)?)\(', name)
    return m.group(1) if m else name[:60]


def find(raw, sub, pattern):
    return sorted(glob.glob(os.path.join(raw, sub, '**', pattern), recursive=True))


def main():
    tag, raw = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    bench_line = None
    log = os.path.join(raw, 'stats.log')
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith('{"metric"'):
                bench_line = json.loads(ln)

    # ---- kernel stats ----
    rows = []
    for p in find(raw, 'stats', '*kernel_stats.csv'):
        rows += list(csv.DictReader(open(p)))
    md = ['# rocprofv3 --kernel-trace --stats: %s' % tag, '']
    if bench_line:
        args_file = os.path.join(raw, 'args.txt')
        cmd = open(args_file).read().strip() if os.path.exists(args_file) else '--steps %d --warmup %d --no-cpu-baseline' % (
            bench_line['steps'], bench_line['warmup'])
        md += ['command: `python bench.py %s`  (workload: %s; %d step(s) in flight)' % (
            cmd, bench_line['config']['workload'], bench_line.get('steps_in_flight', 1)), '',
            'bench line of the profiled run: value %.4g %s, %.4f ms/step, in-bench HIP-event k_mesh %.4f ms' % (
                bench_line['value'], bench_line['unit'], bench_line['ms_per_step'], bench_line['roofline']['kernel_ms']), '']
    md += ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    stats = {}
    for r in rows:
        n = short(r['Name'])
        stats[n] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3, 'total_ms': float(r['TotalDurationNs']) / 1e6}
        md.append('| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %.2f |' % (
            n, r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
            float(r['MaxNs']) / 1e3, float(r['Percentage'])))
    open(os.path.join(ROOT, 'profiles', '%s_kernel_stats.md' % tag), 'w').write('\n'.join(md) + '\n')

    # ---- PMC passes ----
    per = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [values per dispatch]
    for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2'):
        for p in find(raw, sub, '*counter_collection.csv'):
            for r in csv.DictReader(open(p)):
                per[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {'tag': tag, 'kernels': {}}
    for k, cs in per.items():
        out['kernels'][k] = {c: {'mean': sum(v) / len(v), 'n': len(v)} for c, v in cs.items()}
        if k in stats:
            out['kernels'][k]['avg_us'] = stats[k]['avg_us']
    tris = bench_line['config']['triangles'] if bench_line else None
    dom = next((k for k in per if 'k_mesh' in k), None)
    if dom:
        d = out['kernels'][dom]
        fetch_kib = d.get('FETCH_SIZE', {}).get('mean')
        write_kib = d.get('WRITE_SIZE', {}).get('mean')
        # WRITE_SIZE calibration: measured once on k_gather (profiles/r01b_pmc.json), a kernel whose
        # written byte count was known exactly (72 B per triangle, 8-byte stores per lane like k_mesh's
        # output stores): raw KiB * 1024 * 0.99897 = bytes.  k_gather no longer exists (k_mesh writes the
        # soup itself), so the factor is carried over.
        cal = 0.99897
        g = next((k for k in per if 'k_gather' in k), None)
        if g and tris and 'WRITE_SIZE' in out['kernels'][g]:
            cal = (72.0 * tris) / (out['kernels'][g]['WRITE_SIZE']['mean'] * 1024.0)
        hbm = None
        if fetch_kib is not None and write_kib is not None:
            hbm = fetch_kib * 1024.0 * 2.0 + write_kib * 1024.0 * (cal if cal else 1.0)
        out['dominant_kernel'] = dom
        out['fetch_bytes_per_launch_corrected'] = fetch_kib * 2048.0 if fetch_kib is not None else None
        out['write_bytes_per_launch_raw'] = write_kib * 1024.0 if write_kib is not None else None
        out['write_calibration_factor'] = cal
        out['hbm_bytes_per_launch'] = hbm
        out['algorithmic_bytes_per_launch'] = 72.0 * tris if tris else None
    # which source the counters were taken on: bench.py quotes a summary only for the build it belongs to
    sys.path.insert(0, ROOT)
    from sdf_amd import engine as _engine
    out['source_id'] = _engine.source_id()
    json.dump(out, open(os.path.join(ROOT, 'profiles', '%s_pmc.json' % tag), 'w'), indent=1, sort_keys=True)
    print('\n'.join(md))
    print(json.dumps({k: out.get(k) for k in ('dominant_kernel', 'hbm_bytes_per_launch', 'algorithmic_bytes_per_launch',
                                              'write_calibration_factor')}))


if __name__ == '__main__':
    main()
