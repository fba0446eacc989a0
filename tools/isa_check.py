#!/usr/bin/env python3
"""Post-build self-check of libsdf_hip.so (run by __graft_entry__.build(); cross-compiled code, no GPU needed):

    python tools/isa_check.py [path/to/libsdf_hip.so]

The tape interpreters dispatch through a scalar jump table (`s_setpc_b64` into a table of `s_branch`, csrc/sdf_interp.h) and are
built with `-mllvm -structurizecfg-skip-uniform-regions=1`; a toolchain that drops or changes that option produces kernels that
fault on the device, and one that applies it to the OTHER kernels has miscompiled a divergent loop before (k_expand, r03).  This
script takes the gfx950 code objects out of the library's offload bundles, disassembles them (llvm-objdump) and checks, per kernel:

  * every tape-interpreter kernel (k_mesh, k_skip*, k_eval_*, k_estimate_bounds) contains `s_setpc_b64` (the interval interpreters
    k_cull* / k_prune_list dispatch through a switch and are listed only);
  * no kernel of the plain translation units (k_compact, k_scan_*, k_emit2, k_expand, k_pack_slab, k_mc_*, k_field_*, k_stl,
    k_cast_f32, k_collect_headers) contains one -- they must not have been built with the interpreters' option;
  * every kernel the host launches is there at all;
  * the bounds estimate (k_estimate_bounds_w: 64 workgroups of ONE wave that meet through device memory, csrc/sdf_bounds.hip) contains no
    `s_barrier`.

It prints one line per kernel (instructions, s_setpc_b64, scratch_ instructions) and exits non-zero on a violation."""
import os
import re
import struct
import subprocess
import sys
import tempfile



def find_objdump():
    """llvm-objdump of the toolchain that built the library: next to $HIPCC's clang, under $ROCM_PATH, /opt/rocm, or on PATH"""
    import shutil
    cands = []
    hipcc = os.environ.get('HIPCC') or shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    for r in (root, os.environ.get('ROCM_PATH'), '/opt/rocm'):
        if r:
            cands += [os.path.join(r, 'lib', 'llvm', 'bin', 'llvm-objdump'), os.path.join(r, 'llvm', 'bin', 'llvm-objdump')]
    cands.append(shutil.which('llvm-objdump'))
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


OBJDUMP = find_objdump()
INTERP = ('k_mesh', 'k_skip', 'k_eval_points', 'k_eval_grid', 'k_eval_tiles', 'k_estimate_bounds')
INTERVAL = ('k_cull', 'k_prune_list')       # the interval-arithmetic interpreters (sdf_interval.h): a switch, no jump table
PLAIN = ('k_compact', 'k_scan_rows', 'k_scan_items', 'k_emit2', 'k_expand', 'k_pack_slab', 'k_mc_rows', 'k_mc_emit', 'k_field_rows',
         'k_field_emit', 'k_stl', 'k_cast_f32', 'k_collect_headers')


def code_objects(path):
    b = open(path, 'rb').read()
    for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', b):
        off = m.start()
        n = struct.unpack_from('<Q', b, off + 24)[0]
        p = off + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from('<QQQ', b, p)
            p += 24
            triple = b[p:p + ts].decode()
            p += ts
            if 'gfx950' in triple and sz:
                yield b[off + o:off + o + sz]


def kernels(obj_bytes):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(obj_bytes)
        f.flush()
        text = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', f.name], capture_output=True, text=True, check=True).stdout
    name, stats = None, {}
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:$', line)
        if m:
            name = m.group(1)
            stats[name] = [0, 0, 0, 0]
        elif name and '\t' in line:
            op = line.split('\t')[1].split()[0] if len(line.split('\t')) > 1 and line.split('\t')[1].strip() else ''
            if not op or op.startswith('.'):
                continue
            stats[name][0] += 1
            stats[name][1] += op == 's_setpc_b64'
            stats[name][2] += op.startswith('scratch_')
            stats[name][3] += op == 's_barrier'
    return stats


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sdf_amd', 'csrc', 'libsdf_hip.so')
    import shutil
    if OBJDUMP is None or shutil.which('c++filt') is None:   # nothing to check WITH is not a failed check
        print('isa_check: WARNING: %s not found -- the library was NOT checked' % ('llvm-objdump' if OBJDUMP is None else 'c++filt'), file=sys.stderr)
        return 0
    allk = {}
    for co in code_objects(path):
        allk.update(kernels(co))
    names = subprocess.run(['c++filt'], input='\n'.join(allk), capture_output=True, text=True).stdout.split('\n')
    bad, seen = [], set()
    for mangled, name in zip(allk, names):
        n, setpc, scratch, barriers = allk[mangled]
        base = re.sub(r'^void ', '', name).split('(')[0].split('<')[0].split('::')[-1]
        if not base.startswith('k_'):
            continue                                    # (outlined device functions: o_sin, mc33_triangle, ...)
        seen.add(base)
        kind = 'interp' if base.startswith(INTERP) else ('interval' if base.startswith(INTERVAL) else ('plain' if base in PLAIN else 'other'))
        print('%-8s %-90s %7d instr  s_setpc_b64 %3d  scratch_ %4d' % (kind, name[:90], n, setpc, scratch))
        if kind == 'interp' and setpc == 0:
            bad.append('%s: a tape interpreter WITHOUT its jump-table dispatch (s_setpc_b64)' % name[:100])
        if base == 'k_estimate_bounds_w' and barriers != 0:
            bad.append('%s: the bounds estimate is built of single-wave workgroups that synchronise through device memory only: an s_barrier in it '
                       'means its design was changed without this check' % name[:100])
        if kind == 'plain' and setpc != 0:
            bad.append('%s: a plain kernel WITH s_setpc_b64 (built with the interpreters\' structurizer option?)' % name[:100])
    for k in INTERP + INTERVAL + PLAIN:
        if not any(s == k or s.startswith(k) for s in seen):
            bad.append('%s: kernel missing from the library' % k)
    print('%d kernels in %s' % (len([1 for m in allk if re.search(r'k_[a-z]', m)]), path))
    for b in bad:
        print('ISA CHECK FAILED: ' + b, file=sys.stderr)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
