#!/usr/bin/env python3
"""Static instruction mix of the device functions in a gfx950 assembly listing (hipcc --save-temps):

    tools/isa_mix.py FILE.s [name-substring ...]

For each function whose demangled name contains one of the substrings: number of VALU / SALU / branch / LDS /
vector-memory / scalar-memory instructions and the exec-mask operations (divergent control flow).  A static count
says nothing about how often a block runs; it is for comparing two builds of the same function.
"""
import re
import subprocess
import sys
from collections import Counter


def functions(path):
    name, body = None, []
    for line in open(path, errors='replace'):
        m = re.match(r'^([A-Za-z_][\w.$]*):\s*(;.*)?$', line)
        if m and not m.group(1).startswith('.L'):
            if name:
                yield name, body
            name, body = m.group(1), []
        elif line.lstrip().startswith('.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
            if name:
                yield name, body
            name, body = None, []
        elif name and re.match(r'^\s+[a-z]', line):
            body.append(line.split()[0])
    if name:
        yield name, body


def classify(op):
    if op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc')):
        return 'branch'
    if 'saveexec' in op or (op.startswith(('s_or_b64', 's_andn2_b64', 's_and_b64', 's_xor_b64', 's_mov_b64'))):
        return 'salu64/exec'
    if op.startswith(('s_load', 's_buffer_load')):
        return 'smem'
    if op.startswith(('s_waitcnt', 's_nop')):
        return 'wait/nop'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        return op.split('_')[0]
    if op.startswith('v_'):
        return 'valu'
    return 'other'


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    fns = list(functions(path))
    names = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in fns), capture_output=True, text=True).stdout.split('\n')
    for (mangled, body), name in zip(fns, names):
        if not body or (pats and not any(p in name for p in pats)):
            continue
        c = Counter(classify(o) for o in body)
        print('%-70s %6d instr | ' % (name[:70], len(body)) + ' '.join('%s %d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == '__main__':
    main()
