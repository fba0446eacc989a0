"""Lowering of the expression tree to the flat op tape of the HIP interpreter.

The reference evaluates a model by recursive Python closure calls over whole NumPy arrays
(reference sdf/d3.py:24-25 and SURVEY.md 3.4).  The device evaluates it with a small
register machine, one lane per sample (csrc/sdf_interp.h):

    p      current point (3 scalars; 2-D nodes use x, y)
    acc    current distance
    PS[s]  saved points / per-node scratch triples      (static slot numbers)
    DS[s]  saved distances / per-node scratch scalars   (static slot numbers)

The tape is a list of fixed-size instructions ``(op, post, a, b, const_off)`` packed in two
u32 words, plus one float64 constant pool.  Because the tape is straight-line code, every
stack position is known at lowering time, so instructions name their slots explicitly and
the kernel never keeps a dynamic stack pointer.

Value-producing instructions (all leaves and COMB) fold their value ``v`` into the
accumulator through ``post``:

    SET   acc = v                     UNION  acc = min(acc, v)       SUNION  smooth, K
    DIFF  acc = max(acc, -v)          INTER  acc = max(acc, v)       SDIFF / SINTER / BLEND

which is what removes most stack traffic: ``a | b.translate(t) | c`` lowers to
``A; SAVE_P; TRANSLATE; B[post=UNION]; LOAD_P; C[post=UNION]`` without touching DS.
The constant at ``const_off`` is K (always reserved); leaf parameters follow it.
"""
from array import array

import math

import numpy as np

from . import dn
from .ir import Node, unwrap, resolved_k, NODE_OPS

# ---- machine opcodes (shared with csrc/opcodes.h through tools/gen_headers.py) -----------
_LEAVES = [
    'sphere', 'plane', 'box', 'rounded_box', 'wireframe_box', 'torus', 'capsule', 'cylinder',
    'capped_cylinder', 'rounded_cylinder', 'capped_cone', 'rounded_cone', 'ellipsoid',
    'pyramid', 'tetrahedron', 'octahedron', 'dodecahedron', 'icosahedron',
    'circle', 'line', 'rectangle', 'rounded_rectangle', 'equilateral_triangle', 'hexagon',
    'rounded_x', 'polygon', 'vesica', 'texture2d', 'grid3d', 'extern',
]
_MACHINE = (
    ['END'] + ['L_' + n.upper() for n in _LEAVES] + [
        'COMB',            # v = acc, d1 = DS[a]: acc = post(DS[a], acc)
        # point ops
        'TRANSLATE', 'SCALE', 'ROTATE', 'ELONGATE', 'TWIST', 'BEND', 'BEND_LINEAR',
        'BEND_RADIAL', 'WRAP_AROUND', 'CIRC_PREP', 'CIRC_SET', 'REP_PREP', 'REP_SET',
        'TRANSLATE2', 'SCALE2', 'ROTATE2', 'ELONGATE2', 'REVOLVE', 'SETZ0',
        'SAVE_P', 'LOAD_P',
        # distance ops
        'PUSH_D', 'NEG', 'ADDC', 'SUBC', 'MULC', 'SHELL', 'ADD_DS',
        'TRANS_LIN_PRE', 'TRANS_RAD_PRE', 'TRANS_MIX',
        'EXT_PRE', 'EXT_POST', 'EXTTO_PRE', 'EXTTO_MIX', 'SLICE_POST',
        'NOP',             # only its prefixes act (what the interval prepass leaves of a skipped instruction)
    ])
OP = {name: i for i, name in enumerate(_MACHINE)}
OP_NAMES = _MACHINE

POST = {'SET': 0, 'UNION': 1, 'DIFF': 2, 'INTER': 3, 'SUNION': 4, 'SDIFF': 5, 'SINTER': 6, 'BLEND': 7}
POST_NAMES = list(POST)

# An instruction is two u32 words:
#   word 0:  op[0:8] | post[8:11] | RL[11] rl_slot[12:15] | SV[15] sv_slot[16:19] | PD[19] pd_slot[20:23] | a[24:32]
#   word 1:  const offset[0:24] | b[24:32]
# RL / SV / PD are PREFIXES executed before the op, in this order (see peephole()):
#   RL: p = PS[rl_slot]      (what a LOAD_P in front of the instruction would do)
#   SV: PS[sv_slot] = p      (SAVE_P)
#   PD: DS[pd_slot] = acc    (PUSH_D)
# so the bookkeeping ops ride on their neighbours instead of paying a dispatch each.
A_SHIFT = 24
B_SHIFT = 24
COFF_MASK = (1 << 24) - 1
RL_FLAG, RL_SHIFT = 1 << 11, 12
SV_FLAG, SV_SHIFT = 1 << 15, 16
PD_FLAG, PD_SHIFT = 1 << 19, 20
PREFIX_MASK = 0x00FFF800

PRUNE_MAX_INSTR = 256     # csrc/sdf_hip.hip generate_impl: `pruning` needs n_instr <= 256

# hard limits of the default kernel build (csrc/sdf_interp.h NP_SLOTS / ND_SLOTS)
MAX_P_SLOTS = 8
MAX_D_SLOTS = 8

_PURE_TRANSFORMS = {
    # node op -> machine op, for transforms that only rewrite p (no post-processing of acc)
    'translate': 'TRANSLATE', 'rotate': 'ROTATE', 'twist': 'TWIST', 'bend': 'BEND',
    'bend_linear': 'BEND_LINEAR', 'bend_radial': 'BEND_RADIAL', 'wrap_around': 'WRAP_AROUND',
    'translate2': 'TRANSLATE2', 'rotate2': 'ROTATE2', 'revolve': 'REVOLVE',
}
# machine ops that rewrite the current point
_POINT_WRITERS = {'TRANSLATE', 'SCALE', 'ROTATE', 'ELONGATE', 'TWIST', 'BEND', 'BEND_LINEAR', 'BEND_RADIAL',
                  'WRAP_AROUND', 'CIRC_SET', 'REP_SET', 'TRANSLATE2', 'SCALE2', 'ROTATE2', 'ELONGATE2', 'REVOLVE',
                  'SETZ0'}
_BOOL_POST = {
    'union': ('UNION', 'SUNION'), 'difference': ('DIFF', 'SDIFF'),
    'intersection': ('INTER', 'SINTER'), 'blend': (None, 'BLEND'),
}

# rough per-sample arithmetic weights (adds/muls/compares = 1, fma = 2; sqrt/div/trig listed
# separately) used only for the VALU-side figure bench.py prints
_FLOPS = {
    'L_SPHERE': (10, 1), 'L_PLANE': (8, 0), 'L_BOX': (22, 1), 'L_ROUNDED_BOX': (26, 1),
    'L_WIREFRAME_BOX': (70, 3), 'L_TORUS': (9, 2), 'L_CAPSULE': (22, 2), 'L_CYLINDER': (5, 1),
    'L_CAPPED_CYLINDER': (40, 3), 'L_ROUNDED_CYLINDER': (16, 2), 'L_CAPPED_CONE': (50, 5),
    'L_ROUNDED_CONE': (25, 3), 'L_ELLIPSOID': (20, 9), 'L_PYRAMID': (45, 3),
    'ROTATE': (15, 0), 'TRANSLATE': (3, 0), 'CIRC_PREP': (10, 4), 'CIRC_SET': (6, 2),
    'BEND_LINEAR': (20, 1), 'REP_PREP': (9, 3), 'REP_SET': (9, 0),
}


class Tape:
    """a lowered model: ``code`` (uint32, 2 words per instruction), ``consts`` (float64)"""

    def __init__(self, code, consts, n_pslots, n_dslots, dim):
        self.code = np.ascontiguousarray(code, dtype=np.uint32)
        self.consts = np.ascontiguousarray(consts, dtype=np.float64)
        self.n_pslots = n_pslots
        self.n_dslots = n_dslots
        self.dim = dim
        self.rstart = self.lstart = None
        # user closures the tape reads through its L_EXTERN leaves: [(callable, dimension of its points)],
        # leaf k of the tape = entry k
        self.externs = []

    @property
    def n_instr(self):
        return len(self.code) // 2

    def disassemble(self):
        out = []
        for i in range(self.n_instr):
            w0, w1 = int(self.code[2 * i]), int(self.code[2 * i + 1])
            pre = ''
            if w0 & RL_FLAG:
                pre += 'p<-PS[%d]; ' % ((w0 >> RL_SHIFT) & 7)
            if w0 & SV_FLAG:
                pre += 'PS[%d]<-p; ' % ((w0 >> SV_SHIFT) & 7)
            if w0 & PD_FLAG:
                pre += 'DS[%d]<-acc; ' % ((w0 >> PD_SHIFT) & 7)
            out.append('%3d  %s%-14s post=%-6s a=%d b=%d c@%d' % (
                i, pre, OP_NAMES[w0 & 255], POST_NAMES[(w0 >> 8) & 7], w0 >> A_SHIFT, w1 >> B_SHIFT, w1 & COFF_MASK))
        return '\n'.join(out)

    def flop_estimate(self):
        """(plain ops, sqrt/div/transcendental ops) per evaluated sample"""
        plain = special = 0
        for i in range(self.n_instr):
            name = OP_NAMES[int(self.code[2 * i]) & 255]
            a, b = _FLOPS.get(name, (4, 0))
            plain += a
            special += b
            if (int(self.code[2 * i]) >> 8) & 7 >= 4:
                plain += 12
                special += 1
            elif (int(self.code[2 * i]) >> 8) & 7:
                plain += 1
        return plain, special


class _Lowering:
    def __init__(self):
        self.code = []
        self.consts = array('d')
        self.pdepth = self.ddepth = 0
        self.pmax = self.dmax = 0
        # value numbering of the current point: every instruction that rewrites p gives it a new
        # version; a PS slot remembers the version it holds.  A construct that must preserve p
        # first looks for a live slot that already holds the current version (e.g. the save of an
        # enclosing boolean) and borrows it instead of saving the same point again.
        self.pver = 0
        self.next_ver = 1
        self.slot_ver = {}
        # per instruction: (first instruction of the RIGHT operand, first instruction of the LEFT
        # operand chain) for instructions that combine two values with a hard min / max -- what the
        # interval prepass needs to turn "this operand never wins in this batch" into a range of
        # instructions to skip (csrc/sdf_prune.h)
        self.meta = []
        self.chain = []               # stack: index of the first instruction of the enclosing chain
        self.externs = []             # user closures, in leaf order

    # -- emission helpers --
    def emit(self, op, post='SET', a=0, b=0, consts=(), K=0.0, blob=None):
        off = len(self.consts)
        self.consts.append(float(K))
        self.consts.extend(float(c) for c in consts)
        if blob is not None:          # bulk constants (a sampled field) behind the parameters
            self.consts.frombytes(np.ascontiguousarray(blob, dtype=np.float64).tobytes())
        assert 0 <= a < 256 and 0 <= b < 256
        if len(self.consts) > COFF_MASK:
            raise ValueError('model needs more than %d float64 constants (sampled fields included): the tape '
                             'addresses its constant pool with 24 bits' % COFF_MASK)
        self.code.append(OP[op] | (POST[post] << 8) | (a << A_SHIFT))
        self.code.append(off | (b << B_SHIFT))
        self.meta.append(None)
        if op in _POINT_WRITERS:
            self.pver = self.next_ver
            self.next_ver += 1
        elif op == 'LOAD_P':
            self.pver = self.slot_ver[a]
        elif op == 'SAVE_P':
            self.slot_ver[a] = self.pver
        elif op in ('REP_PREP', 'CIRC_PREP'):
            self.slot_ver[a] = None           # the slot holds indices / polar coordinates, not a point

    def save_point(self):
        """a live slot holding the current point: (slot, owned).  Borrows an enclosing construct's
        slot when one already holds this version of p, else allocates one and emits SAVE_P."""
        for s in range(self.pdepth):
            if self.slot_ver.get(s) == self.pver:
                return s, False
        s = self.palloc()
        self.emit('SAVE_P', a=s)
        return s, True

    def note_combine(self, post, rstart):
        """the instruction just emitted folds a right operand (instructions rstart..here-1) into a
        left one (instructions chain start..rstart-1) with a hard or polynomial-smooth min / max"""
        if post in ('UNION', 'DIFF', 'INTER', 'SUNION', 'SDIFF', 'SINTER') and rstart is not None and self.chain:
            self.meta[-1] = (rstart, self.chain[-1])

    def palloc(self):
        s = self.pdepth
        self.pdepth += 1
        self.pmax = max(self.pmax, self.pdepth)
        if self.pdepth > MAX_P_SLOTS:
            raise ValueError('model needs more than %d saved-point slots on the device' % MAX_P_SLOTS)
        return s

    def pfree(self):
        self.pdepth -= 1

    def dalloc(self):
        s = self.ddepth
        self.ddepth += 1
        self.dmax = max(self.dmax, self.ddepth)
        if self.ddepth > MAX_D_SLOTS:
            raise ValueError('model needs more than %d saved-distance slots on the device' % MAX_D_SLOTS)
        return s

    def dfree(self):
        self.ddepth -= 1

    # -- analysis --
    def transparent(self, obj):
        """True when evaluating obj never needs the accumulator for itself, so its final
        leaf can fold straight into the caller's accumulator"""
        n = unwrap(obj)
        if 'L_' + n.op.upper() in OP:
            return True
        if n.op in _PURE_TRANSFORMS:
            return self.transparent(n.children[0])
        return False

    # -- lowering --
    def here(self):
        return len(self.code) // 2

    def value(self, obj, dim, post='SET', K=0.0, _rs=None):
        """emit code that folds obj(p) into acc with `post`; returns True if p is clobbered.
        `_rs`: index of the first instruction that belongs to this operand (default: here)"""
        n = unwrap(obj)
        if post != 'SET' and _rs is None:
            _rs = self.here()
        if post != 'SET' and not self.transparent(n):
            s = self.dalloc()
            self.emit('PUSH_D', a=s)
            dirty = self.value(n, dim, 'SET')
            self.emit('COMB', post, a=s, K=K)
            self.note_combine(post, _rs)
            self.dfree()
            return dirty
        op = n.op
        if n.dim and n.dim != dim:
            raise TypeError('%s is a %d-D node used on %d-D points' % (op, n.dim, dim))
        leaf = 'L_' + op.upper()
        if op == 'extern':            # a user closure: the device reads its value at this leaf's point from a buffer
            self.emit(leaf, post, consts=[len(self.externs)], K=K)
            self.externs.append((n.meta['fn'], dim))
            self.note_combine(post, _rs)
            return False
        if leaf in OP:
            self.emit(leaf, post, consts=n.params, K=K, blob=(n.meta or {}).get('blob'))
            self.note_combine(post, _rs)
            return False
        if op in _PURE_TRANSFORMS:
            self.emit(_PURE_TRANSFORMS[op], consts=n.params)
            self.value(n.children[0], 2 if op == 'revolve' else dim, post, K, _rs=_rs)
            return True
        # everything below runs with post == 'SET'
        if op in _BOOL_POST:
            return self.boolean(n, dim)
        if op == 'scale' or op == 'scale2':
            self.emit('SCALE' if op == 'scale' else 'SCALE2', consts=n.params)
            self.value(n.children[0], dim)
            self.emit('MULC', consts=[n.params[-1]])
            return True
        if op == 'elongate' or op == 'elongate2':
            s = self.dalloc()
            self.emit('ELONGATE' if op == 'elongate' else 'ELONGATE2', a=s, consts=n.params)
            self.value(n.children[0], dim)
            self.emit('ADD_DS', a=s)
            self.dfree()
            return True
        if op == 'negate':
            d = self.value(n.children[0], dim)
            self.emit('NEG')
            return d
        if op in ('dilate', 'erode', 'shell'):
            d = self.value(n.children[0], dim)
            self.emit({'dilate': 'SUBC', 'erode': 'ADDC', 'shell': 'SHELL'}[op], consts=n.params)
            return d
        if op == 'circular_array':
            s = self.palloc()
            # (behind da / delta: what the float64 interpreter's rotation form needs -- the search's rotations, cos / sin of
            # delta -- and whether it applies (pi / 4096 <= da <= pi: 2 .. 8192 sectors), csrc/sdf_interp.h L_CIRC_PREP; the
            # interval forms and the float32 path read c[0] only)
            da = float(n.params[0])
            rot = 1.0 if (0.0 < da <= math.pi and da >= math.pi / 4096.0) else 0.0     # (else: the reference's polar form, PREP and SET alike)
            # Since r04 the float64 interpreter finds the sector WITHOUT atan2 / sincos: the point is turned back by 2^m da,
            # m = M .. 0 (2^M da <= pi < 2^(M+1) da), whenever that leaves it on the counter-clockwise side -- a binary search
            # of k = floor(angle / da) whose by-product is the point turned by -k da, which is all CIRC_SET needs.  The
            # rotations' cos / sin are constants of the instruction: c[2] = M + 1, then (cos, sin)(2^m da) for m = 0 .. M.
            prep = [da, 0.0]
            if rot:
                M = int(math.floor(math.log2(math.pi / da)))
                while (2.0 ** (M + 1)) * da <= math.pi:
                    M += 1
                while (2.0 ** M) * da > math.pi:
                    M -= 1
                prep = [da, 2.0, float(M + 1)]
                for m in range(M + 1):
                    ang = (2.0 ** m) * da            # (exact: a power of two times da)
                    prep += [math.cos(ang), math.sin(ang)]
            self.emit('CIRC_PREP', a=s, consts=prep)
            self.chain.append(self.here())
            self.emit('CIRC_SET', a=s, consts=[da, float(np.cos(da)), float(np.sin(da)), rot])
            self.value(n.children[0], dim)
            rs = self.here()
            self.emit('CIRC_SET', a=s, consts=[0.0, 1.0, 0.0, rot])
            self.value(n.children[0], dim, 'UNION', _rs=rs)
            self.chain.pop()
            self.pfree()
            return True
        if op == 'repeat':
            prm = dn.repeat_params(n, dim)
            nn = int(prm[8])
            s0, owned = self.save_point()
            s1 = self.palloc()
            self.emit('REP_PREP', a=s1, consts=prm[:8])
            self.chain.append(self.here())
            for k in range(nn):
                rs = self.here()
                self.emit('REP_SET', a=s0, b=s1, consts=list(prm[1:4]) + list(prm[9 + 3 * k: 12 + 3 * k]))
                self.value(n.children[0], dim, 'SET' if k == 0 else 'UNION', _rs=rs if k else None)
            self.chain.pop()
            self.pfree()
            if owned:
                self.pfree()
            return True
        if op in ('transition_linear', 'transition_radial'):
            st = self.dalloc()
            self.emit('TRANS_LIN_PRE' if op == 'transition_linear' else 'TRANS_RAD_PRE', a=st, consts=n.params)
            s1 = self.dalloc()
            dirty = self.pair(n.children[0], n.children[1], dim, s1)
            self.emit('TRANS_MIX', a=st, b=s1)
            self.dfree()
            self.dfree()
            return dirty
        if op == 'extrude':
            s = self.dalloc()
            self.emit('EXT_PRE', a=s, consts=n.params)
            self.value(n.children[0], 2)
            self.emit('EXT_POST', a=s)
            self.dfree()
            return True
        if op == 'extrude_to':
            s0 = self.dalloc()
            s1 = self.dalloc()
            s2 = self.dalloc()
            self.emit('EXT_PRE', a=s0, consts=[n.params[1]])
            self.emit('EXTTO_PRE', a=s1, consts=[n.params[0], n.params[2]])
            self.pair(n.children[0], n.children[1], 2, s2)
            self.emit('EXTTO_MIX', a=s1, b=s2)
            self.emit('EXT_POST', a=s0)
            self.dfree(); self.dfree(); self.dfree()
            return True
        if op == 'slice':
            s = self.dalloc()
            self.emit('SETZ0')
            self.pair(n.children[0], n.children[1], 3, s)
            self.emit('SLICE_POST', a=s)
            self.dfree()
            return True
        raise NotImplementedError(op)

    def pair(self, a, b, dim, dslot):
        """acc = b(p) with a(p) parked in DS[dslot]; both see the same p"""
        na = unwrap(a)
        need_save = self.clobbers(na)
        sp, owned = None, False
        if need_save:
            sp, owned = self.save_point()
        self.value(na, dim)
        self.emit('PUSH_D', a=dslot)
        if need_save:
            self.emit('LOAD_P', a=sp)
        dirty = self.value(b, dim)
        if owned:
            self.pfree()
        return dirty

    def clobbers(self, obj):
        n = unwrap(obj)
        if 'L_' + n.op.upper() in OP:
            return False
        if n.op in _BOOL_POST or n.op in ('negate', 'dilate', 'erode', 'shell',
                                          'transition_linear', 'transition_radial'):
            return any(self.clobbers(c) for c in n.children)
        return True

    def boolean(self, n, dim):
        kinds = _BOOL_POST[n.op]
        Ks = resolved_k(n)
        kids = n.children
        dirty_flags = [self.clobbers(c) for c in kids]
        save = any(dirty_flags[:-1])
        sp, owned = None, False
        dirty = False
        self.chain.append(self.here())
        for i, c in enumerate(kids):
            if save and sp is None and dirty_flags[i] and i < len(kids) - 1:
                sp, owned = self.save_point()
            if dirty:
                self.emit('LOAD_P', a=sp)
                dirty = False
            if i == 0:
                dirty = self.value(c, dim)
            else:
                K = Ks[i - 1]
                if K is None:
                    if kinds[0] is None:
                        raise TypeError("unsupported operand type(s) for *: 'NoneType' and 'float'")
                    dirty = self.value(c, dim, kinds[0])
                else:
                    dirty = self.value(c, dim, kinds[1], float(K))
        self.chain.pop()
        if owned:
            self.pfree()
        return dirty


def peephole(code, meta=None):
    """dispatch-count reduction on a lowered tape (list of u32 words, 2 per instruction); the
    arithmetic per sample is unchanged, so results stay bit-identical:

    * `SAVE_P a; SAVE_P b` back to back store the same point twice: the second store is dropped and
      later readers of slot b read slot a instead (slot lifetimes nest, a outlives b);
    * LOAD_P / SAVE_P / PUSH_D are folded into the NEXT instruction as its RL / SV / PD prefixes
      whenever the fixed prefix order (RL, SV, PD, then the op) reproduces the original order.
    """
    ins = [[int(code[i]), int(code[i + 1]), i // 2] for i in range(0, len(code), 2)]   # + original index
    op = lambda w: w & 255
    sa = lambda w: w >> A_SHIFT
    # 1. duplicate saves.  Slots written by SAVE_P are read by LOAD_P (a) and REP_SET (a).
    writers = (OP['SAVE_P'], OP['REP_PREP'], OP['CIRC_PREP'])
    readers = (OP['LOAD_P'], OP['REP_SET'])

    def dedupe(ins):
        out, alias = [], {}
        for k, (w0, w1, orig) in enumerate(ins):
            o, s = op(w0), sa(w0)
            if o in writers:
                for bslot in [x for x, tgt in alias.items() if tgt == s]:
                    # slot s is rewritten: an alias to it may only be dropped if it is dead, i.e.
                    # not read again before it is written again
                    for v0, _, _ in ins[k + 1:]:
                        if sa(v0) == bslot and op(v0) in readers:
                            return None
                        if sa(v0) == bslot and op(v0) in writers:
                            break
                    del alias[bslot]
                alias.pop(s, None)
                if o == OP['SAVE_P'] and out and op(out[-1][0]) == OP['SAVE_P']:
                    alias[s] = sa(out[-1][0])      # (out[-1] is never itself an alias: aliases are dropped)
                    continue
            elif o in readers and s in alias:
                w0 = (w0 & ~(255 << A_SHIFT)) | (alias[s] << A_SHIFT)
            out.append([w0, w1, orig])
        return out

    ins = dedupe(ins) or ins
    # 2. bookkeeping ops -> prefixes of the following instruction
    out = []
    rl = sv = pd = None            # pending prefixes (slot numbers)
    pending_src = []               # the original instructions the pending prefixes came from

    def flush_standalone():
        nonlocal rl, sv, pd, pending_src
        out.extend(pending_src)
        rl = sv = pd = None
        pending_src = []

    for w0, w1, orig in ins:
        o, s = op(w0), sa(w0)
        if s < 8 and not (w0 & PREFIX_MASK):
            if o == OP['LOAD_P']:
                if sv is not None:
                    flush_standalone()          # SAVE then LOAD cannot be expressed as RL-then-SV
                elif rl is not None:
                    pending_src = [x for x in pending_src if op(x[0]) != OP['LOAD_P']]   # first reload is dead
                rl = s
                pending_src.append([w0, w1, orig])
                continue
            if o == OP['SAVE_P']:
                if sv is not None:
                    flush_standalone()
                sv = s
                pending_src.append([w0, w1, orig])
                continue
            if o == OP['PUSH_D']:
                if pd is not None:
                    flush_standalone()
                pd = s
                pending_src.append([w0, w1, orig])
                continue
        if pending_src:
            if o == OP['END']:
                flush_standalone()              # keep them as instructions in front of END
            else:
                if rl is not None:
                    w0 |= RL_FLAG | (rl << RL_SHIFT)
                if sv is not None:
                    w0 |= SV_FLAG | (sv << SV_SHIFT)
                if pd is not None:
                    w0 |= PD_FLAG | (pd << PD_SHIFT)
                rl = sv = pd = None
                pending_src = []
        out.append([w0, w1, orig])
    new_code = [w for w0, w1, _ in out for w in (w0, w1)]
    if meta is None:
        return new_code
    # carry the combine ranges over: an old index maps to the first kept instruction at or after it
    import bisect
    origs = [o for _, _, o in out]
    remap = lambda old: bisect.bisect_left(origs, old)
    new_meta = []
    for _, _, o in out:
        m = meta[o]
        new_meta.append(None if m is None else (remap(m[0]), remap(m[1])))
    return new_code, new_meta


# which instruction fields name PS / DS slots (for the slot census after the peephole pass)
_P_FIELDS = {'SAVE_P': 'a', 'LOAD_P': 'a', 'REP_PREP': 'a', 'REP_SET': 'ab', 'CIRC_PREP': 'a', 'CIRC_SET': 'a'}
_D_FIELDS = {'COMB': 'a', 'PUSH_D': 'a', 'ELONGATE': 'a', 'ELONGATE2': 'a', 'ADD_DS': 'a', 'TRANS_LIN_PRE': 'a',
             'TRANS_RAD_PRE': 'a', 'TRANS_MIX': 'ab', 'EXT_PRE': 'a', 'EXT_POST': 'a', 'EXTTO_PRE': 'a',
             'EXTTO_MIX': 'ab', 'SLICE_POST': 'a'}


def slot_census(code):
    """(number of PS slots, number of DS slots) a tape really touches"""
    np_, nd = 0, 0
    for i in range(0, len(code), 2):
        w0, w1 = int(code[i]), int(code[i + 1])
        name = OP_NAMES[w0 & 255]
        fields = {'a': w0 >> A_SHIFT, 'b': w1 >> B_SHIFT}
        for f in _P_FIELDS.get(name, ''):
            np_ = max(np_, fields[f] + 1)
        for f in _D_FIELDS.get(name, ''):
            nd = max(nd, fields[f] + 1)
        if w0 & RL_FLAG:
            np_ = max(np_, ((w0 >> RL_SHIFT) & 7) + 1)
        if w0 & SV_FLAG:
            np_ = max(np_, ((w0 >> SV_SHIFT) & 7) + 1)
        if w0 & PD_FLAG:
            nd = max(nd, ((w0 >> PD_SHIFT) & 7) + 1)
    return np_, nd


def lower(obj, dim=None):
    """lower an SDF2/SDF3/Node to a :class:`Tape`"""
    root = unwrap(obj)
    if dim is None:
        from .d2 import SDF2
        dim = 2 if isinstance(obj, SDF2) else (root.dim or 3)
    lw = _Lowering()
    lw.value(root, dim)
    lw.emit('END')
    assert lw.pdepth == 0 and lw.ddepth == 0
    code, meta = peephole(lw.code, lw.meta)
    n_p, n_d = slot_census(code)          # the peephole pass can leave slots unused
    assert n_p <= lw.pmax and n_d <= lw.dmax
    t = Tape(np.array(code, dtype=np.uint32), np.frombuffer(lw.consts, dtype=np.float64).copy(), n_p, n_d, dim)
    # per instruction: first instruction of the right operand / of the left chain (0xFFFF: not a
    # prunable combine) -- input of the interval prepass
    # (the prepass handles tapes of up to PRUNE_MAX_INSTR instructions; longer ones carry no ranges and run unpruned)
    t.externs = list(lw.externs)
    if t.n_instr <= PRUNE_MAX_INSTR and not t.externs:
        t.rstart = np.array([0xFFFF if m is None else m[0] for m in meta], dtype=np.uint16)
        t.lstart = np.array([0xFFFF if m is None else m[1] for m in meta], dtype=np.uint16)
    return t
