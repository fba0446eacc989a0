"""Host binding of the HIP library (sdf_amd/csrc -> libsdf_hip.so) through its C ABI
(include/sdf_hip.h).  ctypes only: no torch types cross the boundary; torch is used by
sdf_amd/dist.py for the RCCL exchange, not here.

There is NO CPU fallback: if the shared library is missing, or no MI355X is visible,
every entry point raises.  (The CPU checker under oracle/ is test infrastructure and is
never imported from this package.)
"""
import ctypes
import os
import threading
import weakref

import numpy as np

from . import tape as _tape

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SDF_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libsdf_hip.so')   # (SDF_HIP_LIB: another build of the library, for A/B timing)

PRECISION_F64 = 0
PRECISION_F32 = 1

_c_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_f64p = ctypes.POINTER(ctypes.c_double)
_f32p = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


class SdfExchangeStats(ctypes.Structure):
    """mirror of `sdf_exchange_stats` in include/sdf_hip.h"""
    _fields_ = [(k, ctypes.c_int64) for k in ('n_batches', 'n_skipped', 'n_empty', 'n_nonempty', 'n_triangles', 'n_grid_voxels',
                                               'n_eval_voxels', 'n_ambiguous_cells', 'n_sampled_voxels', 'n_pruned_instrs',
                                               'n_retries', 'chunks', 'world', 'slab_bytes')] + \
               [(k, ctypes.c_double) for k in ('ms_mesh', 'ms_exchange', 'ms_expand', 'ms_total')] + \
               [('per_rank_triangles', ctypes.c_int64 * 64)]


class SdfStats(ctypes.Structure):
    """mirror of `sdf_stats` in include/sdf_hip.h"""
    _fields_ = [
        ('n_batches', _c_i64), ('n_skipped', _c_i64), ('n_empty', _c_i64), ('n_nonempty', _c_i64),
        ('n_triangles', _c_i64), ('n_grid_voxels', _c_i64), ('n_eval_voxels', _c_i64),
        ('n_ambiguous_cells', _c_i64), ('n_work_begin', _c_i64), ('n_work_end', _c_i64),
        ('n_retries', _c_i64),
        ('ms_prepass', ctypes.c_double), ('ms_mesh', ctypes.c_double), ('ms_emit', ctypes.c_double),
        ('ms_total', ctypes.c_double), ('n_pruned_instrs', _c_i64), ('n_batch_instrs', _c_i64),
        ('n_sampled_voxels', _c_i64), ('ms_mesh_device', ctypes.c_double), ('sclk_mhz', ctypes.c_double),
        ('t_mesh_first_us', ctypes.c_double), ('t_mesh_last_us', ctypes.c_double), ('mesh_kernel', ctypes.c_int64),
    ]


# name -> (restype, argtypes); this table is also what tests/test_host.py checks against the header
ABI = {
    'sdf_abi_version': (ctypes.c_int, []),
    'sdf_build_info': (ctypes.c_char_p, []),
    'sdf_last_error': (ctypes.c_char_p, []),
    'sdf_device_count': (ctypes.c_int, []),
    'sdf_device_mem_info': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    'sdf_test_fail_alloc': (ctypes.c_int, [ctypes.c_int]),
    'sdf_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    'sdf_ctx_destroy': (ctypes.c_int, [_vp]),
    'sdf_ctx_set_stream': (ctypes.c_int, [_vp, _vp]),
    'sdf_ctx_set_prune': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_set_cull': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_set_twopass': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_set_tail_order': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_set_defer': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_set_mesh2': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_set_cull_levels': (ctypes.c_int, [_vp, ctypes.c_int]),
    'sdf_ctx_synchronize': (ctypes.c_int, [_vp]),
    'sdf_ctx_trim': (ctypes.c_int, [_vp]),
    'sdf_tape_create': (ctypes.c_int, [_vp, _u32p, ctypes.c_uint32, _f64p, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_vp)]),
    'sdf_tape_set_prune_info': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16),
                                               ctypes.c_uint32]),
    'sdf_tape_destroy': (ctypes.c_int, [_vp]),
    'sdf_eval_points': (ctypes.c_int, [_vp, _vp, _c_i64, ctypes.c_int, _vp, ctypes.c_int]),
    'sdf_eval_points_host': (ctypes.c_int, [_vp, _f64p, _c_i64, ctypes.c_int, _f64p, ctypes.c_int]),
    'sdf_eval_grid_host': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p,
                                          ctypes.c_int, _f64p, ctypes.c_int]),
    'sdf_estimate_bounds': (ctypes.c_int, [_vp, _f64p, ctypes.c_int]),
    'sdf_tape_extern_count': (ctypes.c_int, [_vp]),
    'sdf_eval_extern_points_host': (ctypes.c_int, [_vp, _f64p, _c_i64, ctypes.c_int, _f64p, ctypes.c_int]),
    'sdf_eval_points_extern_host': (ctypes.c_int, [_vp, _f64p, _c_i64, ctypes.c_int, _f64p, _f64p, ctypes.c_int]),
    'sdf_generate_field': (ctypes.c_int, [_vp, _vp, _vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, ctypes.POINTER(_vp)]),
    'sdf_marching_cubes': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp,
                                          _c_i64, ctypes.POINTER(_c_i64)]),
    'sdf_marching_cubes_host': (ctypes.c_int, [_vp, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               _f32p, _c_i64, ctypes.POINTER(_c_i64)]),
    'sdf_generate': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, ctypes.c_int,
                                    ctypes.POINTER(_vp)]),
    'sdf_generate_to_device': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, ctypes.c_int, _vp, _c_i64,
                                              ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_vp)]),
    'sdf_generate_to_device_async': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_int, _vp, ctypes.c_int64, ctypes.POINTER(_vp)]),
    'sdf_mesh_wait': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    'sdf_generate_records': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'sdf_slab_bytes': (ctypes.c_size_t, [_c_i64, _c_i64]),
    'sdf_generate_compact_async': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, ctypes.c_int, _vp, _c_i64,
                                                  _c_i64, ctypes.POINTER(_vp)]),
    'sdf_expand_slabs': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_int, _c_i64, _c_i64, _vp, _c_i64]),
    'sdf_comm_available': (ctypes.c_int, []),
    'sdf_comm_unique_id': (ctypes.c_int, [_vp]),
    'sdf_comm_create': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'sdf_comm_destroy': (ctypes.c_int, [_vp]),
    'sdf_generate_sharded_async': (ctypes.c_int, [_vp, _vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.POINTER(_vp)]),
    'sdf_exchange_wait': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_c_i64)]),
    'sdf_exchange_stats_get': (ctypes.c_int, [_vp, ctypes.POINTER(SdfExchangeStats)]),
    'sdf_exchange_destroy': (ctypes.c_int, [_vp]),
    'sdf_skip_kinds': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int, ctypes.c_int, _c_i64,
                                      _c_i64, ctypes.c_int, _vp]),
    'sdf_generate_from_kinds': (ctypes.c_int, [_vp, _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int, ctypes.c_int,
                                               _c_i64, _c_i64, ctypes.c_int, _vp, ctypes.POINTER(_vp)]),
    'sdf_memcpy_to_host': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    'sdf_mesh_stats': (ctypes.c_int, [_vp, ctypes.POINTER(SdfStats)]),
    'sdf_mesh_triangles': (_c_i64, [_vp]),
    'sdf_mesh_emit_device': (ctypes.c_int, [_vp, _vp]),
    'sdf_mesh_emit_host': (ctypes.c_int, [_vp, _f64p]),
    'sdf_mesh_emit_host_workers': (ctypes.c_int, [_vp, _f64p, ctypes.c_int]),
    'sdf_mesh_emit_host_range': (ctypes.c_int, [_vp, _c_i64, _c_i64, _f64p]),
    'sdf_mesh_batch_offsets': (ctypes.c_int, [_vp, ctypes.POINTER(_c_i64)]),
    'sdf_mesh_emit_stl_host': (ctypes.c_int, [_vp, _vp]),
    'sdf_mesh_adopt_soup': (ctypes.c_int, [_vp, _vp, _c_i64, ctypes.POINTER(_vp)]),
    'sdf_mesh_weld': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    'sdf_mesh_weld_fetch': (ctypes.c_int, [_vp, _f64p, ctypes.POINTER(ctypes.c_int64)]),
    'sdf_host_alloc': (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_vp)]),
    'sdf_host_free': (ctypes.c_int, [_vp]),
    'sdf_mesh_kinds': (ctypes.c_int, [_vp, _u8p]),
    'sdf_mesh_prune_masks': (ctypes.c_int, [_vp, _u32p]),
    'sdf_mesh_destroy': (ctypes.c_int, [_vp]),
}
ABI_VERSION = 9


def build_info():
    """the toolchain that built the loaded library (csrc/build.sh records it: sdf_build_info)"""
    return load_library().sdf_build_info().decode()


def _strip_c_comments(text):
    """C / C++ source without its comments; string and character literals are left alone (a `//` inside a format string or
    a URL is not a comment)"""
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch in '"\'':                                  # a literal: copy to its closing quote, honouring escapes
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == '\\' else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith('//', i):
            j = text.find('\n', i)
            i = n if j < 0 else j
        elif text.startswith('/*', i):
            j = text.find('*/', i + 2)
            i = n if j < 0 else j + 2
        else:
            out.append(ch); i += 1
    return ''.join(out)


def source_id():
    """sha256 (16 hex digits) over the library's sources -- sdf_amd/csrc/*.{h,hip,inc,sh} and the public header
    include/sdf_hip.h -- WITHOUT their comments and blank lines: what a committed rocprofv3 summary was taken on
    (tools/summarize_prof.py writes it, bench.py only quotes a summary whose id is this build's).  Editing a comment does
    not make a profile stale; editing code, a string literal or the exchange's host code does."""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, 'csrc')
    files = sorted(glob.glob(os.path.join(d, '*'))) + [os.path.join(os.path.dirname(here), 'include', 'sdf_hip.h')]
    for f in files:
        ext = f.rsplit('.', 1)[-1]
        if os.path.isfile(f) and ext in ('h', 'hip', 'inc', 'sh'):
            text = open(f, encoding='utf-8', errors='replace').read()
            if ext == 'sh':
                text = re.sub(r'(?m)^\s*#(?!!).*$', '', text)
            else:
                text = _strip_c_comments(text)
            text = '\n'.join(ln.rstrip() for ln in text.split('\n') if ln.strip())
            h.update(os.path.basename(f).encode()); h.update(text.encode())
    return h.hexdigest()[:16]

_lib = None
_lib_lock = threading.Lock()


class SdfHipError(RuntimeError):
    pass


# `sdf_field_fn` of include/sdf_hip.h
FIELD_FN = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _f64p, _c_i64, _f64p)


def call_closure(fn, P):
    """a user closure on (N, d) points -> (N,) float64: the contract of a function decorated with the
    reference's @sdf3 / @op3 (reference README.md:258-295: returns (N,) or (N, 1))"""
    v = np.asarray(fn(P), dtype=np.float64).reshape(-1)
    if v.shape[0] != len(P):
        raise ValueError('a user SDF returned %d values for %d points' % (v.shape[0], len(P)))
    return v


def load_library(path=None):
    """dlopen libsdf_hip.so and attach the prototypes; raises if it was not built"""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise SdfHipError(
                '%s is missing: build the HIP extension first '
                '(python -c "import __graft_entry__ as g; g.build()"); sdf_amd has no CPU path' % p)
        lib = ctypes.CDLL(p)
        for name, (rt, at) in ABI.items():
            fn = getattr(lib, name)
            fn.restype = rt
            fn.argtypes = at
        if lib.sdf_abi_version() != ABI_VERSION:
            raise SdfHipError('libsdf_hip.so ABI %d != binding %d' % (lib.sdf_abi_version(), ABI_VERSION))
        if path is None:
            _lib = lib
        return lib


def _check(lib, rc):
    if rc != 0:
        msg = lib.sdf_last_error()
        raise SdfHipError(msg.decode() if msg else 'sdf_hip error %d' % rc)


def _dp(a, t):
    return a.ctypes.data_as(t)


class DeviceTape:
    """a lowered model resident on the device (`sdf_tape*`)"""

    def __init__(self, eng, tape):
        self.engine = eng
        self.tape = tape
        self.handle = _vp()
        lib = eng.lib
        _check(lib, lib.sdf_tape_create(eng.ctx, _dp(tape.code, _u32p), len(tape.code),
                                        _dp(tape.consts, _f64p), len(tape.consts),
                                        tape.n_pslots, tape.n_dslots, ctypes.byref(self.handle)))
        self._fin = weakref.finalize(self, lib.sdf_tape_destroy, self.handle)
        if tape.rstart is not None and len(tape.rstart) == tape.n_instr:
            u16p = ctypes.POINTER(ctypes.c_uint16)
            _check(lib, lib.sdf_tape_set_prune_info(self.handle, _dp(tape.rstart, u16p), _dp(tape.lstart, u16p),
                                                    tape.n_instr))


class _PinnedBlock:
    """a block of the library's pinned host memory (sdf_host_alloc) that an ndarray can sit on: the
    array keeps this object as its base, and the block goes back to the library's free list when the
    last view of it is gone"""

    def __init__(self, lib, nbytes):
        p = _vp()
        _check(lib, lib.sdf_host_alloc(max(int(nbytes), 1), ctypes.byref(p)))
        self.ptr = p.value
        self.nbytes = int(nbytes)
        self._fin = weakref.finalize(self, lib.sdf_host_free, _vp(self.ptr))

    @property
    def __array_interface__(self):
        return {'shape': (self.nbytes,), 'typestr': '|u1', 'data': (self.ptr, False), 'version': 3}


def pinned_empty(lib, shape, dtype):
    """np.empty in pinned host memory (results of large device-to-host copies land here: PCIe rate
    instead of the ~10 GB/s of fresh pageable memory); falls back to np.empty when pinning fails"""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    if n < (1 << 20):
        return np.empty(shape, dtype)
    try:
        blk = _PinnedBlock(lib, n)
    except SdfHipError:
        return np.empty(shape, dtype)
    return np.asarray(blk)[:n].view(dtype).reshape(shape)


class Mesh:
    """result of one `generate` call (`sdf_mesh*`): triangle soup resident on the device"""

    def __init__(self, eng, handle):
        self.engine = eng
        self.handle = handle
        self._fin = weakref.finalize(self, eng.lib.sdf_mesh_destroy, handle)

    @property
    def n_triangles(self):
        return int(self.engine.lib.sdf_mesh_triangles(self.handle))

    def wait(self):
        """collect a call submitted with generate(wait=False); returns `emitted`"""
        e = ctypes.c_int(0)
        _check(self.engine.lib, self.engine.lib.sdf_mesh_wait(self.handle, ctypes.byref(e)))
        self.emitted = bool(e.value)
        return self.emitted

    def stats(self):
        st = SdfStats()
        _check(self.engine.lib, self.engine.lib.sdf_mesh_stats(self.handle, ctypes.byref(st)))
        d = {k: getattr(st, k) for k, _ in SdfStats._fields_}
        d.update(skipped=st.n_skipped, empty=st.n_empty, nonempty=st.n_nonempty,
                 batches=st.n_batches, triangles=st.n_triangles)
        return d

    def kinds(self):
        """per-batch classification in reference batch order: 0 skipped / 1 empty / 2 nonempty
        (3 = outside this rank's shard)"""
        n = self.stats()['n_batches']
        out = np.empty(max(n, 1), np.uint8)
        _check(self.engine.lib, self.engine.lib.sdf_mesh_kinds(self.handle, _dp(out, _u8p)))
        return out[:n]

    def prune_masks(self):
        """(n_batches, 16) uint32: the interval prepass's skip / forced bits per batch (batch order,
        like kinds(); only the batches that were meshed used them)"""
        n = self.stats()['n_batches']
        out = np.zeros((max(n, 1), 16), np.uint32)
        _check(self.engine.lib, self.engine.lib.sdf_mesh_prune_masks(self.handle, _dp(out, _u32p)))
        return out[:n]

    def points(self, workers=0):
        """(3T, 3) float64 world-space soup in reference order, on the host.  A mesh of `generate(records=True)`
        sends its 16-byte records and `workers` host threads (0: the machine's, at most 32; at most 64) make the soup
        from them (`sdf_mesh_emit_host_workers`); any other mesh copies its float64 soup"""
        t = self.n_triangles
        out = pinned_empty(self.engine.lib, (3 * t, 3), np.float64)
        if t:
            _check(self.engine.lib, self.engine.lib.sdf_mesh_emit_host_workers(self.handle, _dp(out, _f64p), int(workers or 0)))
        return out

    def points_range(self, first_tri, n_tris):
        """rows of triangles [first_tri, first_tri + n_tris) of the soup: (3 * n_tris, 3) float64"""
        out = pinned_empty(self.engine.lib, (3 * int(n_tris), 3), np.float64)
        if n_tris:
            _check(self.engine.lib, self.engine.lib.sdf_mesh_emit_host_range(self.handle, int(first_tri), int(n_tris),
                                                                             _dp(out, _f64p)))
        return out

    def batch_offsets(self):
        """(n_batches + 1,) int64: where each batch's triangles start in this shard's soup (reference
        batch order; equal neighbours = the batch contributed nothing)"""
        n = self.stats()['n_batches']
        out = np.zeros(n + 1, np.int64)
        _check(self.engine.lib, self.engine.lib.sdf_mesh_batch_offsets(self.handle, _dp(out, ctypes.POINTER(_c_i64))))
        return out

    def emit_device(self, device_ptr):
        """write the (3T,3) float64 soup into caller-owned device memory (e.g. a torch tensor)"""
        _check(self.engine.lib, self.engine.lib.sdf_mesh_emit_device(self.handle, _vp(device_ptr)))

    def weld(self):
        """(unique points (U, 3) float64 in lexicographic order, cells (T, 3) int64): what
        `np.unique(points, axis=0, return_inverse=True)` gives the reference's `_mesh`
        (reference sdf/core.py:160-164), sorted and deduplicated on the device"""
        nu = ctypes.c_int64(0)
        _check(self.engine.lib, self.engine.lib.sdf_mesh_weld(self.handle, ctypes.byref(nu)))
        pts = pinned_empty(self.engine.lib, (nu.value, 3), np.float64)
        cells = pinned_empty(self.engine.lib, (self.n_triangles, 3), np.int64)
        if nu.value:
            _check(self.engine.lib, self.engine.lib.sdf_mesh_weld_fetch(self.handle, _dp(pts, _f64p),
                                                                       _dp(cells, ctypes.POINTER(ctypes.c_int64))))
        return pts, cells

    def stl_records(self):
        """T x 50-byte binary STL records (normals computed on the device)"""
        t = self.n_triangles
        out = pinned_empty(self.engine.lib, (50 * t,), np.uint8)
        if t:
            _check(self.engine.lib, self.engine.lib.sdf_mesh_emit_stl_host(self.handle, out.ctypes.data_as(_vp)))
        return out

    def close(self):
        self._fin()


COMM_ID_BYTES = 128          # SDF_COMM_ID_BYTES


class DeviceSoup:
    """a float64 soup in LIBRARY memory on the device (9 doubles per triangle): the result of an exchange step.  It
    exposes `__cuda_array_interface__` (zero-copy view for torch / cupy: `torch.as_tensor(soup, device=...)`) and can
    copy itself to the host.  Valid until the next step is submitted on the same lane of its communicator."""

    def __init__(self, eng, ptr, n_tris, keep=None):
        self.engine, self.ptr, self.n_triangles, self._keep = eng, int(ptr or 0), int(n_tris), keep

    @property
    def __cuda_array_interface__(self):
        return {'shape': (9 * self.n_triangles,), 'typestr': '<f8', 'data': (self.ptr, False), 'version': 2, 'strides': None}

    def to_host(self, first_tri=0, n_tris=None):
        """(3n, 3) float64 ndarray of triangles [first_tri, first_tri + n_tris)"""
        n = self.n_triangles - first_tri if n_tris is None else int(n_tris)
        out = pinned_empty(self.engine.lib, (3 * n, 3), np.float64)
        if n:
            _check(self.engine.lib, self.engine.lib.sdf_memcpy_to_host(self.engine.ctx, out.ctypes.data_as(_vp),
                                                                       _vp(self.ptr + 72 * int(first_tri)), 72 * n))
        return out


class Exchange:
    """one sharded step in flight (`sdf_exchange*`)"""

    def __init__(self, comm, handle, tape):
        self.comm, self.handle, self._tape = comm, handle, tape
        self._fin = weakref.finalize(self, comm.engine.lib.sdf_exchange_destroy, handle)
        comm._live.add(self)          # (a communicator that is closed first takes its steps with it: Comm.close)

    def wait(self):
        """the step's one host synchronisation -> (DeviceSoup, stats dict)"""
        lib = self.comm.engine.lib
        p, n = _vp(), _c_i64(0)
        _check(lib, lib.sdf_exchange_wait(self.handle, ctypes.byref(p), ctypes.byref(n)))
        st = SdfExchangeStats()
        _check(lib, lib.sdf_exchange_stats_get(self.handle, ctypes.byref(st)))
        d = {k: getattr(st, k) for k, _ in SdfExchangeStats._fields_ if k != 'per_rank_triangles'}
        d['per_rank_triangles'] = [int(v) for v in st.per_rank_triangles[:st.world]]
        d.update(batches=st.n_batches, skipped=st.n_skipped, empty=st.n_empty, nonempty=st.n_nonempty, triangles=st.n_triangles,
                 payload='16-byte triangle records (local coordinates) + per-batch transform', exchange='rccl (native)')
        return DeviceSoup(self.comm.engine, p.value, n.value, keep=self), d

    def close(self):
        self._fin()


class Comm:
    """the ranks of one multi-GPU job (`sdf_comm*`): RCCL communicators + persistent exchange buffers, 1 or 2 lanes.
    Creation is collective: every rank calls it with the ids rank 0 drew (`Comm.unique_ids`)."""

    @staticmethod
    def available(lib):
        """(True, '') if librccl loads in this process, else (False, why) -- local, nothing collective"""
        if lib.sdf_comm_available():
            return True, ''
        return False, (lib.sdf_last_error() or b'').decode(errors='replace')

    @staticmethod
    def unique_ids(lib, n_lanes=2):
        buf = ctypes.create_string_buffer(COMM_ID_BYTES * n_lanes)
        for l in range(n_lanes):
            _check(lib, lib.sdf_comm_unique_id(ctypes.cast(ctypes.byref(buf, COMM_ID_BYTES * l), _vp)))
        return buf.raw

    def __init__(self, eng, ids, rank, world):
        self.engine, self.rank, self.world = eng, int(rank), int(world)
        self.n_lanes = len(ids) // COMM_ID_BYTES
        self.handle = _vp()
        raw = ctypes.create_string_buffer(bytes(ids), len(ids))
        _check(eng.lib, eng.lib.sdf_comm_create(eng.ctx, ctypes.cast(raw, _vp), self.n_lanes, self.rank, self.world,
                                                ctypes.byref(self.handle)))
        self._fin = weakref.finalize(self, eng.lib.sdf_comm_destroy, self.handle)
        self._live = weakref.WeakSet()          # steps (Exchange) that still hold a handle into this communicator

    def submit(self, sdf, X, Y, Z, batch_size=32, sparse=True, chunks=1, lane=0):
        eng = self.engine
        dt = eng.tape_for(sdf)
        if dt.tape.externs:
            raise ValueError('a model with user closures is sharded through the host protocol (sdf_amd.dist with HostCodec)')
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
        h = _vp()
        _check(eng.lib, eng.lib.sdf_generate_sharded_async(self.handle, dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y),
                                                           _dp(Z, _f64p), len(Z), int(batch_size), 1 if sparse else 0,
                                                           eng.precision, int(chunks), int(lane), ctypes.byref(h)))
        return Exchange(self, h, dt)

    def close(self):
        """destroy the communicator -- after the steps that point into it (an sdf_exchange handle must not outlive its
        sdf_comm: its destructor looks at the communicator's lanes)"""
        for x in list(self._live):
            x.close()
        self._fin()


class Engine:
    """one HIP context (`sdf_ctx*`) on one device"""

    def __init__(self, device=0):
        self.lib = load_library()
        n = self.lib.sdf_device_count()
        if n <= 0:
            raise SdfHipError('no HIP device visible (sdf_amd has no CPU path): %s'
                              % (self.lib.sdf_last_error() or b'').decode())
        self.device = device
        self.ctx = _vp()
        _check(self.lib, self.lib.sdf_ctx_create(device, ctypes.byref(self.ctx)))
        self._fin = weakref.finalize(self, self.lib.sdf_ctx_destroy, self.ctx)
        self.precision = PRECISION_F64
        self._tapes = weakref.WeakValueDictionary()

    def set_stream(self, stream_ptr):
        _check(self.lib, self.lib.sdf_ctx_set_stream(self.ctx, _vp(stream_ptr)))

    def set_prune(self, enabled):
        """interval prepass of generate on / off (default on; results are identical)"""
        _check(self.lib, self.lib.sdf_ctx_set_prune(self.ctx, int(bool(enabled))))

    def set_cull(self, enabled):
        """interval culling of cell groups inside a batch on / off (default on; results are identical)"""
        _check(self.lib, self.lib.sdf_ctx_set_cull(self.ctx, int(bool(enabled))))

    def set_tail_order(self, on):
        """hand the tail of the work list to the workgroups by descending cost (default) or in order; same results"""
        _check(self.lib, self.lib.sdf_ctx_set_tail_order(self.ctx, int(bool(on))))

    def set_defer(self, on):
        """one-kernel meshing: sparse tiles + deferred emission (default) or dense tiles + parking; same results"""
        _check(self.lib, self.lib.sdf_ctx_set_defer(self.ctx, int(bool(on))))

    def set_mesh2(self, mode):
        """which fused kernel meshes a call: 0 always k_mesh (default: k_mesh2 measured slower, profiles/r06e_two_wg.json); -1 k_mesh2
        (two workgroups of 512 threads per CU) when the previous call of the tape on the same grid found every tile to be its; 1 k_mesh2
        whenever the tape has a variant.  Same results; `stats()['mesh_kernel']` says which ran"""
        _check(self.lib, self.lib.sdf_ctx_set_mesh2(self.ctx, int(mode)))

    def set_cull_levels(self, levels):
        """interval levels of the culling pass: 2, 3 or 0 = the library's choice by the tape (default); same results"""
        _check(self.lib, self.lib.sdf_ctx_set_cull_levels(self.ctx, int(levels)))

    def set_twopass(self, mode):
        """meshing scheme: 0 one kernel (look-back + parking), 1 three kernels (sample / number / emit), -1 the
        library's choice by the tape's length (default); results are identical"""
        _check(self.lib, self.lib.sdf_ctx_set_twopass(self.ctx, int(mode)))

    def synchronize(self):
        _check(self.lib, self.lib.sdf_ctx_synchronize(self.ctx))

    def trim(self):
        """return the library's cached device blocks to the driver (before a job of a very different size)"""
        _check(self.lib, self.lib.sdf_ctx_trim(self.ctx))

    # -- models --
    def tape_for(self, sdf):
        """lower `sdf` NOW (boolean smoothing constants are evaluation-time state in the
        reference, so nothing is cached across calls beyond identical tapes)"""
        if isinstance(sdf, DeviceTape):
            return sdf
        t = sdf if isinstance(sdf, _tape.Tape) else _tape.lower(sdf)
        key = (t.code.tobytes(), t.consts.tobytes(), tuple(id(fn) for fn, _ in t.externs))
        dt = self._tapes.get(key)
        if dt is None:
            dt = DeviceTape(self, t)
            self._tapes[key] = dt
        return dt

    # -- f(P) --
    def eval_points(self, sdf, pts):
        dt = self.tape_for(sdf)
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] not in (2, 3):
            raise ValueError('points must be (N,2) or (N,3)')
        if dt.tape.externs:
            return self._eval_points_hybrid(dt, pts)
        out = np.empty(len(pts), np.float64)
        if len(pts):
            _check(self.lib, self.lib.sdf_eval_points_host(dt.handle, _dp(pts, _f64p), len(pts), pts.shape[1],
                                                           _dp(out, _f64p), self.precision))
        return out

    def _eval_points_hybrid(self, dt, pts):
        """f(P) of a model that contains user closures (`extern` leaves): the device runs the tape twice --
        first to find the point every closure is asked at (its argument after the transforms above it), then,
        with the closures' values (computed here, by the user's own NumPy code), for the result"""
        ext = dt.tape.externs
        n, dim = pts.shape
        if n == 0:
            return np.empty(0, np.float64)
        if dt.tape.n_instr == 2:          # the model IS one closure: nothing for the device to do
            return call_closure(ext[0][0], pts.copy())
        epts = np.empty((len(ext), n, 3), np.float64)
        _check(self.lib, self.lib.sdf_eval_extern_points_host(dt.handle, _dp(pts, _f64p), n, dim, _dp(epts, _f64p),
                                                              self.precision))
        vals = np.empty((len(ext), n), np.float64)
        for k, (fn, d) in enumerate(ext):
            vals[k] = call_closure(fn, np.ascontiguousarray(epts[k, :, :d]))
        out = np.empty(n, np.float64)
        _check(self.lib, self.lib.sdf_eval_points_extern_host(dt.handle, _dp(pts, _f64p), n, dim, _dp(vals, _f64p),
                                                              _dp(out, _f64p), self.precision))
        return out

    def estimate_bounds(self, sdf):
        """((x0, y0, z0), (x1, y1, z1)) of reference sdf/core.py:62-82 in one launch; None for a model with user
        closures (the caller then runs the reference's loop around eval_grid)"""
        dt = self.tape_for(sdf)
        if dt.tape.externs:
            return None
        out = np.zeros(6, np.float64)
        lib = self.lib
        if lib.sdf_estimate_bounds(dt.handle, _dp(out, _f64p), self.precision) != 0:
            msg = (lib.sdf_last_error() or b'').decode()
            if msg.startswith('zero-size array'):
                raise ValueError(msg)                       # what np.argwhere(...).max(axis=0) raises in the reference
            if 'barrier' in msg:
                return None                                 # (the caller runs the reference's loop around eval_grid)
            raise SdfHipError(msg)
        return (tuple(out[:3]), tuple(out[3:]))

    def eval_grid(self, sdf, X, Y, Z):
        dt = self.tape_for(sdf)
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
        if dt.tape.externs:
            G = np.stack(np.meshgrid(X, Y, Z, indexing='ij'), axis=-1).reshape(-1, 3)
            return self._eval_points_hybrid(dt, np.ascontiguousarray(G)).reshape(len(X), len(Y), len(Z))
        out = np.empty((len(X), len(Y), len(Z)), np.float64)
        if out.size:
            _check(self.lib, self.lib.sdf_eval_grid_host(dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y),
                                                         _dp(Z, _f64p), len(Z), _dp(out, _f64p), self.precision))
        return out

    # -- meshing --
    def marching_cubes(self, volume):
        """soup (3T,3) float32 in volume index coordinates of a host volume
        (drop-in for reference sdf/core.py:16-18 `_marching_cubes`; raises nothing on an empty
        result, returns a (0,3) array)"""
        vol = np.ascontiguousarray(volume, dtype=np.float32)
        if vol.ndim != 3:
            raise ValueError('Input volume should be a 3D numpy array.')
        cells = max(1, (max(vol.shape[0], 1) - 1) * (max(vol.shape[1], 1) - 1) * (max(vol.shape[2], 1) - 1))
        cap = min(5 * cells, max(4096, 2 * cells))
        while True:
            out = np.empty((cap, 9), np.float32)
            nt = _c_i64(0)
            _check(self.lib, self.lib.sdf_marching_cubes_host(self.ctx, _dp(vol, _f32p), vol.shape[0], vol.shape[1],
                                                              vol.shape[2], _dp(out, _f32p), cap, ctypes.byref(nt)))
            if nt.value <= cap:
                return out[:nt.value].reshape(-1, 3)
            cap = nt.value

    def generate(self, sdf, X, Y, Z, batch_size=32, sparse=True, shard=(0, 1), out_ptr=None, out_cap=0, wait=True, records=False):
        """mesh the grid X x Y x Z.  records=True (one device, the whole work list, no output buffer): for a
        caller who wants the soup on the HOST -- the triangles are kept as 16-byte records and `mesh.points(workers)`
        expands them on host threads (`sdf_generate_records`).  With out_ptr / out_cap (device memory for 9 * out_cap float64)
        the ordered soup is gathered into it inside the same submission (`mesh.emitted` tells
        whether it fitted); otherwise it stays in the mesh until `points()` / `emit_device()`.
        wait=False (needs out_ptr): the call is only enqueued; `mesh.wait()` -- or any read of the
        mesh -- collects it, so a caller can submit the next job while this one runs."""
        dt = self.tape_for(sdf)
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
        h = _vp()
        emitted = ctypes.c_int(0)
        if dt.tape.externs:
            if out_ptr or not wait:
                raise ValueError('a model with user closures is meshed through the host callback path: no output '
                                 'buffer, no asynchronous submission')
            return self._generate_field(dt, X, Y, Z, batch_size, sparse, shard)
        if not wait:
            if not out_ptr:
                raise ValueError('generate(wait=False) needs an output buffer (out_ptr / out_cap)')
            _check(self.lib, self.lib.sdf_generate_to_device_async(
                dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y), _dp(Z, _f64p), len(Z), int(batch_size),
                1 if sparse else 0, int(shard[0]), int(shard[1]), self.precision, _vp(out_ptr), int(out_cap),
                ctypes.byref(h)))
            m = Mesh(self, h)
            m._tape = dt
            m.emitted = None          # unknown until wait()
            return m
        if records and not out_ptr and tuple(shard) == (0, 1):
            _check(self.lib, self.lib.sdf_generate_records(dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y),
                                                           _dp(Z, _f64p), len(Z), int(batch_size), 1 if sparse else 0,
                                                           self.precision, ctypes.byref(h)))
        elif out_ptr:
            _check(self.lib, self.lib.sdf_generate_to_device(
                dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y), _dp(Z, _f64p), len(Z), int(batch_size),
                1 if sparse else 0, int(shard[0]), int(shard[1]), self.precision, _vp(out_ptr), int(out_cap),
                ctypes.byref(emitted), ctypes.byref(h)))
        else:
            _check(self.lib, self.lib.sdf_generate(dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y),
                                                   _dp(Z, _f64p), len(Z), int(batch_size), 1 if sparse else 0,
                                                   int(shard[0]), int(shard[1]), self.precision, ctypes.byref(h)))
        m = Mesh(self, h)
        m._tape = dt          # keep the device tape alive as long as the mesh
        m.emitted = bool(emitted.value)
        return m


    def skip_kinds(self, sdf, X, Y, Z, batch_size, b_begin, b_end, kinds_ptr):
        """`_skip` for batches [b_begin, b_end): 0 / 255 into the caller's device buffer (one byte per batch of the grid)"""
        dt = self.tape_for(sdf)
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
        _check(self.lib, self.lib.sdf_skip_kinds(dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y), _dp(Z, _f64p), len(Z),
                                                 int(batch_size), int(b_begin), int(b_end), self.precision, _vp(kinds_ptr)))

    def generate_from_kinds(self, sdf, X, Y, Z, batch_size, kinds_ptr, shard=(0, 1)):
        """`generate` (sparse) with the skip test's verdicts already on the device (skip_kinds)"""
        dt = self.tape_for(sdf)
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
        h = _vp()
        _check(self.lib, self.lib.sdf_generate_from_kinds(dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y), _dp(Z, _f64p), len(Z),
                                                          int(batch_size), int(shard[0]), int(shard[1]), self.precision,
                                                          _vp(kinds_ptr), ctypes.byref(h)))
        m = Mesh(self, h)
        m._tape = dt
        m.emitted = False
        return m

    # -- the multi-GPU exchange unit (sdf_amd/dist.py) --
    SLAB_HEADER = ('n_tris', 'n_items', 'overflow', 'n_empty', 'n_nonempty', 'n_eval_voxels', 'n_ambiguous_cells',
                   'n_sampled_voxels', 'n_pruned_instrs', 'n_work_total')      # int64[16] at the head of a slab

    def slab_bytes(self, cap_items, cap_tris):
        return int(self.lib.sdf_slab_bytes(int(cap_items), int(cap_tris)))

    def generate_compact(self, sdf, X, Y, Z, batch_size, sparse, shard, slab_ptr, cap_items, cap_tris):
        """mesh this rank's shard into a slab in caller-owned device memory (enqueue only; the returned mesh
        keeps the call's device buffers alive until it is closed)"""
        dt = self.tape_for(sdf)
        X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
        h = _vp()
        _check(self.lib, self.lib.sdf_generate_compact_async(
            dt.handle, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y), _dp(Z, _f64p), len(Z), int(batch_size),
            1 if sparse else 0, int(shard[0]), int(shard[1]), self.precision, _vp(slab_ptr), int(cap_items), int(cap_tris),
            ctypes.byref(h)))
        m = Mesh(self, h)
        m._tape = dt
        m.emitted = None
        return m

    def adopt_soup(self, device_ptr, n_triangles):
        """a Mesh over a float64 soup that already sits in device memory (n_triangles x 9 doubles, the caller's: it must
        stay alive and complete while the Mesh is used): STL records, weld and host copies then run on it like on a soup
        the library generated -- the gathered soup of a multi-GPU step"""
        h = _vp()
        _check(self.lib, self.lib.sdf_mesh_adopt_soup(self.ctx, _vp(int(device_ptr)), int(n_triangles), ctypes.byref(h)))
        return Mesh(self, h)

    def expand_slabs(self, slab_ptrs, cap_items, cap_tris, out_ptr, out_cap):
        """gathered slabs (device pointers, final order) -> ordered float64 soup at out_ptr (enqueue only)"""
        arr = (_vp * len(slab_ptrs))(*[_vp(int(p)) for p in slab_ptrs])
        _check(self.lib, self.lib.sdf_expand_slabs(self.ctx, arr, len(slab_ptrs), int(cap_items), int(cap_tris),
                                                   _vp(out_ptr), int(out_cap)))

    def _generate_field(self, dt, X, Y, Z, batch_size, sparse, shard):
        """`generate` for a model with user closures: the library drives the reference's batch loop and asks
        this callback for f(P) of the skip test and of every surviving batch (sdf_generate_field)"""
        failure = []

        def field(_user, p_pts, n, p_out):
            try:
                P = np.ctypeslib.as_array(p_pts, shape=(n, 3))
                np.ctypeslib.as_array(p_out, shape=(n,))[:] = self._eval_points_hybrid(dt, P)
                return 0
            except BaseException as e:      # (an exception cannot cross the C frames: it is re-raised below)
                failure.append(e)
                return 1

        cb = FIELD_FN(field)
        h = _vp()
        rc = self.lib.sdf_generate_field(self.ctx, ctypes.cast(cb, _vp), None, _dp(X, _f64p), len(X), _dp(Y, _f64p), len(Y),
                                         _dp(Z, _f64p), len(Z), int(batch_size), 1 if sparse else 0, int(shard[0]),
                                         int(shard[1]), ctypes.byref(h))
        if failure:
            raise failure[0]
        _check(self.lib, rc)
        m = Mesh(self, h)
        m._tape = dt
        m.emitted = False
        return m


_engines = {}


def get_engine(device=None):
    if device is None:
        device = int(os.environ.get('SDF_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    eng = _engines.get(device)
    if eng is None:
        eng = _engines[device] = Engine(device)
    return eng


def evaluate(sdf, p):
    """f(p) on the device -> (N,) float64 (reference sdf/d3.py:24-25 reshapes to (N,1))"""
    return get_engine().eval_points(sdf, p)
