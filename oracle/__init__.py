"""ctypes front of the CPU checker (oracle/sdf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by sdf_amd.  See the header of sdf_oracle.c for what it restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f64p = ctypes.POINTER(ctypes.c_double)
_f32p = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libsdf_oracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.sdf_oracle_eval_tree.argtypes = [_i32p, _f64p, _i32p, ctypes.c_int32, _f64p, ctypes.c_int64,
                                           ctypes.c_int, _f64p]
        L.sdf_oracle_eval_tree.restype = None
        L.sdf_oracle_marching_cubes.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p,
                                                ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        L.sdf_oracle_marching_cubes.restype = ctypes.c_int64
        L.sdf_oracle_generate.argtypes = [_i32p, _f64p, _i32p, ctypes.c_int32,
                                          _f64p, ctypes.c_int, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
        L.sdf_oracle_generate.restype = ctypes.c_void_p
        for name, rt in (('ntri', ctypes.c_int64), ('nbatches', ctypes.c_int64), ('neval', ctypes.c_int64),
                         ('nambiguous', ctypes.c_int64), ('tris', _f64p),
                         ('kinds', ctypes.POINTER(ctypes.c_uint8))):
            fn = getattr(L, 'sdf_oracle_result_' + name)
            fn.argtypes = [ctypes.c_void_p]
            fn.restype = rt
        L.sdf_oracle_result_free.argtypes = [ctypes.c_void_p]
        L.sdf_oracle_result_free.restype = None
        L.sdf_oracle_estimate_bounds.argtypes = [_i32p, _f64p, _i32p, ctypes.c_int32, _f64p]
        L.sdf_oracle_estimate_bounds.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _tree(sdf):
    from sdf_amd.ir import flatten          # the front end only builds the tree; no device code
    nodes, params, children, root = flatten(sdf)
    if len(params) == 0:
        params = np.zeros(1)
    if len(children) == 0:
        children = np.zeros(1, np.int32)
    return (np.ascontiguousarray(nodes), np.ascontiguousarray(params), np.ascontiguousarray(children), root)


def _p(a, t):
    return a.ctypes.data_as(t)


def evaluate(sdf, pts):
    """f(P) -> (N,) float64"""
    nodes, params, children, root = _tree(sdf)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n, dim = pts.shape
    out = np.empty(n, np.float64)
    lib().sdf_oracle_eval_tree(_p(nodes, _i32p), _p(params, _f64p), _p(children, _i32p), root,
                               _p(pts, _f64p), n, dim, _p(out, _f64p))
    return out


def marching_cubes(volume):
    """soup (3T,3) float32 in volume index coordinates; also returns the ambiguous-cell count"""
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    if vol.ndim != 3 or min(vol.shape) < 2:
        return np.zeros((0, 3), np.float32), 0
    if vol.size and (0 < vol.min() or 0 > vol.max()):
        return np.zeros((0, 3), np.float32), 0
    cap = 12 * max(1, (vol.shape[0] - 1) * (vol.shape[1] - 1) * (vol.shape[2] - 1))
    out = np.empty((cap, 9), np.float32)
    namb = ctypes.c_int64(0)
    nt = lib().sdf_oracle_marching_cubes(_p(vol, _f32p), vol.shape[0], vol.shape[1], vol.shape[2],
                                         _p(out, _f32p), cap, ctypes.byref(namb))
    return out[:nt].reshape(-1, 3).copy(), namb.value


class GenerateResult:
    pass


def generate(sdf, X, Y, Z, batch_size=32, sparse=True, batch_range=None):
    nodes, params, children, root = _tree(sdf)
    X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
    b0, b1 = batch_range if batch_range is not None else (0, -1)
    L = lib()
    h = L.sdf_oracle_generate(_p(nodes, _i32p), _p(params, _f64p), _p(children, _i32p), root,
                              _p(X, _f64p), len(X), _p(Y, _f64p), len(Y), _p(Z, _f64p), len(Z),
                              batch_size, 1 if sparse else 0, b0, b1)
    try:
        r = GenerateResult()
        nt = L.sdf_oracle_result_ntri(h)
        nb = L.sdf_oracle_result_nbatches(h)
        r.points = np.ctypeslib.as_array(L.sdf_oracle_result_tris(h), shape=(max(nt, 1) * 9,))[:nt * 9] \
            .reshape(-1, 3).copy()
        r.kinds = np.ctypeslib.as_array(L.sdf_oracle_result_kinds(h), shape=(max(nb, 1),))[:nb].copy()
        r.n_eval = L.sdf_oracle_result_neval(h)
        r.n_ambiguous = L.sdf_oracle_result_nambiguous(h)
        return r
    finally:
        L.sdf_oracle_result_free(h)


def estimate_bounds(sdf):
    nodes, params, children, root = _tree(sdf)
    out = np.zeros(6)
    rc = lib().sdf_oracle_estimate_bounds(_p(nodes, _i32p), _p(params, _f64p), _p(children, _i32p), root,
                                          _p(out, _f64p))
    if rc:
        raise ValueError('zero-size array to reduction operation maximum which has no identity')
    return (tuple(out[:3]), tuple(out[3:]))
