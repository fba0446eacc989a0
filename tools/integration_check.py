#!/usr/bin/env python3
"""INTEGRATION.md sections 2 and 2b as a program (GPU box): the ctypes binding a maintainer of the reference would
write -- raw `ctypes.CDLL`, no sdf_amd.engine -- drives libsdf_hip.so (a) with a lowered tape and (b) with the closure
tree left on the host (`sdf_generate_field`), and both soups must equal the packaged path's."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

L = ctypes.CDLL(os.path.join(ROOT, 'sdf_amd', 'csrc', 'libsdf_hip.so'))
vp, i64, f64p = ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)
L.sdf_last_error.restype = ctypes.c_char_p
L.sdf_mesh_triangles.restype = i64
L.sdf_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
L.sdf_tape_create.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, f64p, ctypes.c_uint32,
                              ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp)]
L.sdf_generate.argtypes = [vp, f64p, ctypes.c_int, f64p, ctypes.c_int, f64p, ctypes.c_int, ctypes.c_int,
                           ctypes.c_int, i64, i64, ctypes.c_int, ctypes.POINTER(vp)]
L.sdf_mesh_triangles.argtypes = [vp]
L.sdf_mesh_emit_host.argtypes = [vp, f64p]
L.sdf_mesh_destroy.argtypes = [vp]
L.sdf_tape_destroy.argtypes = [vp]
FIELD = ctypes.CFUNCTYPE(ctypes.c_int, vp, f64p, i64, f64p)
L.sdf_generate_field.argtypes = [vp, vp, vp, f64p, ctypes.c_int, f64p, ctypes.c_int, f64p, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, i64, i64, ctypes.POINTER(vp)]


def check(rc):
    if rc:
        raise RuntimeError(L.sdf_last_error().decode())


ctx = vp(); check(L.sdf_ctx_create(0, ctypes.byref(ctx)))


def emit(mesh):
    t = L.sdf_mesh_triangles(mesh)
    pts = np.empty((3 * t, 3)); check(L.sdf_mesh_emit_host(mesh, pts.ctypes.data_as(f64p)))
    L.sdf_mesh_destroy(mesh)
    return pts


def generate_on_gpu(code, consts, n_p, n_d, X, Y, Z, batch_size=32, sparse=True):
    tape = vp(); check(L.sdf_tape_create(ctx, code.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(code),
                                         consts.ctypes.data_as(f64p), len(consts), n_p, n_d, ctypes.byref(tape)))
    mesh = vp(); check(L.sdf_generate(tape, X.ctypes.data_as(f64p), len(X), Y.ctypes.data_as(f64p), len(Y),
                                      Z.ctypes.data_as(f64p), len(Z), batch_size, int(sparse), 0, 1, 0, ctypes.byref(mesh)))
    pts = emit(mesh)
    L.sdf_tape_destroy(tape)
    return pts


def generate_closures_on_gpu(sdf, X, Y, Z, batch_size=32, sparse=True):
    def field(_user, p_pts, n, p_out):
        P = np.ctypeslib.as_array(p_pts, shape=(n, 3))
        np.ctypeslib.as_array(p_out, shape=(n,))[:] = sdf(P).reshape(-1)
        return 0
    cb = FIELD(field); mesh = vp()
    check(L.sdf_generate_field(ctx, ctypes.cast(cb, vp), None, X.ctypes.data_as(f64p), len(X), Y.ctypes.data_as(f64p),
                               len(Y), Z.ctypes.data_as(f64p), len(Z), batch_size, int(sparse), 0, 1, ctypes.byref(mesh)))
    return emit(mesh)


if __name__ == '__main__':
    import sdf_amd as s
    from sdf_amd import core, tape
    f = s.sphere(1) & s.box(1.5)
    c = s.cylinder(0.5)
    f -= c.orient(s.X) | c.orient(s.Y) | c.orient(s.Z)
    X, Y, Z, _ = core.grid_axes(((-0.85, -0.85, -0.85), (0.85, 0.85, 0.85)), samples=2 ** 20)
    want = f.generate(bounds=((-0.85, -0.85, -0.85), (0.85, 0.85, 0.85)), samples=2 ** 20, verbose=False)
    t = tape.lower(f)
    a = generate_on_gpu(t.code, t.consts, t.n_pslots, t.n_dslots, X, Y, Z)
    assert np.array_equal(a, want), 'section 2 binding'

    def closure_tree(P):          # the reference's own NumPy closures for this model (sdf/d3.py), kept on the host
        def length(v): return np.linalg.norm(v, axis=1)
        sph = length(P) - 1
        q = np.abs(P) - 0.75
        box = length(np.maximum(q, 0)) + np.minimum(np.amax(q, axis=1), 0)
        d = np.maximum(sph, box)
        cz = length(P[:, [0, 1]]) - 0.5
        cx = length(P[:, [2, 1]]) - 0.5          # orient(X): the cylinder along x (distance to the x axis)
        cy = length(P[:, [0, 2]]) - 0.5
        return np.maximum(d, -np.minimum(np.minimum(cx, cy), cz))
    b = generate_closures_on_gpu(closure_tree, X, Y, Z)
    assert b.shape == want.shape and np.abs(b - want).max() < 1e-12, 'section 2b binding'
    print('INTEGRATION.md bindings ok: %d triangles; tape route bit-identical, closure route max |diff| %.1e'
          % (len(want) // 3, np.abs(b - want).max()))
