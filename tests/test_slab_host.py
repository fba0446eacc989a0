"""The exchange slab's 16-byte triangle record (csrc/sdf_slab.h): the device source's encoder / decoder, built for the
HOST, against the NumPy restatement (sdf_amd/slabcodec.py) -- every triangle of the edge shape survives the round trip bit
for bit in both directions, triangles that have not the shape are recognised by both.  No GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from sdf_amd import slabcodec as sc

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SRC = r'''
#include "sdf_slab.h"
extern "C" int enc(const float *tri, long long n, unsigned *code, float *f, unsigned char *ok) {
    for (long long i = 0; i < n; i++) { Tri16 r; ok[i] = slab_encode16(tri + 9 * i, r) ? 1 : 0; code[i] = r.code; f[3 * i] = r.f[0]; f[3 * i + 1] = r.f[1]; f[3 * i + 2] = r.f[2]; }
    return 0;
}
extern "C" int dec(const unsigned *code, const float *f, long long n, float *tri) {
    for (long long i = 0; i < n; i++) { Tri16 r; r.code = code[i]; r.f[0] = f[3 * i]; r.f[1] = f[3 * i + 1]; r.f[2] = f[3 * i + 2]; slab_decode16(r, tri + 9 * i); }
    return 0;
}
extern "C" long long layout(long long ci, long long ct, long long *o) { SlabLayout L(ci, ct); o[0] = L.prefix_off; o[1] = L.xf_off; o[2] = L.tris_off; o[3] = L.raw_off; o[4] = L.raw_cap; return (long long)L.bytes; }
'''


@pytest.fixture(scope='module')
def lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not installed')
    d = tmp_path_factory.mktemp('slab')
    src, so = str(d / 'slab_host.hip'), str(d / 'libslab_host.so')
    open(src, 'w').write(SRC)
    subprocess.check_call([HIPCC, '--offload-host-only', '-O1', '-std=c++17', '-w', '-fPIC', '-shared', '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', so, src])
    L = ctypes.CDLL(so)
    vp, ll = ctypes.c_void_p, ctypes.c_longlong
    L.enc.argtypes = [vp, ll, vp, vp, vp]
    L.dec.argtypes = [vp, vp, ll, vp]
    L.layout.argtypes = [ll, ll, vp]
    L.layout.restype = ll
    return L


def _edge_triangles(rng, n):
    c = rng.integers(0, 32, (n, 3))
    tri = np.zeros((n, 3, 3), np.float32)
    rows = np.arange(n)
    for k in range(3):
        frac = rng.integers(0, 3, n)
        v = (c + rng.integers(0, 2, (n, 3))).astype(np.float32)
        t = rng.random(n).astype(np.float32)
        t[rng.random(n) < 0.05] = 0.0
        t[rng.random(n) < 0.05] = 1.0
        v[rows, frac] = (c[rows, frac] + t).astype(np.float32)
        tri[:, k, :] = v
    return tri.reshape(n, 9)


def test_tri16_round_trip_device_source_and_numpy(lib):
    rng = np.random.default_rng(7)
    n = 200000
    tri = _edge_triangles(rng, n)
    tri[::101] = rng.uniform(0, 32, (len(tri[::101]), 9)).astype(np.float32)      # vertices inside cells: raw
    tri[5::1009, 4] = np.nan                                                      # a NaN along an edge survives as the float
    code = np.zeros(n, np.uint32); f = np.zeros((n, 3), np.float32); ok = np.zeros(n, np.uint8)
    assert lib.enc(tri.ctypes.data, n, code.ctypes.data, f.ctypes.data, ok.ctypes.data) == 0
    ncode, nf, nok = sc.encode16(tri)
    dok = ok.astype(bool)
    finite = np.isfinite(tri).all(axis=1)
    assert np.array_equal(dok[finite], nok[finite])        # (with a NaN coordinate the two may pick different cells: either is lossless)
    assert nok.mean() > 0.98 and (~nok).sum() > 1000
    back = np.zeros((n, 9), np.float32)
    assert lib.dec(code.ctypes.data, f.ctypes.data, n, back.ctypes.data) == 0
    assert np.array_equal(back[dok].view(np.uint32), tri[dok].view(np.uint32))                      # device enc -> device dec
    assert np.array_equal(sc.decode16(code, f)[dok].view(np.uint32), tri[dok].view(np.uint32))     # device enc -> NumPy dec
    back2 = np.zeros((n, 9), np.float32)
    nf_c = np.ascontiguousarray(nf)
    assert lib.dec(ncode.ctypes.data, nf_c.ctypes.data, n, back2.ctypes.data) == 0
    assert np.array_equal(back2[nok].view(np.uint32), tri[nok].view(np.uint32))                     # NumPy enc -> device dec
    good = dok
    assert not (code[good] & sc.RAW).any()


@pytest.mark.parametrize('ci,ct', [(0, 0), (1, 1), (64, 192), (4096, 1 << 22)])
def test_slab_layout_matches_numpy(ci, ct, lib):
    o = (ctypes.c_longlong * 5)()
    nbytes = lib.layout(ci, ct, ctypes.addressof(o))
    L = sc.layout(ci, ct)
    assert (o[0], o[1], o[2], o[3], o[4], nbytes) == (L['prefix_off'], L['xf_off'], L['tris_off'], L['raw_off'], L['raw_cap'], L['bytes'])
