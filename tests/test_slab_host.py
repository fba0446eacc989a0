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


# ---- the host-side expansion of a slab's records (csrc/sdf_expand_host.h: what `f.generate()` returns is made on host threads) ----
EXPAND_SRC = r'''
#include "sdf_expand_host.h"
extern "C" int expand(const unsigned long long *prefix, const double *xf, const unsigned *rec, const float *raw, long long raw_cap,
                      long long n_items, long long n_tris, double *out, int threads, long long block, int trickle) {
    sdfhost::ExpandJob j;
    j.prefix = prefix; j.xf = xf; j.recs = (const Tri16 *)rec; j.raw = raw; j.raw_cap = raw_cap; j.n_items = n_items; j.n_tris = n_tris;
    j.out = out; j.block = block;
    sdfhost::Pool &p = sdfhost::Pool::get();
    p.start(j, threads - 1);
    if (trickle) {       // the records "arrive" piece by piece, like the copies of sdf_mesh_emit_host_workers
        for (long long a = 0; a < n_tris; a += 3 * block + 1) { j.avail.store(a); std::this_thread::sleep_for(std::chrono::microseconds(200)); }
    }
    j.avail.store(n_tris);
    sdfhost::expand_work(j);
    p.wait(j);
    return (int)(j.done.load() != j.nblocks());
}
'''


@pytest.fixture(scope='module')
def xlib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not installed')
    d = tmp_path_factory.mktemp('expand')
    src, so = str(d / 'expand_host.hip'), str(d / 'libexpand_host.so')
    open(src, 'w').write('#include <chrono>\n' + EXPAND_SRC)
    # (-ffp-contract=off like the library: a fused multiply-add would round once where the device rounds twice)
    subprocess.check_call([HIPCC, '--offload-host-only', '-O2', '-std=c++17', '-w', '-fPIC', '-shared', '-ffp-contract=off', '-pthread',
                           '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', so, src])
    L = ctypes.CDLL(so)
    vp, ll = ctypes.c_void_p, ctypes.c_longlong
    L.expand.argtypes = [vp, vp, vp, vp, ll, ll, ll, vp, ctypes.c_int, ll, ctypes.c_int]
    return L


@pytest.mark.parametrize('threads,block,trickle', [(1, 8192, 0), (4, 257, 1), (7, 64, 0), (3, 100000, 1)])
def test_host_expansion_of_records_is_k_expand(xlib, threads, block, trickle):
    """random work items (some empty, some of one triangle), records of the edge shape + raw ones: the threaded host expansion
    writes float64(local) * scale + offset of the decoded triangles, bit for bit, whatever the threads / block size / arrival"""
    rng = np.random.default_rng(11 + threads)
    n_items = 700
    counts = rng.integers(0, 400, n_items)
    counts[rng.random(n_items) < 0.2] = 0
    counts[rng.random(n_items) < 0.1] = 1
    counts[-1] = 0 if threads == 4 else 5
    n = int(counts.sum())
    incl = np.cumsum(counts).astype(np.uint64)
    prefix = incl | (np.uint64(2) << np.uint64(62))
    prefix[1::7] = incl[1::7] | (np.uint64(1) << np.uint64(62))           # (flag bits above the count are ignored)
    xf = np.concatenate([rng.uniform(-3, 3, (n_items, 3)), rng.uniform(1e-3, 0.1, (n_items, 3)) * rng.choice([-1.0, 1.0], (n_items, 3))], axis=1)
    tri = _edge_triangles(rng, n)
    tri[::53] = rng.uniform(0, 32, (len(tri[::53]), 9)).astype(np.float32)   # vertices inside cells: raw records
    code, f, ok = sc.encode16(tri)
    raw_rows = np.flatnonzero(~ok)
    rec = np.zeros((n, 4), np.uint32)
    rec[:, 0] = code
    rec[:, 1:] = f.view(np.uint32)
    rec[raw_rows, 0] = sc.RAW
    rec[raw_rows, 1] = np.arange(len(raw_rows), dtype=np.uint32)
    raw = np.ascontiguousarray(tri[raw_rows])
    assert len(raw_rows) > 100
    item = np.repeat(np.arange(n_items), counts)
    want = tri.astype(np.float64).reshape(n, 3, 3) * xf[item, None, 3:] + xf[item, None, :3]
    out = np.full((n, 9), np.nan)
    xfc = np.ascontiguousarray(xf)
    assert xlib.expand(prefix.ctypes.data, xfc.ctypes.data, rec.ctypes.data, raw.ctypes.data, len(raw_rows), n_items, n, out.ctypes.data,
                       threads, block, trickle) == 0
    assert np.array_equal(out.view(np.uint64), want.reshape(n, 9).view(np.uint64))
