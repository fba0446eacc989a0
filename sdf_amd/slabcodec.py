"""The multi-GPU exchange slab's triangle record (csrc/sdf_slab.h `Tri16`) restated in NumPy: what a tool or a test
needs to read / write device slabs on the host.  A marching-cubes triangle in a batch's local voxel coordinates has its
three vertices on edges of ONE cell, so of a vertex's three float32 coordinates two are the integers c or c + 1 and one
lies along the edge: 16 bytes keep the three along-edge floats bit for bit + one word (the cell 3 x 6 bits; per vertex
the axis of its float, 2 bits, and the two offsets, 2 bits); a triangle that has not that shape (a vertex inside a cell)
is RAW: its nine floats go to the slab's raw area and the record holds their index."""
import numpy as np

RAW = np.uint32(1 << 31)
RAW_DIV, RAW_MIN = 128, 256


def layout(cap_items, cap_tris):
    """byte offsets of a slab of these capacities: prefix words, transforms, records, raw area; raw capacity; size"""
    prefix_off = 128
    xf_off = prefix_off + cap_items * 8
    tris_off = (xf_off + cap_items * 48 + 15) & ~15
    raw_cap = cap_tris // RAW_DIV + RAW_MIN
    raw_off = tris_off + cap_tris * 16
    return {'prefix_off': prefix_off, 'xf_off': xf_off, 'tris_off': tris_off, 'raw_off': raw_off, 'raw_cap': raw_cap,
            'bytes': (raw_off + raw_cap * 36 + 255) & ~255}


def encode16(tri):
    """(n, 9) float32 local triangles -> (codes uint32 (n,), floats float32 (n, 3), ok bool (n,)); rows with ok False
    are not of the edge shape (store them raw)"""
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 3, 3)
    m = tri.min(axis=1)
    c = np.where((m >= 0) & (m < 64), np.floor(np.where(np.isfinite(m), m, 0.0)), 0).astype(np.int64)      # (n, 3)
    code = (c[:, 0] | (c[:, 1] << 6) | (c[:, 2] << 12)).astype(np.uint32)
    ok = np.ones(len(tri), bool)
    f = np.zeros((len(tri), 3), np.float32)
    cf = c.astype(np.float32)
    for k in range(3):
        v = tri[:, k, :]
        lo, hi = v == cf, v == cf + 1
        isfrac = ~lo & ~hi
        nfrac = isfrac.sum(axis=1)
        ok &= nfrac <= 1
        frac = np.where(nfrac > 0, 2 - np.argmax(isfrac[:, ::-1], axis=1), 0)          # the LAST fractional axis, like the device loop
        a1 = np.where(frac == 0, 1, 0)
        a2 = np.where(frac == 2, 1, 2)
        rows = np.arange(len(tri))
        bits = frac.astype(np.uint32) | (hi[rows, a1].astype(np.uint32) << 2) | (hi[rows, a2].astype(np.uint32) << 3)
        code |= bits << np.uint32(18 + 4 * k)
        f[:, k] = v[rows, frac]
    return code, f, ok


def decode16(code, f):
    """records -> (n, 9) float32 (RAW records decode to garbage: look those up in the raw area)"""
    code = np.asarray(code, np.uint32)
    f = np.asarray(f, np.float32).reshape(-1, 3)
    c = np.stack([code & 63, (code >> 6) & 63, (code >> 12) & 63], axis=1).astype(np.int64)
    out = np.zeros((len(code), 3, 3), np.float32)
    rows = np.arange(len(code))
    for k in range(3):
        v = (code >> np.uint32(18 + 4 * k)) & 15
        frac = (v & 3).astype(np.int64)
        frac = np.minimum(frac, 2)
        a1 = np.where(frac == 0, 1, 0)
        a2 = np.where(frac == 2, 1, 2)
        out[rows, k, frac] = f[:, k]
        out[rows, k, a1] = (c[rows, a1] + ((v >> 2) & 1)).astype(np.float32)
        out[rows, k, a2] = (c[rows, a2] + ((v >> 3) & 1)).astype(np.float32)
    return out.reshape(-1, 9)


def write_triangles(slab, cap_items, cap_tris, tri):
    """encode (n, 9) float32 local triangles into the byte array of one slab (records + raw area); returns n_raw"""
    L = layout(cap_items, cap_tris)
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    code, f, ok = encode16(tri)
    raw_rows = np.flatnonzero(~ok)
    assert len(raw_rows) <= L['raw_cap']
    rec = np.zeros((len(tri), 4), np.uint32)
    rec[:, 0] = code
    rec[:, 1:] = f.view(np.uint32)
    rec[raw_rows, 0] = RAW
    rec[raw_rows, 1] = np.arange(len(raw_rows), dtype=np.uint32)
    rec[raw_rows, 2:] = 0
    slab[L['tris_off']:L['tris_off'] + 16 * len(tri)] = rec.reshape(-1).view(np.uint8)
    if len(raw_rows):
        slab[L['raw_off']:L['raw_off'] + 36 * len(raw_rows)] = tri[raw_rows].reshape(-1).view(np.uint8)
    return len(raw_rows)


def read_triangles(slab, cap_items, cap_tris, n):
    """the first n triangles of a slab as (n, 9) float32"""
    L = layout(cap_items, cap_tris)
    rec = np.frombuffer(bytes(slab[L['tris_off']:L['tris_off'] + 16 * n]), np.uint32).reshape(n, 4)
    out = decode16(rec[:, 0], rec[:, 1:].copy().view(np.float32))
    israw = (rec[:, 0] & RAW) != 0
    if israw.any():
        raw = np.frombuffer(bytes(slab[L['raw_off']:L['raw_off'] + 36 * L['raw_cap']]), np.float32).reshape(-1, 9)
        out[israw] = raw[rec[israw, 1]]
    return out
