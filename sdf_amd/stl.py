"""Binary STL writer (reference sdf/stl.py:4-24): 80 zero bytes, u32 triangle count, then
50-byte records (f32 normal, 3 x f32 vertex, u16 attribute).  Accepts the (3T,3) array
`generate` returns, or any sequence of points like the reference does."""
import struct

import numpy as np

_RECORD = np.dtype([
    ('normal', ('<f', 3)),
    ('points', ('<f', (3, 3))),
    ('attr', '<H'),
])


def stl_records(points):
    """the 50-byte records of `write_binary_stl` as a structured array"""
    tri = np.asarray(points, dtype='float32').reshape((-1, 3, 3))
    rec = np.zeros(len(tri), dtype=_RECORD)
    e1 = tri[:, 1] - tri[:, 0]
    e2 = tri[:, 2] - tri[:, 0]
    n = np.cross(e1, e2)
    n /= np.linalg.norm(n, axis=1).reshape((-1, 1))
    rec['points'] = tri
    rec['normal'] = n
    return rec


def write_binary_stl(path, points):
    n = len(points) // 3
    rec = stl_records(points)
    with open(path, 'wb') as fp:
        fp.write(b'\x00' * 80)
        fp.write(struct.pack('<I', n))
        fp.write(rec.tobytes())


def write_stl_records(path, records):
    """`records`: T x 50 bytes as a flat uint8 array (Engine Mesh.stl_records())"""
    records = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
    with open(path, 'wb') as fp:
        fp.write(b'\x00' * 80)
        fp.write(struct.pack('<I', len(records) // 50))
        fp.write(memoryview(records))
