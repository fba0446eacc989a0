"""Triangle mesh -> SDF leaf (reference sdf/mesh.py:8-113).

The reference voxelises a mesh with OpenVDB into a narrow-band level set, copies the active voxels
into a dense float32 array and evaluates it with scipy's RegularGridInterpolator, falling back to
the distance to the bounding box beyond the band (reference sdf/mesh.py:64-113).  Here the lookup
is the `grid3d` leaf of the op tape: the voxel array and its three axes travel to the device with
the tape's constants and the interpreter does the trilinear interpolation per sample
(csrc/sdf_interp.h L_GRID3D, same operation order as scipy 1.7.1's `_evaluate_linear`).

`grid_sdf` builds the leaf from a ready voxel grid (any producer); `Mesh.sdf` is the reference's
entry point and needs `pyopenvdb` for the voxelisation step only -- without it the import fails
exactly where the reference's does (reference sdf/mesh.py:66).
"""
import numpy as np

from .d3 import SDF3, box
from .ir import Node, unwrap


def grid_sdf(xyz, array, background, bounding_box):
    """the SDF3 the reference's `Mesh.sdf` returns, from its ingredients (reference sdf/mesh.py:88-105):
    `xyz` the three strictly increasing coordinate axes of the voxel centres, `array` the float32
    voxel values of shape (len(X), len(Y), len(Z)), `background` the narrow-band value
    (`grid.background`), `bounding_box` = (a, b) of the mesh for the `box(a=a, b=b)` estimator."""
    X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in xyz)
    A = np.ascontiguousarray(array, dtype=np.float32)          # scipy keeps float32 values as they are
    if A.shape != (len(X), len(Y), len(Z)):
        raise ValueError('There are %d points and %d values in dimension 0' % (len(X), A.shape[0] if A.ndim else 0))
    for i, g in enumerate((X, Y, Z)):
        if len(g) < 2 or not np.all(np.diff(g) > 0):
            raise ValueError('The points in dimension %d must be strictly ascending' % i)
    a, b = bounding_box
    est = unwrap(box(a=a, b=b))                                # Node('box', centre + half size): d3.py:122-134
    params = [len(X), len(Y), len(Z), float(background)] + list(est.params)
    blob = np.concatenate([X, Y, Z, A.astype(np.float64).reshape(-1)])
    f = SDF3(Node('grid3d', params, (), meta={'blob': blob}))
    f.array, f.xyz = A, (X, Y, Z)                              # what the reference hangs on its closure (mesh.py:107-111)
    return f


class Mesh:
    """reference sdf/mesh.py:8-62: points (V, 3), triangles (T, 3) and rigid / scaling transforms"""

    def __init__(self, points, triangles):
        self.points = points
        self.triangles = triangles

    @classmethod
    def from_file(cls, path):
        import meshio
        m = meshio.read(path)
        return cls(m.points, m.cells[0].data)

    @property
    def bounding_box(self):
        lo, hi = self.points.min(axis=0), self.points.max(axis=0)
        return (tuple(lo.tolist()), tuple(hi.tolist()))

    @property
    def size(self):
        lo, hi = self.points.min(axis=0), self.points.max(axis=0)
        return tuple((hi - lo).tolist())

    def transformed(self, matrix):
        h = np.hstack([self.points, np.ones((self.points.shape[0], 1))])
        return Mesh((h @ np.array(matrix).T)[:, :3], self.triangles)

    def scaled(self, scale):
        try:
            sx, sy, sz = scale
        except TypeError:
            sx = sy = sz = scale
        return self.transformed([[sx, 0, 0, 0], [0, sy, 0, 0], [0, 0, sz, 0], [0, 0, 0, 1]])

    def translated(self, offset):
        dx, dy, dz = offset
        return self.transformed([[1, 0, 0, dx], [0, 1, 0, dy], [0, 0, 1, dz], [0, 0, 0, 1]])

    def positioned(self, position, anchor):
        lo, hi = map(np.array, self.bounding_box)
        return self.translated(position - (lo + (hi - lo) * anchor))

    def centered(self):
        return self.positioned((0, 0, 0), (0.5, 0.5, 0.5))

    def sdf(self, voxel_size, half_width=None):
        """reference sdf/mesh.py:64-113; the voxelisation is OpenVDB's (host, like the reference),
        the per-sample lookup runs on the device"""
        import pyopenvdb as vdb

        half_width_voxels = 3
        if half_width is not None:
            half_width_voxels = max(half_width_voxels, int(np.ceil(half_width / voxel_size)))
        grid = vdb.FloatGrid.createLevelSetFromPolygons(
            self.points, triangles=self.triangles,
            transform=vdb.createLinearTransform(voxelSize=voxel_size), halfWidth=half_width_voxels)
        v0, v1 = grid.evalActiveVoxelBoundingBox()
        ijk0, ijk1 = np.array(v0, dtype=int), np.array(v1, dtype=int)
        size = ijk1 - ijk0 + 1
        p0, p1 = grid.transform.indexToWorld(ijk0), grid.transform.indexToWorld(ijk1)
        xyz = tuple(np.linspace(p0[i], p1[i], size[i]) for i in range(3))
        A = np.zeros(size, dtype=np.float32)
        grid.copyToArray(A, ijk=ijk0)
        f = grid_sdf(xyz, A, grid.background, self.bounding_box)
        f.grid = grid
        return f
