"""Mesh -> SDF leaf (reference sdf/mesh.py:64-113).  Out of scope of the hot path
(needs pyopenvdb, used by no benchmark config); the class exists so that
``from sdf import *`` keeps exporting the name."""


class Mesh:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            'sdf_amd does not implement Mesh (voxelised mesh leaves need pyopenvdb); '
            'see DESIGN.md "out of scope"')
