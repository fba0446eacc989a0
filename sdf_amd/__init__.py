"""sdf_amd -- MI355X-native drop-in for the fogleman/sdf sampling + meshing path.

``from sdf_amd import *`` (or ``from sdf import *`` through the alias package at the repo
root) exposes the same names as the reference package (reference sdf/__init__.py:1-27):
the 2-D / 3-D modelling API, the easing module, ``generate / save / sample_slice /
show_slice`` and ``write_binary_stl``.  Models are lowered to an op tape and sampled and
meshed by hand-written HIP kernels (sdf_amd/csrc); there is no CPU evaluation path for the library's own
nodes.  A user-written closure (a function decorated with ``@sdf3`` / ``@op3`` / ``@sdf2`` that returns NumPy
code, reference README.md:258-295) stays the user's code and runs on the host; everything around it runs on
the device.
"""
from . import d2, d3, ease

from .util import *

from .d2 import *

from .d3 import *

from .mesh import Mesh

from .text import (
    measure_image,
    measure_text,
    image,
    text,
)

from .core import (
    generate,
    save,
    sample_slice,
    show_slice,
)

from .stl import (
    write_binary_stl,
)
