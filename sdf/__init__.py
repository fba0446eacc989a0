"""Alias package: ``from sdf import *`` -- the import line of every fogleman/sdf script --
resolves to the MI355X-native implementation in sdf_amd (same names, reference
sdf/__init__.py:1-27)."""
import sys as _sys

import sdf_amd as _impl
from sdf_amd import *  # noqa: F401,F403
from sdf_amd import d2, d3, dn, ease, core, stl, util, text, mesh, progress  # noqa: F401

for _name in ('d2', 'd3', 'dn', 'ease', 'core', 'stl', 'util', 'text', 'mesh', 'progress'):
    _sys.modules[__name__ + '.' + _name] = getattr(_impl, _name)
