"""3-D modelling API: ``SDF3``, the ``sdf3 / op3 / op32`` decorators, primitives and
operators -- same names, signatures and quirks as reference sdf/d3.py, but every builder
records an IR node (sdf_amd/ir.py) instead of closing over NumPy code, so the whole model
can be lowered to the op tape of the HIP interpreter.

Constants are reduced with the same NumPy expressions the reference uses (file:line cited
per builder) so that the values baked into the tape are bit-identical to the ones the
reference closures capture.
"""
import functools

import numpy as np

from . import dn, ease
from .ir import Node, SDFBase

# Constants (reference sdf/d3.py:9-15) -- integer arrays, like the reference

ORIGIN = np.array((0, 0, 0))

X = np.array((1, 0, 0))
Y = np.array((0, 1, 0))
Z = np.array((0, 0, 1))

UP = Z

# SDF class (reference sdf/d3.py:19-46)

_ops = {}


class SDF3(SDFBase):
    """callable wrapper around a Node (or around another wrapper, which the reference
    produces whenever a builder returns an SDF3: slab, box(a=,b=), rotate_to, orient)"""

    def __init__(self, f):
        self.f = f

    def __call__(self, p):
        from . import engine
        return engine.evaluate(self, p).reshape((-1, 1))

    def __getattr__(self, name):
        if name in _ops:
            return functools.partial(_ops[name], self)
        # fall through to the payload: this is how `_k` propagates through
        # wrapper-of-wrapper objects (reference sdf/d3.py:26-31)
        return getattr(self.__dict__['f'], name)

    def __or__(self, other):
        return union(self, other)

    def __and__(self, other):
        return intersection(self, other)

    def __sub__(self, other):
        return difference(self, other)

    def k(self, k=None):
        self._k = k
        return self

    def generate(self, *args, **kwargs):
        from . import core
        return core.generate(self, *args, **kwargs)

    def save(self, path, *args, **kwargs):
        from . import core
        return core.save(path, self, *args, **kwargs)

    def show_slice(self, *args, **kwargs):
        from . import core
        return core.show_slice(self, *args, **kwargs)


def sdf3(f):
    """decorator for primitives (reference sdf/d3.py:48-51)"""
    def wrapper(*args, **kwargs):
        return SDF3(f(*args, **kwargs))
    return wrapper


def op3(f):
    """decorator for 3D->3D operators; registers the method name (reference sdf/d3.py:53-57)"""
    def wrapper(*args, **kwargs):
        return SDF3(f(*args, **kwargs))
    _ops[f.__name__] = wrapper
    return wrapper


def op32(f):
    """decorator for 3D->2D operators (reference sdf/d3.py:59-63)"""
    def wrapper(*args, **kwargs):
        from . import d2
        return d2.SDF2(f(*args, **kwargs))
    _ops[f.__name__] = wrapper
    return wrapper


# Helpers

def _normalize(a):
    return a / np.linalg.norm(a)


def _v3(a):
    """a value the reference would broadcast against an (N,3) array, as 3 floats"""
    return np.broadcast_to(np.asarray(a, dtype=np.float64), (3,))


def _perpendicular(v):
    if v[1] == 0 and v[2] == 0:
        if v[0] == 0:
            raise ValueError('zero vector')
        else:
            return np.cross(v, [0, 1, 0])
    return np.cross(v, [1, 0, 0])


# Primitives

@sdf3
def sphere(radius=1, center=ORIGIN):
    """|p - center| - radius (reference sdf/d3.py:92-96)"""
    return Node('sphere', [radius, *_v3(center)])


@sdf3
def plane(normal=UP, point=ORIGIN):
    """dot(point - p, n) with n normalised at build time (reference sdf/d3.py:98-103)"""
    normal = _normalize(normal)
    return Node('plane', [*_v3(normal), *_v3(point)])


@sdf3
def slab(x0=None, y0=None, z0=None, x1=None, y1=None, z1=None, k=None):
    """intersection of up to six axis planes (reference sdf/d3.py:105-120)"""
    fs = []
    if x0 is not None:
        fs.append(plane(X, (x0, 0, 0)))
    if x1 is not None:
        fs.append(plane(-X, (x1, 0, 0)))
    if y0 is not None:
        fs.append(plane(Y, (0, y0, 0)))
    if y1 is not None:
        fs.append(plane(-Y, (0, y1, 0)))
    if z0 is not None:
        fs.append(plane(Z, (0, 0, z0)))
    if z1 is not None:
        fs.append(plane(-Z, (0, 0, z1)))
    return intersection(*fs, k=k)


@sdf3
def box(size=1, center=ORIGIN, a=None, b=None):
    """exact box distance (reference sdf/d3.py:122-134)"""
    if a is not None and b is not None:
        a = np.array(a)
        b = np.array(b)
        size = b - a
        center = a + size / 2
        return box(size, center)
    size = np.array(size)
    return Node('box', [*_v3(center), *_v3(size / 2)])


@sdf3
def rounded_box(size, radius):
    """reference sdf/d3.py:136-142"""
    size = np.array(size)
    return Node('rounded_box', [*_v3(size / 2), radius])


@sdf3
def wireframe_box(size, thickness):
    """reference sdf/d3.py:144-155"""
    size = np.array(size)
    return Node('wireframe_box', [*_v3(size / 2), thickness / 2])


@sdf3
def torus(r1, r2):
    """reference sdf/d3.py:157-165"""
    return Node('torus', [r1, r2])


@sdf3
def capsule(a, b, radius):
    """segment distance (reference sdf/d3.py:167-176)"""
    a = np.array(a)
    b = np.array(b)
    ba = b - a
    return Node('capsule', [*_v3(a), *_v3(ba), np.dot(ba, ba), radius])


@sdf3
def cylinder(radius):
    """infinite cylinder along Z (reference sdf/d3.py:178-182)"""
    return Node('cylinder', [radius])


@sdf3
def capped_cylinder(a, b, radius):
    """reference sdf/d3.py:184-204"""
    a = np.array(a)
    b = np.array(b)
    ba = b - a
    baba = np.dot(ba, ba)
    return Node('capped_cylinder', [*_v3(a), *_v3(ba), baba, radius, radius * baba, baba * 0.5])


@sdf3
def rounded_cylinder(ra, rb, h):
    """reference sdf/d3.py:206-215"""
    return Node('rounded_cylinder', [ra, rb, h / 2])


@sdf3
def capped_cone(a, b, ra, rb):
    """reference sdf/d3.py:217-237"""
    a = np.array(a)
    b = np.array(b)
    rba = rb - ra
    baba = np.dot(b - a, b - a)
    k = rba * rba + baba
    return Node('capped_cone', [*_v3(a), *_v3(b - a), ra, rb, baba, rba, k])


@sdf3
def rounded_cone(r1, r2, h):
    """reference sdf/d3.py:239-250"""
    b = (r1 - r2) / h
    a = np.sqrt(1 - b * b)
    return Node('rounded_cone', [r1, r2, h, b, a, a * h])


@sdf3
def ellipsoid(size):
    """reference sdf/d3.py:252-259"""
    size = np.array(size)
    return Node('ellipsoid', [*_v3(size), *_v3(size * size)])


@sdf3
def pyramid(h):
    """reference sdf/d3.py:261-282"""
    m2 = h * h + 0.25
    return Node('pyramid', [h, m2, m2 + 0.25])


# Platonic solids (reference sdf/d3.py:286-325)

@sdf3
def tetrahedron(r):
    return Node('tetrahedron', [r, np.sqrt(3)])


@sdf3
def octahedron(r):
    return Node('octahedron', [r, np.tan(np.radians(30))])


@sdf3
def dodecahedron(r):
    x, y, z = _normalize(((1 + np.sqrt(5)) / 2, 1, 0))
    return Node('dodecahedron', [r, x, y, z])


@sdf3
def icosahedron(r):
    r *= 0.8506507174597755
    x, y, z = _normalize(((np.sqrt(5) + 3) / 2, 1, 0))
    w = np.sqrt(3) / 3
    return Node('icosahedron', [r, x, y, z, w])


# Positioning

@op3
def translate(other, offset):
    """reference sdf/d3.py:329-333"""
    return Node('translate', _v3(offset), (other,))


@op3
def scale(other, factor):
    """reference sdf/d3.py:335-345"""
    try:
        x, y, z = factor
    except TypeError:
        x = y = z = factor
    m = min(x, min(y, z))
    return Node('scale', [x, y, z, m], (other,))


@op3
def rotate(other, angle, vector=Z):
    """Rodrigues matrix, applied as p @ M (reference sdf/d3.py:347-360)"""
    x, y, z = _normalize(vector)
    s = np.sin(angle)
    c = np.cos(angle)
    m = 1 - c
    matrix = np.array([
        [m*x*x + c, m*x*y + z*s, m*z*x - y*s],
        [m*x*y - z*s, m*y*y + c, m*y*z + x*s],
        [m*z*x + y*s, m*y*z - x*s, m*z*z + c],
    ]).T
    return Node('rotate', matrix.reshape(-1), (other,))


@op3
def rotate_to(other, a, b):
    """reference sdf/d3.py:362-373 (parallel vectors hand back `other` itself)"""
    a = _normalize(np.array(a))
    b = _normalize(np.array(b))
    dot = np.dot(b, a)
    if dot == 1:
        return other
    if dot == -1:
        return rotate(other, np.pi, _perpendicular(a))
    angle = np.arccos(dot)
    v = _normalize(np.cross(b, a))
    return rotate(other, angle, v)


@op3
def orient(other, axis):
    """reference sdf/d3.py:375-377"""
    return rotate_to(other, UP, axis)


@op3
def circular_array(other, count, offset=0):
    """two evaluations of the pre-translated child at the folded angle and one sector
    before it (reference sdf/d3.py:379-392)"""
    other = other.translate(X * offset)
    da = 2 * np.pi / count
    return Node('circular_array', [da], (other,))


# Alterations

@op3
def elongate(other, size):
    """reference sdf/d3.py:396-405"""
    return Node('elongate', _v3(size), (other,))


@op3
def twist(other, k):
    """reference sdf/d3.py:407-419"""
    return Node('twist', [k], (other,))


@op3
def bend(other, k):
    """reference sdf/d3.py:421-433"""
    return Node('bend', [k], (other,))


@op3
def bend_linear(other, p0, p1, v, e=ease.linear):
    """reference sdf/d3.py:435-445"""
    p0 = np.array(p0)
    p1 = np.array(p1)
    v = -np.array(v)
    ab = p1 - p0
    return Node('bend_linear', [*_v3(p0), *_v3(ab), np.dot(ab, ab), *_v3(v), ease.easing_id(e)],
                (other,))


@op3
def bend_radial(other, r0, r1, dz, e=ease.linear):
    """reference sdf/d3.py:447-457"""
    return Node('bend_radial', [r0, r1 - r0, dz, ease.easing_id(e)], (other,))


@op3
def transition_linear(f0, f1, p0=-Z, p1=Z, e=ease.linear):
    """reference sdf/d3.py:459-470"""
    p0 = np.array(p0)
    p1 = np.array(p1)
    ab = p1 - p0
    return Node('transition_linear', [*_v3(p0), *_v3(ab), np.dot(ab, ab), ease.easing_id(e)],
                (f0, f1))


@op3
def transition_radial(f0, f1, r0=0, r1=1, e=ease.linear):
    """reference sdf/d3.py:472-481"""
    return Node('transition_radial', [r0, r1 - r0, ease.easing_id(e)], (f0, f1))


@op3
def wrap_around(other, x0, x1, r=None, e=ease.linear):
    """reference sdf/d3.py:483-502"""
    p0 = X * x0
    p1 = X * x1
    v = -Y
    if r is None:
        r = np.linalg.norm(p1 - p0) / (2 * np.pi)
    return Node('wrap_around', [*_v3(p0), *_v3(p1 - p0), *_v3(v), r, ease.easing_id(e)], (other,))


# 3D => 2D Operations

@op32
def slice(other):
    """cross-section at z=0 through two thin-slab intersections (reference sdf/d3.py:506-520)"""
    s = slab(z0=-1e-9, z1=1e-9)
    a = other & s
    b = other.negate() & s
    return Node('slice', (), (a, b))


# Common (reference sdf/d3.py:524-532)

union = op3(dn.union)
difference = op3(dn.difference)
intersection = op3(dn.intersection)
blend = op3(dn.blend)
negate = op3(dn.negate)
dilate = op3(dn.dilate)
erode = op3(dn.erode)
shell = op3(dn.shell)
repeat = op3(dn.repeat)
