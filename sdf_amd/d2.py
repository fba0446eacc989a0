"""2-D modelling API: ``SDF2``, ``sdf2 / op2 / op23``, primitives, operators and the
2D->3D operators -- the surface of reference sdf/d2.py, recording IR nodes.
"""
import functools

import numpy as np

from . import dn, ease
from .ir import Node, SDFBase

# Constants (reference sdf/d2.py:9-14)

ORIGIN = np.array((0, 0))

X = np.array((1, 0))
Y = np.array((0, 1))

UP = Y

# SDF class (reference sdf/d2.py:18-38)

_ops = {}


class SDF2(SDFBase):
    def __init__(self, f):
        self.f = f

    def __call__(self, p):
        from . import engine
        return engine.evaluate(self, p).reshape((-1, 1))

    def __getattr__(self, name):
        # unlike SDF3 there is no fall-through to the payload (reference sdf/d2.py:25-29)
        if name in _ops:
            return functools.partial(_ops[name], self)
        raise AttributeError

    def __or__(self, other):
        return union(self, other)

    def __and__(self, other):
        return intersection(self, other)

    def __sub__(self, other):
        return difference(self, other)

    def k(self, k=None):
        self._k = k
        return self


def sdf2(f):
    def wrapper(*args, **kwargs):
        return SDF2(f(*args, **kwargs))
    return wrapper


def op2(f):
    def wrapper(*args, **kwargs):
        return SDF2(f(*args, **kwargs))
    _ops[f.__name__] = wrapper
    return wrapper


def op23(f):
    def wrapper(*args, **kwargs):
        from . import d3
        return d3.SDF3(f(*args, **kwargs))
    _ops[f.__name__] = wrapper
    return wrapper


# Helpers

def _normalize(a):
    return a / np.linalg.norm(a)


def _v2(a):
    return np.broadcast_to(np.asarray(a, dtype=np.float64), (2,))


# Primitives

@sdf2
def circle(radius=1, center=ORIGIN):
    """reference sdf/d2.py:76-80"""
    return Node('circle', [radius, *_v2(center)])


@sdf2
def line(normal=UP, point=ORIGIN):
    """reference sdf/d2.py:82-87"""
    normal = _normalize(normal)
    return Node('line', [*_v2(normal), *_v2(point)])


@sdf2
def slab(x0=None, y0=None, x1=None, y1=None, k=None):
    """reference sdf/d2.py:89-100"""
    fs = []
    if x0 is not None:
        fs.append(line(X, (x0, 0)))
    if x1 is not None:
        fs.append(line(-X, (x1, 0)))
    if y0 is not None:
        fs.append(line(Y, (0, y0)))
    if y1 is not None:
        fs.append(line(-Y, (0, y1)))
    return intersection(*fs, k=k)


@sdf2
def rectangle(size=1, center=ORIGIN, a=None, b=None):
    """reference sdf/d2.py:102-114"""
    if a is not None and b is not None:
        a = np.array(a)
        b = np.array(b)
        size = b - a
        center = a + size / 2
        return rectangle(size, center)
    size = np.array(size)
    return Node('rectangle', [*_v2(center), *_v2(size / 2)])


@sdf2
def rounded_rectangle(size, radius, center=ORIGIN):
    """per-quadrant corner radii; `center` is accepted and ignored exactly like the
    reference does (reference sdf/d2.py:116-134)"""
    try:
        r0, r1, r2, r3 = radius
    except TypeError:
        r0 = r1 = r2 = r3 = radius
    return Node('rounded_rectangle', [*_v2(size / 2), r0, r1, r2, r3])


@sdf2
def equilateral_triangle():
    """reference sdf/d2.py:136-152"""
    k = 3 ** 0.5
    return Node('equilateral_triangle', [k, 1 / k])


@sdf2
def hexagon(r):
    """reference sdf/d2.py:154-165"""
    r *= 3 ** 0.5 / 2
    k = np.array((3 ** 0.5 / -2, 0.5, np.tan(np.pi / 6)))
    two_k = 2 * k[:2]
    return Node('hexagon', [r, k[0], k[1], k[2], two_k[0], two_k[1], -k[2] * r, k[2] * r])


@sdf2
def rounded_x(w, r):
    """reference sdf/d2.py:167-173"""
    return Node('rounded_x', [w, r])


@sdf2
def polygon(points):
    """reference sdf/d2.py:175-196"""
    points = [np.array(p) for p in points]
    flat = [float(len(points))]
    for p in points:
        flat += [float(p[0]), float(p[1])]
    return Node('polygon', flat)


@sdf2
def vesica(r, d):
    """reference sdf/d2.py:198-207"""
    b = np.sqrt(r * r - d * d)
    return Node('vesica', [r, d, b])


# Positioning

@op2
def translate(other, offset):
    """reference sdf/d2.py:211-215"""
    return Node('translate2', _v2(offset), (other,))


@op2
def scale(other, factor):
    """reference sdf/d2.py:217-227"""
    try:
        x, y = factor
    except TypeError:
        x = y = factor
    m = min(x, y)
    return Node('scale2', [x, y, m], (other,))


@op2
def rotate(other, angle):
    """reference sdf/d2.py:229-240"""
    s = np.sin(angle)
    c = np.cos(angle)
    matrix = np.array([
        [c, -s],
        [s, c],
    ]).T
    return Node('rotate2', matrix.reshape(-1), (other,))


@op2
def circular_array(other, count):
    """union of `count` rotated copies (reference sdf/d2.py:242-245)"""
    angles = [i / count * 2 * np.pi for i in range(count)]
    return union(*[other.rotate(a) for a in angles])


# Alterations

@op2
def elongate(other, size):
    """reference sdf/d2.py:249-257"""
    return Node('elongate2', _v2(size), (other,))


# 2D => 3D Operations

@op23
def extrude(other, h):
    """reference sdf/d2.py:261-267"""
    return Node('extrude', [h / 2], (other,))


@op23
def extrude_to(a, b, h, e=ease.linear):
    """reference sdf/d2.py:269-278"""
    return Node('extrude_to', [h, h / 2, ease.easing_id(e)], (a, b))


@op23
def revolve(other, offset=0):
    """reference sdf/d2.py:280-286"""
    return Node('revolve', [offset], (other,))


# Common (reference sdf/d2.py:290-298)

union = op2(dn.union)
difference = op2(dn.difference)
intersection = op2(dn.intersection)
blend = op2(dn.blend)
negate = op2(dn.negate)
dilate = op2(dn.dilate)
erode = op2(dn.erode)
shell = op2(dn.shell)
repeat = op2(dn.repeat)
