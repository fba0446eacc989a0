"""Easing curves usable as the ``e=`` argument of bend_linear / bend_radial /
transition_* / wrap_around / extrude_to (reference sdf/ease.py:3-162, 34 curves).

In the reference an easing is a plain NumPy function.  Here it is an :class:`Easing`
object that carries the id of the curve inside the op tape (the HIP interpreter evaluates
it on the device: csrc/sdf_interp.h `ease_apply`); calling it on a host array still works
and follows the reference formulas, so user scripts that plot or compose easings keep
running.  Ids are shared with csrc/opcodes.h and oracle/node_ops.h.
"""
import numpy as np

_REGISTRY = []


class Easing:
    __slots__ = ('name', 'id', '_fn')

    def __init__(self, name, fn):
        self.name = name
        self.id = len(_REGISTRY)
        self._fn = fn
        _REGISTRY.append(self)

    def __call__(self, t, *args, **kwargs):
        return self._fn(np.asarray(t) if not np.isscalar(t) else t, *args, **kwargs)

    @property
    def __name__(self):
        return self.name

    def __repr__(self):
        return '<easing %s #%d>' % (self.name, self.id)


def _piecewise(split):
    """builder for the in_out_* polynomials: u = 2t; rising half on u<1, falling half after"""
    def deco(pair):
        rise, fall = pair
        def fn(t):
            u = t * 2
            return np.where(u < 1, rise(u), fall(u - 2))
        return fn
    return deco


def _pow(t, n):
    r = t
    for _ in range(n - 1):
        r = r * t
    return r


# polynomial families (reference sdf/ease.py:6-63)
linear = Easing('linear', lambda t: t)
in_quad = Easing('in_quad', lambda t: t * t)
out_quad = Easing('out_quad', lambda t: -t * (t - 2))


def _in_out_quad(t):
    u = 2 * t - 1
    return np.where(t < 0.5, 2 * t * t, -0.5 * (u * (u - 2) - 1))


in_out_quad = Easing('in_out_quad', _in_out_quad)
in_cubic = Easing('in_cubic', lambda t: _pow(t, 3))
out_cubic = Easing('out_cubic', lambda t: _pow(t - 1, 3) + 1)
in_out_cubic = Easing('in_out_cubic', _piecewise(1)((
    lambda u: 0.5 * u * u * u, lambda v: 0.5 * (v * v * v + 2))))
in_quart = Easing('in_quart', lambda t: _pow(t, 4))
out_quart = Easing('out_quart', lambda t: -(_pow(t - 1, 4) - 1))
in_out_quart = Easing('in_out_quart', _piecewise(1)((
    lambda u: 0.5 * u * u * u * u, lambda v: -0.5 * (v * v * v * v - 2))))
in_quint = Easing('in_quint', lambda t: _pow(t, 5))
out_quint = Easing('out_quint', lambda t: _pow(t - 1, 5) + 1)
in_out_quint = Easing('in_out_quint', _piecewise(1)((
    lambda u: 0.5 * u * u * u * u * u, lambda v: 0.5 * (v * v * v * v * v + 2))))

# trigonometric / exponential / circular (reference sdf/ease.py:65-108)
in_sine = Easing('in_sine', lambda t: -np.cos(t * np.pi / 2) + 1)
out_sine = Easing('out_sine', lambda t: np.sin(t * np.pi / 2))
in_out_sine = Easing('in_out_sine', lambda t: -0.5 * (np.cos(np.pi * t) - 1))
in_expo = Easing('in_expo', lambda t: np.where(t == 0, 0.0, 2 ** (10 * (t - 1))))
out_expo = Easing('out_expo', lambda t: np.where(t == 1, 1.0, 1 - 2 ** (-10 * t)))
in_out_expo = Easing('in_out_expo', lambda t: np.where(
    t == 0, 0.0, np.where(t == 1, 1.0, np.where(
        t < 0.5, 0.5 * 2 ** (20 * t - 10), 1 - 0.5 * 2 ** (-20 * t + 10)))))
in_circ = Easing('in_circ', lambda t: -1 * (np.sqrt(1 - t * t) - 1))
out_circ = Easing('out_circ', lambda t: np.sqrt(1 - (t - 1) * (t - 1)))


def _in_out_circ(t):
    u = t * 2
    v = u - 2
    with np.errstate(invalid='ignore'):
        return np.where(u < 1, -0.5 * (np.sqrt(1 - u * u) - 1), 0.5 * (np.sqrt(1 - v * v) + 1))


in_out_circ = Easing('in_out_circ', _in_out_circ)


# elastic / back / bounce / square (reference sdf/ease.py:110-162)
def _in_elastic(t, k=0.5):
    u = t - 1
    return -1 * (2 ** (10 * u) * np.sin((u - k / 4) * (2 * np.pi) / k))


def _out_elastic(t, k=0.5):
    return 2 ** (-10 * t) * np.sin((t - k / 4) * (2 * np.pi / k)) + 1


def _in_out_elastic(t, k=0.5):
    u = t * 2
    v = u - 1
    s = np.sin((v - k / 4) * 2 * np.pi / k)
    return np.where(u < 1, -0.5 * (2 ** (10 * v) * s), 2 ** (-10 * v) * s * 0.5 + 1)


in_elastic = Easing('in_elastic', _in_elastic)
out_elastic = Easing('out_elastic', _out_elastic)
in_out_elastic = Easing('in_out_elastic', _in_out_elastic)

_BACK = 1.70158
in_back = Easing('in_back', lambda t: t * t * ((_BACK + 1) * t - _BACK))
out_back = Easing('out_back', lambda t: (t - 1) * (t - 1) * ((_BACK + 1) * (t - 1) + _BACK) + 1)
_BACK2 = 1.70158 * 1.525
in_out_back = Easing('in_out_back', _piecewise(1)((
    lambda u: 0.5 * (u * u * ((_BACK2 + 1) * u - _BACK2)),
    lambda v: 0.5 * (v * v * ((_BACK2 + 1) * v + _BACK2) + 2))))


def _out_bounce(t):
    return np.where(t < 4 / 11, (121 * t * t) / 16, np.where(
        t < 8 / 11, (363 / 40 * t * t) - (99 / 10 * t) + 17 / 5, np.where(
            t < 9 / 10, (4356 / 361 * t * t) - (35442 / 1805 * t) + 16061 / 1805,
            (54 / 5 * t * t) - (513 / 25 * t) + 268 / 25)))


def _in_bounce(t):
    return 1 - _out_bounce(1 - t)


in_bounce = Easing('in_bounce', _in_bounce)
out_bounce = Easing('out_bounce', _out_bounce)
in_out_bounce = Easing('in_out_bounce', lambda t: np.where(
    t < 0.5, _in_bounce(2 * t) * 0.5, _out_bounce(2 * t - 1) * 0.5 + 0.5))
in_square = Easing('in_square', lambda t: np.where(t < 1, 0.0, 1.0))
out_square = Easing('out_square', lambda t: np.where(t > 0, 1.0, 0.0))
in_out_square = Easing('in_out_square', lambda t: np.where(t < 0.5, 0.0, 1.0))

EASINGS = tuple(_REGISTRY)
EASING_IDS = {e.name: e.id for e in EASINGS}


def easing_id(e):
    """tape id of an ``e=`` argument; only the built-in curves can run on the device"""
    if isinstance(e, Easing):
        return e.id
    from .ir import OpaqueSDFError
    raise OpaqueSDFError('easing %r is a Python callable; only sdf.ease.* curves lower to the tape' % (e,))
