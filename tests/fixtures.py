"""Model fixtures shared by the golden-vector generator (run against the unmodified
reference under /opt/conda/bin/python3.9) and by the parity tests (run against sdf_amd).

Every entry is a small program over the ``from sdf import *`` namespace that leaves the
model in ``f``.  The per-function list follows the argument list the reference uses for
its documentation renders (reference docs/render.py:15-226); the ``ex_*`` entries are the
reference's example scripts (reference examples/*.py) minus their ``save`` call.

This file is data, it must stay importable under Python 3.9 and must not import sdf_amd.
"""

EXAMPLE = """
f = sphere(1) & box(1.5)
c = cylinder(0.5)
f -= c.orient(X) | c.orient(Y) | c.orient(Z)
"""

FIXTURES = {
    # --- BASELINE configs (reference examples/example.py, gearlike.py, weave.py, blobby.py)
    'ex_example': EXAMPLE,
    'ex_gearlike': """
f = sphere(2) & slab(z0=-0.5, z1=0.5).k(0.1)
f -= cylinder(1).k(0.1)
f -= cylinder(0.25).circular_array(16, 2).k(0.1)
""",
    'ex_weave': """
f = rounded_box([3.2, 1, 0.25], 0.1).translate((1.5, 0, 0.0625))
f = f.bend_linear(X * 0.75, X * 2.25, Z * -0.1875, ease.in_out_quad)
f = f.circular_array(3, 0)
f = f.repeat((2.7, 5.4, 0), padding=1)
f |= f.translate((2.7 / 2, 2.7, 0))
f &= cylinder(10)
f |= (cylinder(12) - cylinder(10)) & slab(z0=-0.5, z1=0.5).k(0.25)
""",
    'ex_blobby': """
s = sphere(0.75)
s = s.translate(Z * -3) | s.translate(Z * 3)
s = s.union(capsule(Z * -3, Z * 3, 0.5), k=1)
f = sphere(1.5).union(s.orient(X), s.orient(Y), s.orient(Z), k=1)
""",
    'ex_knurling': """
f = rounded_cylinder(1, 0.1, 5)
x = box((1, 1, 4)).rotate(pi / 4)
x = x.circular_array(24, 1.6)
x = x.twist(0.75) | x.twist(-0.75)
f -= x.k(0.1)
f -= cylinder(0.5).k(0.1)
c = cylinder(0.25).orient(X)
f -= c.translate(Z * -2.5).k(0.1)
f -= c.translate(Z * 2.5).k(0.1)
""",
    'ex_pawn': """
def section(z0, z1, d0, d1, e=ease.linear):
    f = cylinder(d0/2).transition_linear(cylinder(d1/2), Z * z0, Z * z1, e)
    return f & slab(z0=z0, z1=z1)
f = section(0, 0.2, 1, 1.25)
f |= section(0.2, 0.3, 1.25, 1).k(0.05)
f |= rounded_cylinder(0.6, 0.1, 0.2).translate(Z * 0.4).k(0.05)
f |= section(0.5, 1.75, 1, 0.25, ease.out_quad).k(0.01)
f |= section(1.75, 1.85, 0.25, 0.5).k(0.01)
f |= section(1.85, 1.90, 0.5, 0.25).k(0.05)
f |= sphere(0.3).translate(Z * 2.15).k(0.05)
""",
    'ex_custbox': """
WIDTH = 12; HEIGHT = 6; DEPTH = 2; ROWS = 3; COLS = 5
col_spacing = WIDTH / COLS; row_spacing = HEIGHT / ROWS
c = rounded_box((0.2, 1e9, 1.5), 0.1).translate(Z * 1.5 / 2).repeat((col_spacing, 0, 0))
r = rounded_box((1e9, 0.2, 1.75), 0.1).translate(Z * 1.75 / 2).repeat((0, row_spacing, 0))
c = c.translate((col_spacing / 2, 0, 0))
r = r.translate((0, row_spacing / 2, 0))
d = c | r
f = rounded_box((WIDTH - 0.25, HEIGHT - 0.25, 1e9), 0.5)
f &= slab(z0=0.125).k(0.25)
d &= f
f = f.shell(0.25)
f &= slab(z1=DEPTH).k(0.125)
f = f | d
""",
    # --- one entry per documented function (reference docs/render.py)
    'sphere': "f = sphere(1)",
    'sphere_c': "f = sphere(0.7, (0.1, -0.2, 0.3))",
    'box': "f = box(1)",
    'box2': "f = box((1, 2, 3))",
    'box_ab': "f = box(a=(-1, -0.5, 0), b=(0.5, 1, 2))",
    'rounded_box': "f = rounded_box((1, 2, 3), 0.25)",
    'wireframe_box': "f = wireframe_box((1, 2, 3), 0.05)",
    'torus': "f = torus(1, 0.25)",
    'capsule': "f = capsule(-Z, Z, 0.5)",
    'capped_cylinder': "f = capped_cylinder(-Z, Z, 0.5)",
    'rounded_cylinder': "f = rounded_cylinder(0.5, 0.1, 2)",
    'capped_cone': "f = capped_cone(-Z, Z, 1, 0.5)",
    'rounded_cone': "f = rounded_cone(0.75, 0.25, 2)",
    'ellipsoid': "f = ellipsoid((1, 2, 3))",
    'pyramid': "f = pyramid(1)",
    'tetrahedron': "f = tetrahedron(1)",
    'octahedron': "f = octahedron(1)",
    'dodecahedron': "f = dodecahedron(1)",
    'icosahedron': "f = icosahedron(1)",
    'plane': "f = sphere() & plane()",
    'plane2': "f = sphere() & plane((1, 2, 3), (0.1, 0, 0.2))",
    'slab': "f = sphere() & slab(z0=-0.5, z1=0.5, x0=0)",
    'slab_k': "f = sphere() & slab(z0=-0.5, z1=0.5, x0=0, k=0.1)",
    'cylinder': "f = sphere() - cylinder(0.5)",
    'translate': "f = sphere().translate((0, 0, 2))",
    'scale': "f = sphere().scale((1, 2, 3))",
    'scale1': "f = box(1).scale(1.5)",
    'rotate': "f = capped_cylinder(-Z, Z, 0.5).rotate(pi / 4, X)",
    'rotate_to': "f = box((1, 2, 3)).rotate_to((1, 1, 0), (0, 1, 1))",
    'rotate_to_neg': "f = box((1, 2, 3)).rotate_to(Z, -Z)",
    'orient': """
c = capped_cylinder(-Z, Z, 0.25)
f = c.orient(X) | c.orient(Y) | c.orient(Z)
""",
    'union': "f = box((3, 3, 0.5)) | sphere()",
    'difference': "f = box((3, 3, 0.5)) - sphere()",
    'intersection': "f = box((3, 3, 0.5)) & sphere()",
    'smooth_union': "f = box((3, 3, 0.5)) | sphere().k(0.25)",
    'smooth_difference': "f = box((3, 3, 0.5)) - sphere().k(0.25)",
    'smooth_intersection': "f = box((3, 3, 0.5)) & sphere().k(0.25)",
    'union_multi_k': "f = union(sphere(1), box(1.5), torus(1, 0.25).k(0.2), k=None) | capsule(-X, X, 0.3)",
    'repeat': "f = sphere().repeat(3, (1, 1, 0))",
    'repeat_pad': "f = sphere(0.9).repeat((2, 2.5, 0), padding=1)",
    'repeat_all': "f = box(0.8).repeat(1.5, 2, padding=(1, 0, 1))",
    'circular_array': "f = capped_cylinder(-Z, Z, 0.5).circular_array(8, 4)",
    'blend': "f = sphere().blend(box())",
    'blend_k': "f = sphere().blend(box(), torus(1, 0.25), k=0.3)",
    'negate': "f = sphere().negate() & box(3)",
    'dilate': EXAMPLE + "f = f.dilate(0.1)\n",
    'erode': EXAMPLE + "f = f.erode(0.1)\n",
    'shell': "f = sphere().shell(0.05) & plane(-Z)",
    'elongate': EXAMPLE + "f = f.elongate((0.25, 0.5, 0.75))\n",
    'twist': "f = box().twist(pi / 2)",
    'bend': "f = box().bend(1)",
    'bend_linear': "f = capsule(-Z * 2, Z * 2, 0.25).bend_linear(-Z, Z, X, ease.in_out_quad)",
    'bend_radial': "f = box((5, 5, 0.25)).bend_radial(1, 2, -1, ease.in_out_quad)",
    'transition_linear': "f = box().transition_linear(sphere(), e=ease.in_out_quad)",
    'transition_radial': "f = box().transition_radial(sphere(), e=ease.in_out_quad)",
    'wrap_around': "f = box((6, 0.5, 0.5)).orient(Y).wrap_around(-3, 3)",
    # --- 2-D nodes through the 2D->3D operators
    'extrude': "f = hexagon(1).extrude(1)",
    'extrude_to': "f = rectangle(2).extrude_to(circle(1), 2, ease.in_out_quad)",
    'revolve': "f = hexagon(1).revolve(3)",
    'slice': EXAMPLE + "f = f.translate((0, 0, 0.55)).slice().extrude(0.1)\n",
    'circle': "f = circle(1.2, (0.3, -0.1)).extrude(0.5)",
    'line': "f = (circle(1) & line((1, 2), (0.1, 0.2))).extrude(0.5)",
    'slab2': "f = (circle(1) & d2.slab(x0=-0.5, y1=0.25)).extrude(0.5)",
    'rectangle': "f = rectangle((1, 2)).extrude(0.5)",
    'rectangle_ab': "f = rectangle(a=(-1, -0.5), b=(0.5, 1)).extrude(0.5)",
    'rounded_rectangle': "f = rounded_rectangle(np.array((2, 1)), (0.1, 0.2, 0.3, 0.4)).extrude(0.5)",
    'equilateral_triangle': "f = equilateral_triangle().extrude(0.5)",
    'hexagon': "f = hexagon(0.75).extrude(0.5)",
    'rounded_x': "f = rounded_x(1, 0.2).extrude(0.5)",
    'polygon': "f = polygon([(0, 0), (2, 0), (2, 1), (1, 0.4), (0, 1.5)]).extrude(0.5)",
    'vesica': "f = vesica(1, 0.4).extrude(0.5)",
    'ops2d': "f = (rectangle((1, 0.5)).translate((0.25, 0)).rotate(0.3).scale((1.5, 0.8)) | circle(0.4).k(0.1)).extrude(0.5)",
    'circular_array2': "f = rectangle((1, 0.2)).translate((1, 0)).circular_array(5).extrude(0.3)",
    'elongate2': "f = circle(0.5).elongate((0.5, 0.25)).extrude(0.3)",
    'repeat2': "f = circle(0.4).repeat((1, 1.5), (1, 2)).extrude(0.3)",
    'shell2': "f = hexagon(1).shell(0.1).dilate(0.02).erode(0.01).extrude(0.4)",
}

# every easing (reference sdf/ease.py:3-162) through bend_linear
EASINGS = [
    'linear',
    'in_quad', 'out_quad', 'in_out_quad',
    'in_cubic', 'out_cubic', 'in_out_cubic',
    'in_quart', 'out_quart', 'in_out_quart',
    'in_quint', 'out_quint', 'in_out_quint',
    'in_sine', 'out_sine', 'in_out_sine',
    'in_expo', 'out_expo', 'in_out_expo',
    'in_circ', 'out_circ', 'in_out_circ',
    'in_elastic', 'out_elastic', 'in_out_elastic',
    'in_back', 'out_back', 'in_out_back',
    'in_bounce', 'out_bounce', 'in_out_bounce',
    'in_square', 'out_square', 'in_out_square',
]
for _e in EASINGS:
    FIXTURES['ease_' + _e] = (
        "f = capsule(-Z * 2, Z * 2, 0.25).bend_linear(-Z, Z, X, ease.%s)" % _e)


# ---- user-written SDFs: the reference's documented extension point (reference README.md:258-295,
# sdf/d3.py:48-63).  The same user code runs against the reference (tools/make_golden_custom.py) and
# against the implementation under test; the first two closures are the README's own examples. ----
_CUSTOM_PRELUDE = """
@sdf3
def my_sphere(radius=1, center=ORIGIN):
    def f(p):
        return np.linalg.norm(p - center, axis=1) - radius
    return f

@op3
def my_translate(other, offset):
    def f(p):
        return other(p - offset)
    return f

@sdf3
def my_gyroid(scale, thickness):
    def f(p):
        q = p * scale
        g = np.sin(q[:, 0]) * np.cos(q[:, 1]) + np.sin(q[:, 1]) * np.cos(q[:, 2]) + np.sin(q[:, 2]) * np.cos(q[:, 0])
        return (np.abs(g) - thickness).reshape((-1, 1)) / scale
    return f

@sdf2
def my_circle(radius):
    def f(p):
        return np.sqrt(p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) - radius
    return f

@op3
def my_mirror_x(other):
    def f(p):
        q = p.copy()
        q[:, 0] = np.abs(q[:, 0])
        return other(q)
    return f
"""

CUSTOM_FIXTURES = {
    # the README's two examples, alone: the whole model is user code
    'custom_readme': "f = my_sphere(1).my_translate((0.25, -0.5, 0.125))",
    # a user leaf under library operators (the canonical example with the README's sphere)
    'custom_leaf_in_example': """
f = my_sphere(1) & box(1.5)
c = cylinder(0.5)
f -= c.orient(X) | c.orient(Y) | c.orient(Z)
""",
    # a user operator around a library model, library operators around that
    'custom_op_over_library': EXAMPLE + "f = f.my_translate((0.1, 0.2, -0.15)).rotate(pi / 5, X) | sphere(0.3).translate((0, 0, 0.9)).k(0.1)\n",
    # user leaves under transforms, smooth booleans and repeat; two closures, one of them used twice
    'custom_under_transforms': """
s = my_sphere(0.4)
f = s.translate((0.5, 0, 0)) | s.translate((-0.5, 0, 0)).scale(1.25).k(0.2)
f = f.rotate(0.3, Z) & my_gyroid(9.0, 0.35).translate((0.1, 0.1, 0.1))
""",
    # a user 2-D leaf through extrude, a user operator over a user leaf over a library leaf
    'custom_2d_and_nested': """
f = my_circle(0.6).extrude(0.5) - capsule(-Z, Z, 0.25)
f = f | box((0.5, 0.3, 0.3)).translate((0.8, 0, 0)).my_mirror_x().k(0.05)
""",
}
for _k in CUSTOM_FIXTURES:
    CUSTOM_FIXTURES[_k] = _CUSTOM_PRELUDE + CUSTOM_FIXTURES[_k]


def build(name, namespace):
    """exec the fixture program in a copy of `namespace` and return its ``f``."""
    ns = dict(namespace)
    exec(FIXTURES[name] if name in FIXTURES else CUSTOM_FIXTURES[name], ns)
    return ns['f']
