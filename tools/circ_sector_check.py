"""NumPy restatement of the float64 interpreter's sector search for circular_array (csrc/sdf_interp.h L_CIRC_PREP, sdf_amd/tape.py):
the point turned back by 2^m da, m = M .. 0, against the reference's polar form (reference sdf/d3.py:379-392) on random and
near-boundary points.  Run: python tools/circ_sector_check.py"""
import numpy as np, math
def consts_for(da):
    M = int(math.floor(math.log2(math.pi / da)))
    while (2.0 ** (M + 1)) * da <= math.pi: M += 1
    while (2.0 ** M) * da > math.pi: M -= 1
    cs = []
    for m in range(M + 1):
        ang = (2.0 ** m) * da
        cs += [math.cos(ang), math.sin(ang)]
    return M, cs
def search(x, y, da):
    M, cs = consts_for(da)
    neg = np.signbit(y)
    qx, qy = x.copy(), np.where(neg, -y, y)
    for m in range(M, -1, -1):
        cm, sm = cs[2*m], cs[2*m+1]
        rx = qx * cm + qy * sm
        ry = qy * cm - qx * sm
        acc = ry >= 0.0
        qx = np.where(acc, rx, qx); qy = np.where(acc, ry, qy)
    c0, s0 = cs[0], cs[1]
    ux, uy = qx, -qy
    nz = qy != 0.0
    rx = np.where(nz, ux * c0 - uy * s0, ux)
    ry = np.where(nz, ux * s0 + uy * c0, uy)
    gx, gy = np.where(neg, rx, qx), np.where(neg, ry, qy)
    # points ON a coordinate axis: the exact angle, its exact floored remainder, the reference's own expression (round 5;
    # fma(-k, da, A) of the device code is exact, np.fmod is the same number)
    axis = (x == 0.0) | (y == 0.0)
    A = np.where(x == 0.0, np.where(y == 0.0, np.where(np.signbit(x), np.pi, 0.0), np.pi / 2), np.where(np.signbit(x), np.pi, 0.0))
    k = np.floor(A / da)
    rm = np.fmod(A, da)                       # == A - k da exactly once k is the floor
    mm = np.where(neg & (rm != 0.0), da - rm, rm)
    dd = np.abs(x) + np.abs(y)
    gx = np.where(axis, np.cos(mm) * dd, gx); gy = np.where(axis, np.sin(mm) * dd, gy)
    return gx, gy
def reference(x, y, da):      # the reference's polar form (d3.py:379-392), delta = 0 evaluation point
    d = np.hypot(x, y); a = np.arctan2(y, x) % da
    return np.cos(a) * d, np.sin(a) * d
rng = np.random.default_rng(0)
for count in (2, 3, 5, 7, 12, 16, 18, 24, 100):
    da = 2 * np.pi / count
    n = 400000
    x = rng.normal(size=n) * 10 ** rng.uniform(-3, 3, n); y = rng.normal(size=n) * 10 ** rng.uniform(-3, 3, n)
    # add near-boundary points
    k = rng.integers(-count, count, 20000); eps = rng.choice([0, 1e-16, -1e-16, 1e-12, -1e-12, 1e-9, -1e-9], 20000)
    r = 10 ** rng.uniform(-2, 2, 20000)
    xb = r * np.cos(k * da + eps); yb = r * np.sin(k * da + eps)
    x = np.concatenate([x, xb]); y = np.concatenate([y, yb])
    gx, gy = search(x, y, da)
    wx, wy = reference(x, y, da)
    d = np.hypot(x, y)
    err = np.maximum(np.abs(gx - wx), np.abs(gy - wy)) / d
    # points whose sector differs (boundary) show up as ~da-sized rotations: count them separately
    bad = err > 1e-13
    ang_g = np.arctan2(gy, gx)
    print(count, 'max rel err (same sector)', err[~bad].max(), 'sector mismatches', int(bad.sum()), 'residual angle range', ang_g.min(), ang_g.max(), 'da', da)

# points ON the axes: the sector must be the reference's, whatever the count (an asymmetric child tells them apart)
worst = 0.0
for count in range(2, 400):
    da = 2 * np.pi / count
    r = np.array([1e-9, 0.3, 1.0, 2.0, 1e6])
    x = np.concatenate([0 * r, 0 * r, -0.0 * r, -0.0 * r, r, r, -r, -r]); y = np.concatenate([r, -r, r, -r, 0 * r, -0.0 * r, 0 * r, -0.0 * r])
    gx, gy = search(x, y, da)
    wx, wy = reference(x, y, da)
    err = np.maximum(np.abs(gx - wx), np.abs(gy - wy)) / np.hypot(x, y)
    worst = max(worst, err.max())
    assert err.max() < 1e-15, (count, err.max())
print('axis points, 2 .. 399 sectors: max rel err', worst)
