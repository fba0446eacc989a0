/*
 * sdf_oracle.c -- CPU restatement of the reference's sampling + meshing path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (sdf_amd/) never does.
 *
 * It restates, in plain scalar C double arithmetic, what the reference computes with NumPy
 * closures and scikit-image:
 *
 *   eval_node()            the closure tree             reference sdf/d3.py, sdf/d2.py, sdf/dn.py,
 *                                                       sdf/ease.py  (file:line at every case)
 *   oracle_marching_cubes  skimage 0.18.3 measure.marching_cubes(volume, 0) as called by
 *                          reference sdf/core.py:16-18 -- third-party code (Lewiner's MC33) that
 *                          is not in /root/reference; restated from the published algorithm and
 *                          its observed behaviour (oracle/mc33.h) with the
 *                          lookup tables obtained from the installed package
 *                          (oracle/mc_table.h, oracle/mc33_tables.h; tools/derive_mc*_tables.py)
 *   oracle_skip            reference sdf/core.py:28-43
 *   oracle_generate        reference sdf/core.py:45-60 (_worker) + :114-141 (batch loop)
 *   oracle_estimate_bounds reference sdf/core.py:62-82
 *
 * Parity pinning: tests/test_oracle.py checks every function here against tests/golden/ (npz files),
 * which tools/make_golden.py produced by running the unmodified reference (Python 3.9,
 * numpy 1.26.4, scikit-image 0.18.3): values, bounds, batch classification, marching cubes on
 * volumes that reach every Lewiner case, and whole `generate` soups bit for bit.
 *
 * The tree comes from sdf_amd.ir.flatten(): nodes[n][5] = {op, param_off, nparams, child_off,
 * nchildren}, params[], children[].  It is evaluated by recursion, one point at a time -- on
 * purpose a different mechanism from the product's op tape + stack machine.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mc_table.h"
#include "mc33.h"
#include "node_ops.h"

typedef struct {
    const int32_t *nodes;
    const double *params;
    const int32_t *children;
    int32_t root;
} tree_t;

/* ---- NumPy scalar semantics ------------------------------------------------------ */
/* np.minimum / np.maximum propagate NaN: (a < b || isnan(a)) ? a : b */
static inline double np_min(double a, double b) { return (a < b || a != a) ? a : b; }
static inline double np_max(double a, double b) { return (a >= b || a != a) ? a : b; }
/* np.clip = min(max(x, lo), hi) with NaN-propagating helpers (strict compares) */
static inline double np_clip(double x, double lo, double hi) {
    double t = (x != x || x > lo) ? x : lo;
    return (t != t || t < hi) ? t : hi;
}
static inline double np_sign(double x) { return x != x ? x : (x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0)); }
/* Python/NumPy floored modulo (npy_divmod) */
static inline double np_mod(double a, double b) {
    double m = fmod(a, b);
    if (b == 0) return m;
    if (m != 0) { if ((b < 0) != (m < 0)) m += b; }
    else m = copysign(0.0, b);
    return m;
}
/* np.linalg.norm(axis=1): sqrt of the left-to-right sum of squares */
static inline double len2(double x, double y) { return sqrt(x * x + y * y); }
static inline double len3(double x, double y, double z) { return sqrt((x * x + y * y) + z * z); }
/* np.dot((N,3),(3,)) and np.dot((N,3),(3,3)) go through BLAS; the summation used here (fused
 * multiply-adds, first term a plain product) is what OpenBLAS' Haswell kernels do for these
 * shapes; other BLAS builds differ in the last bit, which is why value parity is checked to a
 * few ulp and not bitwise (DESIGN.md section "numerics"). */
static inline double dot3(double ax, double ay, double az, double bx, double by, double bz) {
    return fma(az, bz, fma(ay, by, ax * bx));
}
static inline double dot2(double ax, double ay, double bx, double by) { return fma(ay, by, ax * bx); }

/* ---- easing curves: reference sdf/ease.py:3-162 ---------------------------------- */
static double out_bounce(double t) {
    if (t < 4.0 / 11) return (121 * t * t) / 16;
    if (t < 8.0 / 11) return (363.0 / 40 * t * t) - (99.0 / 10 * t) + 17.0 / 5;
    if (t < 9.0 / 10) return (4356.0 / 361 * t * t) - (35442.0 / 1805 * t) + 16061.0 / 1805;
    return (54.0 / 5 * t * t) - (513.0 / 25 * t) + 268.0 / 25;
}
static double ease_apply(int id, double t) {
    const double pi = 3.141592653589793;
    double u, v, a, b;
    switch (id) {
    case EASE_linear: return t;
    case EASE_in_quad: return t * t;
    case EASE_out_quad: return -t * (t - 2);
    case EASE_in_out_quad:
        u = 2 * t - 1; a = 2 * t * t; b = -0.5 * (u * (u - 2) - 1);
        return t < 0.5 ? a : b;
    case EASE_in_cubic: return t * t * t;
    case EASE_out_cubic: u = t - 1; return u * u * u + 1;
    case EASE_in_out_cubic:
        u = t * 2; v = u - 2;
        return u < 1 ? 0.5 * u * u * u : 0.5 * (v * v * v + 2);
    case EASE_in_quart: return t * t * t * t;
    case EASE_out_quart: u = t - 1; return -(u * u * u * u - 1);
    case EASE_in_out_quart:
        u = t * 2; v = u - 2;
        return u < 1 ? 0.5 * u * u * u * u : -0.5 * (v * v * v * v - 2);
    case EASE_in_quint: return t * t * t * t * t;
    case EASE_out_quint: u = t - 1; return u * u * u * u * u + 1;
    case EASE_in_out_quint:
        u = t * 2; v = u - 2;
        return u < 1 ? 0.5 * u * u * u * u * u : 0.5 * (v * v * v * v * v + 2);
    case EASE_in_sine: return -cos(t * pi / 2) + 1;
    case EASE_out_sine: return sin(t * pi / 2);
    case EASE_in_out_sine: return -0.5 * (cos(pi * t) - 1);
    case EASE_in_expo: return t == 0 ? 0.0 : pow(2.0, 10 * (t - 1));
    case EASE_out_expo: return t == 1 ? 1.0 : 1 - pow(2.0, -10 * t);
    case EASE_in_out_expo:
        if (t == 0) return 0.0;
        if (t == 1) return 1.0;
        return t < 0.5 ? 0.5 * pow(2.0, 20 * t - 10) : 1 - 0.5 * pow(2.0, -20 * t + 10);
    case EASE_in_circ: return -1 * (sqrt(1 - t * t) - 1);
    case EASE_out_circ: u = t - 1; return sqrt(1 - u * u);
    case EASE_in_out_circ:
        u = t * 2; v = u - 2;
        return u < 1 ? -0.5 * (sqrt(1 - u * u) - 1) : 0.5 * (sqrt(1 - v * v) + 1);
    case EASE_in_elastic: {
        const double k = 0.5; u = t - 1;
        return -1 * (pow(2.0, 10 * u) * sin((u - k / 4) * (2 * pi) / k)); }
    case EASE_out_elastic: {
        const double k = 0.5;
        return pow(2.0, -10 * t) * sin((t - k / 4) * (2 * pi / k)) + 1; }
    case EASE_in_out_elastic: {
        const double k = 0.5; u = t * 2; v = u - 1;
        a = -0.5 * (pow(2.0, 10 * v) * sin((v - k / 4) * 2 * pi / k));
        b = pow(2.0, -10 * v) * sin((v - k / 4) * 2 * pi / k) * 0.5 + 1;
        return u < 1 ? a : b; }
    case EASE_in_back: { const double k = 1.70158; return t * t * ((k + 1) * t - k); }
    case EASE_out_back: { const double k = 1.70158; u = t - 1; return u * u * ((k + 1) * u + k) + 1; }
    case EASE_in_out_back: {
        const double k = 1.70158 * 1.525; u = t * 2; v = u - 2;
        return u < 1 ? 0.5 * (u * u * ((k + 1) * u - k)) : 0.5 * (v * v * ((k + 1) * v + k) + 2); }
    case EASE_in_bounce: return 1 - out_bounce(1 - t);
    case EASE_out_bounce: return out_bounce(t);
    case EASE_in_out_bounce:
        return t < 0.5 ? (1 - out_bounce(1 - 2 * t)) * 0.5 : out_bounce(2 * t - 1) * 0.5 + 0.5;
    case EASE_in_square: return t < 1 ? 0.0 : 1.0;
    case EASE_out_square: return t > 0 ? 1.0 : 0.0;
    case EASE_in_out_square: return t < 0.5 ? 0.0 : 1.0;
    }
    return NAN;
}

/* ---- booleans: reference sdf/dn.py:7-58 ------------------------------------------ */
static double fold_boolean(int op, double d1, double d2, int has_k, double K) {
    double h, m;
    switch (op) {
    case NODE_union:
        if (!has_k) return np_min(d1, d2);
        h = np_clip(0.5 + 0.5 * (d2 - d1) / K, 0, 1);
        m = d2 + (d1 - d2) * h;
        return m - K * h * (1 - h);
    case NODE_difference:
        if (!has_k) return np_max(d1, -d2);
        h = np_clip(0.5 - 0.5 * (d2 + d1) / K, 0, 1);
        m = d1 + (-d2 - d1) * h;
        return m + K * h * (1 - h);
    case NODE_intersection:
        if (!has_k) return np_max(d1, d2);
        h = np_clip(0.5 - 0.5 * (d2 - d1) / K, 0, 1);
        m = d2 + (d1 - d2) * h;
        return m + K * h * (1 - h);
    case NODE_blend:
        return K * d2 + (1 - K) * d1;
    }
    return NAN;
}

static double box_like(double qx, double qy, double qz) {
    /* _length(_max(q, 0)) + _min(np.amax(q, axis=1), 0) */
    double mx = np_max(np_max(qx, qy), qz);
    return len3(np_max(qx, 0), np_max(qy, 0), np_max(qz, 0)) + np_min(mx, 0);
}

static double eval_node(const tree_t *t, int id, const double *p);

static inline double child(const tree_t *t, const int32_t *n, int i, const double *p) {
    return eval_node(t, t->children[n[3] + i], p);
}

static double eval_node(const tree_t *t, int id, const double *p) {
    const int32_t *n = t->nodes + 5 * id;
    const double *c = t->params + n[1];
    const double x = p[0], y = p[1], z = p[2];
    double q[3];
    switch (n[0]) {
    /* ---------------- 3-D leaves ---------------- */
    case NODE_sphere:  /* d3.py:92-96 */
        return len3(x - c[1], y - c[2], z - c[3]) - c[0];
    case NODE_plane:   /* d3.py:98-103: np.dot(point - p, normal) */
        return dot3(c[3] - x, c[4] - y, c[5] - z, c[0], c[1], c[2]);
    case NODE_box:     /* d3.py:122-134 */
        return box_like(fabs(x - c[0]) - c[3], fabs(y - c[1]) - c[4], fabs(z - c[2]) - c[5]);
    case NODE_rounded_box: /* d3.py:136-142 */
        return box_like(fabs(x) - c[0] + c[3], fabs(y) - c[1] + c[3], fabs(z) - c[2] + c[3]) - c[3];
    case NODE_wireframe_box: { /* d3.py:144-155 */
        double t2 = c[3];
        double px = fabs(x) - c[0] - t2, py = fabs(y) - c[1] - t2, pz = fabs(z) - c[2] - t2;
        double qx = fabs(px + t2) - t2, qy = fabs(py + t2) - t2, qz = fabs(pz + t2) - t2;
#define WG(a, b, cc) (len3(np_max(a, 0), np_max(b, 0), np_max(cc, 0)) + np_min(np_max(a, np_max(b, cc)), 0))
        double g1 = WG(px, qy, qz), g2 = WG(qx, py, qz), g3 = WG(qx, qy, pz);
#undef WG
        return np_min(np_min(g1, g2), g3); }
    case NODE_torus: { /* d3.py:157-165 */
        double a = len2(x, y) - c[0];
        return len2(a, z) - c[1]; }
    case NODE_capsule: { /* d3.py:167-176 */
        double pax = x - c[0], pay = y - c[1], paz = z - c[2];
        double h = np_clip(dot3(pax, pay, paz, c[3], c[4], c[5]) / c[6], 0, 1);
        return len3(pax - c[3] * h, pay - c[4] * h, paz - c[5] * h) - c[7]; }
    case NODE_cylinder: /* d3.py:178-182 */
        return len2(x, y) - c[0];
    case NODE_capped_cylinder: { /* d3.py:184-204 */
        double bax = c[3], bay = c[4], baz = c[5], baba = c[6];
        double pax = x - c[0], pay = y - c[1], paz = z - c[2];
        double paba = dot3(pax, pay, paz, bax, bay, baz);
        double xx = len3(pax * baba - bax * paba, pay * baba - bay * paba, paz * baba - baz * paba) - c[8];
        double yy = fabs(paba - c[9]) - c[9];
        double x2 = xx * xx, y2 = yy * yy * baba, d;
        if (np_max(xx, yy) < 0) d = -np_min(x2, y2);
        else d = (xx > 0 ? x2 : 0) + (yy > 0 ? y2 : 0);
        return np_sign(d) * sqrt(fabs(d)) / baba; }
    case NODE_rounded_cylinder: { /* d3.py:206-215 */
        double d0 = len2(x, y) - c[0] + c[1];
        double d1 = fabs(z) - c[2] + c[1];
        return np_min(np_max(d0, d1), 0) + len2(np_max(d0, 0), np_max(d1, 0)) - c[1]; }
    case NODE_capped_cone: { /* d3.py:217-237 */
        double ra = c[6], rb = c[7], baba = c[8], rba = c[9], k = c[10];
        double pax = x - c[0], pay = y - c[1], paz = z - c[2];
        double papa = (pax * pax + pay * pay) + paz * paz;
        double paba = dot3(pax, pay, paz, c[3], c[4], c[5]) / baba;
        double xx = sqrt(papa - paba * paba * baba);
        double cax = np_max(0, xx - (paba < 0.5 ? ra : rb));
        double cay = fabs(paba - 0.5) - 0.5;
        double f = np_clip((rba * (xx - ra) + paba * baba) / k, 0, 1);
        double cbx = xx - ra - f * rba;
        double cby = paba - f;
        double s = (cbx < 0 && cay < 0) ? -1 : 1;
        return s * sqrt(np_min(cax * cax + cay * cay * baba, cbx * cbx + cby * cby * baba)); }
    case NODE_rounded_cone: { /* d3.py:239-250 */
        double r1 = c[0], r2 = c[1], h = c[2], b = c[3], a = c[4], ah = c[5];
        double qx = len2(x, y), qy = z;
        double k = dot2(qx, qy, -b, a);
        double c1 = len2(qx, qy) - r1;
        double c2 = len2(qx - 0, qy - h) - r2;
        double c3 = dot2(qx, qy, a, b) - r1;
        return k < 0 ? c1 : (k > ah ? c2 : c3); }
    case NODE_ellipsoid: { /* d3.py:252-259 */
        double k0 = len3(x / c[0], y / c[1], z / c[2]);
        double k1 = len3(x / c[3], y / c[4], z / c[5]);
        return k0 * (k0 - 1) / k1; }
    case NODE_pyramid: { /* d3.py:261-282 */
        double h = c[0], m2 = c[1], m2q = c[2];
        double a0 = fabs(x) - 0.5, a1 = fabs(y) - 0.5;
        if (a1 > a0) { double tmp = a0; a0 = a1; a1 = tmp; }
        double px = a0, py = z, pz = a1;
        double qx = pz, qy = h * py - 0.5 * px, qz = h * px + 0.5 * py;
        double s = np_max(-qx, 0);
        double tt = np_clip((qy - 0.5 * pz) / m2q, 0, 1);
        double a = m2 * ((qx + s) * (qx + s)) + qy * qy;
        double b = m2 * ((qx + 0.5 * tt) * (qx + 0.5 * tt)) + (qy - m2 * tt) * (qy - m2 * tt);
        double d2 = np_min(qy, -qx * m2 - qy * 0.5) > 0 ? 0 : np_min(a, b);
        return sqrt((d2 + qz * qz) / m2) * np_sign(np_max(qz, -py)); }
    case NODE_tetrahedron: /* d3.py:286-293 */
        return (np_max(fabs(x + y) - z, fabs(x - y) + z) - c[0]) / c[1];
    case NODE_octahedron:  /* d3.py:295-299: np.sum(np.abs(p), axis=1) */
        return (((fabs(x) + fabs(y)) + fabs(z)) - c[0]) * c[1];
    case NODE_dodecahedron: { /* d3.py:301-311 */
        double r = c[0], X = c[1], Y = c[2], Z = c[3];
        double ax = fabs(x / r), ay = fabs(y / r), az = fabs(z / r);
        double a = dot3(ax, ay, az, X, Y, Z), b = dot3(ax, ay, az, Z, X, Y), cc = dot3(ax, ay, az, Y, Z, X);
        return (np_max(np_max(a, b), cc) - X) * r; }
    case NODE_icosahedron: { /* d3.py:313-325 */
        double r = c[0], X = c[1], Y = c[2], Z = c[3], w = c[4];
        double ax = fabs(x / r), ay = fabs(y / r), az = fabs(z / r);
        double a = dot3(ax, ay, az, X, Y, Z), b = dot3(ax, ay, az, Z, X, Y), cc = dot3(ax, ay, az, Y, Z, X);
        double d = dot3(ax, ay, az, w, w, w) - X;
        return np_max(np_max(np_max(a, b), cc) - X, d) * r; }
    /* ---------------- 3-D transforms ---------------- */
    case NODE_translate: /* d3.py:329-333 */
        q[0] = x - c[0]; q[1] = y - c[1]; q[2] = z - c[2];
        return child(t, n, 0, q);
    case NODE_scale:     /* d3.py:335-345 */
        q[0] = x / c[0]; q[1] = y / c[1]; q[2] = z / c[2];
        return child(t, n, 0, q) * c[3];
    case NODE_rotate:    /* d3.py:347-360: np.dot(p, matrix), matrix row-major in c[0..8] */
        q[0] = dot3(x, y, z, c[0], c[3], c[6]);
        q[1] = dot3(x, y, z, c[1], c[4], c[7]);
        q[2] = dot3(x, y, z, c[2], c[5], c[8]);
        return child(t, n, 0, q);
    case NODE_circular_array: { /* d3.py:379-392 */
        double da = c[0];
        double d = hypot(x, y);
        double a = np_mod(atan2(y, x), da);
        q[0] = cos(a - da) * d; q[1] = sin(a - da) * d; q[2] = z;
        double d1 = child(t, n, 0, q);
        q[0] = cos(a) * d; q[1] = sin(a) * d; q[2] = z;
        double d2 = child(t, n, 0, q);
        return np_min(d1, d2); }
    case NODE_elongate: { /* d3.py:396-405 */
        double qx = fabs(x) - c[0], qy = fabs(y) - c[1], qz = fabs(z) - c[2];
        double w = np_min(np_max(qx, np_max(qy, qz)), 0);
        q[0] = np_max(qx, 0); q[1] = np_max(qy, 0); q[2] = np_max(qz, 0);
        return child(t, n, 0, q) + w; }
    case NODE_twist: {  /* d3.py:407-419 */
        double cc = cos(c[0] * z), s = sin(c[0] * z);
        q[0] = cc * x - s * y; q[1] = s * x + cc * y; q[2] = z;
        return child(t, n, 0, q); }
    case NODE_bend: {   /* d3.py:421-433 */
        double cc = cos(c[0] * x), s = sin(c[0] * x);
        q[0] = cc * x - s * y; q[1] = s * x + cc * y; q[2] = z;
        return child(t, n, 0, q); }
    case NODE_bend_linear: { /* d3.py:435-445 */
        double tt = np_clip(dot3(x - c[0], y - c[1], z - c[2], c[3], c[4], c[5]) / c[6], 0, 1);
        tt = ease_apply((int)c[10], tt);
        q[0] = x + tt * c[7]; q[1] = y + tt * c[8]; q[2] = z + tt * c[9];
        return child(t, n, 0, q); }
    case NODE_bend_radial: { /* d3.py:447-457 */
        double r = hypot(x, y);
        double tt = np_clip((r - c[0]) / c[1], 0, 1);
        q[0] = x; q[1] = y; q[2] = z - c[2] * ease_apply((int)c[3], tt);
        return child(t, n, 0, q); }
    case NODE_wrap_around: { /* d3.py:483-502 */
        const double pi = 3.141592653589793;
        double d = hypot(x, y) - c[9];
        double a = atan2(y, x);
        double tt = ease_apply((int)c[10], (a + pi) / (2 * pi));
        q[0] = c[0] + c[3] * tt + c[6] * d;
        q[1] = c[1] + c[4] * tt + c[7] * d;
        q[2] = z;
        return child(t, n, 0, q); }
    case NODE_transition_linear: { /* d3.py:459-470 */
        double d1 = child(t, n, 0, p), d2 = child(t, n, 1, p);
        double tt = np_clip(dot3(x - c[0], y - c[1], z - c[2], c[3], c[4], c[5]) / c[6], 0, 1);
        tt = ease_apply((int)c[7], tt);
        return tt * d2 + (1 - tt) * d1; }
    case NODE_transition_radial: { /* d3.py:472-481 */
        double d1 = child(t, n, 0, p), d2 = child(t, n, 1, p);
        double r = hypot(x, y);
        double tt = ease_apply((int)c[2], np_clip((r - c[0]) / c[1], 0, 1));
        return tt * d2 + (1 - tt) * d1; }
    /* ---------------- dimension-agnostic ---------------- */
    case NODE_union: case NODE_difference: case NODE_intersection: case NODE_blend: {
        /* dn.py:7-58; params are (has_k, K) per right operand, resolved by the front end */
        double d1 = child(t, n, 0, p);
        for (int i = 1; i < n[4]; i++) {
            double d2 = child(t, n, i, p);
            d1 = fold_boolean(n[0], d1, d2, c[2 * (i - 1)] != 0, c[2 * (i - 1) + 1]);
        }
        return d1; }
    case NODE_negate: return -child(t, n, 0, p);            /* dn.py:60-63 */
    case NODE_dilate: return child(t, n, 0, p) - c[0];      /* dn.py:65-68 */
    case NODE_erode:  return child(t, n, 0, p) + c[0];      /* dn.py:70-73 */
    case NODE_shell:  return fabs(child(t, n, 0, p)) - c[0]; /* dn.py:75-78 */
    case NODE_repeat: { /* dn.py:80-112; params: dim, s[3], has_count, count[3], nn, n[nn][3] */
        int dim = (int)c[0], nn = (int)c[8];
        double idx[3] = {0, 0, 0};
        for (int i = 0; i < dim; i++) {
            double s = c[1 + i];
            double qq = s != 0 ? p[i] / s : 0.0;
            double r = nearbyint(qq);                      /* np.round: half to even */
            if (c[4] != 0) r = np_clip(r, -c[5 + i], c[5 + i]);
            idx[i] = r;
        }
        double best = 0;
        for (int k = 0; k < nn; k++) {
            q[0] = x; q[1] = y; q[2] = z;
            for (int i = 0; i < dim; i++) q[i] = p[i] - c[1 + i] * (idx[i] + c[9 + 3 * k + i]);
            double d = child(t, n, 0, q);
            best = k == 0 ? d : np_min(best, d);
        }
        return best; }
    /* ---------------- 2-D leaves: point is (x, y) ---------------- */
    case NODE_circle:  /* d2.py:76-80 */
        return len2(x - c[1], y - c[2]) - c[0];
    case NODE_line:    /* d2.py:82-87 */
        return dot2(c[2] - x, c[3] - y, c[0], c[1]);
    case NODE_rectangle: { /* d2.py:102-114 */
        double qx = fabs(x - c[0]) - c[2], qy = fabs(y - c[1]) - c[3];
        return len2(np_max(qx, 0), np_max(qy, 0)) + np_min(np_max(qx, qy), 0); }
    case NODE_rounded_rectangle: { /* d2.py:116-134 */
        double r = 0;
        if (x > 0 && y > 0) r = c[2];
        if (x > 0 && y <= 0) r = c[3];
        if (x <= 0 && y <= 0) r = c[4];
        if (x <= 0 && y > 0) r = c[5];
        double qx = fabs(x) - c[0] + r, qy = fabs(y) - c[1] + r;
        return np_min(np_max(qx, qy), 0) + len2(np_max(qx, 0), np_max(qy, 0)) - r; }
    case NODE_equilateral_triangle: { /* d2.py:136-152 */
        double k = c[0];
        double px = fabs(x) - 1, py = y + c[1];
        if (px + k * py > 0) {
            double nx = (px - k * py) / 2, ny = (-k * px - py) / 2;
            px = nx; py = ny;
        }
        px = px - np_clip(px, -2, 0);
        return -len2(px, py) * np_sign(py); }
    case NODE_hexagon: { /* d2.py:154-165 */
        double r = c[0], k0 = c[1], k1 = c[2];
        double px = fabs(x), py = fabs(y);
        double m = np_min(k0 * px + k1 * py, 0);
        px -= c[4] * m; py -= c[5] * m;
        px -= np_clip(px, c[6], c[7]); py -= (0.0 + r);
        return len2(px, py) * np_sign(py); }
    case NODE_rounded_x: { /* d2.py:167-173 */
        double px = fabs(x), py = fabs(y);
        double qq = np_min(px + py, c[0]) * 0.5;
        return len2(px - qq, py - qq) - c[1]; }
    case NODE_polygon: { /* d2.py:175-196 */
        int np_ = (int)c[0];
        const double *v = c + 1;
        double dx = x - v[0], dy = y - v[1];
        double d = dx * dx + dy * dy;
        double s = 1.0;
        for (int i = 0; i < np_; i++) {
            int j = (i + np_ - 1) % np_;
            double vix = v[2 * i], viy = v[2 * i + 1], vjx = v[2 * j], vjy = v[2 * j + 1];
            double ex = vjx - vix, ey = vjy - viy;
            double wx = x - vix, wy = y - viy;
            double ee = dot2(ex, ey, ex, ey);  /* np.dot(e, e): 1-D ddot */
            double cl = np_clip(dot2(wx, wy, ex, ey) / ee, 0, 1);
            double bx = wx - ex * cl, by = wy - ey * cl;
            d = np_min(d, bx * bx + by * by);
            int c1 = y >= viy, c2 = y < vjy, c3 = ex * wy > ey * wx;
            if ((c1 && c2 && c3) || (!c1 && !c2 && !c3)) s = -s;
        }
        return s * sqrt(d); }
    case NODE_vesica: { /* d2.py:198-207 */
        double r = c[0], d = c[1], b = c[2];
        double px = fabs(x), py = fabs(y);
        if ((py - b) * d > px * b) return len2(px - 0, py - b);
        return len2(px - (-d), py - 0) - r; }
    /* ---------------- 2-D transforms ---------------- */
    case NODE_texture2d: { /* text.py:116-153: bilinear lookup into the distance texture, fallback
                            * rectangle outside it.  c: x0 y0 x1 y1 pw ph px py tw th rect(4) tex[th][tw] */
        long tw = (long)c[8], th = (long)c[9];
        const double *tex = c + 14;
        double u = (x - c[0]) / (c[2] - c[0]);
        double v = (y - c[1]) / (c[3] - c[1]);
        v = 1 - v;
        double fi = u * c[4] + c[6], fj = v * c[5] + c[7];
        /* _bilinear_interpolate (:138-153): np.floor(...).astype(int), then np.clip */
        double gi = floor(fi), gj = floor(fj);
        long a0 = gi != gi ? 0 : (gi < -2 ? -2 : (gi > (double)tw ? tw : (long)gi));
        long b0 = gj != gj ? 0 : (gj < -2 ? -2 : (gj > (double)th ? th : (long)gj));
        long ix0 = a0 < 0 ? 0 : (a0 > tw - 1 ? tw - 1 : a0), ix1 = a0 + 1 < 0 ? 0 : (a0 + 1 > tw - 1 ? tw - 1 : a0 + 1);
        long iy0 = b0 < 0 ? 0 : (b0 > th - 1 ? th - 1 : b0), iy1 = b0 + 1 < 0 ? 0 : (b0 + 1 > th - 1 ? th - 1 : b0 + 1);
        double pa = tex[iy0 * tw + ix0], pb = tex[iy1 * tw + ix0], pc = tex[iy0 * tw + ix1], pd = tex[iy1 * tw + ix1];
        double wa = ((double)ix1 - fi) * ((double)iy1 - fj), wb = ((double)ix1 - fi) * (fj - (double)iy0);
        double wc = (fi - (double)ix0) * ((double)iy1 - fj), wd = (fi - (double)ix0) * (fj - (double)iy0);
        double d = wa * pa + wb * pb + wc * pc + wd * pd;
        double qx = fabs(x - c[10]) - c[12], qy = fabs(y - c[11]) - c[13];
        double qd = len2(np_max(qx, 0), np_max(qy, 0)) + np_min(np_max(qx, qy), 0);
        int outside = (fi < 0) || (fi >= (double)(tw - 1)) || (fj < 0) || (fj >= (double)(th - 1));
        return outside ? qd : d; }
    case NODE_grid3d: { /* mesh.py:96-105 `f`: np.where(e > background, e, interpolator(p)) with e = box(a=a, b=b)
                         * (d3.py:122-134) and scipy 1.7.1 RegularGridInterpolator(method='linear', bounds_error=False,
                         * fill_value=background): _find_indices (searchsorted - 1 clipped to [0, n-2], out of bounds =
                         * x < grid[0] or x > grid[-1]) and _evaluate_linear (values = 0.; for the 8 corners in
                         * itertools.product order: weight = ((1. * wx) * wy) * wz; values += float32 voxel * weight).
                         * c: nx ny nz background box-centre(3) box-half-size(3) X[nx] Y[ny] Z[nz] A[nx][ny][nz] */
        long n3[3] = {(long)c[0], (long)c[1], (long)c[2]};
        double bg = c[3];
        const double *g[3] = {c + 10, c + 10 + n3[0], c + 10 + n3[0] + n3[1]};
        const double *vox = c + 10 + n3[0] + n3[1] + n3[2];
        double qx = fabs(x - c[4]) - c[7], qy = fabs(y - c[5]) - c[8], qz = fabs(z - c[6]) - c[9];
        double e = len3(np_max(qx, 0), np_max(qy, 0), np_max(qz, 0)) + np_min(np_max(np_max(qx, qy), qz), 0);
        double pp[3] = {x, y, z}, w[3];
        long idx[3];
        int oob = 0;
        for (int a = 0; a < 3; a++) {
            long lo = 0, hi = n3[a];
            while (lo < hi) { /* np.searchsorted side='left'; NaN sorts last */
                long mid = (lo + hi) >> 1;
                double gm = g[a][mid];
                if (gm < pp[a] || (pp[a] != pp[a] && gm == gm)) lo = mid + 1; else hi = mid;
            }
            long i = lo - 1; if (i < 0) i = 0; if (i > n3[a] - 2) i = n3[a] - 2;
            idx[a] = i;
            w[a] = (pp[a] - g[a][i]) / (g[a][i + 1] - g[a][i]);
            oob = oob || pp[a] < g[a][0] || pp[a] > g[a][n3[a] - 1];
        }
        double values = 0.0;
        for (int k = 0; k < 8; k++) {
            int o0 = k >> 2, o1 = (k >> 1) & 1, o2 = k & 1;
            double weight = 1.0;
            weight *= o0 ? w[0] : 1 - w[0];
            weight *= o1 ? w[1] : 1 - w[1];
            weight *= o2 ? w[2] : 1 - w[2];
            values += vox[((idx[0] + o0) * n3[1] + (idx[1] + o1)) * n3[2] + (idx[2] + o2)] * weight;
        }
        double d = oob ? bg : values;
        return e > bg ? e : d; }
    case NODE_translate2: q[0] = x - c[0]; q[1] = y - c[1]; q[2] = z; return child(t, n, 0, q);
    case NODE_scale2: q[0] = x / c[0]; q[1] = y / c[1]; q[2] = z; return child(t, n, 0, q) * c[2];
    case NODE_rotate2: /* d2.py:229-240: np.dot(p, matrix), matrix row-major c[0..3] */
        q[0] = dot2(x, y, c[0], c[2]); q[1] = dot2(x, y, c[1], c[3]); q[2] = z;
        return child(t, n, 0, q);
    case NODE_elongate2: { /* d2.py:249-257 */
        double qx = fabs(x) - c[0], qy = fabs(y) - c[1];
        double w = np_min(np_max(qx, qy), 0);
        q[0] = np_max(qx, 0); q[1] = np_max(qy, 0); q[2] = z;
        return child(t, n, 0, q) + w; }
    /* ---------------- dimension changes ---------------- */
    case NODE_extrude: { /* d2.py:261-267 */
        double d = child(t, n, 0, p);
        double w1 = fabs(z) - c[0];
        return np_min(np_max(d, w1), 0) + len2(np_max(d, 0), np_max(w1, 0)); }
    case NODE_extrude_to: { /* d2.py:269-278 */
        double d1 = child(t, n, 0, p), d2 = child(t, n, 1, p);
        double tt = ease_apply((int)c[2], np_clip(z / c[0], -0.5, 0.5) + 0.5);
        double d = d1 + (d2 - d1) * tt;
        double w1 = fabs(z) - c[1];
        return np_min(np_max(d, w1), 0) + len2(np_max(d, 0), np_max(w1, 0)); }
    case NODE_revolve: /* d2.py:280-286 */
        q[0] = len2(x, y) - c[0]; q[1] = z; q[2] = 0;
        return child(t, n, 0, q);
    case NODE_slice: { /* d3.py:506-520 */
        q[0] = x; q[1] = y; q[2] = 0.0;
        double A = child(t, n, 0, q);
        double B = -child(t, n, 1, q);
        return A <= 0 ? B : A; }
    }
    return NAN;
}

/* ================================================================================== */
/* public C API (called through ctypes from tests/ and bench.py's cpu_baseline leg)    */
/* ================================================================================== */

static tree_t mk_tree(const int32_t *nodes, const double *params, const int32_t *children, int32_t root) {
    tree_t t = {nodes, params, children, root};
    return t;
}

/* f(P): reference sdf/d3.py:24-25 */
void sdf_oracle_eval_tree(const int32_t *nodes, const double *params, const int32_t *children,
                          int32_t root, const double *pts, int64_t n, int dim, double *out) {
    tree_t t = mk_tree(nodes, params, children, root);
    for (int64_t i = 0; i < n; i++) {
        double p[3] = {pts[i * dim], pts[i * dim + 1], dim > 2 ? pts[i * dim + 2] : 0.0};
        out[i] = eval_node(&t, root, p);
    }
}

/* ---- marching cubes ------------------------------------------------------------- */
/* one cell's triangles appended to out (3 floats per vertex, volume axis order).
 * vertex on the edge between samples v_lo (offset 0) and v_hi (offset 1):
 *   w = 1/(eps + |v|), t = w_hi / (w_lo + w_hi), coordinate = float(base + t)    */
static const double MC_EPS = 2.220446049250313e-16;

static inline int mc_config(const float *v, int64_t s0, int64_t s1) {
    /* bit c = 4*o0+2*o1+o2 set when sample > 0 (value <= 0 is inside) */
    int cfg = 0;
    for (int c = 0; c < 8; c++) {
        float val = v[(c >> 2) * s0 + ((c >> 1) & 1) * s1 + (c & 1)];
        if (val > 0.0f) cfg |= 1 << c;
    }
    return cfg;
}

static inline void mc_vertex(const float *v, int64_t s0, int64_t s1, int i0, int i1, int i2, int e, float *o) {
    int axis = e >> 2, oa = (e >> 1) & 1, ob = e & 1;
    int off[3];
    if (axis == 0) { off[0] = 0; off[1] = oa; off[2] = ob; }
    else if (axis == 1) { off[0] = oa; off[1] = 0; off[2] = ob; }
    else { off[0] = oa; off[1] = ob; off[2] = 0; }
    int64_t stride = axis == 0 ? s0 : (axis == 1 ? s1 : 1);
    int64_t base = off[0] * s0 + off[1] * s1 + off[2];
    double vlo = (double)v[base], vhi = (double)v[base + stride];
    double wlo = 1.0 / (MC_EPS + fabs(vlo)), whi = 1.0 / (MC_EPS + fabs(vhi));
    double tt = whi / (wlo + whi);
    double pos[3] = {(double)(i0 + off[0]), (double)(i1 + off[1]), (double)(i2 + off[2])};
    pos[axis] = (double)(axis == 0 ? i0 : (axis == 1 ? i1 : i2)) + tt;
    o[0] = (float)pos[0]; o[1] = (float)pos[1]; o[2] = (float)pos[2];
}

/* centre vertex of a cell (Lewiner's 13th vertex) as skimage places it: the centre of mass of
 * the 8 corners weighted by w = 1/(eps + |v|), float64 arithmetic, stored as float32 */
static void mc_centre_vertex(const double *lv, int i0, int i1, int i2, float *o) {
    static const int X[8] = {0, 1, 1, 0, 0, 1, 1, 0}, Y[8] = {0, 0, 1, 1, 0, 0, 1, 1}, Z[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    double fx = 0, fy = 0, fz = 0, ff = 0;
    for (int p = 0; p < 8; p++) {
        double w = 1.0 / (MC_EPS + fabs(lv[p]));
        fx += X[p] * w; fy += Y[p] * w; fz += Z[p] * w; ff += w;
    }
    /* Lewiner x, y, z = volume axes 2, 1, 0 */
    o[0] = (float)((double)i0 + fz / ff); o[1] = (float)((double)i1 + fy / ff); o[2] = (float)((double)i2 + fx / ff);
}

/* volume: C-order float32 (n0,n1,n2).  Returns the triangle count; writes up to cap
 * triangles (9 floats each) to out.  Counts also the cells whose sign configuration is
 * ambiguous (they are tiled by Lewiner's tests, mc33.h; all others by the classic table, which
 * is what Lewiner's tables reduce to there -- tools/derive_mc33_tables.py checks that).
 * Pre-checks of skimage: any dim < 2, or level outside [min,max] -> no surface. */
int64_t sdf_oracle_marching_cubes(const float *vol, int n0, int n1, int n2, float *out, int64_t cap,
                                  int64_t *n_ambiguous) {
    int64_t nt = 0, namb = 0;
    if (n0 < 2 || n1 < 2 || n2 < 2) { if (n_ambiguous) *n_ambiguous = 0; return 0; }
    const int64_t s1 = n2, s0 = (int64_t)n1 * n2;
    for (int i0 = 0; i0 < n0 - 1; i0++)
        for (int i1 = 0; i1 < n1 - 1; i1++)
            for (int i2 = 0; i2 < n2 - 1; i2++) {
                const float *v = vol + i0 * s0 + i1 * s1 + i2;
                int cfg = mc_config(v, s0, s1);
                if (cfg == 0 || cfg == 255) continue;
                if (!MC_AMBIGUOUS[cfg]) {
                    int k = MC_NTRI[cfg];
                    for (int j = 0; j < 3 * k; j++)
                        if (nt + j / 3 < cap) mc_vertex(v, s0, s1, i0, i1, i2, MC_TRI[cfg][j], out + (nt + j / 3) * 9 + (j % 3) * 3);
                    nt += k;
                    continue;
                }
                namb++;
                double lv[8];   /* the cell in Lewiner corner order */
                for (int p = 0; p < 8; p++) {
                    int c = MC33_CORNER[p];
                    lv[p] = (double)v[(c >> 2) * s0 + ((c >> 1) & 1) * s1 + (c & 1)];
                }
                const signed char *til;
                int k = mc33_cell(lv, &til);
                for (int j = 0; j < k; j++) {
                    if (nt + j >= cap) continue;
                    for (int q = 0; q < 3; q++) {
                        int e = til[3 * j + 2 - q];   /* gradient_direction='descent' flips every face */
                        float *o = out + (nt + j) * 9 + q * 3;
                        if (e == 12) mc_centre_vertex(lv, i0, i1, i2, o);
                        else mc_vertex(v, s0, s1, i0, i1, i2, MC33_EDGE[e], o);
                    }
                }
                nt += k;
            }
    if (n_ambiguous) *n_ambiguous = namb;
    return nt;
}

/* ---- the batch pipeline ---------------------------------------------------------- */
/* reference sdf/core.py:28-43 */
static int oracle_skip(const tree_t *t, const double *X, int nx, const double *Y, int ny, const double *Z, int nz) {
    double x0 = X[0], x1 = X[nx - 1], y0 = Y[0], y1 = Y[ny - 1], z0 = Z[0], z1 = Z[nz - 1];
    double c[3] = {(x0 + x1) / 2, (y0 + y1) / 2, (z0 + z1) / 2};
    double r = fabs(eval_node(t, t->root, c));
    double d = len3(c[0] - x0, c[1] - y0, c[2] - z0);
    if (r <= d) return 0;
    double v[8];
    int k = 0;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int cc = 0; cc < 2; cc++) {
        double p[3] = {a ? x1 : x0, b ? y1 : y0, cc ? z1 : z0};
        v[k++] = eval_node(t, t->root, p);
    }
    int pos = v[0] > 0;
    for (k = 0; k < 8; k++) if (pos ? !(v[k] > 0) : !(v[k] < 0)) return 0;
    return 1;
}

typedef struct {
    double *tris;       /* 9 doubles per triangle, world coordinates */
    int64_t ntri, cap;
    uint8_t *kinds;     /* per batch: 0 skipped, 1 empty, 2 nonempty */
    int64_t nbatches;
    int64_t n_eval;     /* grid samples evaluated (batches that were not skipped) */
    int64_t n_ambiguous;
} oracle_result;

/* reference sdf/core.py:110-141 + _worker :45-60.  X, Y, Z are the np.arange axes. */
oracle_result *sdf_oracle_generate(const int32_t *nodes, const double *params, const int32_t *children,
                                   int32_t root, const double *X, int nx, const double *Y, int ny,
                                   const double *Z, int nz, int batch, int sparse,
                                   int64_t batch_begin, int64_t batch_end) {
    tree_t t = mk_tree(nodes, params, children, root);
    oracle_result *R = (oracle_result *)calloc(1, sizeof(*R));
    int bx = (nx + batch - 1) / batch, by = (ny + batch - 1) / batch, bz = (nz + batch - 1) / batch;
    R->nbatches = (int64_t)bx * by * bz;
    R->kinds = (uint8_t *)calloc(R->nbatches ? R->nbatches : 1, 1);
    R->cap = 1 << 16;
    R->tris = (double *)malloc(sizeof(double) * 9 * R->cap);
    int m = batch + 1;
    float *vol = (float *)malloc(sizeof(float) * m * m * m);
    float *loc = (float *)malloc(sizeof(float) * 9 * 12 * (size_t)batch * batch * batch);
    if (batch_end < 0 || batch_end > R->nbatches) batch_end = R->nbatches;
    for (int64_t b = batch_begin; b < batch_end; b++) {
        int ibx = (int)(b / ((int64_t)by * bz)), iby = (int)((b / bz) % by), ibz = (int)(b % bz);
        const double *Xs = X + ibx * batch, *Ys = Y + iby * batch, *Zs = Z + ibz * batch;
        int lx = nx - ibx * batch; if (lx > m) lx = m;
        int ly = ny - iby * batch; if (ly > m) ly = m;
        int lz = nz - ibz * batch; if (lz > m) lz = m;
        if (sparse && oracle_skip(&t, Xs, lx, Ys, ly, Zs, lz)) { R->kinds[b] = 0; continue; }
        for (int i = 0; i < lx; i++) for (int j = 0; j < ly; j++) for (int k = 0; k < lz; k++) {
            double p[3] = {Xs[i], Ys[j], Zs[k]};
            vol[((int64_t)i * ly + j) * lz + k] = (float)eval_node(&t, root, p);
        }
        R->n_eval += (int64_t)lx * ly * lz;
        /* skimage: "Surface level must be within volume data range" -> ValueError -> empty */
        int64_t namb = 0, k = 0;
        {
            float mn = INFINITY, mx = -INFINITY;
            for (int64_t i = 0; i < (int64_t)lx * ly * lz; i++) { if (vol[i] < mn) mn = vol[i]; if (vol[i] > mx) mx = vol[i]; }
            if (lx >= 2 && ly >= 2 && lz >= 2 && !(0.0f < mn) && !(0.0f > mx))
                k = sdf_oracle_marching_cubes(vol, lx, ly, lz, loc, (int64_t)12 * batch * batch * batch, &namb);
        }
        R->n_ambiguous += namb;
        if (k == 0) { R->kinds[b] = 1; continue; }
        R->kinds[b] = 2;
        if (R->ntri + k > R->cap) {
            while (R->ntri + k > R->cap) R->cap *= 2;
            R->tris = (double *)realloc(R->tris, sizeof(double) * 9 * R->cap);
        }
        /* points * scale + offset, f32 local -> f64 world (core.py:58-60) */
        double sc[3] = {Xs[1] - Xs[0], Ys[1] - Ys[0], Zs[1] - Zs[0]};
        double of[3] = {Xs[0], Ys[0], Zs[0]};
        double *o = R->tris + 9 * R->ntri;
        for (int64_t i = 0; i < 3 * k; i++)
            for (int a = 0; a < 3; a++) o[3 * i + a] = (double)loc[3 * i + a] * sc[a] + of[a];
        R->ntri += k;
    }
    free(vol); free(loc);
    return R;
}

int64_t sdf_oracle_result_ntri(const oracle_result *R) { return R->ntri; }
int64_t sdf_oracle_result_nbatches(const oracle_result *R) { return R->nbatches; }
int64_t sdf_oracle_result_neval(const oracle_result *R) { return R->n_eval; }
int64_t sdf_oracle_result_nambiguous(const oracle_result *R) { return R->n_ambiguous; }
const double *sdf_oracle_result_tris(const oracle_result *R) { return R->tris; }
const uint8_t *sdf_oracle_result_kinds(const oracle_result *R) { return R->kinds; }
void sdf_oracle_result_free(oracle_result *R) { if (R) { free(R->tris); free(R->kinds); free(R); } }

/* reference sdf/core.py:62-82.  np.linspace(a, b, 16)[i] = a + i*step with step=(b-a)/15 and the
 * last sample forced to b.  Returns 0 on success, 1 when no cell passed the threshold (the
 * reference raises on `where.max` of an empty array there). */
int sdf_oracle_estimate_bounds(const int32_t *nodes, const double *params, const int32_t *children,
                               int32_t root, double *out6) {
    tree_t t = mk_tree(nodes, params, children, root);
    const int s = 16;
    double lo[3] = {-1e9, -1e9, -1e9}, hi[3] = {1e9, 1e9, 1e9};
    double prev = -1; int have_prev = 0;
    for (int it = 0; it < 32; it++) {
        double ax[3][16], d[3];
        for (int a = 0; a < 3; a++) {
            double step = (hi[a] - lo[a]) / (s - 1);
            for (int i = 0; i < s; i++) ax[a][i] = lo[a] + i * step;
            ax[a][s - 1] = hi[a];
            d[a] = ax[a][1] - ax[a][0];
        }
        double threshold = len3(d[0], d[1], d[2]) / 2;
        if (have_prev && threshold == prev) break;
        prev = threshold; have_prev = 1;
        int mn[3] = {s, s, s}, mx[3] = {-1, -1, -1};
        for (int i = 0; i < s; i++) for (int j = 0; j < s; j++) for (int k = 0; k < s; k++) {
            double p[3] = {ax[0][i], ax[1][j], ax[2][k]};
            double v = eval_node(&t, root, p);
            if (fabs(v) <= threshold) {
                int id[3] = {i, j, k};
                for (int a = 0; a < 3; a++) { if (id[a] < mn[a]) mn[a] = id[a]; if (id[a] > mx[a]) mx[a] = id[a]; }
            }
        }
        if (mx[0] < 0) return 1;
        for (int a = 0; a < 3; a++) {
            double l0 = lo[a];
            hi[a] = l0 + mx[a] * d[a] + d[a] / 2;
            lo[a] = l0 + mn[a] * d[a] - d[a] / 2;
        }
    }
    for (int a = 0; a < 3; a++) { out6[a] = lo[a]; out6[3 + a] = hi[a]; }
    return 0;
}
