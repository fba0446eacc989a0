// Host-side check of the interval primitives of sdf_amd/csrc/sdf_interval.h that go through libm
// or lose monotonicity (circular_array, repeat, easing): every value computed at a sampled point of
// a box must lie inside the interval computed for the box.  Runs on the CPU (the primitives are
// __host__ __device__); built and run by tests/test_interval_host.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "sdf_interval.h"

using namespace sdfk;

static double pymod(double a, double b) {   // npy_divmod's remainder (sdf_interp.h s_mod)
    double m = std::fmod(a, b);
    if (b == 0.0) return m;
    if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; }
    else m = std::copysign(0.0, b);
    return m;
}
static bool in(const Ival &i, double v) { return v >= i.lo && v <= i.hi; }
static long fails = 0, checks = 0;
static double tight_sum = 0; static long tight_n = 0;
#define CHECK(cond, ...) do { checks++; if (!(cond)) { if (fails++ < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    auto pick = [&](double lo, double hi) { return lo + (hi - lo) * U(rng); };
    // ---- circular_array ----
    const int counts[] = {1, 2, 3, 4, 5, 7, 16, 40};
    for (int it = 0; it < 200000; it++) {
        const int count = counts[rng() % 8];
        const double da = 2 * M_PI / count;
        Ival x, y;
        const int kind = it % 8;
        const double sz = std::pow(10.0, pick(-4, 0.5));
        double cx = pick(-3, 3), cy = pick(-3, 3);
        if (kind == 1) cy = 0;                       // on the x axis (both signs of x)
        if (kind == 2) { cx = 0; cy = 0; }           // around the origin
        if (kind == 3) { const double r = pick(0.1, 3), t = da * (double)(rng() % (count + 1)) - M_PI; cx = r * cos(t); cy = r * sin(t); }   // on sector boundaries
        x = Ival{cx - sz * U(rng), cx + sz * U(rng)};
        y = Ival{cy - sz * U(rng), cy + sz * U(rng)};
        if (kind == 4) x.hi = x.lo;                  // degenerate
        if (kind == 5) { y.lo = -0.0; }              // a box that ends on the axis
        if (kind == 6) { y.lo = 0.0; x.hi = fmin(x.hi, -0.01); x.lo = fmin(x.lo, x.hi); }
        if (kind == 7) { y.hi = -0.0; y.lo = fmin(y.lo, y.hi); }
        Ival d, a;
        ia::circ_prep(x, y, da, d, a);
        Ival X0, Y0, X1, Y1;
        ia::circ_set(d, a, da, X0, Y0);
        ia::circ_set(d, a, 0.0, X1, Y1);
        if (a.hi - a.lo < da) { tight_sum += (a.hi - a.lo) / da; tight_n++; }
        for (int s = 0; s < 24; s++) {
            double px = s & 1 ? x.lo : x.hi, py = s & 2 ? y.lo : y.hi;
            if (s >= 4) { px = pick(x.lo, x.hi); py = pick(y.lo, y.hi); }
            if (s >= 16 && s < 20) py = s & 1 ? 0.0 : -0.0;
            if (!(px >= x.lo && px <= x.hi && py >= y.lo && py <= y.hi)) continue;
            const double pd = std::hypot(px, py), pa = pymod(std::atan2(py, px), da);
            CHECK(in(d, pd), "d: box x[%g,%g] y[%g,%g] p(%g,%g) d=%.17g not in [%.17g,%.17g]", x.lo, x.hi, y.lo, y.hi, px, py, pd, d.lo, d.hi);
            CHECK(in(a, pa), "a: count %d box x[%.17g,%.17g] y[%.17g,%.17g] p(%.17g,%.17g) a=%.17g not in [%.17g,%.17g]", count, x.lo, x.hi, y.lo, y.hi, px, py, pa, a.lo, a.hi);
            const double q0x = cos(pa - da) * pd, q0y = sin(pa - da) * pd, q1x = cos(pa - 0.0) * pd, q1y = sin(pa - 0.0) * pd;
            CHECK(in(X0, q0x) && in(Y0, q0y), "set(da): p(%g,%g) -> (%.17g,%.17g) not in x[%.17g,%.17g] y[%.17g,%.17g]", px, py, q0x, q0y, X0.lo, X0.hi, Y0.lo, Y0.hi);
            CHECK(in(X1, q1x) && in(Y1, q1y), "set(0): p(%g,%g) -> (%.17g,%.17g) not in x[%.17g,%.17g] y[%.17g,%.17g]", px, py, q1x, q1y, X1.lo, X1.hi, Y1.lo, Y1.hi);
        }
    }
    // ---- atan2 over a box (wrap_around) ----
    for (int it = 0; it < 200000; it++) {
        const double sz = std::pow(10.0, pick(-4, 0.5));
        double cx = pick(-3, 3), cy = it % 3 ? pick(-3, 3) : 0.0;
        if (it % 7 == 0) cx = 0;
        Ival x{cx - sz * U(rng), cx + sz * U(rng)}, y{cy - sz * U(rng), cy + sz * U(rng)};
        if (it % 5 == 0) { y.lo = 0.0; y.hi = fabs(y.hi); }
        if (it % 11 == 0) { y.hi = -0.0; y.lo = -fabs(y.lo); }
        const Ival a = ia::atan2_range(x, y);
        for (int s2 = 0; s2 < 12; s2++) {
            double px = s2 & 1 ? x.lo : x.hi, py = s2 & 2 ? y.lo : y.hi;
            if (s2 >= 4) { px = pick(x.lo, x.hi); py = pick(y.lo, y.hi); }
            if (s2 >= 8 && 0.0 >= y.lo && 0.0 <= y.hi) py = s2 & 1 ? 0.0 : -0.0;
            CHECK(in(a, std::atan2(py, px)), "atan2: box x[%.17g,%.17g] y[%.17g,%.17g] p(%.17g,%.17g) a=%.17g not in [%.17g,%.17g]", x.lo, x.hi, y.lo, y.hi, px, py, std::atan2(py, px), a.lo, a.hi);
        }
    }
    // ---- sin / cos ranges over arbitrary angle intervals ----
    for (int it = 0; it < 200000; it++) {
        const double l = pick(-7, 7), w = std::pow(10.0, pick(-6, 1));
        const Ival ang{l, l + w * U(rng)};
        Ival sn, cs;
        ia::sincos_range(ang, sn, cs);
        for (int s = 0; s < 16; s++) {
            const double t = s == 0 ? ang.lo : (s == 1 ? ang.hi : pick(ang.lo, ang.hi));
            CHECK(in(sn, sin(t)) && in(cs, cos(t)), "sincos: [%g,%g] t=%g", ang.lo, ang.hi, t);
        }
    }
    // ---- easing curves on [0, 1] ----
    for (int id = 0; id < EASE_COUNT; id++) {
        bool known;
        ia::ease_value(id, 0.5, known);
        if (!known) continue;
        for (int it = 0; it < 20000; it++) {
            double a = U(rng), b = U(rng);
            if (it % 5 == 0) a = 0.5; if (it % 7 == 0) b = 1.0; if (it % 11 == 0) a = 0.0;
            const Ival t{fmin(a, b), fmax(a, b)};
            const Ival e = ia::ease01(id, t);
            for (int s = 0; s < 12; s++) {
                const double u = s == 0 ? t.lo : (s == 1 ? t.hi : pick(t.lo, t.hi));
                bool k;
                const double v = ia::ease_value(id, u, k);
                CHECK(in(e, v), "ease %d: t[%g,%g] u=%.17g v=%.17g not in [%.17g,%.17g]", id, t.lo, t.hi, u, v, e.lo, e.hi);
            }
        }
    }
    // ---- repeat: index = clip(rint(p / s)), p' = p - s * (index + n) ----
    for (int it = 0; it < 200000; it++) {
        const double s = (rng() & 1 ? 1 : -1) * std::pow(10.0, pick(-2, 1)), n = (double)((int)(rng() % 3) - 1);
        const double c = pick(-20, 20), w = std::pow(10.0, pick(-4, 1));
        const bool clip = rng() & 1; const double cnt = (double)(rng() % 4);
        const Ival p{c - w * U(rng), c + w * U(rng)};
        Ival r = ia::rint_(ia::divc(p, s));
        if (clip) r = ia::clipc(r, -cnt, cnt);
        const Ival q = ia::sub(p, ia::mulc(ia::addc(r, n), s));
        for (int k = 0; k < 12; k++) {
            const double u = k == 0 ? p.lo : (k == 1 ? p.hi : pick(p.lo, p.hi));
            double idx = std::rint(u / s);
            if (clip) idx = fmin(fmax(idx, -cnt), cnt);
            const double v = u - s * (idx + n);
            CHECK(in(r, idx) && in(q, v), "repeat: p[%g,%g] s=%g u=%g idx=%g v=%g not in [%g,%g]", p.lo, p.hi, s, u, idx, v, q.lo, q.hi);
        }
    }
    // ---- polynomial smooth union / difference / intersection (sdf_interp.h post_combine) ----
    for (int it = 0; it < 300000; it++) {
        const double K = std::pow(10.0, pick(-2, 0.3)), c1 = pick(-2, 2), c2 = it % 3 ? pick(-2, 2) : c1 + pick(-1.5, 1.5) * K;
        const double w = std::pow(10.0, pick(-4, 0));
        const Ival d1{c1 - w * U(rng), c1 + w * U(rng)}, d2{c2 - w * U(rng), c2 + w * U(rng)};
        const uint32_t posts[3] = {POST_SUNION, POST_SDIFF, POST_SINTER};
        for (int q = 0; q < 3; q++) {
            const Ival r = ia_post(posts[q], d1, d2, K);
            for (int k = 0; k < 10; k++) {
                const double a = k == 0 ? d1.lo : (k == 1 ? d1.hi : pick(d1.lo, d1.hi)), b = k == 0 ? d2.hi : (k == 1 ? d2.lo : pick(d2.lo, d2.hi));
                auto clip = [](double x) { return fmin(fmax(x, 0.0), 1.0); };
                double h, m, v;
                if (q == 0) { h = clip(0.5 + 0.5 * (b - a) / K); m = b + (a - b) * h; v = m - K * h * (1.0 - h); }
                else if (q == 1) { h = clip(0.5 - 0.5 * (b + a) / K); m = a + (-b - a) * h; v = m + K * h * (1.0 - h); }
                else { h = clip(0.5 - 0.5 * (b - a) / K); m = b + (a - b) * h; v = m + K * h * (1.0 - h); }
                CHECK(in(r, v), "smooth %d: K=%g d1[%g,%g] d2[%g,%g] a=%g b=%g v=%.17g not in [%.17g,%.17g]", q, K, d1.lo, d1.hi, d2.lo, d2.hi, a, b, v, r.lo, r.hi);
            }
            tight_sum += 0; 
        }
    }
    // ---- leaves composed of interval steps: point values (the formulas of sdf_interp.h) inside ia_leaf's interval ----
    {
        auto smin = [](double a, double b) { return (a < b || a != a) ? a : b; };
        auto smax = [](double a, double b) { return (a >= b || a != a) ? a : b; };
        auto sclip = [](double x, double lo, double hi) { double t = (x != x || x > lo) ? x : lo; return (t != t || t < hi) ? t : hi; };
        auto ssign = [](double x) { return x != x ? x : (x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0)); };
        auto l2 = [](double x, double y) { return std::sqrt(x * x + y * y); };
        auto l3 = [](double x, double y, double z) { return std::sqrt((x * x + y * y) + z * z); };
        auto d3 = [](double ax, double ay, double az, double bx, double by, double bz) { return std::fma(az, bz, std::fma(ay, by, ax * bx)); };
        auto d2 = [](double ax, double ay, double bx, double by) { return std::fma(ay, by, ax * bx); };
        auto leaf = [&](uint32_t op, const double *c, double x, double y, double z) -> double {
            switch (op) {
            // the common leaves (ia_leaf): corner-exact forms, checked without a tolerance
            case OP_L_SPHERE: return l3(x - c[1], y - c[2], z - c[3]) - c[0];
            case OP_L_PLANE: return fma(c[5] - z, c[2], fma(c[4] - y, c[1], (c[3] - x) * c[0]));
            case OP_L_BOX: { const double qx = fabs(x - c[0]) - c[3], qy = fabs(y - c[1]) - c[4], qz = fabs(z - c[2]) - c[5];
                return l3(smax(qx, 0), smax(qy, 0), smax(qz, 0)) + smin(smax(smax(qx, qy), qz), 0); }
            case OP_L_ROUNDED_BOX: { const double qx = fabs(x) - c[0] + c[3], qy = fabs(y) - c[1] + c[3], qz = fabs(z) - c[2] + c[3];
                return l3(smax(qx, 0), smax(qy, 0), smax(qz, 0)) + smin(smax(smax(qx, qy), qz), 0) - c[3]; }
            case OP_L_TORUS: return l2(l2(x, y) - c[0], z) - c[1];
            case OP_L_CYLINDER: return l2(x, y) - c[0];
            case OP_L_ROUNDED_CYLINDER: { const double d0 = l2(x, y) - c[0] + c[1], d1 = fabs(z) - c[2] + c[1];
                return smin(smax(d0, d1), 0) + l2(smax(d0, 0), smax(d1, 0)) - c[1]; }
            case OP_L_CAPSULE: { const double pax = x - c[0], pay = y - c[1], paz = z - c[2];
                const double h = sclip(fma(paz, c[5], fma(pay, c[4], pax * c[3])) / c[6], 0.0, 1.0);
                return l3(pax - c[3] * h, pay - c[4] * h, paz - c[5] * h) - c[7]; }
            case OP_L_OCTAHEDRON: return (((fabs(x) + fabs(y)) + fabs(z)) - c[0]) * c[1];
            case OP_L_CIRCLE: return l2(x - c[1], y - c[2]) - c[0];
            case OP_L_LINE: return fma(c[3] - y, c[1], (c[2] - x) * c[0]);
            case OP_L_RECTANGLE: { const double qx = fabs(x - c[0]) - c[2], qy = fabs(y - c[1]) - c[3];
                return l2(smax(qx, 0), smax(qy, 0)) + smin(smax(qx, qy), 0); }
            case OP_L_WIREFRAME_BOX: {
                const double t2 = c[3];
                const double px = fabs(x) - c[0] - t2, py = fabs(y) - c[1] - t2, pz = fabs(z) - c[2] - t2;
                const double qx = fabs(px + t2) - t2, qy = fabs(py + t2) - t2, qz = fabs(pz + t2) - t2;
                auto g = [&](double a, double b, double cc) { return l3(smax(a, 0), smax(b, 0), smax(cc, 0)) + smin(smax(a, smax(b, cc)), 0); };
                return smin(smin(g(px, qy, qz), g(qx, py, qz)), g(qx, qy, pz)); }
            case OP_L_CAPPED_CYLINDER: {
                const double bax = c[3], bay = c[4], baz = c[5], baba = c[6];
                const double pax = x - c[0], pay = y - c[1], paz = z - c[2];
                const double paba = d3(pax, pay, paz, bax, bay, baz);
                const double xx = l3(pax * baba - bax * paba, pay * baba - bay * paba, paz * baba - baz * paba) - c[8];
                const double yy = fabs(paba - c[9]) - c[9];
                const double x2 = xx * xx, y2 = yy * yy * baba;
                const double din = -smin(x2, y2);
                const double dout = (xx > 0 ? x2 : 0.0) + (yy > 0 ? y2 : 0.0);
                const double d = smax(xx, yy) < 0 ? din : dout;
                return ssign(d) * std::sqrt(fabs(d)) / baba; }
            case OP_L_ROUNDED_CONE: {
                const double r1 = c[0], r2 = c[1], h = c[2], b = c[3], a = c[4], ah = c[5];
                const double qx = l2(x, y), qy = z;
                const double k = d2(qx, qy, -b, a);
                const double c1 = l2(qx, qy) - r1, c2 = l2(qx - 0.0, qy - h) - r2, c3 = d2(qx, qy, a, b) - r1;
                return k < 0 ? c1 : (k > ah ? c2 : c3); }
            case OP_L_ELLIPSOID: {
                const double k0 = l3(x / c[0], y / c[1], z / c[2]), k1 = l3(x / c[3], y / c[4], z / c[5]);
                return k0 * (k0 - 1.0) / k1; }
            case OP_L_TETRAHEDRON: return (smax(fabs(x + y) - z, fabs(x - y) + z) - c[0]) / c[1];
            case OP_L_DODECAHEDRON: case OP_L_ICOSAHEDRON: {
                const double r = c[0], X = c[1], Y = c[2], Z = c[3];
                const double ax = fabs(x / r), ay = fabs(y / r), az = fabs(z / r);
                const double a = d3(ax, ay, az, X, Y, Z), b = d3(ax, ay, az, Z, X, Y), cc = d3(ax, ay, az, Y, Z, X);
                if (op == OP_L_DODECAHEDRON) return (smax(smax(a, b), cc) - X) * r;
                return smax(smax(smax(a, b), cc) - X, d3(ax, ay, az, c[4], c[4], c[4]) - X) * r; }
            case OP_L_ROUNDED_RECTANGLE: {
                const bool xp = x > 0, yp = y > 0;
                double r = 0;
                if (xp && yp) r = c[2]; if (xp && !yp) r = c[3]; if (!xp && !yp) r = c[4]; if (!xp && yp) r = c[5];
                const double qx = fabs(x) - c[0] + r, qy = fabs(y) - c[1] + r;
                return smin(smax(qx, qy), 0) + l2(smax(qx, 0), smax(qy, 0)) - r; }
            case OP_L_EQUILATERAL_TRIANGLE: {
                const double k = c[0];
                double px = fabs(x) - 1.0, py = y + c[1];
                const bool w = px + k * py > 0;
                const double nx = (px - k * py) / 2.0, ny = (-k * px - py) / 2.0;
                if (w) { px = nx; py = ny; }
                px = px - sclip(px, -2.0, 0.0);
                return -l2(px, py) * ssign(py); }
            case OP_L_HEXAGON: {
                const double r = c[0], k0 = c[1], k1 = c[2];
                double px = fabs(x), py = fabs(y);
                const double m = smin(k0 * px + k1 * py, 0);
                px = px - c[4] * m; py = py - c[5] * m;
                px = px - sclip(px, c[6], c[7]); py = py - (0.0 + r);
                return l2(px, py) * ssign(py); }
            case OP_L_ROUNDED_X: {
                const double px = fabs(x), py = fabs(y);
                const double qq = smin(px + py, c[0]) * 0.5;
                return l2(px - qq, py - qq) - c[1]; }
            case OP_L_VESICA: {
                const double r = c[0], d = c[1], b = c[2];
                const double px = fabs(x), py = fabs(y);
                return (py - b) * d > px * b ? l2(px - 0.0, py - b) : l2(px - (-d), py - 0.0) - r; }
            case OP_L_CAPPED_CONE: {
                const double ra = c[6], rb = c[7], baba = c[8], rba = c[9], k = c[10];
                const double pax = x - c[0], pay = y - c[1], paz = z - c[2];
                const double papa = (pax * pax + pay * pay) + paz * paz;
                const double paba = d3(pax, pay, paz, c[3], c[4], c[5]) / baba;
                const double xx = std::sqrt(papa - paba * paba * baba);
                const double cax = smax(0.0, xx - (paba < 0.5 ? ra : rb));
                const double cay = fabs(paba - 0.5) - 0.5;
                const double f = sclip((rba * (xx - ra) + paba * baba) / k, 0.0, 1.0);
                const double cbx = xx - ra - f * rba, cby = paba - f;
                const double sg = (cbx < 0 && cay < 0) ? -1.0 : 1.0;
                return sg * std::sqrt(smin(cax * cax + cay * cay * baba, cbx * cbx + cby * cby * baba)); }
            case OP_L_PYRAMID: {
                const double h = c[0], m2 = c[1], m2q = c[2];
                const double b0 = fabs(x) - 0.5, b1 = fabs(y) - 0.5;
                const bool sw = b1 > b0;
                const double a0 = sw ? b1 : b0, a1 = sw ? b0 : b1;
                const double px = a0, py = z, pz = a1;
                const double qx = pz, qy = h * py - 0.5 * px, qz = h * px + 0.5 * py;
                const double sv = smax(-qx, 0.0);
                const double tt = sclip((qy - 0.5 * pz) / m2q, 0.0, 1.0);
                const double a = m2 * ((qx + sv) * (qx + sv)) + qy * qy;
                const double b = m2 * ((qx + 0.5 * tt) * (qx + 0.5 * tt)) + (qy - m2 * tt) * (qy - m2 * tt);
                const double dd2 = smin(qy, -qx * m2 - qy * 0.5) > 0 ? 0.0 : smin(a, b);
                return std::sqrt((dd2 + qz * qz) / m2) * ssign(smax(qz, -py)); }
            case OP_L_POLYGON: {
                const int np_ = (int)c[0];
                const double *pv = c + 1;
                const double dx = x - pv[0], dy = y - pv[1];
                double d = dx * dx + dy * dy, sg = 1.0;
                for (int i = 0; i < np_; i++) {
                    const int j = (i + np_ - 1) % np_;
                    const double vix = pv[2 * i], viy = pv[2 * i + 1], vjx = pv[2 * j], vjy = pv[2 * j + 1];
                    const double ex = vjx - vix, ey = vjy - viy, wx = x - vix, wy = y - viy;
                    const double ee = std::fma(ey, ey, ex * ex);
                    const double cl = sclip(d2(wx, wy, ex, ey) / ee, 0.0, 1.0);
                    const double bx = wx - ex * cl, by = wy - ey * cl;
                    d = smin(d, bx * bx + by * by);
                    const bool c1 = y >= viy, c2 = y < vjy, c3 = ex * wy > ey * wx;
                    if ((c1 && c2 && c3) || (!c1 && !c2 && !c3)) sg = -sg;
                }
                return sg * std::sqrt(d); }
            case OP_L_TEXTURE2D: {
                const int tw = (int)c[8], th = (int)c[9];
                const double *tex = c + 14;
                const double u = (x - c[0]) / (c[2] - c[0]);
                double vv = (y - c[1]) / (c[3] - c[1]);
                vv = 1.0 - vv;
                const double fi = u * c[4] + c[6], fj = vv * c[5] + c[7];
                const double gi = fi == fi ? sclip(std::floor(fi), -2.0, (double)tw) : 0.0, gj = fj == fj ? sclip(std::floor(fj), -2.0, (double)th) : 0.0;
                const int a0 = (int)gi, b0 = (int)gj;
                const int ix0 = std::min(std::max(a0, 0), tw - 1), ix1 = std::min(std::max(a0 + 1, 0), tw - 1);
                const int iy0 = std::min(std::max(b0, 0), th - 1), iy1 = std::min(std::max(b0 + 1, 0), th - 1);
                const double pa = tex[iy0 * tw + ix0], pb = tex[iy1 * tw + ix0], pc = tex[iy0 * tw + ix1], pd = tex[iy1 * tw + ix1];
                const double wa = ((double)ix1 - fi) * ((double)iy1 - fj), wb = ((double)ix1 - fi) * (fj - (double)iy0);
                const double wc = (fi - (double)ix0) * ((double)iy1 - fj), wd = (fi - (double)ix0) * (fj - (double)iy0);
                const double d = wa * pa + wb * pb + wc * pc + wd * pd;
                const double qx = fabs(x - c[10]) - c[12], qy = fabs(y - c[11]) - c[13];
                const double q = l2(smax(qx, 0), smax(qy, 0)) + smin(smax(qx, qy), 0);
                const bool outside = (fi < 0) || (fi >= (double)(tw - 1)) || (fj < 0) || (fj >= (double)(th - 1));
                return outside ? q : d; }
            default: return 0.0;
            }
        };
        const uint32_t ops[] = {OP_L_SPHERE, OP_L_PLANE, OP_L_BOX, OP_L_ROUNDED_BOX, OP_L_TORUS, OP_L_CYLINDER, OP_L_ROUNDED_CYLINDER, OP_L_CAPSULE,
                                OP_L_OCTAHEDRON, OP_L_CIRCLE, OP_L_LINE, OP_L_RECTANGLE,
                                OP_L_TEXTURE2D, OP_L_CAPPED_CONE, OP_L_PYRAMID, OP_L_POLYGON,OP_L_WIREFRAME_BOX, OP_L_CAPPED_CYLINDER, OP_L_ROUNDED_CONE, OP_L_ELLIPSOID, OP_L_TETRAHEDRON, OP_L_DODECAHEDRON,
                                OP_L_ICOSAHEDRON, OP_L_ROUNDED_RECTANGLE, OP_L_EQUILATERAL_TRIANGLE, OP_L_HEXAGON, OP_L_ROUNDED_X, OP_L_VESICA};
        long decided = 0, boxes = 0;
        for (uint32_t op : ops) {
            for (int it = 0; it < 60000; it++) {
                double c[14 + 24 * 20];
                for (int k = 0; k < 16; k++) c[k] = pick(0.1, 1.2);
                if (op == OP_L_PLANE || op == OP_L_LINE) for (int k = 0; k < 6; k++) c[k] = pick(-1.5, 1.5);            // normal, point
                if (op == OP_L_SPHERE || op == OP_L_CIRCLE || op == OP_L_BOX || op == OP_L_RECTANGLE) for (int k = 1; k < 3; k++) c[k] = pick(-1, 1);   // (centres)
                if (op == OP_L_CAPSULE) {                // a, ba, |ba|^2, radius (d3.py:167-176)
                    for (int k = 0; k < 6; k++) c[k] = pick(-1, 1);
                    c[6] = (c[3] * c[3] + c[4] * c[4]) + c[5] * c[5]; c[7] = pick(0.05, 0.5);
                    if (!(c[6] > 1e-3)) continue;
                }
                if (op == OP_L_CAPPED_CYLINDER) {      // a, ba, baba, -, radius, baba / 2 (d3.py:184-204)
                    for (int k = 0; k < 6; k++) c[k] = pick(-1, 1);
                    c[6] = (c[3] * c[3] + c[4] * c[4]) + c[5] * c[5]; c[8] = pick(0.05, 0.6) * c[6]; c[9] = c[6] * 0.5;
                    if (!(c[6] > 1e-3)) continue;
                }
                if (op == OP_L_TEXTURE2D) {              // a 24 x 20 picture over [-1.5, 1.5] x [-1, 1] with a margin of 2 pixels
                    const int tw = 24, th = 20;
                    c[0] = -1.5; c[1] = -1.0; c[2] = 1.5; c[3] = 1.0; c[4] = tw - 5; c[5] = th - 5; c[6] = 2; c[7] = 2; c[8] = tw; c[9] = th;
                    c[10] = 0; c[11] = 0; c[12] = 1.5; c[13] = 1.0;
                    for (int k = 0; k < tw * th; k++) c[14 + k] = pick(-0.5, 0.5) + 0.02 * (k % tw);
                }
                if (op == OP_L_CAPPED_CONE) {            // a, ba, ra, rb, baba, rba, k (d3.py:217-237)
                    for (int k = 0; k < 6; k++) c[k] = pick(-1, 1);
                    c[6] = pick(0.1, 0.8); c[7] = pick(0.0, 0.6);
                    c[8] = (c[3] * c[3] + c[4] * c[4]) + c[5] * c[5]; c[9] = c[7] - c[6]; c[10] = c[9] * c[9] + c[8];
                    if (!(c[8] > 1e-3)) continue;
                }
                if (op == OP_L_PYRAMID) { c[0] = pick(0.3, 2.0); c[1] = c[0] * c[0] + 0.25; c[2] = c[1] + 0.25; }
                if (op == OP_L_POLYGON) {                // a star-ish polygon of 3..7 vertices
                    const int n = 3 + (int)(rng() % 5);
                    c[0] = n;
                    for (int k = 0; k < n; k++) { const double a = 2 * M_PI * k / n + pick(-0.2, 0.2), rr = pick(0.4, 1.5); c[1 + 2 * k] = rr * cos(a); c[2 + 2 * k] = rr * sin(a); }
                }
                if (op == OP_L_ROUNDED_CONE) { c[3] = pick(-0.9, 0.9); c[4] = std::sqrt(1 - c[3] * c[3]); c[5] = c[4] * c[2]; }
                if (op == OP_L_HEXAGON) { c[1] = -0.866025404; c[2] = 0.5; c[4] = 2 * c[1]; c[5] = 2 * c[2]; c[6] = -0.577350269 * c[0]; c[7] = 0.577350269 * c[0]; }
                if (op == OP_L_EQUILATERAL_TRIANGLE) { c[0] = 1.7320508; c[1] = pick(0.1, 1.0); }
                if (op == OP_L_VESICA) { c[1] = pick(0.05, 0.9) * c[0]; c[2] = std::sqrt(c[0] * c[0] - c[1] * c[1]); }
                const double sz = std::pow(10.0, pick(-3.5, 0.3));
                const double cx = it % 4 ? pick(-2, 2) : 0.0, cy = it % 5 ? pick(-2, 2) : 0.0, cz = it % 6 ? pick(-2, 2) : 0.0;
                Ival X{cx - sz * U(rng), cx + sz * U(rng)}, Y{cy - sz * U(rng), cy + sz * U(rng)}, Z{cz - sz * U(rng), cz + sz * U(rng)};
                if (it % 9 == 0) X.hi = X.lo;
                const Ival v = ia_is_rare_leaf(op) ? ia_leaf_rare(op, c, X, Y, Z) : ia_leaf(op, c, X, Y, Z);
                boxes++; if (v.lo > 0 || v.hi < 0) decided++;
                for (int k = 0; k < 14; k++) {
                    const double px = k & 1 ? (k < 8 ? X.lo : pick(X.lo, X.hi)) : (k < 8 ? X.hi : pick(X.lo, X.hi));
                    const double py = k & 2 ? (k < 8 ? Y.lo : pick(Y.lo, Y.hi)) : (k < 8 ? Y.hi : pick(Y.lo, Y.hi));
                    const double pz = k & 4 ? (k < 8 ? Z.lo : pick(Z.lo, Z.hi)) : (k < 8 ? Z.hi : pick(Z.lo, Z.hi));
                    const double pv = leaf(op, c, px, py, pz);
                    CHECK(pv != pv || in(v, pv), "leaf %u: box x[%.17g,%.17g] y[%.17g,%.17g] z[%.17g,%.17g] p(%.17g,%.17g,%.17g) v=%.17g not in [%.17g,%.17g]",
                          op, X.lo, X.hi, Y.lo, Y.hi, Z.lo, Z.hi, px, py, pz, pv, v.lo, v.hi);
                }
            }
        }
        printf("leaf boxes %ld, sign decided for %ld\n", boxes, decided);
        CHECK(decided * 2 > boxes, "the leaf intervals decide fewer than half of the boxes: are the forms reached?");
    }
    // ---- the voxel-grid leaf (L_GRID3D: mesh.py:96-105): trilinear look-up in scipy's operation order, box estimator ----
    {
        auto grid_value = [&](const double *c, double x, double y, double z) -> double {       // sdf_interp.h grid3d_lookup + L_L_GRID3D
            const int n[3] = {(int)c[0], (int)c[1], (int)c[2]};
            const double bg = c[3];
            const double *g[3] = {c + 10, c + 10 + n[0], c + 10 + n[0] + n[1]};
            const double *vox = c + 10 + n[0] + n[1] + n[2];
            const double qx = fabs(x - c[4]) - c[7], qy = fabs(y - c[5]) - c[8], qz = fabs(z - c[6]) - c[9];
            auto mx = [](double a, double b) { return (a >= b || a != a) ? a : b; };
            auto mn = [](double a, double b) { return (a < b || a != a) ? a : b; };
            const double e = std::sqrt((mx(qx, 0) * mx(qx, 0) + mx(qy, 0) * mx(qy, 0)) + mx(qz, 0) * mx(qz, 0)) + mn(mx(mx(qx, qy), qz), 0);
            const double p[3] = {x, y, z};
            int idx[3]; double w[3]; bool oob = false;
            for (int a = 0; a < 3; a++) {
                int lo = 0, hi = n[a];
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (g[a][mid] < p[a]) lo = mid + 1; else hi = mid; }
                int i = lo - 1; if (i < 0) i = 0; if (i > n[a] - 2) i = n[a] - 2;
                idx[a] = i; w[a] = (p[a] - g[a][i]) / (g[a][i + 1] - g[a][i]);
                oob = oob || p[a] < g[a][0] || p[a] > g[a][n[a] - 1];
            }
            double acc = 0.0;
            for (int q = 0; q < 8; q++) {
                const int o0 = q >> 2, o1 = (q >> 1) & 1, o2 = q & 1;
                double wt = o0 ? w[0] : 1 - w[0];
                wt *= o1 ? w[1] : 1 - w[1];
                wt *= o2 ? w[2] : 1 - w[2];
                acc += vox[((size_t)(idx[0] + o0) * n[1] + (idx[1] + o1)) * n[2] + (idx[2] + o2)] * wt;
            }
            const double d = oob ? bg : acc;
            return e > bg ? e : d;
        };
        long decided = 0, boxes = 0;
        for (int it = 0; it < 40000; it++) {
            const int n0 = 4 + (int)(rng() % 9), n1 = 4 + (int)(rng() % 9), n2 = 4 + (int)(rng() % 9);
            static double c[10 + 36 + 12 * 12 * 12];
            c[0] = n0; c[1] = n1; c[2] = n2;
            const double bg = pick(0.05, 0.4);
            c[3] = bg;
            c[4] = pick(-0.1, 0.1); c[5] = pick(-0.1, 0.1); c[6] = pick(-0.1, 0.1);             // estimator box: centre, half size
            c[7] = pick(0.5, 0.9); c[8] = pick(0.5, 0.9); c[9] = pick(0.5, 0.9);
            double *g = c + 10;
            const int nn[3] = {n0, n1, n2};
            for (int a = 0; a < 3; a++) { double t = -1.0; for (int i = 0; i < nn[a]; i++) { g[i] = t; t += pick(0.05, 2.2 / nn[a] * 1.6); } g += nn[a]; }   // non-uniform, ascending
            const double cx0 = pick(-0.3, 0.3), cy0 = pick(-0.3, 0.3), cz0 = pick(-0.3, 0.3), rr = pick(0.3, 0.7);
            const double *gx = c + 10, *gy = gx + n0, *gz = gy + n1;
            for (int i = 0; i < n0; i++) for (int j = 0; j < n1; j++) for (int k = 0; k < n2; k++) {
                double d = std::sqrt((gx[i] - cx0) * (gx[i] - cx0) + (gy[j] - cy0) * (gy[j] - cy0) + (gz[k] - cz0) * (gz[k] - cz0)) - rr;
                if (it % 7 == 0) d = pick(-bg, bg);                                                  // noise now and then
                g[((size_t)i * n1 + j) * n2 + k] = (double)(float)std::fmin(std::fmax(d, -bg), bg);   // a narrow band, float32 voxels
            }
            const double sz = std::pow(10.0, pick(-3.0, 0.2));
            const double cx = it % 4 ? pick(-1.6, 1.6) : gx[rng() % n0], cy = it % 5 ? pick(-1.6, 1.6) : gy[rng() % n1], cz = it % 6 ? pick(-1.6, 1.6) : gz[rng() % n2];
            Ival X{cx - sz * U(rng), cx + sz * U(rng)}, Y{cy - sz * U(rng), cy + sz * U(rng)}, Z{cz - sz * U(rng), cz + sz * U(rng)};
            if (it % 9 == 0) X.hi = X.lo;
            const Ival v = ia_leaf_rare(OP_L_GRID3D, c, X, Y, Z);
            boxes++; if (v.lo > 0 || v.hi < 0) decided++;
            for (int k = 0; k < 14; k++) {
                const double px = k & 1 ? (k < 8 ? X.lo : pick(X.lo, X.hi)) : (k < 8 ? X.hi : pick(X.lo, X.hi));
                const double py = k & 2 ? (k < 8 ? Y.lo : pick(Y.lo, Y.hi)) : (k < 8 ? Y.hi : pick(Y.lo, Y.hi));
                const double pz = k & 4 ? (k < 8 ? Z.lo : pick(Z.lo, Z.hi)) : (k < 8 ? Z.hi : pick(Z.lo, Z.hi));
                const double pv = grid_value(c, px, py, pz);
                CHECK(pv != pv || in(v, pv), "grid3d: box x[%.17g,%.17g] y[%.17g,%.17g] z[%.17g,%.17g] p(%.17g,%.17g,%.17g) v=%.17g not in [%.17g,%.17g]",
                      X.lo, X.hi, Y.lo, Y.hi, Z.lo, Z.hi, px, py, pz, pv, v.lo, v.hi);
            }
        }
        printf("grid boxes %ld, sign decided for %ld\n", boxes, decided);
        CHECK(decided * 4 > boxes, "the grid leaf's interval decides fewer than a quarter of the boxes");
    }
    // ---- the arithmetic primitives, bounds and constants from a set with the special values in it: the result holds
    // no NaN, is not empty, and encloses the operation at points of the operands (a point at an infinite bound
    // stands for "any value": there the check is that the bound on that side is infinite as well) ----
    {
        const double inf = __builtin_inf(), nan = __builtin_nan("");
        const double sp[] = {-inf, -1e308, -3.5, -1.0, -1e-300, -0.0, 0.0, 1e-300, 0.5, 1.0, 7.25, 1e308, inf};
        const int nsp = (int)(sizeof sp / sizeof sp[0]);
        auto ival = [&]() { int i = (int)(rng() % nsp), j = (int)(rng() % nsp); if (i > j) { const int t = i; i = j; j = t; } return Ival{sp[i], sp[j]}; };
        auto konst = [&]() { const unsigned r = (unsigned)(rng() % 16); return r == 0 ? nan : sp[rng() % nsp]; };
        auto point = [&](const Ival &a) {   // a finite point of a (the finite end, or something large, where a bound is infinite)
            const double lo = a.lo == -inf ? (a.hi == inf ? -1e3 : fmin(a.hi, 0.0) - 1e3) : a.lo, hi = a.hi == inf ? fmax(lo, 0.0) + 1e3 : a.hi;
            const unsigned r = (unsigned)(rng() % 4);
            const double t = U(rng);
            return r == 0 ? lo : (r == 1 ? hi : fmin(fmax(lo * (1.0 - t) + hi * t, lo), hi));
        };
        auto sane = [&](const Ival &r) { return r.lo == r.lo && r.hi == r.hi && r.lo <= r.hi; };
        auto holds = [&](const Ival &r, double v) { return v != v || (v >= r.lo && v <= r.hi); };   // (a NaN result of the point operation: inf - inf at a point, nothing to enclose)
        for (int it = 0; it < 400000; it++) {
            const Ival a = ival(), b = ival();
            if ((a.lo == inf) || (a.hi == -inf) || (b.lo == inf) || (b.hi == -inf)) continue;   // (an interval AT infinity holds no finite point)
            const double c = konst(), u = point(a), v = point(b);
            Ival r;
            r = ia::add(a, b); CHECK(sane(r) && holds(r, u + v), "add [%g,%g]+[%g,%g] = [%g,%g]", a.lo, a.hi, b.lo, b.hi, r.lo, r.hi);
            r = ia::sub(a, b); CHECK(sane(r) && holds(r, u - v), "sub [%g,%g]-[%g,%g] = [%g,%g]", a.lo, a.hi, b.lo, b.hi, r.lo, r.hi);
            r = ia::addc(a, c); CHECK(sane(r) && holds(r, u + c), "addc [%g,%g]+%g = [%g,%g]", a.lo, a.hi, c, r.lo, r.hi);
            r = ia::subc(a, c); CHECK(sane(r) && holds(r, u - c), "subc [%g,%g]-%g = [%g,%g]", a.lo, a.hi, c, r.lo, r.hi);
            r = ia::csub(c, a); CHECK(sane(r) && holds(r, c - u), "csub %g-[%g,%g] = [%g,%g]", c, a.lo, a.hi, r.lo, r.hi);
            r = ia::mulc(a, c); CHECK(sane(r) && holds(r, u * c), "mulc [%g,%g]*%g = [%g,%g] (%g)", a.lo, a.hi, c, r.lo, r.hi, u * c);
            r = ia::divc(a, c); CHECK(sane(r) && (c == 0.0 || holds(r, u / c)), "divc [%g,%g]/%g = [%g,%g] (%g)", a.lo, a.hi, c, r.lo, r.hi, u / c);
            r = ia::fmac(a, c, b); CHECK(sane(r) && holds(r, fma(u, c, v)), "fmac [%g,%g]*%g+[%g,%g] = [%g,%g] (%g)", a.lo, a.hi, c, b.lo, b.hi, r.lo, r.hi, fma(u, c, v));
            r = ia::sqr(a); CHECK(sane(r) && r.lo >= 0.0 && holds(r, u * u), "sqr [%g,%g] = [%g,%g] (%g)", a.lo, a.hi, r.lo, r.hi, u * u);
            r = ia::abs_(a); CHECK(sane(r) && r.lo >= 0.0 && holds(r, fabs(u)), "abs [%g,%g] = [%g,%g]", a.lo, a.hi, r.lo, r.hi);
            r = ia::neg(a); CHECK(sane(r) && holds(r, -u), "neg");
            r = ia::min_(a, b); CHECK(sane(r) && holds(r, fmin(u, v)), "min");
            r = ia::max_(a, b); CHECK(sane(r) && holds(r, fmax(u, v)), "max");
            r = ia::dot3c(a, b, a, c, 0.5, -2.0); CHECK(sane(r) && holds(r, fma(u, -2.0, fma(v, 0.5, u * c))), "dot3c [%g,%g] [%g,%g] c=%g = [%g,%g]", a.lo, a.hi, b.lo, b.hi, c, r.lo, r.hi);
            r = ia::len3(a, b, Ival{u, u}); CHECK(sane(r) && holds(r, sqrt((u * u + v * v) + u * u)), "len3 [%g,%g] [%g,%g] = [%g,%g]", a.lo, a.hi, b.lo, b.hi, r.lo, r.hi);
        }
    }
    // ---- interval product ----
    for (int it = 0; it < 100000; it++) {
        const Ival a{pick(-3, 3), 0}, b{pick(-3, 3), 0};
        Ival A{a.lo, a.lo + pick(0, 2)}, B{b.lo, b.lo + pick(0, 2)};
        const Ival P = ia::mul(A, B);
        for (int k = 0; k < 8; k++) { const double u = pick(A.lo, A.hi), v = pick(B.lo, B.hi); CHECK(in(P, u * v), "mul"); }
    }
    printf("checks %ld fails %ld; narrowed angle ranges: %ld (mean width %.3f of a sector)\n", checks, fails, tight_n, tight_n ? tight_sum / tight_n : 0.0);
    return fails ? 1 : 0;
}
