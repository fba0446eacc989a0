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
