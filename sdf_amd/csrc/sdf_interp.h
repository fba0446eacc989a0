// sdf_interp.h -- the CDNA4 op-tape interpreter: one lane evaluates NS samples.
//
// Replaces the reference's recursive NumPy closure calls (reference sdf/d3.py:24-25 and every
// `def f(p)` in sdf/d3.py, sdf/d2.py, sdf/dn.py, sdf/ease.py).  The tape (sdf_amd/tape.py) is
// straight-line code that is identical for every lane, so all control flow here is
// wave-uniform: instruction words and constants are fetched through the scalar cache
// (s_load), the opcode switch is a scalar branch tree, and the slot numbers that index the
// PS / DS register files are wave-uniform too.  Data-dependent choices inside an op are
// selects (v_cndmask), never branches.  Machine state per sample:
//     (x, y, z)  current point         acc      current distance
//     PS[s]      saved points          DS[s]    saved distances       (static slots)
// Every lane carries NS samples (Vec<T, NS>, sdf_vec.h): the decode cost of an instruction is
// paid once per NS * 64 samples and the NS dependency chains interleave in the VALU.
//
// Arithmetic follows the reference's NumPy expression order operation by operation (no
// contraction: the translation units are built with -ffp-contract=off; fused multiply-adds appear
// only where NumPy itself goes through BLAS, see dot3), so that in T = double the values agree
// with the reference to the last bit for every correctly-rounded operation.  The formulas carry
// the reference file:line they restate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opcodes.h"
#include "sdf_vec.h"

namespace sdfk {

// ---- scalar primitives -------------------------------------------------------------------
// sqrt() / sqrtf() are the correctly rounded ocml forms (the __*sqrt_rn intrinsics may map to the
// native approximation)
SDF_DEV double m_sqrt(double x) { return sqrt(x); }
SDF_DEV float m_sqrt(float x) { return sqrtf(x); }
SDF_DEV double m_fma(double a, double b, double c) { return fma(a, b, c); }
SDF_DEV float m_fma(float a, float b, float c) { return fmaf(a, b, c); }
SDF_DEV double m_fabs(double x) { return fabs(x); }
SDF_DEV float m_fabs(float x) { return fabsf(x); }
SDF_DEV double m_floor(double x) { return floor(x); }
SDF_DEV float m_floor(float x) { return floorf(x); }
SDF_DEV double m_rint(double x) { return rint(x); }
SDF_DEV float m_rint(float x) { return rintf(x); }
// The libm bodies (ocml: argument reduction, polynomial tables) are large -- inlined at every use
// they made the trig-capable kernels 370 KB of code.  They are outlined: one copy per translation
// unit, reached by s_swappc; the call overhead is small next to the functions themselves.
#define SDF_OUTLINE static __device__ __attribute__((noinline))
SDF_OUTLINE double o_sin(double x) { return sin(x); }
SDF_OUTLINE float o_sin(float x) { return sinf(x); }
SDF_OUTLINE double o_cos(double x) { return cos(x); }
SDF_OUTLINE float o_cos(float x) { return cosf(x); }
SDF_OUTLINE double o_atan2(double y, double x) { return atan2(y, x); }
SDF_OUTLINE float o_atan2(float y, float x) { return atan2f(y, x); }
SDF_OUTLINE double o_hypot(double x, double y) { return hypot(x, y); }
SDF_OUTLINE float o_hypot(float x, float y) { return hypotf(x, y); }
SDF_OUTLINE double o_fmod(double x, double y) { return fmod(x, y); }
SDF_OUTLINE float o_fmod(float x, float y) { return fmodf(x, y); }
SDF_OUTLINE double o_pow2(double x) { return pow(2.0, x); }
SDF_OUTLINE float o_pow2(float x) { return powf(2.0f, x); }
SDF_OUTLINE void o_sincos(float x, float *s, float *c) { sincosf(x, s, c); }
struct SinCos64 { double s, c; };    // (returned in registers: out-parameters of an outlined function live in scratch memory)
SDF_OUTLINE SinCos64 o_sincos64(double x) { SinCos64 r; sincos(x, &r.s, &r.c); return r; }
// circular_array's polar form for one float64 point (reference sdf/d3.py:381-383): .s = hypot(x, y), .c = atan2(y, x) mod da (NumPy's floored modulo)
SDF_OUTLINE SinCos64 o_circ_polar64(double x, double y, double da) {
    SinCos64 r; r.s = hypot(x, y);
    const double a = atan2(y, x);
    double m = fmod(a, da);
    if (da != 0.0) { if (m != 0.0) { if ((da < 0.0) != (m < 0.0)) m += da; } else m = copysign(0.0, da); }   // (s_mod)
    r.c = m; return r;
}

// sin and cos of a float64 angle, INLINE and branch-free: what `circular_array` runs twice per child evaluation
// (reference sdf/d3.py:379-392 -- weave at 2^33: 36 of them per sample before pruning), and twist / bend.  The ocml
// sincos is ~200 instructions behind a call; around the call the compiler has to park the interpreter's machine state
// (72 VGPRs for two samples with four saved points) in the half of the register file the calling convention preserves,
// and the rest went to scratch -- the 4-slot kernels spilled 88 - 226 registers because of it.
// Reduction: k = rint(x * 2/pi), r + lo = x - k * pi/2 with pi/2 in three 53-bit pieces: the first product is exact
// in the fma and its difference exactly representable, the second step keeps its rounding and product errors
// (TwoSum + fma), the third piece goes into the tail.  Kernels: the fdlibm / FreeBSD msun polynomials (__kernel_sin,
// __kernel_cos with tail) on |r| <= pi/4.  Measured against an 80-bit reference over 2e7 arguments in [-4e5, 4e5] incl.
// neighbourhoods of multiples of pi/2: max error 0.79 ulp (glibc: 0.56); 2.3 % of the results differ from glibc's in
// the last bit -- the same class of difference as ocml vs glibc, which the parity tolerances for libm models cover
// (DESIGN.md section 5).  |x| >= 1e6, NaN and infinities take the ocml path (a wave-uniform branch never taken by
// angles that come out of atan2 / np_mod).
SDF_DEV void sincos64_dd(double x, double x_lo, double &sn, double &cs) {      // sin / cos of x + x_lo
    const double P1 = 1.5707963267948966, P2 = 6.123233995736766e-17, P3 = -1.4973849048591698e-33;
    const double q = rint(x * 0.6366197723675814);
    const double r0 = fma(-q, P1, x);
    const double p2 = q * P2, b = -p2;
    const double r = r0 + b;
    const double bb = r - r0;
    double lo = (r0 - (r - bb)) + (b - bb);      // TwoSum: the rounding error of r0 - p2
    lo = lo - fma(q, P2, -p2);                   // the product's own error, exact
    lo = fma(-q, P3, lo) + x_lo;
    const double z = r * r, v = z * r;
    const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                 -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double s = r - ((z * (0.5 * lo - v * rs) - lo) - v * -1.66666666666666324348e-01);
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                          2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + (z * rc - r * lo));
    const int n = (int)q;
    double ss = (n & 1) ? c : s, cc = (n & 1) ? s : c;
    ss = (n & 2) ? -ss : ss;
    cc = ((n + 1) & 2) ? -cc : cc;
    const bool big = !(fabs(x) < 1e6);
    if (__builtin_expect(__ballot(big) != 0ull, 0)) {        // (wave-uniform)
        const SinCos64 o = o_sincos64(x);
        ss = big ? o.s : ss; cc = big ? o.c : cc;
    }
    sn = ss; cs = cc;
}
SDF_DEV void sincos64(double x, double &sn, double &cs) { sincos64_dd(x, 0.0, sn, cs); }

// atan2 in float64, INLINE and branch-free, ONE division: with a = min(|x|, |y|), b = max(|x|, |y|), t = a / b in [0, 1]
// is reduced like fdlibm's atan -- t < 7/16: atan(t); t < 11/16: atan(1/2) + atan((2t - 1) / (2 + t)); else
// atan(1) + atan((t - 1) / (t + 1)) -- but the quotients are formed from a and b directly ((2a - b) / (2b + a),
// (a - b) / (a + b): both numerators are exact by Sterbenz' lemma in their ranges), then fdlibm's degree-11
// polynomial and hi / lo constants; pi/2 - u for |y| > |x|, pi - u for x < 0 (sign bit: atan2(+0, -0) = pi), the sign
// of y.  Max error 1.48 ulp against an 80-bit reference over 3e7 points (ocml: 104 VALU instructions behind a call).
SDF_DEV double atan2_64(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const bool swap = ay > ax;
    const double a = swap ? ax : ay, b = swap ? ay : ax;
    const bool id1 = !(16.0 * a < 7.0 * b), id2 = !(16.0 * a < 11.0 * b);
    double num = id1 ? 2.0 * a - b : a, den = id1 ? 2.0 * b + a : b;
    double hi = id1 ? 4.63647609000806093515e-01 : 0.0, lo = id1 ? 2.26987774529616870924e-17 : 0.0;
    num = id2 ? a - b : num; den = id2 ? a + b : den;
    hi = id2 ? 7.85398163397448278999e-01 : hi; lo = id2 ? 3.06161699786838301793e-17 : lo;
    double q = num / den;
    q = b == 0.0 ? 0.0 : q;                                   // atan2(+-0, +-0)
    const bool both_inf = a == __builtin_inf();
    q = both_inf ? 0.0 : q; hi = both_inf ? 7.85398163397448278999e-01 : hi; lo = both_inf ? 3.06161699786838301793e-17 : lo;
    const double z = q * q, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    double u = hi - ((q * (s1 + s2) - lo) - q);               // atan(a / b) in [0, pi/4]
    u = swap ? 1.57079632679489655800e+00 - (u - 6.12323399573676603587e-17) : u;
    u = __builtin_signbit(x) ? 3.1415926535897931160e+00 - (u - 1.2246467991473531772e-16) : u;
    return copysign(u, y);
}
SDF_DEV double m_sin(double x) { return o_sin(x); }
SDF_DEV float m_sin(float x) { return o_sin(x); }
SDF_DEV double m_cos(double x) { return o_cos(x); }
SDF_DEV float m_cos(float x) { return o_cos(x); }
SDF_DEV double m_atan2(double y, double x) { return o_atan2(y, x); }
SDF_DEV float m_atan2(float y, float x) { return o_atan2(y, x); }
SDF_DEV double m_hypot(double x, double y) { return o_hypot(x, y); }
SDF_DEV float m_hypot(float x, float y) { return o_hypot(x, y); }
SDF_DEV double m_fmod(double x, double y) { return o_fmod(x, y); }
SDF_DEV float m_fmod(float x, float y) { return o_fmod(x, y); }
SDF_DEV double m_pow2(double x) { return o_pow2(x); }
SDF_DEV float m_pow2(float x) { return o_pow2(x); }
SDF_DEV double m_copysign(double x, double y) { return copysign(x, y); }
SDF_DEV float m_copysign(float x, float y) { return copysignf(x, y); }

// NumPy scalar semantics: np.minimum / np.maximum / np.clip propagate NaN
template <typename T> SDF_DEV T s_min(T a, T b) { return (a < b || a != a) ? a : b; }
template <typename T> SDF_DEV T s_max(T a, T b) { return (a >= b || a != a) ? a : b; }
template <typename T> SDF_DEV T s_clip(T x, T lo, T hi) {
    T t = (x != x || x > lo) ? x : lo;
    return (t != t || t < hi) ? t : hi;
}
template <typename T> SDF_DEV T s_sign(T x) { return x != x ? x : (x > T(0) ? T(1) : (x < T(0) ? T(-1) : T(0))); }
// Python / NumPy floored modulo (npy_divmod)
template <typename T> SDF_DEV T s_mod(T a, T b) {
    T m = m_fmod(a, b);
    if (b == T(0)) return m;
    if (m != T(0)) { if ((b < T(0)) != (m < T(0))) m += b; }
    else m = m_copysign(T(0), b);
    return m;
}

// ---- the same, element-wise over Vec ------------------------------------------------------
SDF_VEC_MAP1(m_sqrt, m_sqrt(x))
SDF_VEC_MAP1(m_fabs, m_fabs(x))
SDF_VEC_MAP1(m_rint, m_rint(x))
SDF_VEC_MAP1(m_sin, m_sin(x))
SDF_VEC_MAP1(m_cos, m_cos(x))
SDF_VEC_MAP1(m_pow2, m_pow2(x))
// sin and cos of the same angle share the argument reduction (float64: sincos64 above; float32: ocml)
template <int N> SDF_DEV void m_sincos(const Vec<float, N> &a, Vec<float, N> &s, Vec<float, N> &c) {
    SDF_UNROLL for (int i = 0; i < N; i++) o_sincos(a.v[i], &s.v[i], &c.v[i]);
}
template <int N> SDF_DEV void m_sincos(const Vec<double, N> &a, Vec<double, N> &s, Vec<double, N> &c) {
    SDF_UNROLL for (int i = 0; i < N; i++) sincos64(a.v[i], s.v[i], c.v[i]);
}
SDF_VEC_MAP1(np_sign, s_sign(x))
SDF_VEC_MAP2(m_atan2, m_atan2(x, y))
SDF_VEC_MAP2(m_hypot, m_hypot(x, y))
SDF_VEC_MAP2(np_min, s_min(x, y))
SDF_VEC_MAP2(np_max, s_max(x, y))
SDF_VEC_MAP2(np_mod, s_mod(x, y))
template <typename T, int N> SDF_DEV Vec<T, N> np_clip(const Vec<T, N> &a, T lo, T hi) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = s_clip(a.v[i], lo, hi); return r;
}
template <typename T, int N> SDF_DEV Vec<T, N> np_clip(const Vec<T, N> &a, const Vec<T, N> &lo, const Vec<T, N> &hi) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = s_clip(a.v[i], lo.v[i], hi.v[i]); return r;
}

// np.linalg.norm(axis=1): sqrt of the left-to-right sum of squares
template <typename V> SDF_DEV V len2(const V &x, const V &y) { return m_sqrt(x * x + y * y); }
template <typename V> SDF_DEV V len3(const V &x, const V &y, const V &z) { return m_sqrt((x * x + y * y) + z * z); }
// np.dot((N,3),(3,)) / np.dot((N,3),(3,3)): BLAS kernels accumulate with fused multiply-adds
template <typename T, int N>
SDF_DEV Vec<T, N> dot3(const Vec<T, N> &ax, const Vec<T, N> &ay, const Vec<T, N> &az, T bx, T by, T bz) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = m_fma(az.v[i], bz, m_fma(ay.v[i], by, ax.v[i] * bx)); return r;
}
template <typename T, int N> SDF_DEV Vec<T, N> dot2(const Vec<T, N> &ax, const Vec<T, N> &ay, T bx, T by) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = m_fma(ay.v[i], by, ax.v[i] * bx); return r;
}
template <typename T> SDF_DEV T dot2s(T ax, T ay, T bx, T by) { return m_fma(ay, by, ax * bx); }

// ---- easing curves: reference sdf/ease.py:3-162 ------------------------------------------
template <typename T, int N> SDF_DEV Vec<T, N> out_bounce(const Vec<T, N> &t) {
    const Vec<T, N> a = (T(121) * t * t) / T(16);
    const Vec<T, N> b = (T(363.0 / 40) * t * t) - (T(99.0 / 10) * t) + T(17.0 / 5);
    const Vec<T, N> c = (T(4356.0 / 361) * t * t) - (T(35442.0 / 1805) * t) + T(16061.0 / 1805);
    const Vec<T, N> d = (T(54.0 / 5) * t * t) - (T(513.0 / 25) * t) + T(268.0 / 25);
    return vsel(t < T(4.0 / 11), a, vsel(t < T(8.0 / 11), b, vsel(t < T(9.0 / 10), c, d)));
}

template <typename T, bool FULL, int N> SDF_DEV Vec<T, N> ease_apply(int id, const Vec<T, N> &t) {
    typedef Vec<T, N> V;
    const T pi = T(3.141592653589793);
    V u, v, a, b;
    switch (id) {   // id is wave-uniform
    case EASE_linear: return t;
    case EASE_in_quad: return t * t;
    case EASE_out_quad: return -t * (t - T(2));
    case EASE_in_out_quad:
        u = T(2) * t - T(1); a = T(2) * t * t; b = T(-0.5) * (u * (u - T(2)) - T(1));
        return vsel(t < T(0.5), a, b);
    case EASE_in_cubic: return t * t * t;
    case EASE_out_cubic: u = t - T(1); return u * u * u + T(1);
    case EASE_in_out_cubic:
        u = t * T(2); v = u - T(2);
        return vsel(u < T(1), T(0.5) * u * u * u, T(0.5) * (v * v * v + T(2)));
    case EASE_in_quart: return t * t * t * t;
    case EASE_out_quart: u = t - T(1); return -(u * u * u * u - T(1));
    case EASE_in_out_quart:
        u = t * T(2); v = u - T(2);
        return vsel(u < T(1), T(0.5) * u * u * u * u, T(-0.5) * (v * v * v * v - T(2)));
    case EASE_in_quint: return t * t * t * t * t;
    case EASE_out_quint: u = t - T(1); return u * u * u * u * u + T(1);
    case EASE_in_out_quint:
        u = t * T(2); v = u - T(2);
        return vsel(u < T(1), T(0.5) * u * u * u * u * u, T(0.5) * (v * v * v * v * v + T(2)));
    case EASE_in_circ: return T(-1) * (m_sqrt(T(1) - t * t) - T(1));
    case EASE_out_circ: u = t - T(1); return m_sqrt(T(1) - u * u);
    case EASE_in_out_circ:
        u = t * T(2); v = u - T(2);
        return vsel(u < T(1), T(-0.5) * (m_sqrt(T(1) - u * u) - T(1)), T(0.5) * (m_sqrt(T(1) - v * v) + T(1)));
    case EASE_in_back: { const T k = T(1.70158); return t * t * ((k + T(1)) * t - k); }
    case EASE_out_back: { const T k = T(1.70158); u = t - T(1); return u * u * ((k + T(1)) * u + k) + T(1); }
    case EASE_in_out_back: {
        const T k = T(1.70158 * 1.525); u = t * T(2); v = u - T(2);
        return vsel(u < T(1), T(0.5) * (u * u * ((k + T(1)) * u - k)), T(0.5) * (v * v * ((k + T(1)) * v + k) + T(2))); }
    case EASE_in_bounce: return T(1) - out_bounce(T(1) - t);
    case EASE_out_bounce: return out_bounce(t);
    case EASE_in_out_bounce:
        return vsel(t < T(0.5), (T(1) - out_bounce(T(1) - T(2) * t)) * T(0.5), out_bounce(T(2) * t - T(1)) * T(0.5) + T(0.5));
    case EASE_in_square: return vsel_s(t < T(1), T(0), T(1));
    case EASE_out_square: return vsel_s(t > T(0), T(1), T(0));
    case EASE_in_out_square: return vsel_s(t < T(0.5), T(0), T(1));
    default: break;
    }
    if constexpr (FULL) {   // curves that need sin / cos / 2**x
        switch (id) {
        case EASE_in_sine: return -m_cos(t * pi / T(2)) + T(1);
        case EASE_out_sine: return m_sin(t * pi / T(2));
        case EASE_in_out_sine: return T(-0.5) * (m_cos(pi * t) - T(1));
        case EASE_in_expo: return vsel(t == T(0), T(0), m_pow2(T(10) * (t - T(1))));
        case EASE_out_expo: return vsel(t == T(1), T(1), T(1) - m_pow2(T(-10) * t));
        case EASE_in_out_expo:
            a = vsel(t < T(0.5), T(0.5) * m_pow2(T(20) * t - T(10)), T(1) - T(0.5) * m_pow2(T(-20) * t + T(10)));
            return vsel(t == T(0), T(0), vsel(t == T(1), T(1), a));
        case EASE_in_elastic: {
            const T k = T(0.5); u = t - T(1);
            return T(-1) * (m_pow2(T(10) * u) * m_sin((u - k / T(4)) * (T(2) * pi) / k)); }
        case EASE_out_elastic: {
            const T k = T(0.5);
            return m_pow2(T(-10) * t) * m_sin((t - k / T(4)) * (T(2) * pi / k)) + T(1); }
        case EASE_in_out_elastic: {
            const T k = T(0.5); u = t * T(2); v = u - T(1);
            a = T(-0.5) * (m_pow2(T(10) * v) * m_sin((v - k / T(4)) * T(2) * pi / k));
            b = m_pow2(T(-10) * v) * m_sin((v - k / T(4)) * T(2) * pi / k) * T(0.5) + T(1);
            return vsel(u < T(1), a, b); }
        default: break;
        }
    }
    return t - t + T(__builtin_nan(""));   // unknown id
}

// ---- boolean folds: reference sdf/dn.py:7-58 ----------------------------------------------
template <typename T, int N> SDF_DEV Vec<T, N> post_combine(uint32_t post, const Vec<T, N> &d1, const Vec<T, N> &d2, T K) {
    Vec<T, N> h, m;
    switch (post) {   // wave-uniform
    case POST_SET: return real_move(d2);   // acc and the leaf value keep their own registers (see real_move)
    case POST_UNION: return np_min(d1, d2);
    case POST_DIFF: return np_max(d1, -d2);
    case POST_INTER: return np_max(d1, d2);
    case POST_SUNION:
        h = np_clip(T(0.5) + T(0.5) * (d2 - d1) / K, T(0), T(1));
        m = d2 + (d1 - d2) * h;
        return m - K * h * (T(1) - h);
    case POST_SDIFF:
        h = np_clip(T(0.5) - T(0.5) * (d2 + d1) / K, T(0), T(1));
        m = d1 + (-d2 - d1) * h;
        return m + K * h * (T(1) - h);
    case POST_SINTER:
        h = np_clip(T(0.5) - T(0.5) * (d2 - d1) / K, T(0), T(1));
        m = d2 + (d1 - d2) * h;
        return m + K * h * (T(1) - h);
    case POST_BLEND: return K * d2 + (T(1) - K) * d1;
    }
    return d2;
}

// A small per-lane register file addressed by a WAVE-UNIFORM slot number.  A plain array indexed
// by a run-time value would be placed in scratch memory by the compiler (and so would a struct
// whose members are picked by a switch: the optimiser merges the arms into one load through a phi
// of member addresses).  Slots are therefore passed around BY VALUE and picked by a chain of
// wave-uniform branches; the empty asm statements keep the arms from being merged back into
// selects or pointer phis.  S is the number of slots the kernel variant provides (the host picks
// the smallest variant that fits the tape: fewer slots = fewer VGPRs).
template <typename V, int S> struct RegFile {
    V r0, r1, r2, r3, r4, r5, r6, r7;
};
#define SDF_RF_ARM(K) asm volatile("; slot " #K)
template <typename V, int S> SDF_DEV void rf_clear(RegFile<V, S> &f, const V &z) {
    f.r0 = z;
    if constexpr (S > 1) f.r1 = z;
    if constexpr (S > 2) f.r2 = z;
    if constexpr (S > 3) f.r3 = z;
    if constexpr (S > 4) f.r4 = z;
    if constexpr (S > 5) f.r5 = z;
    if constexpr (S > 6) f.r6 = z;
    if constexpr (S > 7) f.r7 = z;
}
// run STMT(rK) for the member rK selected by the wave-uniform slot number s
#define SDF_RF_PICK(S_, s, STMT)                                              \
    do {                                                                      \
        if (S_ > 1 && (s) == 1) { SDF_RF_ARM(1); STMT(r1); SDF_RF_ARM(1); }       \
        else if (S_ > 2 && (s) == 2) { SDF_RF_ARM(2); STMT(r2); SDF_RF_ARM(2); }  \
        else if (S_ > 3 && (s) == 3) { SDF_RF_ARM(3); STMT(r3); SDF_RF_ARM(3); }  \
        else if (S_ > 4 && (s) == 4) { SDF_RF_ARM(4); STMT(r4); SDF_RF_ARM(4); }  \
        else if (S_ > 5 && (s) == 5) { SDF_RF_ARM(5); STMT(r5); SDF_RF_ARM(5); }  \
        else if (S_ > 6 && (s) == 6) { SDF_RF_ARM(6); STMT(r6); SDF_RF_ARM(6); }  \
        else if (S_ > 7 && (s) == 7) { SDF_RF_ARM(7); STMT(r7); SDF_RF_ARM(7); }  \
        else { SDF_RF_ARM(0); STMT(r0); SDF_RF_ARM(0); }                          \
    } while (0)
static_assert(SDF_NP_SLOTS == 8 && SDF_ND_SLOTS == 8, "RegFile holds at most eight slots");

template <typename V> SDF_DEV V box_like(const V &qx, const V &qy, const V &qz) {
    // _length(_max(q, 0)) + _min(np.amax(q, axis=1), 0)
    typedef decltype(qx.v[0] + qx.v[0]) T;
    const V mx = np_max(np_max(qx, qy), qz);
    return len3(np_max(qx, T(0)), np_max(qy, T(0)), np_max(qz, T(0))) + np_min(mx, T(0));
}

#define SDF_JT_ROW(NAME) "s_branch %l[L_" #NAME "]\n\t"
#define SDF_JT_LABEL(NAME) L_##NAME,

// The voxel look-up of the grid leaf (reference sdf/mesh.py:88-103): scipy 1.7.1 RegularGridInterpolator(method='linear',
// bounds_error=False, fill_value=background) over float32 voxels.  `_find_indices`: i = searchsorted(grid, x) - 1
// clipped to [0, n - 2], w = (x - grid[i]) / (grid[i + 1] - grid[i]), out of bounds = x < grid[0] or x > grid[-1];
// `_evaluate_linear`: values = 0.; for the 8 corners in itertools.product order: weight = ((1. * wx) * wy) * wz,
// values += voxel * weight.  Outlined like the libm bodies: its index arithmetic would otherwise cost every
// kernel variant registers at its tightest point.
// c: nx ny nz | background | box centre (3) | box half size (3) | X[nx] Y[ny] Z[nz] | A[nx][ny][nz]
template <typename T>
static __device__ __attribute__((noinline)) T grid3d_lookup(const T *__restrict__ c, T px, T py, T pz) {
    const int n0 = (int)c[0], n1 = (int)c[1], n2 = (int)c[2];
    const T *g = c + 10;
    const T *vox = g + n0 + n1 + n2;
    const T p[3] = {px, py, pz};
    const int n[3] = {n0, n1, n2};
    int idx[3];
    T w[3];
    bool oob = false;
    SDF_UNROLL
    for (int a = 0; a < 3; a++) {
        int lo = 0, hi = n[a];
        while (lo < hi) {           // np.searchsorted side='left'; NaN sorts behind everything, like NumPy
            const int mid = (lo + hi) >> 1;
            const T gm = g[mid];
            if (gm < p[a] || (p[a] != p[a] && gm == gm)) lo = mid + 1; else hi = mid;
        }
        const int i = min(max(lo - 1, 0), n[a] - 2);
        idx[a] = i;
        w[a] = (p[a] - g[i]) / (g[i + 1] - g[i]);
        oob = oob || p[a] < g[0] || p[a] > g[n[a] - 1];
        g += n[a];
    }
    T acc8 = T(0);
    SDF_UNROLL
    for (int q = 0; q < 8; q++) {
        const int o0 = q >> 2, o1 = (q >> 1) & 1, o2 = q & 1;
        T wt = o0 ? w[0] : T(1) - w[0];
        wt = wt * (o1 ? w[1] : T(1) - w[1]);
        wt = wt * (o2 ? w[2] : T(1) - w[2]);
        acc8 = acc8 + vox[((size_t)(idx[0] + o0) * n1 + (idx[1] + o1)) * n2 + (idx[2] + o2)] * wt;
    }
    return oob ? c[3] : acc8;
}

// User closures (L_EXTERN leaves, reference README.md:258-295): the values of a closure at the leaf's points
// are computed on the host by the user's own code.  A kernel that supports them passes an ExtIO: in `dump`
// mode the leaf stores its current point (the host then calls the closure on those points), in read mode it
// takes the closure's value from the buffer.  Every other kernel passes NoExt and the leaf yields NaN (the
// host never launches those kernels with a tape that has such leaves).
struct NoExt { static constexpr bool enabled = false; };
struct ExtIO {
    static constexpr bool enabled = true;
    double *buf;          // dump: [leaf][sample][3] points; read: [leaf][sample] values
    long long n, i;       // samples in the launch, this lane's sample
    bool dump;
};

// Run the whole tape for NS samples per lane.  FULL=false builds leave out the ops that need
// sin/cos/atan2/hypot/fmod/pow (their ocml bodies cost registers); NP / ND are the register-file
// sizes; the host picks the variant from the opcodes and slot counts of the tape.
template <typename T, bool FULL, int NP, int ND, int NS, typename EXT = NoExt>
__device__ __forceinline__ Vec<T, NS> run_tape(const uint32_t *__restrict__ code, const T *__restrict__ consts,
                                               Vec<T, NS> x, Vec<T, NS> y, Vec<T, NS> z, EXT ext = EXT()) {
    typedef Vec<T, NS> V;
    V acc(T(0));
    RegFile<V, NP> PSx, PSy, PSz;
    RegFile<V, ND> DS;
    rf_clear(PSx, V(T(0))); rf_clear(PSy, V(T(0))); rf_clear(PSz, V(T(0))); rf_clear(DS, V(T(0)));

#define SDF_DGET_(R) _dst = DS.R
#define SDF_DSET_(R) DS.R = _val
#define SDF_PGET_(R) _px = PSx.R; _py = PSy.R; _pz = PSz.R
#define SDF_PSET_(R) PSx.R = _px; PSy.R = _py; PSz.R = _pz
// copies BETWEEN state variables go through late_bind (sdf_vec.h): a plain copy would let the
// register coalescer merge e.g. a PS slot with the current point, which then costs every op a move
#define SDF_PRELOAD_(R) move_into(x, PSx.R); move_into(y, PSy.R); move_into(z, PSz.R)
#define SDF_PSAVE_(R) move_into(PSx.R, x); move_into(PSy.R, y); move_into(PSz.R, z)
#define SDF_DPUSH_(R) move_into(DS.R, acc)
#define SDF_DGETM_(R) _dst = real_move(DS.R)
#define SDF_PGETM_(R) _px = real_move(PSx.R); _py = real_move(PSy.R); _pz = real_move(PSz.R)
// reads of a slot into live state are explicit moves (real_move); writes of COMPUTED values are
// late-bound (no instruction), writes of live state (SAVE_P, PUSH_D) are explicit moves
#define DGET(OUT, s) do { V _dst; SDF_RF_PICK(ND, s, SDF_DGETM_); OUT = _dst; } while (0)
#define DSET(s, ...) do { V _val = (__VA_ARGS__); late_bind(_val); SDF_RF_PICK(ND, s, SDF_DSET_); } while (0)
#define DSETM(s, A) do { const V _val = real_move(A); SDF_RF_PICK(ND, s, SDF_DSET_); } while (0)
#define PGET(s, X, Y, Z) do { V _px, _py, _pz; SDF_RF_PICK(NP, s, SDF_PGETM_); X = _px; Y = _py; Z = _pz; } while (0)
#define PSET(s, X, Y, Z) do { V _px = (X), _py = (Y), _pz = (Z); late_bind(_px, _py, _pz); SDF_RF_PICK(NP, s, SDF_PSET_); } while (0)
#define PSETM(s, X, Y, Z) do { const V _px = real_move(X), _py = real_move(Y), _pz = real_move(Z); SDF_RF_PICK(NP, s, SDF_PSET_); } while (0)

    // The next instruction's words are requested before the current one executes, so their
    // scalar-cache latency overlaps this instruction's arithmetic (the host pads the code with a
    // second END so the look-ahead of END itself stays inside the buffer).  The loop is a plain
    // do-while with one latch: END sets `done` like any other op instead of leaving from the
    // middle, which keeps the control-flow graph around the jump table reducible and copy-free.
    // (the two words of an instruction travel as ONE loop-carried 64-bit value that is taken apart
    // at the top of the next pass: a separately carried `coff` made the compiler copy it right
    // after the load, i.e. wait for the load it had just issued)
    const unsigned long long *code64 = reinterpret_cast<const unsigned long long *>(code);
    unsigned long long nw = code64[0];
    uint32_t pc = 1;
    bool done = false;
    do {
        // word 0: op[0:8] post[8:11] RL[11] slot[12:15] SV[15] slot[16:19] PD[19] slot[20:23] a[24:32]
        // word 1: constant offset[0:24] b[24:32]                                  (sdf_amd/tape.py)
        const uint32_t w0 = __builtin_amdgcn_readfirstlane((uint32_t)nw);
        const uint32_t w1 = __builtin_amdgcn_readfirstlane((uint32_t)(nw >> 32));
        const uint32_t op = w0 & 255u, post = (w0 >> 8) & 7u, sa = w0 >> 24, sb = w1 >> 24;
        const T *c = consts + (w1 & 0xFFFFFFu) + 1;          // c[-1] is K
        nw = code64[pc];
        pc += 1;
        // prefixes (tape.py peephole): the bookkeeping a LOAD_P / SAVE_P / PUSH_D instruction in
        // front of this one would do, without paying a dispatch for it
        if (w0 & 0x0FF800u) {   // (in-place moves: the path without prefixes pays nothing at the merge)
            if (w0 & 0x000800u) { const uint32_t s = (w0 >> 12) & 7u; SDF_RF_PICK(NP, s, SDF_PRELOAD_); }
            if (w0 & 0x008000u) { const uint32_t s = (w0 >> 16) & 7u; SDF_RF_PICK(NP, s, SDF_PSAVE_); }
            if (w0 & 0x080000u) { const uint32_t s = (w0 >> 20) & 7u; SDF_RF_PICK(ND, s, SDF_DPUSH_); }
        }
        // Dispatch: ONE indirect jump through a table of s_branch instructions (the compiler only
        // offers a compare-and-branch tree for `switch`, and every taken branch costs an instruction
        // buffer refill).  s_getpc returns the address A of the instruction after it; the table
        // starts at A + 12 (three 4-byte SALU instructions), so the target is A + 4 * (op + 3).
        // s_lshl2_add_u32 leaves the carry of the 32-bit add in SCC for the s_addc.
        // A leaf computes v and jumps to `fold`; every other op jumps to `next`.
        V v;
        const uint32_t jt = op + 3u;
        asm goto(
            "s_getpc_b64 vcc\n\t"
            "s_lshl2_add_u32 vcc_lo, %0, vcc_lo\n\t"
            "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
            "s_setpc_b64 vcc\n\t"
            SDF_OPCODE_LIST(SDF_JT_ROW)
            : : "s"(jt) : "vcc", "scc" : SDF_OPCODE_LIST(SDF_JT_LABEL) L_BAD);
        goto next;   // never reached: the asm always jumps (kept inside the loop for the CFG passes)
        {
        // ---------------- 3-D leaves ----------------
        L_L_SPHERE:   // d3.py:92-96
            v = len3(x - c[1], y - c[2], z - c[3]) - c[0]; goto fold;
        L_L_PLANE:    // d3.py:98-103
            v = dot3(c[3] - x, c[4] - y, c[5] - z, c[0], c[1], c[2]); goto fold;
        L_L_BOX:      // d3.py:122-134
            v = box_like(m_fabs(x - c[0]) - c[3], m_fabs(y - c[1]) - c[4], m_fabs(z - c[2]) - c[5]); goto fold;
        L_L_ROUNDED_BOX:  // d3.py:136-142
            v = box_like(m_fabs(x) - c[0] + c[3], m_fabs(y) - c[1] + c[3], m_fabs(z) - c[2] + c[3]) - c[3]; goto fold;
        L_L_WIREFRAME_BOX: {  // d3.py:144-155
            const T t2 = c[3];
            const V px = m_fabs(x) - c[0] - t2, py = m_fabs(y) - c[1] - t2, pz = m_fabs(z) - c[2] - t2;
            const V qx = m_fabs(px + t2) - t2, qy = m_fabs(py + t2) - t2, qz = m_fabs(pz + t2) - t2;
            auto g = [](const V &a, const V &b, const V &cc) {
                return len3(np_max(a, T(0)), np_max(b, T(0)), np_max(cc, T(0))) + np_min(np_max(a, np_max(b, cc)), T(0));
            };
            v = np_min(np_min(g(px, qy, qz), g(qx, py, qz)), g(qx, qy, pz)); goto fold; }
        L_L_TORUS: {  // d3.py:157-165
            const V a = len2(x, y) - c[0];
            v = len2(a, z) - c[1]; goto fold; }
        L_L_CAPSULE: {  // d3.py:167-176
            const V pax = x - c[0], pay = y - c[1], paz = z - c[2];
            const V h = np_clip(dot3(pax, pay, paz, c[3], c[4], c[5]) / c[6], T(0), T(1));
            v = len3(pax - c[3] * h, pay - c[4] * h, paz - c[5] * h) - c[7]; goto fold; }
        L_L_CYLINDER:  // d3.py:178-182
            v = len2(x, y) - c[0]; goto fold;
        L_L_CAPPED_CYLINDER: {  // d3.py:184-204
            const T bax = c[3], bay = c[4], baz = c[5], baba = c[6];
            const V pax = x - c[0], pay = y - c[1], paz = z - c[2];
            const V paba = dot3(pax, pay, paz, bax, bay, baz);
            const V xx = len3(pax * baba - bax * paba, pay * baba - bay * paba, paz * baba - baz * paba) - c[8];
            const V yy = m_fabs(paba - c[9]) - c[9];
            const V x2 = xx * xx, y2 = yy * yy * baba;
            const V din = -np_min(x2, y2);
            const V dout = vsel(xx > T(0), x2, T(0)) + vsel(yy > T(0), y2, T(0));
            const V d = vsel(np_max(xx, yy) < T(0), din, dout);
            v = np_sign(d) * m_sqrt(m_fabs(d)) / baba; goto fold; }
        L_L_ROUNDED_CYLINDER: {  // d3.py:206-215
            const V d0 = len2(x, y) - c[0] + c[1];
            const V dd1 = m_fabs(z) - c[2] + c[1];
            v = np_min(np_max(d0, dd1), T(0)) + len2(np_max(d0, T(0)), np_max(dd1, T(0))) - c[1]; goto fold; }
        L_L_CAPPED_CONE: {  // d3.py:217-237
            const T ra = c[6], rb = c[7], baba = c[8], rba = c[9], k = c[10];
            const V pax = x - c[0], pay = y - c[1], paz = z - c[2];
            const V papa = (pax * pax + pay * pay) + paz * paz;
            const V paba = dot3(pax, pay, paz, c[3], c[4], c[5]) / baba;
            const V xx = m_sqrt(papa - paba * paba * baba);
            const V cax = np_max(T(0), xx - vsel_s(paba < T(0.5), ra, rb));
            const V cay = m_fabs(paba - T(0.5)) - T(0.5);
            const V f = np_clip((rba * (xx - ra) + paba * baba) / k, T(0), T(1));
            const V cbx = xx - ra - f * rba;
            const V cby = paba - f;
            const V s = vsel_s((cbx < T(0)) & (cay < T(0)), T(-1), T(1));
            v = s * m_sqrt(np_min(cax * cax + cay * cay * baba, cbx * cbx + cby * cby * baba)); goto fold; }
        L_L_ROUNDED_CONE: {  // d3.py:239-250
            const T r1 = c[0], r2 = c[1], h = c[2], b = c[3], a = c[4], ah = c[5];
            const V qx = len2(x, y), qy = z;
            const V k = dot2(qx, qy, -b, a);
            const V c1 = len2(qx, qy) - r1;
            const V c2 = len2(qx - T(0), qy - h) - r2;
            const V c3 = dot2(qx, qy, a, b) - r1;
            v = vsel(k < T(0), c1, vsel(k > ah, c2, c3)); goto fold; }
        L_L_ELLIPSOID: {  // d3.py:252-259
            const V k0 = len3(x / c[0], y / c[1], z / c[2]);
            const V k1 = len3(x / c[3], y / c[4], z / c[5]);
            v = k0 * (k0 - T(1)) / k1; goto fold; }
        L_L_PYRAMID: {  // d3.py:261-282
            const T h = c[0], m2 = c[1], m2q = c[2];
            const V b0 = m_fabs(x) - T(0.5), b1 = m_fabs(y) - T(0.5);
            const Mask<NS> sw = b1 > b0;
            const V a0 = vsel(sw, b1, b0), a1 = vsel(sw, b0, b1);
            const V px = a0, py = z, pz = a1;
            const V qx = pz, qy = h * py - T(0.5) * px, qz = h * px + T(0.5) * py;
            const V s = np_max(-qx, T(0));
            const V tt = np_clip((qy - T(0.5) * pz) / m2q, T(0), T(1));
            const V a = m2 * ((qx + s) * (qx + s)) + qy * qy;
            const V b = m2 * ((qx + T(0.5) * tt) * (qx + T(0.5) * tt)) + (qy - m2 * tt) * (qy - m2 * tt);
            const V dd2 = vsel(np_min(qy, -qx * m2 - qy * T(0.5)) > T(0), T(0), np_min(a, b));
            v = m_sqrt((dd2 + qz * qz) / m2) * np_sign(np_max(qz, -py)); goto fold; }
        L_L_TETRAHEDRON:  // d3.py:286-293
            v = (np_max(m_fabs(x + y) - z, m_fabs(x - y) + z) - c[0]) / c[1]; goto fold;
        L_L_OCTAHEDRON:   // d3.py:295-299
            v = (((m_fabs(x) + m_fabs(y)) + m_fabs(z)) - c[0]) * c[1]; goto fold;
        L_L_DODECAHEDRON: {  // d3.py:301-311
            const T r = c[0], X = c[1], Y = c[2], Z = c[3];
            const V ax = m_fabs(x / r), ay = m_fabs(y / r), az = m_fabs(z / r);
            const V a = dot3(ax, ay, az, X, Y, Z), b = dot3(ax, ay, az, Z, X, Y), cc = dot3(ax, ay, az, Y, Z, X);
            v = (np_max(np_max(a, b), cc) - X) * r; goto fold; }
        L_L_ICOSAHEDRON: {  // d3.py:313-325
            const T r = c[0], X = c[1], Y = c[2], Z = c[3], w = c[4];
            const V ax = m_fabs(x / r), ay = m_fabs(y / r), az = m_fabs(z / r);
            const V a = dot3(ax, ay, az, X, Y, Z), b = dot3(ax, ay, az, Z, X, Y), cc = dot3(ax, ay, az, Y, Z, X);
            const V d = dot3(ax, ay, az, w, w, w) - X;
            v = np_max(np_max(np_max(a, b), cc) - X, d) * r; goto fold; }
        // ---------------- 2-D leaves: the point is (x, y) ----------------
        L_L_CIRCLE:  // d2.py:76-80
            v = len2(x - c[1], y - c[2]) - c[0]; goto fold;
        L_L_LINE:    // d2.py:82-87
            v = dot2(c[2] - x, c[3] - y, c[0], c[1]); goto fold;
        L_L_RECTANGLE: {  // d2.py:102-114
            const V qx = m_fabs(x - c[0]) - c[2], qy = m_fabs(y - c[1]) - c[3];
            v = len2(np_max(qx, T(0)), np_max(qy, T(0))) + np_min(np_max(qx, qy), T(0)); goto fold; }
        L_L_ROUNDED_RECTANGLE: {  // d2.py:116-134 (later assignments win, as in the reference)
            const Mask<NS> xp = x > T(0), yp = y > T(0);
            V r(T(0));
            r = vsel(xp & yp, c[2], r);
            r = vsel(xp & !yp, c[3], r);
            r = vsel(!xp & !yp, c[4], r);
            r = vsel(!xp & yp, c[5], r);
            const V qx = m_fabs(x) - c[0] + r, qy = m_fabs(y) - c[1] + r;
            v = np_min(np_max(qx, qy), T(0)) + len2(np_max(qx, T(0)), np_max(qy, T(0))) - r; goto fold; }
        L_L_EQUILATERAL_TRIANGLE: {  // d2.py:136-152
            const T k = c[0];
            V px = m_fabs(x) - T(1), py = y + c[1];
            const Mask<NS> w = px + k * py > T(0);
            const V nx = (px - k * py) / T(2), ny = (-k * px - py) / T(2);
            px = vsel(w, nx, px); py = vsel(w, ny, py);
            px = px - np_clip(px, T(-2), T(0));
            v = -len2(px, py) * np_sign(py); goto fold; }
        L_L_HEXAGON: {  // d2.py:154-165
            const T r = c[0], k0 = c[1], k1 = c[2];
            V px = m_fabs(x), py = m_fabs(y);
            const V m = np_min(k0 * px + k1 * py, T(0));
            px = px - c[4] * m; py = py - c[5] * m;
            px = px - np_clip(px, c[6], c[7]); py = py - (T(0) + r);
            v = len2(px, py) * np_sign(py); goto fold; }
        L_L_ROUNDED_X: {  // d2.py:167-173
            const V px = m_fabs(x), py = m_fabs(y);
            const V qq = np_min(px + py, c[0]) * T(0.5);
            v = len2(px - qq, py - qq) - c[1]; goto fold; }
        L_L_POLYGON: {  // d2.py:175-196
            const int np_ = (int)c[0];
            const T *pv = c + 1;
            const V dx = x - pv[0], dy = y - pv[1];
            V d = dx * dx + dy * dy;
            V s(T(1));
            for (int i = 0; i < np_; i++) {
                const int j = (i + np_ - 1) % np_;
                const T vix = pv[2 * i], viy = pv[2 * i + 1], vjx = pv[2 * j], vjy = pv[2 * j + 1];
                const T ex = vjx - vix, ey = vjy - viy;
                const V wx = x - vix, wy = y - viy;
                const T ee = dot2s(ex, ey, ex, ey);
                const V cl = np_clip(dot2(wx, wy, ex, ey) / ee, T(0), T(1));
                const V bx = wx - ex * cl, by = wy - ey * cl;
                d = np_min(d, bx * bx + by * by);
                const Mask<NS> c1 = y >= viy, c2 = y < vjy, c3 = ex * wy > ey * wx;
                s = vsel((c1 & c2 & c3) | (!c1 & !c2 & !c3), -s, s);
            }
            v = s * m_sqrt(d); goto fold; }
        L_L_VESICA: {  // d2.py:198-207
            const T r = c[0], d = c[1], b = c[2];
            const V px = m_fabs(x), py = m_fabs(y);
            v = vsel((py - b) * d > px * b, len2(px - T(0), py - b), len2(px - (-d), py - T(0)) - r); goto fold; }
        L_L_TEXTURE2D: {  // text.py:116-153 (`f` of `_sdf`) + :138-153 (`_bilinear_interpolate`)
            // c: x0 y0 x1 y1 | pw ph px py | tw th | fallback rectangle (cx cy hx hy) | texture[th][tw]
            const int tw = (int)c[8], th = (int)c[9];
            const T *tex = c + 14;
            const V u = (x - c[0]) / (c[2] - c[0]);
            V vv = (y - c[1]) / (c[3] - c[1]);
            vv = T(1) - vv;
            const V ti = u * c[4] + c[6], tj = vv * c[5] + c[7];
            V d;
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const T fi = ti.v[k], fj = tj.v[k];
                // floor, then np.clip to the texture; the float clamp only keeps the int conversion in
                // range (every value below -1 / above tw clips to the same indices)
                const T gi = fi == fi ? s_clip(m_floor(fi), T(-2), (T)tw) : T(0);
                const T gj = fj == fj ? s_clip(m_floor(fj), T(-2), (T)th) : T(0);
                const int a0 = (int)gi, b0 = (int)gj;
                const int ix0 = min(max(a0, 0), tw - 1), ix1 = min(max(a0 + 1, 0), tw - 1);
                const int iy0 = min(max(b0, 0), th - 1), iy1 = min(max(b0 + 1, 0), th - 1);
                const T pa = tex[iy0 * tw + ix0], pb = tex[iy1 * tw + ix0], pc = tex[iy0 * tw + ix1], pd = tex[iy1 * tw + ix1];
                const T wa = ((T)ix1 - fi) * ((T)iy1 - fj), wb = ((T)ix1 - fi) * (fj - (T)iy0);
                const T wc = (fi - (T)ix0) * ((T)iy1 - fj), wd = (fi - (T)ix0) * (fj - (T)iy0);
                d.v[k] = wa * pa + wb * pb + wc * pc + wd * pd;
            }
            const V qx = m_fabs(x - c[10]) - c[12], qy = m_fabs(y - c[11]) - c[13];       // d2.py:102-114
            const V q = len2(np_max(qx, T(0)), np_max(qy, T(0))) + np_min(np_max(qx, qy), T(0));
            const Mask<NS> outside = (ti < T(0)) | (ti >= (T)(tw - 1)) | (tj < T(0)) | (tj >= (T)(th - 1));
            v = vsel(outside, q, d); goto fold; }
        L_L_GRID3D: {  // mesh.py:96-105: np.where(e > background, e, interpolator(p)); e = box(a, b) (d3.py:122-134)
            const V e = box_like(m_fabs(x - c[4]) - c[7], m_fabs(y - c[5]) - c[8], m_fabs(z - c[6]) - c[9]);
            V d;
            SDF_UNROLL for (int k = 0; k < NS; k++) d.v[k] = grid3d_lookup<T>(c, x.v[k], y.v[k], z.v[k]);
            v = vsel(e > c[3], e, d); goto fold; }
        L_L_EXTERN: {  // a user closure: its value at this leaf's point comes from the host (ExtIO)
            if constexpr (EXT::enabled) {
                static_assert(NS == 1, "extern leaves: one sample per lane");
                const long long k = (long long)c[0];
                if (ext.dump) {
                    double *o = ext.buf + (k * ext.n + ext.i) * 3;
                    o[0] = (double)x.v[0]; o[1] = (double)y.v[0]; o[2] = (double)z.v[0];
                    v = V(T(0));
                } else {
                    v = V((T)ext.buf[k * ext.n + ext.i]);
                }
            } else {
                v = V(T(__builtin_nan("")));
            }
            goto fold; }
        // ---------------- fold a parked distance ----------------
        L_COMB: { V d1; DGET(d1, sa); acc = post_combine(post, d1, acc, c[-1]); goto next; }
        // ---------------- point ops ----------------
        L_TRANSLATE:  // d3.py:329-333
            x = x - c[0]; y = y - c[1]; z = z - c[2]; goto next;
        L_SCALE:      // d3.py:335-345
            x = x / c[0]; y = y / c[1]; z = z / c[2]; goto next;
        L_ROTATE: {   // d3.py:347-360: p @ M, M row-major
            V nx = dot3(x, y, z, c[0], c[3], c[6]);
            V ny = dot3(x, y, z, c[1], c[4], c[7]);
            V nz = dot3(x, y, z, c[2], c[5], c[8]);
            late_bind(nx, ny, nz);
            x = nx; y = ny; z = nz; goto next; }
        L_ELONGATE: {  // d3.py:396-405
            const V qx = m_fabs(x) - c[0], qy = m_fabs(y) - c[1], qz = m_fabs(z) - c[2];
            DSET(sa, np_min(np_max(qx, np_max(qy, qz)), T(0)));
            x = np_max(qx, T(0)); y = np_max(qy, T(0)); z = np_max(qz, T(0)); goto next; }
        L_BEND_LINEAR: {  // d3.py:435-445
            V tt = np_clip(dot3(x - c[0], y - c[1], z - c[2], c[3], c[4], c[5]) / c[6], T(0), T(1));
            // (the curves the example models bend with -- ease.linear is bend_linear's default, weave.py passes
            // ease.in_out_quad -- without going through ease_apply's 34-way switch: in the trig-capable kernels that is
            // an out-of-line function, and a call here parks the whole machine state, see sincos64)
            const int eid = (int)c[10];
            if (eid == EASE_in_out_quad) {
                const V u = T(2) * tt - T(1), a = T(2) * tt * tt, b = T(-0.5) * (u * (u - T(2)) - T(1));
                tt = vsel(tt < T(0.5), a, b);
            } else if (eid != EASE_linear) {
                tt = ease_apply<T, FULL, NS>(eid, tt);
            }
            x = x + tt * c[7]; y = y + tt * c[8]; z = z + tt * c[9]; goto next; }
        L_REP_PREP: {  // dn.py:80-112: cell index of p
            const int dim = (int)c[0];
            V idx[3] = {V(T(0)), V(T(0)), V(T(0))};
            const V pp[3] = {x, y, z};
            SDF_UNROLL
            for (int i = 0; i < 3; i++) {
                if (i < dim) {
                    const T s = c[1 + i];
                    V r = m_rint(s != T(0) ? pp[i] / s : V(T(0)));
                    if (c[4] != T(0)) r = np_clip(r, -c[5 + i], c[5 + i]);
                    idx[i] = r;
                }
            }
            PSET(sa, idx[0], idx[1], idx[2]); goto next; }
        L_REP_SET:   // p = p0 - spacing * (index + n)
        {   V ax, ay, az, bx, by, bz;
            PGET(sa, ax, ay, az); PGET(sb, bx, by, bz);
            x = ax - c[0] * (bx + c[3]);
            y = ay - c[1] * (by + c[4]);
            z = az - c[2] * (bz + c[5]); goto next; }
        L_TRANSLATE2: x = x - c[0]; y = y - c[1]; goto next;   // d2.py:211-215
        L_SCALE2: x = x / c[0]; y = y / c[1]; goto next;       // d2.py:217-227
        L_ROTATE2: {  // d2.py:229-240
            V nx = dot2(x, y, c[0], c[2]), ny = dot2(x, y, c[1], c[3]);
            late_bind(nx, ny);
            x = nx; y = ny; goto next; }
        L_ELONGATE2: {  // d2.py:249-257
            const V qx = m_fabs(x) - c[0], qy = m_fabs(y) - c[1];
            DSET(sa, np_min(np_max(qx, qy), T(0)));
            x = np_max(qx, T(0)); y = np_max(qy, T(0)); goto next; }
        L_REVOLVE: {  // d2.py:280-286
            V nx = len2(x, y) - c[0], ny = z;
            late_bind(nx, ny);
            y = ny; x = nx; z = V(T(0)); goto next; }
        L_SETZ0: z = V(T(0)); goto next;                        // d3.py:513
        L_SAVE_P: PSETM(sa, x, y, z); goto next;
        L_LOAD_P: PGET(sa, x, y, z); goto next;
        // ---------------- distance ops ----------------
        L_PUSH_D: DSETM(sa, acc); goto next;
        L_NEG: acc = -acc; goto next;                            // dn.py:60-63
        L_ADDC: acc = acc + c[0]; goto next;                     // dn.py:70-73
        L_SUBC: acc = acc - c[0]; goto next;                     // dn.py:65-68
        L_MULC: acc = acc * c[0]; goto next;                     // d3.py:344
        L_SHELL: acc = m_fabs(acc) - c[0]; goto next;            // dn.py:75-78
        L_ADD_DS: { V t; DGET(t, sa); acc = acc + t; goto next; }   // d3.py:405
        L_TRANS_LIN_PRE: {  // d3.py:459-470
            const V tt = np_clip(dot3(x - c[0], y - c[1], z - c[2], c[3], c[4], c[5]) / c[6], T(0), T(1));
            DSET(sa, ease_apply<T, FULL, NS>((int)c[7], tt)); goto next; }
        L_TRANS_MIX: {  // t * d2 + (1 - t) * d1
            V tt, dd; DGET(tt, sa); DGET(dd, sb);
            acc = tt * acc + (T(1) - tt) * dd; goto next; }
        L_EXT_PRE: DSET(sa, m_fabs(z) - c[0]); goto next;      // d2.py:264-266
        L_EXT_POST: {  // d2.py:267
            V w1; DGET(w1, sa);
            acc = np_min(np_max(acc, w1), T(0)) + len2(np_max(acc, T(0)), np_max(w1, T(0))); goto next; }
        L_EXTTO_PRE:   // d2.py:274
            DSET(sa, ease_apply<T, FULL, NS>((int)c[1], np_clip(z / c[0], T(-0.5), T(0.5)) + T(0.5))); goto next;
        L_EXTTO_MIX: {  // d2.py:275
            V dd1, tt; DGET(dd1, sb); DGET(tt, sa);
            acc = dd1 + (acc - dd1) * tt; goto next; }
        L_NOP: goto next;   // (sdf_prune.h: a skipped instruction whose prefixes still have to run)
        L_SLICE_POST: {  // d3.py:515-519
            V A; DGET(A, sa); const V B = -acc;
            acc = vsel(A <= T(0), B, A); goto next; }
        L_TWIST: if constexpr (FULL) {  // d3.py:407-419
            V cc, s; m_sincos(c[0] * z, s, cc);
            V nx = cc * x - s * y, ny = s * x + cc * y;
            late_bind(nx, ny);
            x = nx; y = ny; } goto next;
        L_BEND: if constexpr (FULL) {   // d3.py:421-433
            V cc, s; m_sincos(c[0] * x, s, cc);
            V nx = cc * x - s * y, ny = s * x + cc * y;
            late_bind(nx, ny);
            x = nx; y = ny; } goto next;
        L_BEND_RADIAL: if constexpr (FULL) {  // d3.py:447-457
            const V r = m_hypot(x, y);
            const V tt = np_clip((r - c[0]) / c[1], T(0), T(1));
            z = z - c[2] * ease_apply<T, FULL, NS>((int)c[3], tt); } goto next;
        L_WRAP_AROUND: if constexpr (FULL) {  // d3.py:483-502
            const T pi = T(3.141592653589793);
            const V d = m_hypot(x, y) - c[9];
            const V a = m_atan2(y, x);
            const V tt = ease_apply<T, FULL, NS>((int)c[10], (a + pi) / (T(2) * pi));
            V nx = c[0] + c[3] * tt + c[6] * d, ny = c[1] + c[4] * tt + c[7] * d;
            late_bind(nx, ny);
            x = nx; y = ny; } goto next;
        // circular_array (d3.py:379-392): d = hypot(x, y), a = arctan2(y, x) % da, then the child at
        // (cos(a - delta) * d, sin(a - delta) * d, z) for delta = da and delta = 0.  In float64 the same two points are
        // reached by ROTATING (x, y): with k = floor(arctan2(y, x) / da) -- the sector NumPy's floored modulo puts the point
        // in -- (cos(a - delta) d, sin(a - delta) d) is (x, y) turned by -(k da + delta).  CIRC_PREP leaves the point turned by
        // -k da in the saved-point slot (found by a binary search of rotations, below: no atan2, no hypot, no fmod, no sin /
        // cos); CIRC_SET is four products with the host's cos / sin of delta (c[1], c[2]; delta = 0: the saved point
        // itself).  c[1] == 2 (PREP) / c[3] != 0 (SET) say that the lowering found pi / 4096 <= da <= pi (else, and in
        // float32, the polar form).  (r03 found k with an inline atan2 and the rotation with an inline sincos of k da:
        // ~200 vector instructions per sample, a quarter of weave's interpreter time.)
        L_CIRC_PREP: if constexpr (FULL) {
            bool polar = true;
            if constexpr (sizeof(T) == 8) {
                if (c[1] == T(2)) {   // (uniform) the sector by a binary search of rotations: no atan2, no sin / cos
                    // With phi = the point's angle mirrored into [0, pi] (y -> |y|), turn the point back by 2^m da for
                    // m = M .. 0 whenever it stays on the counter-clockwise side (ry >= 0): afterwards it has been turned
                    // by -floor(phi / da) da and its angle r lies in [0, da).  For y >= 0 that is the wanted point; for
                    // y < 0 the angle is -phi = -(k' + 1) da + (da - r): mirror back and turn forward by da (r = 0: mirror
                    // only).  Against the reference's own expression order: <= 7e-16 of the radius over 4e6 random and
                    // near-boundary points for 2 .. 100 sectors (tools/circ_sector_check.py), the sector equal except within
                    // ~1e-16 rad of a boundary, like the atan2 form before it; ~60 vector instructions per sample instead
                    // of ~200 (r04: weave 2^33, gearlike 2^30).
                    polar = false;
                    const int nst = (int)c[2];
                    V xr, yr;
                    SDF_UNROLL
                    for (int i = 0; i < NS; i++) {
                        const bool neg = __builtin_signbit(y.v[i]);
                        double qx = x.v[i], qy = neg ? -y.v[i] : y.v[i];
                        for (int m = nst - 1; m >= 0; m--) {     // (uniform trip count; the constants are scalar loads)
                            const double cm = c[3 + 2 * m], sm = c[4 + 2 * m];
                            const double rx = qx * cm + qy * sm, ry = qy * cm - qx * sm;
                            const bool acc = ry >= 0.0;
                            qx = acc ? rx : qx; qy = acc ? ry : qy;
                        }
                        const double ux = qx, uy = -qy, c0 = c[3], s0 = c[4];
                        const bool fwd = qy != 0.0;
                        const double nx = fwd ? ux * c0 - uy * s0 : ux, ny = fwd ? ux * s0 + uy * c0 : uy;
                        xr.v[i] = neg ? nx : qx;
                        yr.v[i] = neg ? ny : qy;
                        // A point ON a coordinate axis (x == 0 or y == 0: whole planes of a grid like np.arange(-1, 1, 0.01))
                        // has an angle that arctan2 returns EXACTLY (0, +-pi/2, +-pi), and the reference's floored modulo of
                        // it is exact too -- 0.0 wherever da divides the angle in floating point (4 | count: the quarter
                        // turns), else a remainder of a few 1e-16 on ONE side of the boundary.  The search above lands within
                        // 1e-16 rad of that boundary on a side of its own (the accepted quarter turn leaves qy ~ 6e-17, not 0),
                        // which an asymmetric child turns into a different sector's value (r04 advisor finding).  Such lanes
                        // take the reference's own expression: A = the exact angle, m = A mod da by an exact remainder
                        // (fma(-k, da, A) with k corrected to the floor: the remainder of two doubles is a double), the point
                        // (cos(m) d, sin(m) d) with d = |x| + |y| = hypot exactly.  (wave-uniform branch; no call)
                        const bool axis = x.v[i] == 0.0 || y.v[i] == 0.0;
                        if (__builtin_expect(__ballot(axis) != 0ull, 0)) {
                            const double xa = x.v[i], ya = y.v[i], dda = c[0];
                            const double A = xa == 0.0 ? (ya == 0.0 ? (__builtin_signbit(xa) ? 3.141592653589793 : 0.0) : 1.5707963267948966)
                                                       : (__builtin_signbit(xa) ? 3.141592653589793 : 0.0);   // |arctan2(y, x)| on an axis
                            double k = floor(A / dda);
                            double rm = fma(-k, dda, A);                       // (the quotient may be one off: the sign tells)
                            k = rm < 0.0 ? k - 1.0 : (rm >= dda ? k + 1.0 : k);
                            rm = fma(-k, dda, A);                              // exact: 0 <= A - k da < da is a double
                            // arctan2 carries y's sign: fmod(-A, da) = -rm, and NumPy's floored modulo adds da to a negative
                            // remainder (npy_divmod); a zero remainder is +0.0
                            const double mm = (neg && rm != 0.0) ? dda - rm : rm;
                            double sn, cs;
                            sincos64(mm, sn, cs);
                            const double dd = fabs(xa) + fabs(ya);
                            xr.v[i] = axis ? cs * dd : xr.v[i];
                            yr.v[i] = axis ? sn * dd : yr.v[i];
                        }
                    }
                    PSET(sa, xr, yr, z);
                }
            }
            if constexpr (sizeof(T) == 8) {
                // (any other da -- one sector, or thousands -- takes the reference's polar form, out of line: ONE call site, so
                // that the search above keeps the machine state in registers)
                if (__builtin_expect(polar, 0)) {
                    V d, a;
                    SDF_UNROLL for (int i = 0; i < NS; i++) { const SinCos64 r = o_circ_polar64((double)x.v[i], (double)y.v[i], (double)c[0]); d.v[i] = (T)r.s; a.v[i] = (T)r.c; }
                    PSET(sa, d, a, z);
                    polar = false;
                }
            }
            if (polar) PSET(sa, m_hypot(x, y), np_mod(m_atan2(y, x), c[0]), z); } goto next;
        L_CIRC_SET: if constexpr (FULL) {   // p = (cos(a - delta) * d, sin(a - delta) * d, z)
            bool polar = true;
            if constexpr (sizeof(T) == 8) {
                if (c[3] != T(0)) {   // (uniform)
                    polar = false;
                    V xr, yr, z0; PGET(sa, xr, yr, z0);
                    if (c[0] == T(0)) { x = xr; y = yr; }
                    else { V nx = xr * c[1] + yr * c[2], ny = yr * c[1] - xr * c[2]; late_bind(nx, ny); x = nx; y = ny; }
                    z = z0;
                }
            }
            if (polar) {
                V d, a0, z0; PGET(sa, d, a0, z0);
                const V ang = a0 - c[0];
                V sn, cs; m_sincos(ang, sn, cs);
                x = cs * d; y = sn * d; z = z0; } } goto next;
        L_TRANS_RAD_PRE: if constexpr (FULL) {  // d3.py:472-481
            const V r = m_hypot(x, y);
            DSET(sa, ease_apply<T, FULL, NS>((int)c[2], np_clip((r - c[0]) / c[1], T(0), T(1)))); } goto next;
        L_END: done = true; goto next;
        L_BAD: goto next;
        }
        fold:
        acc = post_combine(post, acc, v, c[-1]);
        late_bind(acc);
        next:
        // the prefetched words are first touched HERE (the empty asm consumes them), i.e. the wait
        // for the scalar load issued at the top of this pass sits behind the op it overlapped with;
        // without it the compiler moves the loop-carried copy right behind the load
        asm volatile("" : "+s"(nw));
    } while (!done);
    return acc;
}

#undef DGET
#undef DSET
#undef DSETM
#undef PGET
#undef PSET
#undef PSETM

// one sample per lane, the largest register files: the variant the non-hot kernels use
template <typename T, bool FULL>
__device__ __forceinline__ T run_tape1(const uint32_t *__restrict__ code, const T *__restrict__ consts, T x, T y, T z) {
    return run_tape<T, FULL, SDF_NP_SLOTS, SDF_ND_SLOTS, 1>(code, consts, Vec<T, 1>(x), Vec<T, 1>(y), Vec<T, 1>(z)).v[0];
}
// the same with user closures (L_EXTERN leaves) served from / dumped to a buffer
template <typename T, bool FULL>
__device__ __forceinline__ T run_tape1_ext(const uint32_t *__restrict__ code, const T *__restrict__ consts, T x, T y, T z, ExtIO io) {
    return run_tape<T, FULL, SDF_NP_SLOTS, SDF_ND_SLOTS, 1, ExtIO>(code, consts, Vec<T, 1>(x), Vec<T, 1>(y), Vec<T, 1>(z), io).v[0];
}

}  // namespace sdfk
