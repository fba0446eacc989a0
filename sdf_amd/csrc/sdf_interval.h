// sdf_interval.h -- the tape in interval arithmetic: which instructions can a batch skip
// (prepass, sdf_prune.h), and where inside a batch can the surface not be (k_cull, sdf_device.h cull_tasks)?
//
// Inside one 33^3 batch most of a CSG tree is irrelevant: far from the cylinders of the canonical
// example, `max(sphere & box, -cylinders)` is decided by its left operand at EVERY sample, so the
// five instructions that evaluate the cylinders only burn VALU cycles.  The prepass runs the tape
// once per batch in INTERVAL arithmetic over the batch's box of sample coordinates
// (8 lanes per batch, one octant each; a decision must hold in all eight) and, at every hard
// min / max, compares the operand intervals:
//     the right operand can never win  ->  its instructions are skipped in this batch
//     the left  operand can never win  ->  the instructions of the left chain are skipped and the
//                                          combine is "forced" to take the right value
// (sdf_amd/tape.py records for every combine where its operands start) and then writes the batch
// its OWN tape with those instructions left out (`compact_tape`), which k_mesh's interpreter runs
// instead of the model's tape -- a pruned instruction costs nothing at all there.  Tapes longer
// than 256 instructions are not pruned.
//
// This never changes a result.  The basic interval operations perform the SAME floating-point operations,
// in the same order, as the float64 interpreter (sdf_interp.h) does for that op, on the interval's end
// points: +, -, *, /, sqrt, fma and rint are correctly rounded, hence monotone in each argument, so the range
// of an operation over a box of arguments is spanned by its values at the corners -- the interval contains
// every value the interpreter can produce for a sample in the box (no outward rounding is needed, and none
// is done).  Ops that go through libm on the device (hypot / atan2 / sin / cos: circular_array, twist, bend,
// wrap_around, ...) or whose floating-point form is not monotone term by term (easing curves, the smooth
// min / max) take the exact range of the real function widened by 1e-12; the rarer leaves are compositions
// of interval steps (ia_leaf_rare).  An operand of a hard min / max is only dropped when its interval lies
// STRICTLY on the losing side of the other's, in which case min / max returns the other operand bit for
// bit; an operand of a smooth one only in its exact direction (ia_decide).  Anything that could be NaN, and
// the few ops without an interval form (non-monotone easings), yields the whole real
// line, which never licenses a skip.  The parity tests (bit-identical soups against the CPU checker and the
// reference goldens, random CSG / array / leaf sweeps with the passes on and off) run with the prepass
// active; tests/test_interval_host.py checks the enclosure property of the primitives on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opcodes.h"

// the interval primitives also compile for the host: tests/test_interval_host.py builds them into a
// small CPU program and checks the enclosures against sampled points
#define SDF_IA __host__ __device__ __forceinline__

namespace sdfk {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void ia_sincos(double x, double *s, double *c) { sincos64(x, *s, *c); }   // (the interpreter's own, sdf_interp.h: inline, no call)
__device__ __forceinline__ double ia_atan2(double y, double x) { return atan2_64(y, x); }   // (inline, 1.5 ulp: the callers widen by 1e-12)
#else
inline void ia_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
inline double ia_atan2(double y, double x) { return atan2(y, x); }
#endif

struct Ival {
    double lo, hi;
};

namespace ia {

SDF_IA Ival top() { return Ival{-__builtin_inf(), __builtin_inf()}; }
SDF_IA Ival pt(double c) { return Ival{c, c}; }
SDF_IA bool bad(const Ival &a) { return !(a.lo <= a.hi); }                 // NaN or empty
SDF_IA Ival fix(const Ival &a) { return bad(a) ? top() : a; }
SDF_IA Ival wide(double lo, double hi) { return fix(Ival{lo, hi}); }   // (NaN -> the whole line)
// Intervals never hold a NaN; an operation that can produce one (inf - inf, 0 * inf) maps it to "no bound".
// For the operations whose lower result depends on lower bounds only and whose upper result on upper bounds only
// (sums, differences, products with a constant: monotone term by term) that can be done PER BOUND, with one
// instruction each -- maxNum / minNum return the operand that is not a NaN: a lower bound that came out as NaN
// is -inf, an upper one +inf, and the other bound stays what it is (it was computed from valid bounds of its
// own side).  The general form (`wide`: compare, then four selects) used to be two thirds of the instructions
// of a leaf like the box.
SDF_IA Ival side(double lo, double hi) { return Ival{fmax(lo, -__builtin_inf()), fmin(hi, __builtin_inf())}; }
SDF_IA Ival add(const Ival &a, const Ival &b) { return side(a.lo + b.lo, a.hi + b.hi); }
SDF_IA Ival sub(const Ival &a, const Ival &b) { return side(a.lo - b.hi, a.hi - b.lo); }
SDF_IA Ival addc(const Ival &a, double c) { return side(a.lo + c, a.hi + c); }        // (a NaN constant: both bounds NaN -> the whole line)
SDF_IA Ival subc(const Ival &a, double c) { return side(a.lo - c, a.hi - c); }
SDF_IA Ival csub(double c, const Ival &a) { return side(c - a.hi, c - a.lo); }
SDF_IA Ival neg(const Ival &a) { return Ival{-a.hi, -a.lo}; }
// a * c (== c * a): the sign of the (wave-uniform) constant says which end gives which bound
SDF_IA Ival mulc(const Ival &a, double c) {
    return c >= 0 ? side(a.lo * c, a.hi * c) : side(a.hi * c, a.lo * c);
}
// fma(x, c, r) with a constant c: monotone in x (direction = sign of c) and in r
SDF_IA Ival fmac(const Ival &x, double c, const Ival &r) {
    return c >= 0 ? side(fma(x.lo, c, r.lo), fma(x.hi, c, r.hi)) : side(fma(x.hi, c, r.lo), fma(x.lo, c, r.hi));
}
SDF_IA Ival divc(const Ival &a, double c) {
    if (!(c != 0.0)) return top();                                   // 0 or NaN
    return c > 0 ? side(a.lo / c, a.hi / c) : side(a.hi / c, a.lo / c);   // (inf / inf)
}
// x * x: with m <= |x| <= M over the interval (m = 0 when it straddles zero), [m * m, M * M] -- the rounded
// square is monotone in |x|; no products of infinities with zero, no comparisons
SDF_IA Ival sqr(const Ival &a) {
    const double m = fmax(fmax(a.lo, -a.hi), 0.0), M = fmax(-a.lo, a.hi);
    return Ival{m * m, M * M};
}
SDF_IA Ival sqrt_(const Ival &a) {
    // only sums of squares get here: never negative, never NaN unless a bound already is
    if (bad(a) || a.lo < 0) return top();
    return Ival{sqrt(a.lo), sqrt(a.hi)};
}
SDF_IA Ival abs_(const Ival &a) {   // (branch-free: the three cases of the sign are these two maxima)
    return Ival{fmax(fmax(a.lo, -a.hi), 0.0), fmax(-a.lo, a.hi)};
}
SDF_IA Ival min_(const Ival &a, const Ival &b) { return Ival{fmin(a.lo, b.lo), fmin(a.hi, b.hi)}; }
SDF_IA Ival max_(const Ival &a, const Ival &b) { return Ival{fmax(a.lo, b.lo), fmax(a.hi, b.hi)}; }
SDF_IA Ival maxc(const Ival &a, double c) { return Ival{fmax(a.lo, c), fmax(a.hi, c)}; }
SDF_IA Ival minc(const Ival &a, double c) { return Ival{fmin(a.lo, c), fmin(a.hi, c)}; }
SDF_IA Ival clip01(const Ival &a) { return Ival{fmin(fmax(a.lo, 0.0), 1.0), fmin(fmax(a.hi, 0.0), 1.0)}; }
// sdf_interp.h len2 / len3: sqrt(x*x + y*y), sqrt((x*x + y*y) + z*z)
SDF_IA Ival len2(const Ival &x, const Ival &y) { return sqrt_(add(sqr(x), sqr(y))); }
SDF_IA Ival len3(const Ival &x, const Ival &y, const Ival &z) { return sqrt_(add(add(sqr(x), sqr(y)), sqr(z))); }
// sdf_interp.h dot3 / dot2: fma(z, c, fma(y, b, x * a))
SDF_IA Ival dot3c(const Ival &x, const Ival &y, const Ival &z, double a, double b, double c) {
    return fmac(z, c, fmac(y, b, mulc(x, a)));
}
SDF_IA Ival dot2c(const Ival &x, const Ival &y, double a, double b) { return fmac(y, b, mulc(x, a)); }

// ---- ops that go through libm, or whose floating-point form is not monotone term by term ------
// hypot / atan2 / sin / cos are not correctly rounded on the device (ocml: a few ulp), so their
// interval forms are the exact range of the real function over the box, widened by a margin
// (1e-12 relative / absolute) that is four orders above any libm error and still far below anything
// that decides a group (the culling test asks for |bound| > 1e-30, the operand test for a strict
// gap).  Correctly rounded steps around them (a - delta, cos * d, x + t * c) stay exact corner forms.
SDF_IA bool finite_(const Ival &a) { return a.lo - a.lo == 0.0 && a.hi - a.hi == 0.0; }
SDF_IA Ival pad(const Ival &a, double rel, double abs_) {
    if (bad(a)) return top();
    const double m = rel * fmax(fabs(a.lo), fabs(a.hi)) + abs_;
    return wide(a.lo - m, a.hi + m);
}
// a * b for two intervals: the rounded product is monotone in each factor, so the corners span it
SDF_IA Ival mul(const Ival &a, const Ival &b) {
    const double p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    if (p0 != p0 || p1 != p1 || p2 != p2 || p3 != p3) return top();
    return wide(fmin(fmin(p0, p1), fmin(p2, p3)), fmax(fmax(p0, p1), fmax(p2, p3)));
}
SDF_IA Ival rint_(const Ival &a) { return wide(rint(a.lo), rint(a.hi)); }
SDF_IA Ival hull(const Ival &a, const Ival &b) { return (bad(a) || bad(b)) ? top() : Ival{fmin(a.lo, b.lo), fmax(a.hi, b.hi)}; }
// a / b for b > 0: the rounded quotient is monotone in each argument, so the corners span it
SDF_IA Ival div_pos(const Ival &a, const Ival &b) {
    if (bad(a) || bad(b) || !(b.lo > 0.0)) return top();
    const double q0 = a.lo / b.lo, q1 = a.lo / b.hi, q2 = a.hi / b.lo, q3 = a.hi / b.hi;
    if (q0 != q0 || q1 != q1 || q2 != q2 || q3 != q3) return top();
    return wide(fmin(fmin(q0, q1), fmin(q2, q3)), fmax(fmax(q0, q1), fmax(q2, q3)));
}
// np.sign over an interval: -1, 0 or 1 at each end (monotone)
SDF_IA Ival sign_(const Ival &a) {
    if (bad(a)) return Ival{-1.0, 1.0};
    return Ival{a.lo > 0.0 ? 1.0 : (a.lo < 0.0 ? -1.0 : 0.0), a.hi > 0.0 ? 1.0 : (a.hi < 0.0 ? -1.0 : 0.0)};
}
// p - clip(p, lo, hi): non-decreasing in p (p - lo below, 0 inside, p - hi above), one rounding per value
SDF_IA Ival sub_clip(const Ival &p, double lo, double hi) {
    if (bad(p) || lo != lo || hi != hi) return top();
    return wide(p.lo - fmin(fmax(p.lo, lo), hi), p.hi - fmin(fmax(p.hi, lo), hi));
}
// cond ? a : b where cond = (l > r): decided when the intervals do not overlap, else both are possible
SDF_IA Ival sel_gt(const Ival &l, const Ival &r, const Ival &a, const Ival &b) {
    if (bad(l) || bad(r)) return top();
    if (l.lo > r.hi) return a;
    if (l.hi <= r.lo) return b;
    return hull(a, b);
}
// s_clip(x, lo, hi) of sdf_interp.h for non-NaN x: min(max(x, lo), hi), monotone in x
SDF_IA Ival clipc(const Ival &a, double lo, double hi) {
    if (lo != lo || hi != hi) return top();
    return wide(fmin(fmax(a.lo, lo), hi), fmin(fmax(a.hi, lo), hi));
}
// does [l, h] (already widened by the caller) contain p + k * period for an integer k?
SDF_IA bool hits(double l, double h, double p, double period) { return ceil((l - p) / period) <= floor((h - p) / period); }
// ranges of sin and cos over an angle interval
SDF_IA void sincos_range(const Ival &ang, Ival &sn, Ival &cs) {
    const double two_pi = 6.283185307179586, pi = 3.141592653589793;
    sn = cs = Ival{-1.0, 1.0};
    if (!finite_(ang) || !(ang.hi - ang.lo < 6.0)) { sn = cs = pad(sn, 0.0, 1e-12); return; }
    double sl, cl, sh, ch;
    ia_sincos(ang.lo, &sl, &cl);
    ia_sincos(ang.hi, &sh, &ch);
    sn = Ival{fmin(sl, sh), fmax(sl, sh)};
    cs = Ival{fmin(cl, ch), fmax(cl, ch)};
    const double l = ang.lo - 1e-9, h = ang.hi + 1e-9;
    if (hits(l, h, 0.0, two_pi)) cs.hi = 1.0;
    if (hits(l, h, pi, two_pi)) cs.lo = -1.0;
    if (hits(l, h, 0.5 * pi, two_pi)) sn.hi = 1.0;
    if (hits(l, h, -0.5 * pi, two_pi)) sn.lo = -1.0;
    sn = pad(sn, 0.0, 1e-12);
    cs = pad(cs, 0.0, 1e-12);
}
// circular_array, first half (sdf_interp.h L_CIRC_PREP, d3.py:379-392): d = hypot(x, y),
// a = atan2(y, x) mod da (Python's floored modulo, so a lies in [0, da])
SDF_IA void circ_prep(const Ival &x, const Ival &y, double da, Ival &d, Ival &a) {
    d = a = top();
    if (!finite_(x) || !finite_(y) || !(da > 0.0) || !(da < 7.0)) return;
    d = pad(len2(x, y), 1e-12, 1e-300);
    if (bad(d)) { d = top(); return; }
    d.lo = fmax(d.lo, 0.0);
    a = Ival{0.0, da};
    // a box that touches the origin or the negative x axis (-0.0 counts) sees the jump of atan2
    if (x.lo <= 0.0 && y.lo <= 0.0 && y.hi >= 0.0) return;
    // elsewhere the angle has no stationary point and is monotone along every edge of the box
    const double t0 = ia_atan2(y.lo, x.lo), t1 = ia_atan2(y.lo, x.hi), t2 = ia_atan2(y.hi, x.lo), t3 = ia_atan2(y.hi, x.hi);
    const double tl = fmin(fmin(t0, t1), fmin(t2, t3)) - 1e-12, th = fmax(fmax(t0, t1), fmax(t2, t3)) + 1e-12;
    if (!(tl <= th)) return;
    const double kl = floor(tl / da), kh = floor(th / da);
    if (kl != kh) return;                                   // the box straddles a sector boundary
    const double lo = tl - kl * da, hi = th - kl * da;
    if (!(lo > 1e-9) || !(da - hi > 1e-9)) return;          // (too close to a boundary to trust kl)
    a = Ival{lo - 1e-12, hi + 1e-12};
}
// range of atan2(y, x) over a box: [-pi, pi] when the box touches the origin or the negative x axis, else the
// corners' (see circ_prep), widened
SDF_IA Ival atan2_range(const Ival &x, const Ival &y) {
    const double pi = 3.141592653589793;
    if (!finite_(x) || !finite_(y)) return top();
    if (x.lo <= 0.0 && y.lo <= 0.0 && y.hi >= 0.0) return Ival{-pi, pi};
    const double t0 = ia_atan2(y.lo, x.lo), t1 = ia_atan2(y.lo, x.hi), t2 = ia_atan2(y.hi, x.lo), t3 = ia_atan2(y.hi, x.hi);
    const double tl = fmin(fmin(t0, t1), fmin(t2, t3)) - 1e-12, th = fmax(fmax(t0, t1), fmax(t2, t3)) + 1e-12;
    if (!(tl <= th)) return Ival{-pi, pi};
    return Ival{fmax(tl, -pi), fmin(th, pi)};
}
// second half (L_CIRC_SET): p = (cos(a - delta) * d, sin(a - delta) * d, z)
SDF_IA void circ_set(const Ival &d, const Ival &a, double delta, Ival &x, Ival &y) {
    x = y = top();
    if (!finite_(d) || !finite_(a) || delta != delta) return;
    Ival sn, cs;
    sincos_range(subc(a, delta), sn, cs);
    x = mul(cs, d);
    y = mul(sn, d);
}
// easing curves that are non-decreasing on [0, 1] and need no trigonometry (sdf_interp.h ease_apply,
// ease.py): values at the end points, widened (the floating-point forms are not monotone to the ulp)
SDF_IA double ease_value(int id, double t, bool &known) {
    double u, v;
    known = true;
    switch (id) {
    case EASE_linear: return t;
    case EASE_in_quad: return t * t;
    case EASE_out_quad: return -t * (t - 2.0);
    case EASE_in_out_quad: u = 2.0 * t - 1.0; return t < 0.5 ? 2.0 * t * t : -0.5 * (u * (u - 2.0) - 1.0);
    case EASE_in_cubic: return t * t * t;
    case EASE_out_cubic: u = t - 1.0; return u * u * u + 1.0;
    case EASE_in_out_cubic: u = t * 2.0; v = u - 2.0; return u < 1.0 ? 0.5 * u * u * u : 0.5 * (v * v * v + 2.0);
    case EASE_in_quart: return t * t * t * t;
    case EASE_out_quart: u = t - 1.0; return -(u * u * u * u - 1.0);
    case EASE_in_out_quart: u = t * 2.0; v = u - 2.0; return u < 1.0 ? 0.5 * u * u * u * u : -0.5 * (v * v * v * v - 2.0);
    case EASE_in_quint: return t * t * t * t * t;
    case EASE_out_quint: u = t - 1.0; return u * u * u * u * u + 1.0;
    case EASE_in_out_quint: u = t * 2.0; v = u - 2.0; return u < 1.0 ? 0.5 * u * u * u * u * u : 0.5 * (v * v * v * v * v + 2.0);
    case EASE_in_circ: return -1.0 * (sqrt(fmax(1.0 - t * t, 0.0)) - 1.0);
    case EASE_out_circ: u = t - 1.0; return sqrt(fmax(1.0 - u * u, 0.0));
    case EASE_in_out_circ:
        u = t * 2.0; v = u - 2.0;
        return u < 1.0 ? -0.5 * (sqrt(fmax(1.0 - u * u, 0.0)) - 1.0) : 0.5 * (sqrt(fmax(1.0 - v * v, 0.0)) + 1.0);
    case EASE_in_square: return t < 1.0 ? 0.0 : 1.0;
    case EASE_out_square: return t > 0.0 ? 1.0 : 0.0;
    case EASE_in_out_square: return t < 0.5 ? 0.0 : 1.0;
    default: known = false; return 0.0;
    }
}
SDF_IA Ival ease01(int id, const Ival &t) {
    if (bad(t) || t.lo < 0.0 || t.hi > 1.0) return top();
    bool k0, k1;
    const double a = ease_value(id, t.lo, k0), b = ease_value(id, t.hi, k1);
    if (!k0 || !k1 || a != a || b != b) return top();
    return Ival{fmin(a, b) - 1e-12, fmax(a, b) + 1e-12};
}

}  // namespace ia

// Machine state of the interval run: the current point and distance live in registers, the saved
// points / distances in (dynamic) LDS, [slot][component][thread] -- a run-time slot number would put
// a per-thread array into scratch memory.  The host sizes it for the slots the tape uses.
enum { PRUNE_BLOCK = 256 };
struct IaShared {
    double *base;
    int n_p;
    int stride;     // threads that share the array
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ __forceinline__ static int lane_slot() { return (int)threadIdx.x; }
#elif defined(SDF_HOST_SIMT)
    static int lane_slot() { return sdf_host_simt_tid(); }   // (host threads playing a workgroup: tests/native/cull_tasks_host.py)
#else
    static int lane_slot() { return 0; }      // (host build of the interval run: one box at a time, tests/native/interval_tape_host.hip)
#endif
    SDF_IA double &ps(uint32_t slot, int k) { return base[(slot * 6 + k) * stride + lane_slot()]; }
    SDF_IA double &ds(uint32_t slot, int k) { return base[(n_p * 6 + slot * 2 + k) * stride + lane_slot()]; }
};
__host__ __device__ inline size_t prune_lds_bytes(int n_p, int n_d) { return (size_t)(6 * n_p + 2 * n_d) * PRUNE_BLOCK * 8; }


// ops with an interval form below (ia_leaf / ia_run_tape); a tape made of these only is worth the
// cell-group pass of k_cull, any other op turns everything behind it into the whole line
__host__ __device__ inline bool ia_has_form(uint32_t op) {
    switch (op) {
    case OP_END: case OP_L_SPHERE: case OP_L_PLANE: case OP_L_BOX: case OP_L_ROUNDED_BOX: case OP_L_TORUS: case OP_L_CYLINDER:
    case OP_L_ROUNDED_CYLINDER: case OP_L_CAPSULE: case OP_L_OCTAHEDRON: case OP_L_CIRCLE: case OP_L_LINE: case OP_L_RECTANGLE:
    case OP_L_WIREFRAME_BOX: case OP_L_CAPPED_CYLINDER: case OP_L_ROUNDED_CONE: case OP_L_ELLIPSOID: case OP_L_TETRAHEDRON:
    case OP_L_DODECAHEDRON: case OP_L_ICOSAHEDRON: case OP_L_ROUNDED_RECTANGLE: case OP_L_EQUILATERAL_TRIANGLE: case OP_L_HEXAGON:
    case OP_L_ROUNDED_X: case OP_L_VESICA: case OP_L_CAPPED_CONE: case OP_L_PYRAMID: case OP_L_POLYGON: case OP_L_TEXTURE2D:
    case OP_L_GRID3D:
    case OP_COMB: case OP_TRANSLATE: case OP_SCALE: case OP_ROTATE: case OP_ELONGATE: case OP_TRANSLATE2: case OP_SCALE2:
    case OP_ROTATE2: case OP_ELONGATE2: case OP_REVOLVE: case OP_SETZ0: case OP_SAVE_P: case OP_LOAD_P: case OP_PUSH_D: case OP_NOP:
    case OP_NEG: case OP_ADDC: case OP_SUBC: case OP_MULC: case OP_SHELL: case OP_ADD_DS: case OP_EXT_PRE: case OP_EXT_POST:
    case OP_REP_PREP: case OP_REP_SET: case OP_CIRC_PREP: case OP_CIRC_SET: case OP_BEND_LINEAR:
    case OP_TWIST: case OP_BEND: case OP_BEND_RADIAL: case OP_TRANS_LIN_PRE: case OP_TRANS_RAD_PRE: case OP_TRANS_MIX:
    case OP_EXTTO_PRE: case OP_EXTTO_MIX: case OP_SLICE_POST: case OP_WRAP_AROUND:
        return true;
    default: return false;
    }
}

SDF_IA Ival ia_box_like(const Ival &qx, const Ival &qy, const Ival &qz) {
    using namespace ia;
    const Ival mx = max_(max_(qx, qy), qz);
    return add(len3(maxc(qx, 0.0), maxc(qy, 0.0), maxc(qz, 0.0)), minc(mx, 0.0));
}

// value interval of a leaf (the formulas of sdf_interp.h, operation by operation), or the whole
// line for leaves without an interval form
SDF_IA Ival ia_leaf(uint32_t op, const double *c, const Ival &x, const Ival &y, const Ival &z) {
    using namespace ia;
    switch (op) {
    case OP_L_SPHERE: return subc(len3(subc(x, c[1]), subc(y, c[2]), subc(z, c[3])), c[0]);
    case OP_L_PLANE: return dot3c(csub(c[3], x), csub(c[4], y), csub(c[5], z), c[0], c[1], c[2]);
    case OP_L_BOX:
        return ia_box_like(subc(abs_(subc(x, c[0])), c[3]), subc(abs_(subc(y, c[1])), c[4]), subc(abs_(subc(z, c[2])), c[5]));
    case OP_L_ROUNDED_BOX:
        return subc(ia_box_like(addc(subc(abs_(x), c[0]), c[3]), addc(subc(abs_(y), c[1]), c[3]), addc(subc(abs_(z), c[2]), c[3])), c[3]);
    case OP_L_TORUS: return subc(len2(subc(len2(x, y), c[0]), z), c[1]);
    case OP_L_CYLINDER: return subc(len2(x, y), c[0]);
    case OP_L_ROUNDED_CYLINDER: {
        const Ival d0 = addc(subc(len2(x, y), c[0]), c[1]), d1 = addc(subc(abs_(z), c[2]), c[1]);
        return subc(add(minc(max_(d0, d1), 0.0), len2(maxc(d0, 0.0), maxc(d1, 0.0))), c[1]);
    }
    case OP_L_CAPSULE: {
        const Ival pax = subc(x, c[0]), pay = subc(y, c[1]), paz = subc(z, c[2]);
        const Ival h = clip01(divc(dot3c(pax, pay, paz, c[3], c[4], c[5]), c[6]));
        return subc(len3(sub(pax, mulc(h, c[3])), sub(pay, mulc(h, c[4])), sub(paz, mulc(h, c[5]))), c[7]);
    }
    case OP_L_OCTAHEDRON: return mulc(subc(add(add(abs_(x), abs_(y)), abs_(z)), c[0]), c[1]);
    case OP_L_CIRCLE: return subc(len2(subc(x, c[1]), subc(y, c[2])), c[0]);
    case OP_L_LINE: return dot2c(csub(c[2], x), csub(c[3], y), c[0], c[1]);
    case OP_L_RECTANGLE: {
        const Ival qx = subc(abs_(subc(x, c[0])), c[2]), qy = subc(abs_(subc(y, c[1])), c[3]);
        return add(len2(maxc(qx, 0.0), maxc(qy, 0.0)), minc(max_(qx, qy), 0.0));
    }
    default: return top();

    }
}

// The less common leaves, in a function of their own: a kernel's register budget is that of its
// hungriest callee, and these would cost the cell-group pass (k_cull) a wave of occupancy on every tape.
// Kernels are instantiated with and without them (RARE); the host picks by the tape's content.
SDF_IA bool ia_is_rare_leaf(uint32_t op) {
    switch (op) {
    case OP_L_WIREFRAME_BOX: case OP_L_CAPPED_CYLINDER: case OP_L_ROUNDED_CONE: case OP_L_ELLIPSOID: case OP_L_TETRAHEDRON:
    case OP_L_DODECAHEDRON: case OP_L_ICOSAHEDRON: case OP_L_ROUNDED_RECTANGLE: case OP_L_EQUILATERAL_TRIANGLE: case OP_L_HEXAGON:
    case OP_L_ROUNDED_X: case OP_L_VESICA: case OP_L_CAPPED_CONE: case OP_L_PYRAMID: case OP_L_POLYGON: case OP_L_TEXTURE2D:
    case OP_L_GRID3D:
        return true;
    default: return false;
    }
}
__host__ __device__ __attribute__((noinline)) inline Ival ia_leaf_rare(uint32_t op, const double *c, const Ival &x, const Ival &y, const Ival &z) {
    using namespace ia;
    switch (op) {
    // The leaves below are compositions of the interval operations above, step by step in the order of
    // sdf_interp.h: every step encloses the rounded result of its operation over the step's argument
    // intervals, so the composition encloses the leaf (dependencies between sub-expressions only cost
    // tightness).  A data-dependent choice (vsel) is decided when the intervals of its condition do not
    // overlap and is the hull of both arms otherwise.
    case OP_L_WIREFRAME_BOX: {
        const double t2 = c[3];
        const Ival px = subc(subc(abs_(x), c[0]), t2), py = subc(subc(abs_(y), c[1]), t2), pz = subc(subc(abs_(z), c[2]), t2);
        const Ival qx = subc(abs_(addc(px, t2)), t2), qy = subc(abs_(addc(py, t2)), t2), qz = subc(abs_(addc(pz, t2)), t2);
        const Ival g0 = add(len3(maxc(px, 0.0), maxc(qy, 0.0), maxc(qz, 0.0)), minc(max_(px, max_(qy, qz)), 0.0));
        const Ival g1 = add(len3(maxc(qx, 0.0), maxc(py, 0.0), maxc(qz, 0.0)), minc(max_(qx, max_(py, qz)), 0.0));
        const Ival g2 = add(len3(maxc(qx, 0.0), maxc(qy, 0.0), maxc(pz, 0.0)), minc(max_(qx, max_(qy, pz)), 0.0));
        return min_(min_(g0, g1), g2);
    }
    case OP_L_CAPPED_CYLINDER: {
        const double bax = c[3], bay = c[4], baz = c[5], baba = c[6];
        if (!(baba > 0.0)) return top();
        const Ival pax = subc(x, c[0]), pay = subc(y, c[1]), paz = subc(z, c[2]);
        const Ival paba = dot3c(pax, pay, paz, bax, bay, baz);
        const Ival xx = subc(len3(sub(mulc(pax, baba), mulc(paba, bax)), sub(mulc(pay, baba), mulc(paba, bay)),
                                  sub(mulc(paz, baba), mulc(paba, baz))), c[8]);
        const Ival yy = subc(abs_(subc(paba, c[9])), c[9]);
        const Ival x2 = sqr(xx), y2 = mulc(sqr(yy), baba);
        const Ival zero = pt(0.0);
        const Ival din = neg(min_(x2, y2));
        const Ival dout = add(sel_gt(xx, zero, x2, zero), sel_gt(yy, zero, y2, zero));
        const Ival d = sel_gt(zero, max_(xx, yy), din, dout);          // max(xx, yy) < 0
        if (bad(d)) return top();
        // sign(d) * sqrt(|d|) / baba is non-decreasing in d
        const double lo = (d.lo > 0.0 ? 1.0 : (d.lo < 0.0 ? -1.0 : 0.0)) * sqrt(fabs(d.lo)) / baba;
        const double hi = (d.hi > 0.0 ? 1.0 : (d.hi < 0.0 ? -1.0 : 0.0)) * sqrt(fabs(d.hi)) / baba;
        return wide(lo, hi);
    }
    case OP_L_ROUNDED_CONE: {
        const double r1 = c[0], r2 = c[1], h = c[2], b = c[3], a = c[4], ah = c[5];
        const Ival qx = len2(x, y), qy = z;
        const Ival k = dot2c(qx, qy, -b, a);
        const Ival c1 = subc(len2(qx, qy), r1), c2 = subc(len2(qx, subc(qy, h)), r2), c3 = subc(dot2c(qx, qy, a, b), r1);
        if (bad(k) || ah != ah) return top();
        Ival r{0.0, 0.0};                                               // the hull of the arms k can reach
        bool any = false;
        if (k.lo < 0.0) { r = c1; any = true; }
        if (k.hi >= 0.0) {
            if (k.hi > ah) { r = any ? hull(r, c2) : c2; any = true; }
            if (fmax(k.lo, 0.0) <= ah) { r = any ? hull(r, c3) : c3; any = true; }
        }
        return any ? fix(r) : top();
    }
    case OP_L_ELLIPSOID: {
        const Ival k0 = len3(divc(x, c[0]), divc(y, c[1]), divc(z, c[2]));
        const Ival k1 = len3(divc(x, c[3]), divc(y, c[4]), divc(z, c[5]));
        return div_pos(mul(k0, subc(k0, 1.0)), k1);
    }
    case OP_L_TETRAHEDRON:
        return divc(subc(max_(sub(abs_(add(x, y)), z), add(abs_(sub(x, y)), z)), c[0]), c[1]);
    case OP_L_DODECAHEDRON: case OP_L_ICOSAHEDRON: {
        const double r = c[0], X = c[1], Y = c[2], Z = c[3];
        const Ival ax = abs_(divc(x, r)), ay = abs_(divc(y, r)), az = abs_(divc(z, r));
        const Ival a = dot3c(ax, ay, az, X, Y, Z), b = dot3c(ax, ay, az, Z, X, Y), cc = dot3c(ax, ay, az, Y, Z, X);
        const Ival m = subc(max_(max_(a, b), cc), X);
        if (op == OP_L_DODECAHEDRON) return mulc(m, r);
        return mulc(max_(m, subc(dot3c(ax, ay, az, c[4], c[4], c[4]), X)), r);
    }
    case OP_L_ROUNDED_RECTANGLE: {
        if (bad(x) || bad(y)) return top();
        // the corner radius of the quadrant the point is in: x > 0 / y > 0 decide it (d2.py:116-134)
        const bool xt = x.hi > 0.0, xf = !(x.lo > 0.0), yt = y.hi > 0.0, yf = !(y.lo > 0.0);
        double rl = __builtin_inf(), rh = -__builtin_inf();
        if (xt && yt) { rl = fmin(rl, c[2]); rh = fmax(rh, c[2]); }
        if (xt && yf) { rl = fmin(rl, c[3]); rh = fmax(rh, c[3]); }
        if (xf && yf) { rl = fmin(rl, c[4]); rh = fmax(rh, c[4]); }
        if (xf && yt) { rl = fmin(rl, c[5]); rh = fmax(rh, c[5]); }
        if (!(rl <= rh)) return top();
        const Ival r{rl, rh};
        const Ival qx = add(subc(abs_(x), c[0]), r), qy = add(subc(abs_(y), c[1]), r);
        return sub(add(minc(max_(qx, qy), 0.0), len2(maxc(qx, 0.0), maxc(qy, 0.0))), r);
    }
    case OP_L_EQUILATERAL_TRIANGLE: {
        const double k = c[0];
        Ival px = subc(abs_(x), 1.0), py = addc(y, c[1]);
        const Ival w = add(px, mulc(py, k));
        const Ival nx = divc(sub(px, mulc(py, k)), 2.0), ny = divc(sub(mulc(px, -k), py), 2.0);
        const Ival zero = pt(0.0);
        const Ival sx = sel_gt(w, zero, nx, px), sy = sel_gt(w, zero, ny, py);
        px = sub_clip(sx, -2.0, 0.0); py = sy;
        return mul(neg(len2(px, py)), sign_(py));
    }
    case OP_L_HEXAGON: {
        const double r = c[0], k0 = c[1], k1 = c[2];
        Ival px = abs_(x), py = abs_(y);
        const Ival m = minc(add(mulc(px, k0), mulc(py, k1)), 0.0);
        px = sub(px, mulc(m, c[4])); py = sub(py, mulc(m, c[5]));
        px = sub_clip(px, c[6], c[7]); py = subc(py, 0.0 + r);
        return mul(len2(px, py), sign_(py));
    }
    case OP_L_ROUNDED_X: {
        const Ival px = abs_(x), py = abs_(y);
        const Ival qq = mulc(minc(add(px, py), c[0]), 0.5);
        return subc(len2(sub(px, qq), sub(py, qq)), c[1]);
    }
    case OP_L_VESICA: {
        const double r = c[0], d = c[1], b = c[2];
        const Ival px = abs_(x), py = abs_(y);
        const Ival v1 = len2(px, subc(py, b)), v2 = subc(len2(subc(px, -d), py), r);
        return sel_gt(mulc(subc(py, b), d), mulc(px, b), v1, v2);
    }
    case OP_L_CAPPED_CONE: {   // d3.py:217-237
        const double bax = c[3], bay = c[4], baz = c[5], ra = c[6], rb = c[7], baba = c[8], rba = c[9], k = c[10];
        if (!(baba > 0.0)) return top();
        const Ival pax = subc(x, c[0]), pay = subc(y, c[1]), paz = subc(z, c[2]);
        const Ival papa = add(add(sqr(pax), sqr(pay)), sqr(paz));
        const Ival paba = divc(dot3c(pax, pay, paz, bax, bay, baz), baba);
        // xx = sqrt(papa - paba * paba * baba) cancels: its argument is the squared distance from the axis,
        // |pa x ba|^2 / baba, which the cross product bounds without cancellation; the interpreter's value differs
        // from that real number by at most a few roundings of papa.  Too close to the axis to keep the computed
        // argument positive: no statement.
        const Ival crx = sub(mulc(pay, baz), mulc(paz, bay)), cry = sub(mulc(paz, bax), mulc(pax, baz)), crz = sub(mulc(pax, bay), mulc(pay, bax));
        const Ival A = divc(add(add(sqr(crx), sqr(cry)), sqr(crz)), baba);
        if (bad(A) || bad(papa) || !finite_(papa) || !finite_(A)) return top();
        const double delta = 1e-13 * papa.hi;
        if (!(A.lo - delta > 0.0)) return top();
        const Ival xx{sqrt(A.lo - delta), sqrt(A.hi + delta)};
        const Ival half = pt(0.5), zero = pt(0.0);
        const Ival cax = maxc(sub(xx, sel_gt(half, paba, pt(ra), pt(rb))), 0.0);          // paba < 0.5 ? ra : rb
        const Ival cay = subc(abs_(subc(paba, 0.5)), 0.5);
        const Ival f = clip01(divc(add(mulc(subc(xx, ra), rba), mulc(paba, baba)), k));
        const Ival cbx = sub(subc(xx, ra), mulc(f, rba)), cby = sub(paba, f);
        const Ival m = min_(add(sqr(cax), mulc(sqr(cay), baba)), add(sqr(cbx), mulc(sqr(cby), baba)));
        const Ival r = sqrt_(m);
        if (bad(cbx) || bad(cay) || bad(r)) return top();
        if (cbx.hi < 0.0 && cay.hi < 0.0) return neg(r);
        if (cbx.lo >= 0.0 || cay.lo >= 0.0) return r;
        return Ival{-r.hi, r.hi};
    }
    case OP_L_PYRAMID: {       // d3.py:261-282
        const double h = c[0], m2 = c[1], m2q = c[2];
        if (!(m2 > 0.0)) return top();
        const Ival b0 = subc(abs_(x), 0.5), b1 = subc(abs_(y), 0.5);
        const Ival px = max_(b0, b1), py = z, pz = min_(b0, b1);                            // (the swap picks the larger one)
        const Ival qx = pz, qy = sub(mulc(py, h), mulc(px, 0.5)), qz = add(mulc(px, h), mulc(py, 0.5));
        const Ival sv = maxc(neg(qx), 0.0);
        const Ival tt = clip01(divc(sub(qy, mulc(pz, 0.5)), m2q));
        const Ival a = add(mulc(sqr(add(qx, sv)), m2), sqr(qy));
        const Ival b = add(mulc(sqr(add(qx, mulc(tt, 0.5))), m2), sqr(sub(qy, mulc(tt, m2))));
        const Ival zero = pt(0.0);
        const Ival dd2 = sel_gt(min_(qy, sub(mulc(neg(qx), m2), mulc(qy, 0.5))), zero, zero, min_(a, b));
        return mul(sqrt_(divc(add(dd2, sqr(qz)), m2)), sign_(max_(qz, neg(py))));
    }
    case OP_L_TEXTURE2D: {     // text.py:116-153: bilinear look-up inside the picture, a rectangle's distance outside
        // c: x0 y0 x1 y1 | pw ph px py | tw th | fallback rectangle (cx cy hx hy) | texture[th][tw]
        const int tw = (int)c[8], th = (int)c[9];
        const double *tex = c + 14;
        if (tw < 2 || th < 2 || bad(x) || bad(y)) return top();
        const Ival u = divc(subc(x, c[0]), c[2] - c[0]);
        const Ival vv = csub(1.0, divc(subc(y, c[1]), c[3] - c[1]));
        const Ival ti = addc(mulc(u, c[4]), c[6]), tj = addc(mulc(vv, c[5]), c[7]);
        const Ival qx = subc(abs_(subc(x, c[10])), c[12]), qy = subc(abs_(subc(y, c[11])), c[13]);
        const Ival q = add(len2(maxc(qx, 0.0), maxc(qy, 0.0)), minc(max_(qx, qy), 0.0));
        if (bad(ti) || bad(tj) || !finite_(ti) || !finite_(tj)) return top();
        const double wi = (double)(tw - 1), hj = (double)(th - 1);
        if (ti.hi < 0.0 || ti.lo >= wi || tj.hi < 0.0 || tj.lo >= hj) return q;          // outside for every point of the box
        // inside, the four weights lie in [0, 1] and add up to 1: the value is a convex combination of the four
        // texels around the point (to a few roundings); over the box: of the texels the box can reach
        const int i0 = (int)floor(fmax(ti.lo, 0.0)), i1 = (int)fmin(floor(fmin(ti.hi, wi)) + 1.0, wi);
        const int j0 = (int)floor(fmax(tj.lo, 0.0)), j1 = (int)fmin(floor(fmin(tj.hi, hj)) + 1.0, hj);
        if ((long long)(i1 - i0 + 1) * (long long)(j1 - j0 + 1) > 1024) return top();     // (a box that large decides nothing anyway)
        double lo = __builtin_inf(), hi = -__builtin_inf();
        for (int j = j0; j <= j1; j++)
            for (int i = i0; i <= i1; i++) { const double p = tex[(size_t)j * tw + i]; lo = fmin(lo, p); hi = fmax(hi, p); if (p != p) return top(); }
        if (!(lo <= hi)) return top();
        const Ival d = pad(Ival{lo, hi}, 1e-12, 1e-300);
        const bool all_inside = ti.lo >= 0.0 && ti.hi < wi && tj.lo >= 0.0 && tj.hi < hj;
        return all_inside ? d : hull(d, q);
    }
    case OP_L_GRID3D: {        // mesh.py:96-105: the box estimator beyond `background`, else the trilinear look-up
        // c: nx ny nz | background | box centre (3) | box half size (3) | X[nx] Y[ny] Z[nz] | A[nx][ny][nz]
        const int n[3] = {(int)c[0], (int)c[1], (int)c[2]};
        const double bg = c[3];
        if (n[0] < 2 || n[1] < 2 || n[2] < 2 || bad(x) || bad(y) || bad(z) || !(bg == bg)) return top();
        const Ival e = ia_box_like(subc(abs_(subc(x, c[4])), c[7]), subc(abs_(subc(y, c[5])), c[8]), subc(abs_(subc(z, c[6])), c[9]));
        if (bad(e)) return top();
        if (e.lo > bg) return e;                              // the estimator wins at every point of the box
        // the voxels a point of the box can interpolate between: per axis, cells searchsorted(lo) - 1 .. searchsorted(hi) - 1
        const double *g[3] = {c + 10, c + 10 + n[0], c + 10 + n[0] + n[1]};
        const double *vox = c + 10 + n[0] + n[1] + n[2];
        const Ival *pv[3] = {&x, &y, &z};
        int i0[3], i1[3];
        bool oob = false;
        for (int a = 0; a < 3; a++) {
            const double plo = pv[a]->lo, phi = pv[a]->hi;
            int lo = 0, hi = n[a];
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (g[a][mid] < plo) lo = mid + 1; else hi = mid; }
            i0[a] = lo - 1 < 0 ? 0 : (lo - 1 > n[a] - 2 ? n[a] - 2 : lo - 1);
            lo = 0; hi = n[a];
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (g[a][mid] < phi) lo = mid + 1; else hi = mid; }
            i1[a] = (lo - 1 < 0 ? 0 : (lo - 1 > n[a] - 2 ? n[a] - 2 : lo - 1)) + 1;
            oob = oob || plo < g[a][0] || phi > g[a][n[a] - 1];
        }
        if ((long long)(i1[0] - i0[0] + 1) * (long long)(i1[1] - i0[1] + 1) * (long long)(i1[2] - i0[2] + 1) > 4096) return top();
        double lo = __builtin_inf(), hi = -__builtin_inf();
        for (int i = i0[0]; i <= i1[0]; i++)
            for (int j = i0[1]; j <= i1[1]; j++)
                for (int k = i0[2]; k <= i1[2]; k++) {
                    const double p = vox[((size_t)i * n[1] + j) * n[2] + k];
                    if (p != p) return top();
                    lo = fmin(lo, p); hi = fmax(hi, p);
                }
        if (!(lo <= hi)) return top();
        // inside the grid the eight weights lie in [0, 1] and add up to 1 (to a few roundings): a convex combination
        // of the reachable voxels; outside it the fill value
        Ival d = pad(Ival{lo, hi}, 1e-12, 1e-300);
        if (oob) d = hull(d, pt(bg));
        return e.hi <= bg ? d : hull(e, d);
    }
    case OP_L_POLYGON: {       // d2.py:175-196
        const int np_ = (int)c[0];
        const double *pv = c + 1;
        if (np_ < 1 || bad(x) || bad(y) || !finite_(x) || !finite_(y)) return top();
        Ival d = add(sqr(subc(x, pv[0])), sqr(subc(y, pv[1])));
        double sgn = 1.0;                                    // the interpreter's winding sign at the corner (x.lo, y.lo)
        double scale2 = x.hi * x.hi + y.hi * y.hi + x.lo * x.lo + y.lo * y.lo;
        for (int i = 0; i < np_; i++) {
            const int j = (i + np_ - 1) % np_;
            const double vix = pv[2 * i], viy = pv[2 * i + 1], vjx = pv[2 * j], vjy = pv[2 * j + 1];
            const double ex = vjx - vix, ey = vjy - viy;
            const Ival wx = subc(x, vix), wy = subc(y, viy);
            const double ee = fma(ey, ey, ex * ex);
            const Ival cl = clip01(divc(dot2c(wx, wy, ex, ey), ee));
            const Ival bx = sub(wx, mulc(cl, ex)), by = sub(wy, mulc(cl, ey));
            d = min_(d, add(sqr(bx), sqr(by)));
            const double pwx = x.lo - vix, pwy = y.lo - viy;
            const bool c1 = y.lo >= viy, c2 = y.lo < vjy, c3 = ex * pwy > ey * pwx;
            if ((c1 && c2 && c3) || (!c1 && !c2 && !c3)) sgn = -sgn;
            scale2 += vix * vix + viy * viy;
        }
        const Ival r = sqrt_(d);
        if (bad(d) || bad(r) || !(scale2 == scale2)) return top();
        // The winding sign only changes across the polygon's boundary (the y-range tests of neighbouring edges are
        // exact complements; the side test of an edge only counts inside its y-range, i.e. at the edge itself): a box
        // that stays clear of the boundary by more than the rounding of those tests has one sign, the corner's.
        if (d.lo > 1e-24 * (1.0 + scale2)) return sgn > 0.0 ? r : neg(r);
        return Ival{-r.hi, r.hi};
    }
    default: return top();
    }
}

// interval of post(d1, d2) (sdf_interp.h post_combine)
SDF_IA Ival ia_post(uint32_t post, const Ival &d1, const Ival &d2, double K) {
    using namespace ia;
    switch (post) {
    case POST_SET: return d2;
    case POST_UNION: return min_(d1, d2);
    case POST_DIFF: return max_(d1, neg(d2));
    case POST_INTER: return max_(d1, d2);
    case POST_BLEND: return add(mulc(d2, K), mulc(d1, 1.0 - K));
    default: break;
    }
    // polynomial smooth min / max (dn.py:7-50), for K > 0: with e = the second operand (negated for a
    // difference) and t = (e - d1) / K, the blend weight is h = clip((1 + t) / 2) and the result is
    //     min(d1, e) - (K / 4) (1 - |t|)^2   resp.   max(d1, e) + (K / 4) (1 - |t|)^2      for |t| <= 1,
    // the hard result beyond.  The correction falls with |t|, so the range of |e - d1| over the box
    // bounds it from both sides; the margin is far above the rounding of the few operations.
    if (!(K > 0.0) || bad(d1) || bad(d2)) return top();
    const Ival e = post == POST_SDIFF ? neg(d2) : d2;
    const Ival ad = abs_(sub(e, d1));
    const double tmin = fmin(ad.lo / K, 1.0), tmax = fmin(ad.hi / K, 1.0);
    const double cmax = 0.25 * K * (1.0 - tmin) * (1.0 - tmin), cmin = 0.25 * K * (1.0 - tmax) * (1.0 - tmax);
    Ival r = post == POST_SUNION ? min_(d1, e) : max_(d1, e);
    if (post == POST_SUNION) { r.lo -= cmax; r.hi -= cmin; } else { r.lo += cmin; r.hi += cmax; }
    const double m = 1e-9 * (fabs(r.lo) + fabs(r.hi) + K);
    return fix(Ival{r.lo - m, r.hi + m});
}

// 0 keep both, 1 the right operand never wins, 2 the left operand never wins.
// The polynomial smooth forms (sdf_interp.h post_combine) have ONE exact direction each: when the
// blend weight h clips to 0 over the whole box,
//     smooth union / intersection:  m = d2 + (d1 - d2) * 0 = d2 + (+-0),  result = m -+ K * 0 * 1  ->  d2 + (+0.0)
//     smooth difference:            m = d1 + (-d2 - d1) * 0 = d1 + (-0),  result = m + K * 0 * 1   ->  d1 + (+0.0)
// bit for bit (the other direction, h = 1, gives d2 + (d1 - d2), which is not d1 in floating point).  h is
// evaluated here with the interpreter's own operations on the end points (all monotone), so "h == 0
// everywhere" is exact; compact_tape then replaces the combine by `+ (+0.0)` on the surviving operand.
SDF_IA int ia_decide(uint32_t post, const Ival &left, const Ival &right, double K) {
    using namespace ia;
    switch (post) {
    case POST_UNION: return right.lo > left.hi ? 1 : (left.lo > right.hi ? 2 : 0);
    case POST_INTER: return right.hi < left.lo ? 1 : (left.hi < right.lo ? 2 : 0);
    case POST_DIFF: return -right.lo < left.lo ? 1 : (left.hi < -right.hi ? 2 : 0);       // max(left, -right)
    case POST_SUNION: case POST_SINTER: case POST_SDIFF: {
        if (!(K > 0.0) || !finite_(left) || !finite_(right)) return 0;
        Ival e;
        if (post == POST_SUNION) e = addc(divc(mulc(sub(right, left), 0.5), K), 0.5);       // 0.5 + 0.5 * (d2 - d1) / K
        else if (post == POST_SINTER) e = csub(0.5, divc(mulc(sub(right, left), 0.5), K));  // 0.5 - 0.5 * (d2 - d1) / K
        else e = csub(0.5, divc(mulc(add(right, left), 0.5), K));                           // 0.5 - 0.5 * (d2 + d1) / K
        if (bad(e) || !(e.hi <= 0.0)) return 0;
        return post == POST_SDIFF ? 1 : 2;
    }
    default: return 0;
    }
}

__device__ __forceinline__ void mask_set_range(uint32_t *m, int a, int b) {   // bits a..b inclusive, 0 <= a <= b < 256
    for (int k = 0; k < 8; k++) {
        const int lo = a - 32 * k, hi = b - 32 * k;
        if (hi < 0 || lo > 31) continue;
        m[k] |= (0xFFFFFFFFu >> (31 - (hi > 31 ? 31 : hi))) & (0xFFFFFFFFu << (lo < 0 ? 0 : lo));
    }
}

// (the masks live in registers: no run-time indexing of the arrays)
__device__ __forceinline__ void mask_set_bit(uint32_t *m, int i) {
    for (int k = 0; k < 8; k++) if ((i >> 5) == k) m[k] |= 1u << (i & 31);
}
__device__ __forceinline__ uint32_t mask_word(const uint32_t *m, int idx) {
    uint32_t r = 0;
    for (int k = 0; k < 8; k++) r = idx == k ? m[k] : r;
    return r;
}

SDF_IA void ia_keep_bit(uint32_t *m, int i) {   // (mask_set_bit, host-compilable)
    for (int k = 0; k < 8; k++) if ((i >> 5) == k) m[k] |= 1u << (i & 31);
}

// One pass of the tape over the box (x, y, z); returns the interval of the model's value.  All
// lanes of the wave run the same tape (uniform control flow).  With DECIDE the 8 lanes of a batch
// agree on which operands to drop with a ballot and every one of them records it: masks[0..8) skip
// bits, masks[8..16) forced bits (without DECIDE rstart / lstart / masks are not touched).
// (Without DECIDE the run also compiles for the host: tests/test_interval_host.py runs whole tapes over random boxes
// and checks the CPU checker's point values against the intervals.)
template <bool DECIDE, bool FULL, bool RARE>
SDF_IA Ival ia_run_tape(const uint32_t *__restrict__ code, const double *__restrict__ consts,
                        const uint16_t *__restrict__ rstart, const uint16_t *__restrict__ lstart, int n_instr,
                        Ival x, Ival y, Ival z, bool live, IaShared sh, int n_d, uint32_t *masks) {
    using namespace ia;
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63, gbase = lane & ~7;
#endif
    Ival acc = pt(0.0);
    for (int i = 0; i < sh.n_p; i++)
        for (int k = 0; k < 6; k++) sh.ps(i, k) = 0.0;
    for (int i = 0; i < n_d; i++) { sh.ds(i, 0) = 0.0; sh.ds(i, 1) = 0.0; }
    uint32_t keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // point bookkeeping is never skipped (see below)
    // (slot numbers were validated against n_p / n_d when the tape was created)
    auto psave = [&](uint32_t k) {
        sh.ps(k, 0) = x.lo; sh.ps(k, 1) = x.hi; sh.ps(k, 2) = y.lo; sh.ps(k, 3) = y.hi; sh.ps(k, 4) = z.lo; sh.ps(k, 5) = z.hi;
    };
    auto pload = [&](uint32_t k) {
        x = Ival{sh.ps(k, 0), sh.ps(k, 1)}; y = Ival{sh.ps(k, 2), sh.ps(k, 3)}; z = Ival{sh.ps(k, 4), sh.ps(k, 5)};
    };
    auto dsave = [&](uint32_t k, const Ival &v) { sh.ds(k, 0) = v.lo; sh.ds(k, 1) = v.hi; };
    auto dload = [&](uint32_t k) { return Ival{sh.ds(k, 0), sh.ds(k, 1)}; };
    for (int ip = 0; ip < n_instr; ip++) {
        const uint32_t w0 = code[2 * ip], w1 = code[2 * ip + 1];
        const uint32_t op = w0 & 255u, post = (w0 >> 8) & 7u, sa = (w0 >> 24) & 7u;
        const double *c = consts + (w1 & 0xFFFFFFu) + 1;
        if (op == OP_END) break;
        if (w0 & 0x000800u) pload((w0 >> 12) & 7u);
        if (w0 & 0x008000u) psave((w0 >> 16) & 7u);
        if (w0 & 0x080000u) dsave((w0 >> 20) & 7u, acc);
        const bool is_leaf = op >= OP_L_SPHERE && op < OP_COMB;
        if (is_leaf || op == OP_COMB) {
            const Ival left = is_leaf ? acc : dload(sa);
            Ival right = acc;
            if (is_leaf) {
                if (RARE && ia_is_rare_leaf(op)) right = ia_leaf_rare(op, c, x, y, z);
                else right = ia_leaf(op, c, x, y, z);
            }
            int rs = 0xFFFF, ls = 0xFFFF;
            if constexpr (DECIDE) { rs = rstart[ip]; ls = lstart[ip]; }
#if defined(__HIP_DEVICE_COMPILE__)
            if (DECIDE && rs != 0xFFFF && ls != 0xFFFF && rs <= ip && ls <= rs) {
                const int d = live ? ia_decide(post, left, right, c[-1]) : 3;     // dead lanes agree with everything
                const unsigned long long b1 = __ballot(d == 1 || d == 3), b2 = __ballot(d == 2 || d == 3);
                if (((b1 >> gbase) & 0xFFull) == 0xFFull) {
                    if (post < POST_SUNION) mask_set_range(masks, rs, ip);
                    else {                                   // smooth: the combine itself stays, as `+ (+0.0)`
                        if (rs < ip) mask_set_range(masks, rs, ip - 1);
                        mask_set_bit(masks + 8, ip);
                    }
                } else if (((b2 >> gbase) & 0xFFull) == 0xFFull && rs > ls) {
                    mask_set_range(masks, ls, rs - 1);
                    mask_set_bit(masks + 8, ip);
                }
            }
#endif
            acc = ia_post(post, left, right, c[-1]);
            continue;
        }
        switch (op) {
        case OP_TRANSLATE: x = subc(x, c[0]); y = subc(y, c[1]); z = subc(z, c[2]); break;
        case OP_SCALE: x = divc(x, c[0]); y = divc(y, c[1]); z = divc(z, c[2]); break;
        case OP_ROTATE: {
            const Ival nx = dot3c(x, y, z, c[0], c[3], c[6]);
            const Ival ny = dot3c(x, y, z, c[1], c[4], c[7]);
            const Ival nz = dot3c(x, y, z, c[2], c[5], c[8]);
            x = nx; y = ny; z = nz; break; }
        case OP_ELONGATE: {
            const Ival qx = subc(abs_(x), c[0]), qy = subc(abs_(y), c[1]), qz = subc(abs_(z), c[2]);
            dsave(sa, minc(max_(qx, max_(qy, qz)), 0.0));
            x = maxc(qx, 0.0); y = maxc(qy, 0.0); z = maxc(qz, 0.0); break; }
        case OP_TRANSLATE2: x = subc(x, c[0]); y = subc(y, c[1]); break;
        case OP_SCALE2: x = divc(x, c[0]); y = divc(y, c[1]); break;
        case OP_ROTATE2: {
            const Ival nx = dot2c(x, y, c[0], c[2]), ny = dot2c(x, y, c[1], c[3]);
            x = nx; y = ny; break; }
        case OP_ELONGATE2: {
            const Ival qx = subc(abs_(x), c[0]), qy = subc(abs_(y), c[1]);
            dsave(sa, minc(max_(qx, qy), 0.0));
            x = maxc(qx, 0.0); y = maxc(qy, 0.0); break; }
        case OP_REVOLVE: { const Ival nx = subc(len2(x, y), c[0]); y = z; x = nx; z = pt(0.0); break; }
        case OP_SETZ0: z = pt(0.0); break;
        case OP_SAVE_P: psave(sa); if constexpr (DECIDE) ia_keep_bit(keep, ip); break;
        case OP_LOAD_P: pload(sa); if constexpr (DECIDE) ia_keep_bit(keep, ip); break;
        case OP_PUSH_D: dsave(sa, acc); break;
        case OP_NOP: break;
        case OP_NEG: acc = neg(acc); break;
        case OP_ADDC: acc = addc(acc, c[0]); break;
        case OP_SUBC: acc = subc(acc, c[0]); break;
        case OP_MULC: acc = mulc(acc, c[0]); break;
        case OP_SHELL: acc = subc(abs_(acc), c[0]); break;
        case OP_ADD_DS: acc = add(acc, dload(sa)); break;
        case OP_EXT_PRE: dsave(sa, subc(abs_(z), c[0])); break;
        case OP_EXT_POST: {
            const Ival w = dload(sa);
            acc = add(minc(max_(acc, w), 0.0), len2(maxc(acc, 0.0), maxc(w, 0.0))); break; }
        case OP_BEND_LINEAR: {   // sdf_interp.h L_BEND_LINEAR (d3.py:435-445); the easing id is c[10]
            const Ival tt = ease01((int)c[10], clip01(divc(dot3c(subc(x, c[0]), subc(y, c[1]), subc(z, c[2]), c[3], c[4], c[5]), c[6])));
            x = add(x, mulc(tt, c[7])); y = add(y, mulc(tt, c[8])); z = add(z, mulc(tt, c[9])); break; }
        case OP_REP_PREP: {      // L_REP_PREP (dn.py:80-112): PS[sa] = the cell index of p, per axis
            const int dim = (int)c[0];
            Ival idx[3] = {pt(0.0), pt(0.0), pt(0.0)};
            const Ival pp[3] = {x, y, z};
            for (int i = 0; i < 3; i++) {
                if (i >= dim) continue;
                const double s = c[1 + i];
                Ival r = s != 0.0 ? rint_(divc(pp[i], s)) : pt(0.0);
                if (c[4] != 0.0) r = clipc(r, -c[5 + i], c[5 + i]);
                idx[i] = r;
            }
            sh.ps(sa, 0) = idx[0].lo; sh.ps(sa, 1) = idx[0].hi; sh.ps(sa, 2) = idx[1].lo; sh.ps(sa, 3) = idx[1].hi;
            sh.ps(sa, 4) = idx[2].lo; sh.ps(sa, 5) = idx[2].hi; break; }
        case OP_REP_SET: {       // L_REP_SET: p = p0 - spacing * (index + n); the index is taken as independent of p0
            const uint32_t sb = (w1 >> 24) & 7u;
            const Ival ax{sh.ps(sa, 0), sh.ps(sa, 1)}, ay{sh.ps(sa, 2), sh.ps(sa, 3)}, az{sh.ps(sa, 4), sh.ps(sa, 5)};
            const Ival bx{sh.ps(sb, 0), sh.ps(sb, 1)}, by{sh.ps(sb, 2), sh.ps(sb, 3)}, bz{sh.ps(sb, 4), sh.ps(sb, 5)};
            x = sub(ax, mulc(addc(bx, c[3]), c[0]));
            y = sub(ay, mulc(addc(by, c[4]), c[1]));
            z = sub(az, mulc(addc(bz, c[5]), c[2])); break; }
        case OP_TRANS_LIN_PRE:   // L_TRANS_LIN_PRE (d3.py:459-470): DS[sa] = ease(clip(dot(p - p0, v) / |v|^2))
            dsave(sa, ease01((int)c[7], clip01(divc(dot3c(subc(x, c[0]), subc(y, c[1]), subc(z, c[2]), c[3], c[4], c[5]), c[6])))); break;
        case OP_TRANS_MIX: {     // L_TRANS_MIX: t * d2 + (1 - t) * d1
            const uint32_t sb = (w1 >> 24) & 7u;
            const Ival tt = dload(sa), dd = dload(sb);
            acc = add(mul(tt, acc), mul(csub(1.0, tt), dd)); break; }
        case OP_EXTTO_PRE:       // L_EXTTO_PRE (d2.py:274)
            dsave(sa, ease01((int)c[1], addc(clipc(divc(z, c[0]), -0.5, 0.5), 0.5))); break;
        case OP_EXTTO_MIX: {     // L_EXTTO_MIX (d2.py:275): d1 + (d2 - d1) * t
            const uint32_t sb = (w1 >> 24) & 7u;
            const Ival dd1 = dload(sb), tt = dload(sa);
            acc = add(dd1, mul(sub(acc, dd1), tt)); break; }
        case OP_SLICE_POST: {    // L_SLICE_POST (d3.py:515-519): A <= 0 ? -acc : A
            const Ival A = dload(sa), B = neg(acc);
            if (bad(A) || bad(B)) acc = top();
            else if (A.hi <= 0.0) acc = B;
            else if (A.lo > 0.0) acc = A;
            else acc = Ival{fmin(A.lo, B.lo), fmax(A.hi, B.hi)};
            break; }
        case OP_TWIST: case OP_BEND: case OP_BEND_RADIAL: case OP_TRANS_RAD_PRE: case OP_WRAP_AROUND:   // trig-capable builds only
            if constexpr (FULL) {
                if (op == OP_WRAP_AROUND) {                 // L_WRAP_AROUND (d3.py:483-502)
                    const double pi = 3.141592653589793;
                    Ival r = finite_(x) && finite_(y) ? pad(len2(x, y), 1e-12, 1e-300) : top();
                    if (!bad(r)) r.lo = fmax(r.lo, 0.0);
                    const Ival d = subc(r, c[9]);
                    const Ival u = divc(addc(atan2_range(x, y), pi), 2.0 * pi);
                    const Ival tt = bad(u) ? top() : ease01((int)c[10], Ival{fmax(u.lo, 0.0), fmin(u.hi, 1.0)});
                    const Ival nx = add(addc(mulc(tt, c[3]), c[0]), mulc(d, c[6])), ny = add(addc(mulc(tt, c[4]), c[1]), mulc(d, c[7]));
                    x = nx; y = ny;
                } else
                if (op == OP_TWIST || op == OP_BEND) {      // L_TWIST / L_BEND (d3.py:407-433): rotation by c0 * z resp. c0 * x
                    Ival sn, cs;
                    sincos_range(mulc(op == OP_TWIST ? z : x, c[0]), sn, cs);
                    const Ival nx = sub(mul(cs, x), mul(sn, y)), ny = add(mul(sn, x), mul(cs, y));
                    x = nx; y = ny;
                } else {                                    // hypot(x, y) goes through libm: widened
                    Ival r = finite_(x) && finite_(y) ? pad(len2(x, y), 1e-12, 1e-300) : top();
                    if (!bad(r)) r.lo = fmax(r.lo, 0.0);
                    const Ival tt = clip01(divc(subc(r, c[0]), c[1]));
                    if (op == OP_BEND_RADIAL) z = sub(z, mulc(ease01((int)c[3], tt), c[2]));   // L_BEND_RADIAL (d3.py:447-457)
                    else dsave(sa, ease01((int)c[2], tt));                                    // L_TRANS_RAD_PRE (d3.py:472-481)
                }
                break;
            }
            [[fallthrough]];
        case OP_CIRC_PREP: case OP_CIRC_SET:   // circular_array (d3.py:379-392); trig-capable builds only
            if constexpr (FULL) {
                if (op == OP_CIRC_PREP) {          // L_CIRC_PREP: PS[sa] = (hypot(x, y), atan2(y, x) mod da, z)
                    Ival d, a;
                    circ_prep(x, y, c[0], d, a);
                    sh.ps(sa, 0) = d.lo; sh.ps(sa, 1) = d.hi; sh.ps(sa, 2) = a.lo; sh.ps(sa, 3) = a.hi;
                    sh.ps(sa, 4) = z.lo; sh.ps(sa, 5) = z.hi;
                } else {                           // L_CIRC_SET: p = (cos(a - delta) * d, sin(a - delta) * d, z)
                    const Ival d{sh.ps(sa, 0), sh.ps(sa, 1)}, a{sh.ps(sa, 2), sh.ps(sa, 3)};
                    circ_set(d, a, c[0], x, y);
                    z = Ival{sh.ps(sa, 4), sh.ps(sa, 5)};
                }
                break;
            }
            [[fallthrough]];
        default:
            // an op without an interval form: everything it may write becomes unknown
            x = y = z = top(); acc = top();
            for (int k = 0; k < n_d; k++) dsave(k, top());
            if (op == OP_REP_PREP || op == OP_CIRC_PREP) psave(sa);
            break;
        }
    }
    // A boolean saves the point its operands share inside its FIRST operand and restores it between
    // operands (tape.py `boolean`): those moves must happen even when the operand around them is
    // skipped.  As prefixes they survive on a NOP (compact_tape); as stand-alone instructions they
    // are exempted here.
    if constexpr (DECIDE)
        for (int k = 0; k < 8; k++) masks[k] &= ~keep[k];
    return acc;
}

// The batch's tape: the model's tape without the skipped instructions.
//   * a skipped instruction that carries prefixes leaves a NOP with those prefixes (the point / distance
//     moves of the surrounding constructs ride on their neighbours, tape.py peephole);
//   * a decided smooth combine becomes `acc + (+0.0)` behind the surviving operand (ia_decide);
//   * a forced hard min / max IS its right operand: post becomes SET (a forced COMB would then copy acc
//     onto itself and disappears); a forced difference is -right: SET, then NEG.  Forcing always skips at
//     least one instruction of the left chain, so the result never grows beyond the original length.
// Every lane of a batch holds the same masks and runs this loop (uniform: scalar loads of the
// tape); only lane 0 of the batch stores.  Returns the number of instructions (END included).
__device__ __forceinline__ int compact_tape(const unsigned long long *__restrict__ code64, int n_instr, const uint32_t *masks,
                                            unsigned long long *__restrict__ out, bool store, uint32_t zero_off) {
    const unsigned long long PREFIX = 0x00FFF800ull, POSTM = 0x700ull;
    const unsigned long long ADD0 = ((unsigned long long)zero_off << 32) | OP_ADDC;   // acc + (+0.0)
    int n = 0;
    auto put = [&](unsigned long long w) { if (store) out[n] = w; n++; };
    uint32_t skip_w = 0, force_w = 0;
    for (int ip = 0; ip < n_instr; ip++) {
        const unsigned long long w = code64[ip];
        const uint32_t op = (uint32_t)w & 255u, post = ((uint32_t)w >> 8) & 7u;
        if ((ip & 31) == 0) { skip_w = mask_word(masks, ip >> 5); force_w = mask_word(masks + 8, ip >> 5); }
        const bool skip = skip_w & 1u, forced = force_w & 1u;
        skip_w >>= 1; force_w >>= 1;
        if (skip) {
            if (w & PREFIX) put((w & PREFIX) | OP_NOP);
        } else if (!forced) {
            put(w);
        } else if (post == POST_SDIFF) {                      // smooth difference whose right operand was dropped
            put((w & PREFIX) | ADD0);
        } else if (op == OP_COMB) {
            if (post == POST_DIFF) put((w & PREFIX) | OP_NEG);
            else if (post >= POST_SUNION) put((w & PREFIX) | ADD0);
            else if (w & PREFIX) put((w & PREFIX) | OP_NOP);
        } else {
            put(w & ~POSTM);                                  // a leaf: post = SET
            if (post == POST_DIFF) put(OP_NEG);
            else if (post >= POST_SUNION) put(ADD0);
        }
    }
    return n;
}

}  // namespace sdfk
