// sdf_vec.h -- Vec<T, N>: N independent samples per lane.
//
// The tape interpreter (sdf_interp.h) decodes one instruction per wave and then executes it for
// N samples per lane, so the scalar decode work (s_load of the instruction words and constants,
// the opcode branch tree) is amortised over N * 64 samples and the N dependency chains (float64
// sqrt / divide are ~20-instruction serial sequences) interleave in the VALU.  Everything here is
// element-wise; N = 1 degenerates to plain scalars.  Comparisons produce Mask<N>, data-dependent
// choices go through vsel() -- there is no per-lane control flow in the interpreter.
#pragma once
#include <hip/hip_runtime.h>

namespace sdfk {

#define SDF_DEV __device__ __forceinline__
#define SDF_UNROLL _Pragma("unroll")

template <int N> struct Mask {
    bool m[N];
};
template <typename T, int N> struct Vec {
    T v[N];
    SDF_DEV Vec() {}
    SDF_DEV Vec(T s) { SDF_UNROLL for (int i = 0; i < N; i++) v[i] = s; }
};

#define SDF_VEC_BINOP(OP)                                                                               \
    template <typename T, int N> SDF_DEV Vec<T, N> operator OP(const Vec<T, N> &a, const Vec<T, N> &b) { \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = a.v[i] OP b.v[i]; return r; }       \
    template <typename T, int N> SDF_DEV Vec<T, N> operator OP(const Vec<T, N> &a, T b) {                \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = a.v[i] OP b; return r; }            \
    template <typename T, int N> SDF_DEV Vec<T, N> operator OP(T a, const Vec<T, N> &b) {                \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = a OP b.v[i]; return r; }
SDF_VEC_BINOP(+)
SDF_VEC_BINOP(-)
SDF_VEC_BINOP(*)
SDF_VEC_BINOP(/)
#undef SDF_VEC_BINOP

#define SDF_VEC_CMP(OP)                                                                                \
    template <typename T, int N> SDF_DEV Mask<N> operator OP(const Vec<T, N> &a, const Vec<T, N> &b) { \
        Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = a.v[i] OP b.v[i]; return r; }       \
    template <typename T, int N> SDF_DEV Mask<N> operator OP(const Vec<T, N> &a, T b) {                \
        Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = a.v[i] OP b; return r; }            \
    template <typename T, int N> SDF_DEV Mask<N> operator OP(T a, const Vec<T, N> &b) {                \
        Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = a OP b.v[i]; return r; }
SDF_VEC_CMP(<)
SDF_VEC_CMP(<=)
SDF_VEC_CMP(>)
SDF_VEC_CMP(>=)
SDF_VEC_CMP(==)
SDF_VEC_CMP(!=)
#undef SDF_VEC_CMP

template <typename T, int N> SDF_DEV Vec<T, N> operator-(const Vec<T, N> &a) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = -a.v[i]; return r;
}
template <int N> SDF_DEV Mask<N> operator&(const Mask<N> &a, const Mask<N> &b) {
    Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = a.m[i] && b.m[i]; return r;
}
template <int N> SDF_DEV Mask<N> operator|(const Mask<N> &a, const Mask<N> &b) {
    Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = a.m[i] || b.m[i]; return r;
}
template <int N> SDF_DEV Mask<N> operator!(const Mask<N> &a) {
    Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = !a.m[i]; return r;
}
// mask != mask (exclusive or), used by the floored modulo
template <int N> SDF_DEV Mask<N> mask_xor(const Mask<N> &a, const Mask<N> &b) {
    Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = a.m[i] != b.m[i]; return r;
}
template <int N> SDF_DEV Mask<N> mask_all(bool b) {
    Mask<N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.m[i] = b; return r;
}

template <typename T, int N> SDF_DEV Vec<T, N> vsel(const Mask<N> &c, const Vec<T, N> &a, const Vec<T, N> &b) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = c.m[i] ? a.v[i] : b.v[i]; return r;
}
template <typename T, int N> SDF_DEV Vec<T, N> vsel(const Mask<N> &c, const Vec<T, N> &a, T b) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = c.m[i] ? a.v[i] : b; return r;
}
template <typename T, int N> SDF_DEV Vec<T, N> vsel(const Mask<N> &c, T a, const Vec<T, N> &b) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = c.m[i] ? a : b.v[i]; return r;
}
template <typename T, int N> SDF_DEV Vec<T, N> vsel_s(const Mask<N> &c, T a, T b) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = c.m[i] ? a : b; return r;
}

// "Late binding" of new machine state.  When an op computes the new point from the old one
// (rotate: every output needs every input), the new values are born while the old ones are still
// live, so the register coalescer cannot give the loop-carried state ONE home register and pays
// copies in EVERY op of the interpreter instead.  The empty asm re-defines the values after the
// last use of the old state (it depends on all of them at once), which confines the copies to
// the op that needs them.
template <typename T, int N> SDF_DEV void late_bind(Vec<T, N> &a) {
    SDF_UNROLL for (int i = 0; i < N; i++) asm volatile("" : "+v"(a.v[i]));
}
template <typename T> SDF_DEV void late_bind(Vec<T, 1> &a, Vec<T, 1> &b) { asm volatile("" : "+v"(a.v[0]), "+v"(b.v[0])); }
template <typename T> SDF_DEV void late_bind(Vec<T, 2> &a, Vec<T, 2> &b) {
    asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(b.v[0]), "+v"(b.v[1]));
}
template <typename T> SDF_DEV void late_bind(Vec<T, 1> &a, Vec<T, 1> &b, Vec<T, 1> &c) {
    asm volatile("" : "+v"(a.v[0]), "+v"(b.v[0]), "+v"(c.v[0]));
}
template <typename T> SDF_DEV void late_bind(Vec<T, 2> &a, Vec<T, 2> &b, Vec<T, 2> &c) {
    asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(c.v[0]), "+v"(c.v[1]));
}

template <typename T> SDF_DEV void late_bind(Vec<T, 3> &a, Vec<T, 3> &b) {
    asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]));
}
template <typename T> SDF_DEV void late_bind(Vec<T, 3> &a, Vec<T, 3> &b, Vec<T, 3> &c) {
    asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(c.v[0]), "+v"(c.v[1]), "+v"(c.v[2]));
}
// A copy the register coalescer cannot see through (an explicit v_mov): used where a value moves
// from one piece of machine state to another and the two must keep their own home registers.
SDF_DEV double real_move(double x) { double r; asm volatile("v_mov_b64 %0, %1" : "=v"(r) : "v"(x)); return r; }
SDF_DEV float real_move(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
template <typename T, int N> SDF_DEV Vec<T, N> real_move(const Vec<T, N> &a) {
    Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) r.v[i] = real_move(a.v[i]); return r;
}

// dst = src as an explicit move INTO dst's register ("+v": the result is tied to dst's old home),
// so a conditional state update merges with the not-taken path without any copy there
SDF_DEV void move_into(double &dst, double src) { asm volatile("v_mov_b64 %0, %1" : "+v"(dst) : "v"(src)); }
SDF_DEV void move_into(float &dst, float src) { asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "v"(src)); }
template <typename T, int N> SDF_DEV void move_into(Vec<T, N> &dst, const Vec<T, N> &src) {
    SDF_UNROLL for (int i = 0; i < N; i++) move_into(dst.v[i], src.v[i]);
}

// element-wise application of a scalar function
#define SDF_VEC_MAP1(NAME, EXPR)                                                 \
    template <typename T, int N> SDF_DEV Vec<T, N> NAME(const Vec<T, N> &a) {    \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) { const T x = a.v[i]; r.v[i] = (EXPR); } return r; }
#define SDF_VEC_MAP2(NAME, EXPR)                                                                      \
    template <typename T, int N> SDF_DEV Vec<T, N> NAME(const Vec<T, N> &a, const Vec<T, N> &b) {     \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) { const T x = a.v[i], y = b.v[i]; r.v[i] = (EXPR); } return r; } \
    template <typename T, int N> SDF_DEV Vec<T, N> NAME(const Vec<T, N> &a, T y) {                    \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) { const T x = a.v[i]; r.v[i] = (EXPR); } return r; } \
    template <typename T, int N> SDF_DEV Vec<T, N> NAME(T x, const Vec<T, N> &b) {                    \
        Vec<T, N> r; SDF_UNROLL for (int i = 0; i < N; i++) { const T y = b.v[i]; r.v[i] = (EXPR); } return r; }

}  // namespace sdfk
