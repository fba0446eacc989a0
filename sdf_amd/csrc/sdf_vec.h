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
