// sdf_mc33.h -- device side of Lewiner's marching-cubes case resolution for AMBIGUOUS cells.
//
// skimage.measure.marching_cubes(volume, 0) (reference sdf/core.py:16-18) is Lewiner's MC33: for
// the sign configurations with an ambiguous face or body (MC_AMBIGUOUS in mc_table.h) the tiling is
// chosen from the sample VALUES by face tests and an interior test; every other configuration is
// the classic table the kernels already use.  Ambiguous cells are rare in distance fields (weave
// at 2^22: 291 of 164,999 surface cells), so this code runs on a few diverged lanes; it is kept
// out of line so it costs the hot kernels no registers.
//
// The decision procedure follows Lewiner, Lopes, Vieira, Tavares (JGT 2003) with the deviations of
// the scikit-image 0.18.3 port, established by probing it (tests/golden/mc33_volumes.npz): epsilon
// is np.spacing(1.0) and is added to the interior test's denominators, and the two saddle branches
// of the interior test answer 0 where the original falls through.  Tables: mc33_tables.h (flat
// array `t`, offsets MC33_OFF_*), Lewiner's cube numbering: corner p = (x,y,z) with x = volume
// axis 2, y = axis 1, z = axis 0.
#pragma once
#include <hip/hip_runtime.h>

#include "mc33_tables.h"

namespace sdfk {

#define MC33_EPS 2.220446049250313e-16

__device__ __forceinline__ bool mc33_test_face(const double *v, int face) {
    const int f = face < 0 ? -face : face;
    // corners (A, B, C, D) of face f
    const int ia = f == 1 ? 0 : f == 2 ? 1 : f == 3 ? 2 : f == 4 ? 3 : f == 5 ? 0 : 4;
    const int ib = f == 1 ? 4 : f == 2 ? 5 : f == 3 ? 6 : f == 4 ? 7 : f == 5 ? 3 : 7;
    const int ic = f == 1 ? 5 : f == 2 ? 6 : f == 3 ? 7 : f == 4 ? 4 : f == 5 ? 2 : 6;
    const int id = f == 1 ? 1 : f == 2 ? 2 : f == 3 ? 3 : f == 4 ? 0 : f == 5 ? 1 : 5;
    const double A = v[ia], B = v[ib], C = v[ic], D = v[id];
    const double acbd = A * C - B * D;
    if (acbd > -MC33_EPS && acbd < MC33_EPS) return face >= 0;
    return (double)face * A * acbd >= 0;
}

__device__ __forceinline__ bool mc33_test_interior(const double *v, int kase, int s, int edge) {
    double t, At = 0, Bt, Ct, Dt;
    if (kase == 4 || kase == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC33_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        // t on the reference edge (e0 -> e1), then the three parallel edges (B, C, D)
        // rows: e0, e1, B0, B1, C0, C1, D0, D1
        static const signed char R[12][8] = {
            {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
            {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
            {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
        if (edge < 0 || edge > 11) return s < 0;
        const signed char *r = R[edge];
        t = v[r[0]] / (v[r[0]] - v[r[1]] + MC33_EPS);
        Bt = v[r[2]] + (v[r[3]] - v[r[2]]) * t;
        Ct = v[r[4]] + (v[r[5]] - v[r[4]]) * t;
        Dt = v[r[6]] + (v[r[7]] - v[r[6]]) * t;
    }
    const int test = (At >= 0 ? 1 : 0) | (Bt >= 0 ? 2 : 0) | (Ct >= 0 ? 4 : 0) | (Dt >= 0 ? 8 : 0);
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    case 5: return (At * Ct - Bt * Dt < MC33_EPS) ? (s > 0) : false;      // skimage: 0 on the failed saddle
    case 10: return (At * Ct - Bt * Dt >= MC33_EPS) ? (s > 0) : false;
    default: return s < 0;   // 7, 11, 13, 14, 15
    }
}

// Selects the tiling of one ambiguous cell: v[p] = the 8 samples in Lewiner corner order (float64
// copies of the float32 samples), t = the flat table.  Returns the triangle count (<= 12) and the
// offset of 3 * count vertex ids (0..11 edges, 12 = centre vertex) in Lewiner's order.
__device__ __noinline__ int mc33_cell(const double *v, const signed char *t, int *tiling_off) {
    int idx = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) if (v[p] > 0) idx |= 1 << p;
    const int kase = t[MC33_OFF_CASES + 2 * idx], cfg = t[MC33_OFF_CASES + 2 * idx + 1];
    int sub = 0;
#define T(NAME, ROW, N) do { *tiling_off = MC33_OFF_##NAME + (ROW); return (N); } while (0)
#define FACE(NAME, I) mc33_test_face(v, t[MC33_OFF_##NAME + (I)])
    switch (kase) {
    case 1: T(TILING1, cfg * 3, 1);
    case 2: T(TILING2, cfg * 6, 2);
    case 3:
        if (FACE(TEST3, cfg)) T(TILING3_2, cfg * 12, 4);
        T(TILING3_1, cfg * 6, 2);
    case 4:
        if (mc33_test_interior(v, 4, t[MC33_OFF_TEST4 + cfg], -1)) T(TILING4_1, cfg * 6, 2);
        T(TILING4_2, cfg * 18, 6);
    case 5: T(TILING5, cfg * 9, 3);
    case 6:
        if (FACE(TEST6, cfg * 3)) T(TILING6_2, cfg * 15, 5);
        if (mc33_test_interior(v, 6, t[MC33_OFF_TEST6 + cfg * 3 + 1], t[MC33_OFF_TEST6 + cfg * 3 + 2])) T(TILING6_1_1, cfg * 9, 3);
        T(TILING6_1_2, cfg * 27, 9);
    case 7:
        if (FACE(TEST7, cfg * 5 + 0)) sub += 1;
        if (FACE(TEST7, cfg * 5 + 1)) sub += 2;
        if (FACE(TEST7, cfg * 5 + 2)) sub += 4;
        switch (sub) {
        case 0: T(TILING7_1, cfg * 9, 3);
        case 1: T(TILING7_2, (cfg * 3 + 0) * 15, 5);
        case 2: T(TILING7_2, (cfg * 3 + 1) * 15, 5);
        case 3: T(TILING7_3, (cfg * 3 + 0) * 27, 9);
        case 4: T(TILING7_2, (cfg * 3 + 2) * 15, 5);
        case 5: T(TILING7_3, (cfg * 3 + 1) * 27, 9);
        case 6: T(TILING7_3, (cfg * 3 + 2) * 27, 9);
        default:
            if (mc33_test_interior(v, 7, t[MC33_OFF_TEST7 + cfg * 5 + 3], t[MC33_OFF_TEST7 + cfg * 5 + 4])) T(TILING7_4_2, cfg * 27, 9);
            T(TILING7_4_1, cfg * 15, 5);
        }
    case 8: T(TILING8, cfg * 6, 2);
    case 9: T(TILING9, cfg * 12, 4);
    case 10:
        if (FACE(TEST10, cfg * 3 + 0)) {
            if (FACE(TEST10, cfg * 3 + 1)) T(TILING10_1_1_, cfg * 12, 4);
            T(TILING10_2, cfg * 24, 8);
        }
        if (FACE(TEST10, cfg * 3 + 1)) T(TILING10_2_, cfg * 24, 8);
        if (mc33_test_interior(v, 10, t[MC33_OFF_TEST10 + cfg * 3 + 2], -1)) T(TILING10_1_1, cfg * 12, 4);
        T(TILING10_1_2, cfg * 24, 8);
    case 11: T(TILING11, cfg * 12, 4);
    case 12:
        if (FACE(TEST12, cfg * 4 + 0)) {
            if (FACE(TEST12, cfg * 4 + 1)) T(TILING12_1_1_, cfg * 12, 4);
            T(TILING12_2, cfg * 24, 8);
        }
        if (FACE(TEST12, cfg * 4 + 1)) T(TILING12_2_, cfg * 24, 8);
        if (mc33_test_interior(v, 12, t[MC33_OFF_TEST12 + cfg * 4 + 2], t[MC33_OFF_TEST12 + cfg * 4 + 3])) T(TILING12_1_1, cfg * 12, 4);
        T(TILING12_1_2, cfg * 24, 8);
    case 13: {
        for (int k = 0; k < 6; k++) if (FACE(TEST13, cfg * 7 + k)) sub += 1 << k;
        const int sc = t[MC33_OFF_SUBCONFIG13 + sub];
        if (sc == 0) T(TILING13_1, cfg * 12, 4);
        if (sc >= 1 && sc <= 6) T(TILING13_2, (cfg * 6 + sc - 1) * 18, 6);
        if (sc >= 7 && sc <= 18) T(TILING13_3, (cfg * 12 + sc - 7) * 30, 10);
        if (sc >= 19 && sc <= 22) T(TILING13_4, (cfg * 4 + sc - 19) * 36, 12);
        if (sc >= 23 && sc <= 26) {
            const int k = sc - 23;
            if (mc33_test_interior(v, 13, t[MC33_OFF_TEST13 + cfg * 7 + 6], t[MC33_OFF_TILING13_5_1 + (cfg * 4 + k) * 18]))
                T(TILING13_5_1, (cfg * 4 + k) * 18, 6);
            T(TILING13_5_2, (cfg * 4 + k) * 30, 10);
        }
        if (sc >= 27 && sc <= 38) T(TILING13_3_, (cfg * 12 + sc - 27) * 30, 10);
        if (sc >= 39 && sc <= 44) T(TILING13_2_, (cfg * 6 + sc - 39) * 18, 6);
        if (sc == 45) T(TILING13_1_, cfg * 12, 4);
        break;
    }
    case 14: T(TILING14, cfg * 12, 4);
    default: break;
    }
#undef T
#undef FACE
    *tiling_off = 0;
    return 0;
}

// the 8 samples of the cell whose corner 0 is at `corner` (strides s0, s1, 1), Lewiner order
__device__ __forceinline__ void mc33_load_cell(const float *corner, int s0, int s1, double *lv) {
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int c = (p == 0) ? 0 : (p == 1) ? 1 : (p == 2) ? 3 : (p == 3) ? 2 : (p == 4) ? 4 : (p == 5) ? 5 : (p == 6) ? 7 : 6;   // MC33_CORNER
        lv[p] = (double)corner[(c >> 2) * s0 + ((c >> 1) & 1) * s1 + (c & 1)];
    }
}

// centre vertex (Lewiner's vertex 12) as skimage places it: centre of mass of the corners
// weighted by 1 / (eps + |v|), float64, stored as float32
__device__ __forceinline__ void mc33_centre_vertex(const double *lv, int i0, int i1, int i2, float *o) {
    double fx = 0, fy = 0, fz = 0, ff = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const double w = 1.0 / (MC33_EPS + fabs(lv[p]));
        const int X = (p == 1 || p == 2 || p == 5 || p == 6), Y = (p == 2 || p == 3 || p == 6 || p == 7), Z = p >= 4;
        fx += (double)X * w; fy += (double)Y * w; fz += (double)Z * w; ff += w;
    }
    o[0] = (float)((double)i0 + fz / ff); o[1] = (float)((double)i1 + fy / ff); o[2] = (float)((double)i2 + fx / ff);
}

}  // namespace sdfk
