/*
 * mc33.h -- CPU restatement of the per-cell decision procedure of Lewiner's marching cubes
 * ("Efficient implementation of Marching Cubes' cases with topological guarantees", Lewiner,
 * Lopes, Vieira, Tavares, JGT 2003) as scikit-image 0.18.3 executes it for
 * `measure.marching_cubes(volume, 0)` (the call of reference sdf/core.py:16-18).
 *
 * TEST INFRASTRUCTURE (part of the CPU checker).  scikit-image is a third-party dependency that
 * is not in /root/reference and whose Cython kernel is only on disk as a binary; this file
 * restates the published algorithm (face test, interior test, case/subcase selection over the
 * lookup tables of oracle/mc33_tables.h) and is pinned against what skimage actually returns:
 * tests/golden/mc33_volumes.npz (tools/make_golden_mc33.py) holds volumes that reach every case,
 * including the ones with a centre vertex, and the soups skimage produced for them.
 *
 * Cube numbering is Lewiner's (see tools/derive_mc33_tables.py): v[p], p = 0..7.
 * skimage specifics found by probing: its FLT_EPSILON is np.spacing(1.0) = 2.22e-16 (the same
 * constant as in the edge interpolation); cell values are float64 copies of the float32 samples;
 * the interior test adds FLT_EPSILON to its denominators and answers 0 where Lewiner's code falls
 * through (marked below).
 */
#ifndef SDF_ORACLE_MC33_H
#define SDF_ORACLE_MC33_H

#include <math.h>

#include "mc33_tables.h"

#define MC33_FLT_EPSILON 2.220446049250313e-16

/* face test: sign of the bilinear saddle on a face decides whether the positive corners are
 * joined across it; `face` is +-(1..6), negative = inverted answer */
static int mc33_test_face(const double *v, int face) {
    double A, B, C, D;
    int f = face < 0 ? -face : face;
    switch (f) {
    case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
    case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
    case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
    case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
    case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
    case 6: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;
    default: return 0;
    }
    double acbd = A * C - B * D;
    if (acbd > -MC33_FLT_EPSILON && acbd < MC33_FLT_EPSILON) return face >= 0;
    return face * A * acbd >= 0;
}

/* interior test: does the positive (s > 0 convention of the tables) region connect through the
 * cube?  `edge_hint` is the reference edge for cases 6, 7, 12, 13 */
static int mc33_test_interior(const double *v, int kase, int s, int edge) {
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0, a, b;
    int test = 0;
    if (kase == 4 || kase == 10) {
        a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC33_FLT_EPSILON);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        switch (edge) {
        case 0: t = v[0] / (v[0] - v[1] + MC33_FLT_EPSILON); At = 0; Bt = v[3] + (v[2] - v[3]) * t; Ct = v[7] + (v[6] - v[7]) * t; Dt = v[4] + (v[5] - v[4]) * t; break;
        case 1: t = v[1] / (v[1] - v[2] + MC33_FLT_EPSILON); At = 0; Bt = v[0] + (v[3] - v[0]) * t; Ct = v[4] + (v[7] - v[4]) * t; Dt = v[5] + (v[6] - v[5]) * t; break;
        case 2: t = v[2] / (v[2] - v[3] + MC33_FLT_EPSILON); At = 0; Bt = v[1] + (v[0] - v[1]) * t; Ct = v[5] + (v[4] - v[5]) * t; Dt = v[6] + (v[7] - v[6]) * t; break;
        case 3: t = v[3] / (v[3] - v[0] + MC33_FLT_EPSILON); At = 0; Bt = v[2] + (v[1] - v[2]) * t; Ct = v[6] + (v[5] - v[6]) * t; Dt = v[7] + (v[4] - v[7]) * t; break;
        case 4: t = v[4] / (v[4] - v[5] + MC33_FLT_EPSILON); At = 0; Bt = v[7] + (v[6] - v[7]) * t; Ct = v[3] + (v[2] - v[3]) * t; Dt = v[0] + (v[1] - v[0]) * t; break;
        case 5: t = v[5] / (v[5] - v[6] + MC33_FLT_EPSILON); At = 0; Bt = v[4] + (v[7] - v[4]) * t; Ct = v[0] + (v[3] - v[0]) * t; Dt = v[1] + (v[2] - v[1]) * t; break;
        case 6: t = v[6] / (v[6] - v[7] + MC33_FLT_EPSILON); At = 0; Bt = v[5] + (v[4] - v[5]) * t; Ct = v[1] + (v[0] - v[1]) * t; Dt = v[2] + (v[3] - v[2]) * t; break;
        case 7: t = v[7] / (v[7] - v[4] + MC33_FLT_EPSILON); At = 0; Bt = v[6] + (v[5] - v[6]) * t; Ct = v[2] + (v[1] - v[2]) * t; Dt = v[3] + (v[0] - v[3]) * t; break;
        case 8: t = v[0] / (v[0] - v[4] + MC33_FLT_EPSILON); At = 0; Bt = v[3] + (v[7] - v[3]) * t; Ct = v[2] + (v[6] - v[2]) * t; Dt = v[1] + (v[5] - v[1]) * t; break;
        case 9: t = v[1] / (v[1] - v[5] + MC33_FLT_EPSILON); At = 0; Bt = v[0] + (v[4] - v[0]) * t; Ct = v[3] + (v[7] - v[3]) * t; Dt = v[2] + (v[6] - v[2]) * t; break;
        case 10: t = v[2] / (v[2] - v[6] + MC33_FLT_EPSILON); At = 0; Bt = v[1] + (v[5] - v[1]) * t; Ct = v[0] + (v[4] - v[0]) * t; Dt = v[3] + (v[7] - v[3]) * t; break;
        case 11: t = v[3] / (v[3] - v[7] + MC33_FLT_EPSILON); At = 0; Bt = v[2] + (v[6] - v[2]) * t; Ct = v[1] + (v[5] - v[1]) * t; Dt = v[0] + (v[4] - v[0]) * t; break;
        default: return s < 0;
        }
    }
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    /* tests 5 and 10: Lewiner's code falls through to `return s < 0` when the saddle test fails;
     * skimage's port returns 0 there (found by probing 4000 random case-4 cells, see
     * tests/golden/mc33_volumes.npz) -- restated as skimage behaves */
    case 5: if (At * Ct - Bt * Dt < MC33_FLT_EPSILON) return s > 0; return 0;
    case 10: if (At * Ct - Bt * Dt >= MC33_FLT_EPSILON) return s > 0; return 0;
    default: break;   /* 7, 11, 13, 14, 15 */
    }
    return s < 0;
}

/* selects the tiling of one cell: returns the triangle count, *tiling = 3 * count vertex ids
 * (0..11 edges, 12 centre vertex) in Lewiner's order */
static int mc33_cell(const double *v, const signed char **tiling) {
    int idx = 0;
    for (int p = 0; p < 8; p++) if (v[p] > 0) idx |= 1 << p;
    const int kase = MC33_CASES[2 * idx], cfg = MC33_CASES[2 * idx + 1];
    int sub = 0;
#define T(NAME, STRIDE, N) do { *tiling = MC33_##NAME + (STRIDE); return (N); } while (0)
    switch (kase) {
    case 1: T(TILING1, cfg * 3, 1);
    case 2: T(TILING2, cfg * 6, 2);
    case 3:
        if (mc33_test_face(v, MC33_TEST3[cfg])) T(TILING3_2, cfg * 12, 4);
        T(TILING3_1, cfg * 6, 2);
    case 4:
        if (mc33_test_interior(v, 4, MC33_TEST4[cfg], -1)) T(TILING4_1, cfg * 6, 2);
        T(TILING4_2, cfg * 18, 6);
    case 5: T(TILING5, cfg * 9, 3);
    case 6:
        if (mc33_test_face(v, MC33_TEST6[cfg * 3 + 0])) T(TILING6_2, cfg * 15, 5);
        if (mc33_test_interior(v, 6, MC33_TEST6[cfg * 3 + 1], MC33_TEST6[cfg * 3 + 2])) T(TILING6_1_1, cfg * 9, 3);
        T(TILING6_1_2, cfg * 27, 9);
    case 7:
        if (mc33_test_face(v, MC33_TEST7[cfg * 5 + 0])) sub += 1;
        if (mc33_test_face(v, MC33_TEST7[cfg * 5 + 1])) sub += 2;
        if (mc33_test_face(v, MC33_TEST7[cfg * 5 + 2])) sub += 4;
        switch (sub) {
        case 0: T(TILING7_1, cfg * 9, 3);
        case 1: T(TILING7_2, (cfg * 3 + 0) * 15, 5);
        case 2: T(TILING7_2, (cfg * 3 + 1) * 15, 5);
        case 3: T(TILING7_3, (cfg * 3 + 0) * 27, 9);
        case 4: T(TILING7_2, (cfg * 3 + 2) * 15, 5);
        case 5: T(TILING7_3, (cfg * 3 + 1) * 27, 9);
        case 6: T(TILING7_3, (cfg * 3 + 2) * 27, 9);
        default:
            if (mc33_test_interior(v, 7, MC33_TEST7[cfg * 5 + 3], MC33_TEST7[cfg * 5 + 4])) T(TILING7_4_2, cfg * 27, 9);
            T(TILING7_4_1, cfg * 15, 5);
        }
    case 8: T(TILING8, cfg * 6, 2);
    case 9: T(TILING9, cfg * 12, 4);
    case 10:
        if (mc33_test_face(v, MC33_TEST10[cfg * 3 + 0])) {
            if (mc33_test_face(v, MC33_TEST10[cfg * 3 + 1])) T(TILING10_1_1_, cfg * 12, 4);
            T(TILING10_2, cfg * 24, 8);
        }
        if (mc33_test_face(v, MC33_TEST10[cfg * 3 + 1])) T(TILING10_2_, cfg * 24, 8);
        if (mc33_test_interior(v, 10, MC33_TEST10[cfg * 3 + 2], -1)) T(TILING10_1_1, cfg * 12, 4);
        T(TILING10_1_2, cfg * 24, 8);
    case 11: T(TILING11, cfg * 12, 4);
    case 12:
        if (mc33_test_face(v, MC33_TEST12[cfg * 4 + 0])) {
            if (mc33_test_face(v, MC33_TEST12[cfg * 4 + 1])) T(TILING12_1_1_, cfg * 12, 4);
            T(TILING12_2, cfg * 24, 8);
        }
        if (mc33_test_face(v, MC33_TEST12[cfg * 4 + 1])) T(TILING12_2_, cfg * 24, 8);
        if (mc33_test_interior(v, 12, MC33_TEST12[cfg * 4 + 2], MC33_TEST12[cfg * 4 + 3])) T(TILING12_1_1, cfg * 12, 4);
        T(TILING12_1_2, cfg * 24, 8);
    case 13: {
        for (int k = 0; k < 6; k++) if (mc33_test_face(v, MC33_TEST13[cfg * 7 + k])) sub += 1 << k;
        const int sc = MC33_SUBCONFIG13[sub];
        if (sc == 0) T(TILING13_1, cfg * 12, 4);
        if (sc >= 1 && sc <= 6) T(TILING13_2, (cfg * 6 + sc - 1) * 18, 6);
        if (sc >= 7 && sc <= 18) T(TILING13_3, (cfg * 12 + sc - 7) * 30, 10);
        if (sc >= 19 && sc <= 22) T(TILING13_4, (cfg * 4 + sc - 19) * 36, 12);
        if (sc >= 23 && sc <= 26) {
            const int k = sc - 23;
            if (mc33_test_interior(v, 13, MC33_TEST13[cfg * 7 + 6], MC33_TILING13_5_1[(cfg * 4 + k) * 18]))
                T(TILING13_5_1, (cfg * 4 + k) * 18, 6);
            T(TILING13_5_2, (cfg * 4 + k) * 30, 10);
        }
        if (sc >= 27 && sc <= 38) T(TILING13_3_, (cfg * 12 + sc - 27) * 30, 10);
        if (sc >= 39 && sc <= 44) T(TILING13_2_, (cfg * 6 + sc - 39) * 18, 6);
        if (sc == 45) T(TILING13_1_, cfg * 12, 4);
        *tiling = 0; return 0;   /* impossible subconfiguration (-1 in the table) */
    }
    case 14: T(TILING14, cfg * 12, 4);
    default: break;
    }
#undef T
    *tiling = 0;
    return 0;
}

#endif /* SDF_ORACLE_MC33_H */
