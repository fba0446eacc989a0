// sdf_bounds.hip -- `_estimate_bounds` (reference sdf/core.py:62-82) as ONE launch: up to 32 rounds of a 16^3 probe grid (np.linspace
// per axis: lo + i * step, the last sample forced to hi), threshold = |d| / 2, the box of the samples with |f| <= threshold, grown by
// half a probe cell -- float64 throughout, operation by operation like the reference (and the oracle's restatement).  The host loop
// it replaces paid a kernel launch, two copies and a synchronisation per round: 3 ms per model.  out[0..6) = lo, hi; out[6] = 1 when
// a round found no sample within its threshold (the reference raises there: `where.max` of an empty array).
//
// The rounds are a dependent chain (a round's grid is the previous round's hit box), so what counts is the latency of ONE round: 4096
// probes through the tape, then everybody needs the hit box.  Measured (ms per call incl. the host's share, per model: example /
// gearlike / blobby / weave / knurling / pawn; r05h, the last two rows r05u on one box, alternating):
//   four workgroups, six atomic maxima + arrival counter + polling + fences per round (r02 - r04)   0.80 /  -   /  -   / 4.0  /  -   /  -
//   one workgroup, four probes per lane one after the other (`__syncthreads` only)                  0.78: the barrier was not the cost alone
//   one workgroup of 512, eight probes per lane, four at a time through the interpreter (NS = 4)    0.58 / 0.80 / 1.10 / 3.5* / 1.63 / 0.86*
//   two workgroups, two probes per lane at a time (NS = 2)                                          0.50 / 0.61 / 0.68 / 3.5* / 1.03 / 0.87*
//   four workgroups of 1024, a probe per lane, the box as ONE word per workgroup and round (r05h)   0.43 / 0.55 / 0.56 / 3.48 / 0.84 / 0.87
//   SIXTY-FOUR workgroups of ONE wave, no barrier, the box as a bit mask per axis (this file)       0.17 / 0.21 / 0.29 / 2.13 / 0.36 / 0.45
//   (* = the tape's register file has no room for NS > 1: the four-workgroup kernel ran)
// What the table says: a tape instruction of the one-sample interpreter is a latency chain (the scalar fetch of its words, the
// dispatch, a few dependent float64 operations), and a wave that shares its SIMD with three others -- 1024 lanes on one compute unit
// -- waits for them at every link of it: ~ 500 cycles per instruction whatever it computes.  Alone on its SIMD the same wave takes a
// third of that, and a workgroup of one wave has nothing to synchronise with but the other sixty-three once per round.
// A translation unit of its own: built in parallel with the others (build.sh), with the interpreters' structurizer option.
#include "sdf_interp.h"
#include "sdf_bounds.h"

using namespace sdfk;

// The round's 4096 probes as SIXTY-FOUR workgroups of ONE wave each -- a probe per lane, a wave per compute unit.  A workgroup of one
// wave needs no `__syncthreads` at all: every lane carries the whole state (lo, hi, the previous threshold) in its own
// registers and does the round's scalar arithmetic itself -- the same operations in the same order on the same values in every lane of
// every wave, so all of them take the same decisions.  The hit box travels as a BIT MASK (bit i of a 16-bit field per axis: some probe
// with index i along that axis lies within the threshold), so that waves and lanes are combined by OR -- one wave reduction in the
// vector ALU (the row shifts of block_exclusive_scan, sdf_device.h) instead of six maxima; min / max index = ctz / clz of a field.
// Per round ONE word per wave goes to device memory (a relaxed store: the word is all there is to see), lane l polls wave l's word of
// the round (relaxed loads: one coalesced 512-byte read per poll), a second wave reduction gives everybody the box.  The words carry
// the call's 16-bit tag instead of being zeroed before every call (the host clears them when the tag wraps).  out[6] as above.
__device__ __forceinline__ unsigned wave_or_u32(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
enum { BOUNDS_WAVES = 64 };
template <typename T, bool FULL, int NP, int ND>
__global__ __launch_bounds__(64) void k_estimate_bounds_w(const uint32_t *__restrict__ code, const T *__restrict__ consts, double *__restrict__ out,
                                                          unsigned long long *__restrict__ work, unsigned tag) {
    const int lane = threadIdx.x, q = (int)blockIdx.x * 64 + lane;
    const int i = q >> 8, j = (q >> 4) & 15, k = q & 15;      // this lane's probe of every round: (X[i], Y[j], Z[k])
    double lo0 = -1e9, lo1 = -1e9, lo2 = -1e9, hi0 = 1e9, hi1 = 1e9, hi2 = 1e9, prev = -1.0;
    for (int it = 0; it < 32; it++) {
        // np.linspace(lo, hi, 16): lo + i * step, the last sample forced to hi; d = X[1] - X[0]
        const double s0 = (hi0 - lo0) / 15.0, s1 = (hi1 - lo1) / 15.0, s2 = (hi2 - lo2) / 15.0;
        const double d0 = (lo0 + 1.0 * s0) - (lo0 + 0.0 * s0), d1 = (lo1 + 1.0 * s1) - (lo1 + 0.0 * s1), d2 = (lo2 + 1.0 * s2) - (lo2 + 0.0 * s2);
        const double thr = sqrt((d0 * d0 + d1 * d1) + d2 * d2) / 2;
        if (it > 0 && thr == prev) break;                      // (every lane of every wave alike)
        prev = thr;
        const double x = i == 15 ? hi0 : lo0 + (double)i * s0, y = j == 15 ? hi1 : lo1 + (double)j * s1, z = k == 15 ? hi2 : lo2 + (double)k * s2;
        const T v = run_tape<T, FULL, NP, ND, 1>(code, consts, Vec<T, 1>((T)x), Vec<T, 1>((T)y), Vec<T, 1>((T)z)).v[0];
        const bool hit = fabs((double)v) <= thr;
        const unsigned mine_lo = wave_or_u32(hit ? (1u << i) | (1u << (16 + j)) : 0u), mine_hi = wave_or_u32(hit ? 1u << k : 0u);
        unsigned long long *slot = work + (size_t)it * BOUNDS_WAVES;
        if (lane == 0)
            __hip_atomic_store(&slot[blockIdx.x], ((unsigned long long)tag << 48) | ((unsigned long long)mine_hi << 32) | mine_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long w = 0;
        for (unsigned spins = 0;; spins++) {
            w = __hip_atomic_load(&slot[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(w >> 48) == tag)) break;      // (wave-uniform)
            // (a wave that gives up -- the others were not co-resident for seconds -- must not go on with a partial box)
            if (spins >= (1u << 22)) { if (lane == 0) out[6] = 2.0; return; }
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned m_lo = wave_or_u32((unsigned)w), m_hi = wave_or_u32((unsigned)(w >> 32)) & 0xFFFFu;
        const unsigned mx_ = m_lo & 0xFFFFu, my_ = m_lo >> 16, mz_ = m_hi;
        if (mx_ == 0) { if (blockIdx.x == 0 && lane == 0) out[6] = 1.0; return; }       // no probe within the threshold (a hit sets a bit of every field)
        const double l0 = lo0, l1 = lo1, l2 = lo2;
        hi0 = l0 + (double)(31 - __clz((int)mx_)) * d0 + d0 / 2; lo0 = l0 + (double)(__ffs((int)mx_) - 1) * d0 - d0 / 2;
        hi1 = l1 + (double)(31 - __clz((int)my_)) * d1 + d1 / 2; lo1 = l1 + (double)(__ffs((int)my_) - 1) * d1 - d1 / 2;
        hi2 = l2 + (double)(31 - __clz((int)mz_)) * d2 + d2 / 2; lo2 = l2 + (double)(__ffs((int)mz_) - 1) * d2 - d2 / 2;
    }
    if (blockIdx.x == 0 && lane == 0) {
        out[0] = lo0; out[1] = lo1; out[2] = lo2; out[3] = hi0; out[4] = hi1; out[5] = hi2; out[6] = 0.0;
    }
}

template <typename T, bool FULL>
static int launch_bounds_file(hipStream_t stream, const uint32_t *code, const T *consts, double *out, unsigned long long *work, unsigned tag) {
    hipLaunchKernelGGL((k_estimate_bounds_w<T, FULL, SDF_NP_SLOTS, SDF_ND_SLOTS>), dim3(BOUNDS_WAVES), dim3(64), 0, stream, code, consts, out, work, tag);
    return (int)hipGetLastError();
}

int sdf_launch_bounds(int f64, int full, hipStream_t stream, const uint32_t *code, const void *consts, double *out, void *work, unsigned tag) {
    unsigned long long *w = (unsigned long long *)work;
    if (f64) return full ? launch_bounds_file<double, true>(stream, code, (const double *)consts, out, w, tag)
                         : launch_bounds_file<double, false>(stream, code, (const double *)consts, out, w, tag);
    return full ? launch_bounds_file<float, true>(stream, code, (const float *)consts, out, w, tag)
                : launch_bounds_file<float, false>(stream, code, (const float *)consts, out, w, tag);
}
