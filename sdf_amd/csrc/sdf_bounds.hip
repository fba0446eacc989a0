// sdf_bounds.hip -- `_estimate_bounds` (reference sdf/core.py:62-82) as ONE launch of ONE workgroup: up to 32 rounds of a 16^3 probe
// grid (np.linspace per axis: lo + i * step, the last sample forced to hi), threshold = |d| / 2, the box of the samples with
// |f| <= threshold, grown by half a probe cell -- float64 throughout, operation by operation like the reference (and the oracle's
// restatement).  The host loop it replaces paid a kernel launch, two copies and a synchronisation per round: 3 ms per model.
// out[0..6) = lo, hi; out[6] = 1 when a round found no sample within its threshold (the reference raises there: `where.max` of an
// empty array).
//
// The rounds are a dependent chain (a round's grid is the previous round's hit box), so what counts is the latency of ONE round: 4096
// probes through the tape.  r02 - r04: four workgroups of 1024 lanes, a probe per lane, meeting at a barrier in device memory
// after every round: 25 us per round, 0.8 ms per model -- three times the meshing of 512^3.  r05, first try: one workgroup, four
// probes per lane ONE AFTER THE OTHER (`__syncthreads` only): 24 us per round all the same -- the barrier was never the cost; a
// tape instruction of the one-sample interpreter takes ~ 500 cycles whatever it computes (the scalar fetch of its words, the
// dispatch), and four passes of ~ 25 instructions are 50 k cycles.  Hence: the four probes of a lane go through the interpreter
// TOGETHER (NS = 4: the decode is paid once per four samples) wherever the tape's register file leaves room for four float64
// samples per lane -- the (1,1) and (2,2) files, at 512 threads (eight probes per lane, two passes): every BASELINE model but
// weave / pawn -- two at a time for the (4,4) file; the (8,8) file keeps the four-workgroup scheme (k_estimate_bounds_wide below).  A translation unit of its own: a tape interpreter per
// (precision, trig, file), built in parallel with the others (build.sh), with the interpreters' structurizer option.
#include "sdf_interp.h"
#include "sdf_bounds.h"

using namespace sdfk;

template <typename T, bool FULL, int NP, int ND, int NS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_estimate_bounds(const uint32_t *__restrict__ code, const T *__restrict__ consts, double *__restrict__ out) {
    constexpr int PER_LANE = 4096 / BLOCK;
    static_assert((NS == 1 || NS == 2 || NS == 4) && PER_LANE % NS == 0, "4096 / BLOCK probes per lane, NS at a time");
    __shared__ double ax[3][16];
    __shared__ double lo[3], hi[3], d[3], thr, prev;
    __shared__ int box[6], stop;
    const int tid = threadIdx.x;
    if (tid < 3) { lo[tid] = -1e9; hi[tid] = 1e9; }
    if (tid == 0) { prev = -1.0; stop = 0; }
    __syncthreads();
    for (int it = 0; it < 32; it++) {
        if (tid < 48) {
            const int a = tid >> 4, i = tid & 15;
            const double step = (hi[a] - lo[a]) / 15.0;
            ax[a][i] = i == 15 ? hi[a] : lo[a] + (double)i * step;
        }
        if (tid < 6) box[tid] = 0;      // maxima of 16 - index (lower corner) and index + 1 (upper corner): 0 = no hit
        __syncthreads();
        if (tid == 0) {
            for (int a = 0; a < 3; a++) d[a] = ax[a][1] - ax[a][0];
            const double t = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) / 2;
            if (it > 0 && t == prev) stop = 1;
            prev = t; thr = t;
        }
        __syncthreads();
        if (stop) break;
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
#pragma unroll 1
        for (int p0 = 0; p0 < PER_LANE; p0 += NS) {
            Vec<T, NS> px, py, pz;
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const int q = (p0 + k) * BLOCK + tid;
                px.v[k] = (T)ax[0][q >> 8]; py.v[k] = (T)ax[1][(q >> 4) & 15]; pz.v[k] = (T)ax[2][q & 15];
            }
            const Vec<T, NS> v = run_tape<T, FULL, NP, ND, NS>(code, consts, px, py, pz);
            SDF_UNROLL
            for (int k = 0; k < NS; k++) {
                const int q = (p0 + k) * BLOCK + tid;
                const int i = q >> 8, j = (q >> 4) & 15, kk = q & 15;
                if (fabs((double)v.v[k]) <= thr) {
                    b0 = max(b0, 16 - i); b1 = max(b1, 16 - j); b2 = max(b2, 16 - kk);
                    b3 = max(b3, i + 1); b4 = max(b4, j + 1); b5 = max(b5, kk + 1);
                }
            }
        }
        if (b3) {
            atomicMax(&box[0], b0); atomicMax(&box[1], b1); atomicMax(&box[2], b2);
            atomicMax(&box[3], b3); atomicMax(&box[4], b4); atomicMax(&box[5], b5);
        }
        __syncthreads();
        if (box[3] == 0) { if (tid == 0) out[6] = 1.0; return; }       // no probe within the threshold
        if (tid < 3) {
            const double l0 = lo[tid];
            const int mn = 16 - box[tid], mx = box[3 + tid] - 1;
            hi[tid] = l0 + (double)mx * d[tid] + d[tid] / 2;
            lo[tid] = l0 + (double)mn * d[tid] - d[tid] / 2;
        }
        __syncthreads();
    }
    if (tid < 3) { out[tid] = lo[tid]; out[3 + tid] = hi[tid]; }
    if (tid == 0) out[6] = 0.0;
}

// The (8,8) register file -- the deepest tapes (weave, pawn: saved points nested four deep and more) -- leaves room for ONE float64
// sample per lane: four passes per lane in one workgroup were measured at 15 ms for weave against 4 ms (r03), so these tapes keep
// the r02 scheme: FOUR workgroups share a round's 4096 probes (one per lane) and keep in lockstep through a counter in device
// memory (they are co-resident on any gfx950: four workgroups, 256 compute units); every workgroup carries the whole state -- the
// same arithmetic on the same reduced indices -- so nothing but the hit box is exchanged: work[0] = arrivals at the barrier,
// work[1 + 6 * round ..] = the round's hit box, as maxima of 16 - index (lower corner) and index + 1 (upper corner) so that a
// zeroed buffer is "no hit".  out[6] = 2: the workgroups did not meet (the device was held by other kernels for seconds).
template <typename T, bool FULL>
__global__ __launch_bounds__(1024) void k_estimate_bounds_wide(const uint32_t *__restrict__ code, const T *__restrict__ consts, double *__restrict__ out,
                                                          int *__restrict__ work) {
    // FOUR workgroups share a round's 4096 probes (one per lane) and keep in lockstep through a counter in device
    // memory (they are co-resident on any gfx950: four workgroups, 256 compute units); every workgroup carries the
    // whole state -- the same arithmetic on the same reduced indices -- so nothing but the hit box is exchanged:
    // work[0] = arrivals at the barrier, work[1 + 6 * round ..] = the round's hit box, as maxima of 16 - index
    // (lower corner) and index + 1 (upper corner) so that a zeroed buffer is "no hit"
    __shared__ double ax[3][16];
    __shared__ double lo[3], hi[3], d[3], thr, prev;
    __shared__ int box[6], stop;
    const int tid = threadIdx.x;
    if (tid < 3) { lo[tid] = -1e9; hi[tid] = 1e9; }
    if (tid == 0) { prev = -1.0; stop = 0; }
    __syncthreads();
    for (int it = 0; it < 32; it++) {
        if (tid < 48) {
            const int a = tid >> 4, i = tid & 15;
            const double step = (hi[a] - lo[a]) / 15.0;
            ax[a][i] = i == 15 ? hi[a] : lo[a] + (double)i * step;
        }
        if (tid < 6) box[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            for (int a = 0; a < 3; a++) d[a] = ax[a][1] - ax[a][0];
            const double t = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) / 2;
            if (it > 0 && t == prev) stop = 1;
            prev = t; thr = t;
        }
        __syncthreads();
        if (stop) break;                                       // (every workgroup takes the same decision)
        {
            const int q = (int)blockIdx.x * 1024 + tid;
            const int i = q >> 8, j = (q >> 4) & 15, k = q & 15;
            const double v = (double)run_tape1<T, FULL>(code, consts, (T)ax[0][i], (T)ax[1][j], (T)ax[2][k]);
            if (fabs(v) <= thr) {
                atomicMax(&box[0], 16 - i); atomicMax(&box[1], 16 - j); atomicMax(&box[2], 16 - k);
                atomicMax(&box[3], i + 1); atomicMax(&box[4], j + 1); atomicMax(&box[5], k + 1);
            }
        }
        __syncthreads();
        int *slot = work + 1 + 6 * it;
        if (tid < 6 && box[tid]) atomicMax(&slot[tid], box[tid]);
        __syncthreads();
        if (tid == 0) {                                         // the round's barrier over the four workgroups
            __threadfence();
            atomicAdd(&work[0], 1);
            unsigned spins = 0;
            while (__hip_atomic_load(&work[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4 * (it + 1) && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(2);
            // (a workgroup that gives up -- the four were not co-resident for seconds: other kernels held the compute
            // units -- must not go on with a partial hit box: workgroup 0 reports it, the host falls back to its loop)
            if (spins >= (1u << 26)) stop = 2;
            __threadfence();
        }
        __syncthreads();
        if (stop == 2) { if (blockIdx.x == 0 && tid == 0) out[6] = 2.0; return; }
        if (tid < 6) box[tid] = __hip_atomic_load(&slot[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (box[3] == 0) { if (blockIdx.x == 0 && tid == 0) out[6] = 1.0; return; }       // no probe within the threshold
        if (tid < 3) {
            const double l0 = lo[tid];
            const int mn = 16 - box[tid], mx = box[3 + tid] - 1;
            hi[tid] = l0 + (double)mx * d[tid] + d[tid] / 2;
            lo[tid] = l0 + (double)mn * d[tid] - d[tid] / 2;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (tid < 3) { out[tid] = lo[tid]; out[3 + tid] = hi[tid]; }
        if (tid == 0) out[6] = 0.0;
    }
}

template <typename T, bool FULL>
static int launch_bounds_file(int slots, hipStream_t stream, const uint32_t *code, const T *consts, double *out, int *work) {
    // (four float64 samples per lane need ~ 200 vector registers with the small files: 512 threads -- two waves per SIMD, the whole
    // register file of a lane -- and eight probes per lane in two passes; at 1024 threads the compiler spilled 200 - 700 registers)
    switch (slots) {
    case 0: hipLaunchKernelGGL((k_estimate_bounds<T, FULL, 1, 1, 4, 512>), dim3(1), dim3(512), 0, stream, code, consts, out); break;
    case 1: hipLaunchKernelGGL((k_estimate_bounds<T, FULL, 2, 2, 4, 512>), dim3(1), dim3(512), 0, stream, code, consts, out); break;
    case 2: hipLaunchKernelGGL((k_estimate_bounds<T, FULL, 4, 4, 2, 512>), dim3(1), dim3(512), 0, stream, code, consts, out); break;
    default:
        if (hipMemsetAsync(work, 0, (1 + 6 * 32) * sizeof(int), stream) != hipSuccess) return (int)hipGetLastError();
        hipLaunchKernelGGL((k_estimate_bounds_wide<T, FULL>), dim3(4), dim3(1024), 0, stream, code, consts, out, work);
        break;
    }
    return (int)hipGetLastError();
}

int sdf_launch_bounds(int f64, int full, int slots, hipStream_t stream, const uint32_t *code, const void *consts, double *out, int *work) {
    if (f64) return full ? launch_bounds_file<double, true>(slots, stream, code, (const double *)consts, out, work)
                         : launch_bounds_file<double, false>(slots, stream, code, (const double *)consts, out, work);
    // (float32 probing -- `Engine.precision` for sdf_eval_* -- keeps the one-sample interpreter: not a hot path)
    return full ? launch_bounds_file<float, true>(3, stream, code, (const float *)consts, out, work)
                : launch_bounds_file<float, false>(3, stream, code, (const float *)consts, out, work);
}
