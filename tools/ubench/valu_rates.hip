// micro-benchmark (GPU box): issue cost of the vector instructions k_mesh is made of, per SIMD, at one and at four waves per SIMD
// (k_mesh's occupancy), with eight independent chains per wave (throughput) and with one dependent chain (latency).
// Build + run: see tools/ubench/run.sh.  Prints cycles per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// OPS(name, asm with operands d (dst/src pair), s (second source pair), 64-bit or 32-bit registers)
template <int OP, bool DEP>
__global__ __launch_bounds__(1024) void k_rate(long long *out, double *sink, int iters) {
    double d[8], s = 1.000001 + threadIdx.x * 1e-9;
    float f[8], fs = 1.0001f + threadIdx.x * 1e-6f;
    unsigned u[8], us = 3u + (threadIdx.x & 3u);
    unsigned long long q[8];
    for (int i = 0; i < 8; i++) { d[i] = 1.0 + i * 0.125 + threadIdx.x * 1e-7; f[i] = 1.0f + i * 0.125f; u[i] = 0x9E3779B9u * (i + 1) + threadIdx.x; q[i] = 0x9E3779B97F4A7C15ull * (i + 1) + threadIdx.x; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define IDX(i) (DEP ? 0 : i)
#define ONE(i) \
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[IDX(i)]) : "v"(s)); \
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[IDX(i)]) : "v"(s)); \
        if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[IDX(i)]) : "v"(s)); \
        if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[IDX(i)]) : "v"(fs)); \
        if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 5) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[IDX(i)]) : "v"(us)); \
        if (OP == 6) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q[IDX(i)]) : "v"(us)); \
        if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 10) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[IDX(i)]) : "v"(f[IDX(i)])); \
        if (OP == 11) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f[IDX(i)]) : "v"(d[IDX(i)])); \
        if (OP == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[IDX(i)])); \
        if (OP == 13) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[IDX(i)])); \
        if (OP == 14) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[IDX(i)])); \
        if (OP == 15) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[IDX(i)]) : "v"(us) : ); \
        if (OP == 17) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[IDX(i)]) : "v"(s)); \
        if (OP == 18) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[IDX(i)]), "v"(s) : "vcc"); \
        if (OP == 19) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 20) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 21) asm volatile("v_bfe_u32 %0, %0, %1, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 22) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[IDX(i)]) : "v"(u[IDX(i)])); \
        if (OP == 23) asm volatile("v_mov_b32 %0, %1" : "+v"(u[IDX(i)]) : "v"(us)); \
        if (OP == 24) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[IDX(i)]) : "v"(s) : "vcc"); \
        if (OP == 25) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(d[IDX(i)]) : "v"(s)); \
        if (OP == 26) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[IDX(i)]) : "v"(us)); \
        if (OP == 27) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[IDX(i)]) : "v"(s)); \
        if (OP == 28) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sreg[IDX(i) & 3]) : : "scc");
        unsigned sreg[4] = {1u, 2u, 3u, 4u};
        REP8(ONE) REP8(ONE)
        if (OP == 28) asm volatile("" :: "s"(sreg[0]), "s"(sreg[1]), "s"(sreg[2]), "s"(sreg[3]));
    }
    const long long t1 = clock64();
    double acc = 0;
    for (int i = 0; i < 8; i++) acc += d[i] + f[i] + u[i] + (double)q[i];
    if (acc == 12345.678) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP, bool DEP>
static void run(const char *name, long long *d_out, double *d_sink) {
    const int iters = 256;      // x 16 instructions
    for (int threads : {256, 1024}) {   // 1 and 4 waves per SIMD, one workgroup per CU
        hipLaunchKernelGGL((k_rate<OP, DEP>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
        (void)hipDeviceSynchronize();
        const int nw = 256 * threads / 64;
        std::vector<long long> o(nw);
        (void)hipMemcpy(o.data(), d_out, nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nw; i++) s += (double)o[i];
        const double per_wave = s / nw / (iters * 16.0);
        printf("%-16s %s  %d wave(s)/SIMD: %6.2f cycles per instruction and wave -> %6.2f per instruction and SIMD\n", name, DEP ? "dependent  " : "independent", threads / 256,
               per_wave, per_wave / (threads / 256));
    }
}

#define BOTH(OP, NAME) run<OP, false>(NAME, d_out, d_sink); run<OP, true>(NAME, d_out, d_sink);
int main() {
    long long *d_out; double *d_sink;
    (void)hipMalloc(&d_out, 256 * 16 * 8); (void)hipMalloc(&d_sink, 8);
    BOTH(0, "v_add_f64") BOTH(1, "v_mul_f64") BOTH(2, "v_fma_f64") BOTH(3, "v_add_f32") BOTH(4, "v_add_u32") BOTH(5, "v_lshrrev_b64") BOTH(6, "v_lshlrev_b64")
    BOTH(7, "v_alignbit_b32") BOTH(8, "v_mul_lo_u32") BOTH(9, "v_mad_u32_u24") BOTH(10, "v_cvt_f64_f32") BOTH(11, "v_cvt_f32_f64") BOTH(12, "v_rcp_f64")
    BOTH(13, "v_sqrt_f64") BOTH(14, "v_rsq_f64") BOTH(15, "v_bcnt_u32_b32") BOTH(16, "v_cndmask_b32") BOTH(17, "v_min_f64") BOTH(18, "v_cmp_lt_f64") BOTH(19, "v_and_b32")
    BOTH(20, "v_add3_u32") BOTH(21, "v_bfe_u32") BOTH(22, "v_cvt_f64_u32") BOTH(23, "v_mov_b32") BOTH(24, "v_div_scale_f64") BOTH(25, "v_div_fixup_f64") BOTH(26, "v_ldexp_f64")
    BOTH(27, "v_pk_add_f32") BOTH(28, "s_add_u32")
    return 0;
}
