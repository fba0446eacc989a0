// micro-benchmark (GPU box), second sheet: the selects, compares, lane moves and LDS reads around k_mesh's arithmetic -- in particular what a
// v_cndmask_b32 costs next to a plain VALU instruction (valu_rates.hip measured 12 cycles per SIMD against 2 - 3).  Each body is sixteen
// instructions (or pairs / triples as named) over eight independent registers; printed: cycles per BODY ELEMENT and SIMD at 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(1024) void k_rate(long long *out, double *sink, int iters) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    double d[8], s = 1.000001 + threadIdx.x * 1e-9;
    unsigned u[8], us = 3u + (threadIdx.x & 3u), addr = (threadIdx.x * 4u) & 16380u;
    for (int i = 0; i < 8; i++) { d[i] = 1.0 + i * 0.125 + threadIdx.x * 1e-7; u[i] = 0x9E3779B9u * (i + 1) + threadIdx.x; }
    unsigned sr4[4] = {1u, 2u, 3u, 4u};
    unsigned long long m = 0x5555555555555555ull;
    asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define ONE(i) \
        if (OP == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(us)); \
        if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(us), "s"(m)); \
        if (OP == 2) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32_e32 %2, %2, %3, vcc" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us) : "vcc"); \
        if (OP == 3) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(us) : "vcc"); \
        if (OP == 4) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i])); \
        if (OP == 5) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i])); \
        if (OP == 6) { unsigned sr; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr) : "v"(u[i])); asm volatile("" :: "s"(sr)); } \
        if (OP == 7) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(addr)); \
        if (OP == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(addr & ~7u)); \
        if (OP == 9) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(us)); \
        if (OP == 10) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(us)); \
        if (OP == 11) asm volatile("v_ffbl_b32 %0, %0" : "+v"(u[i])); \
        if (OP == 12) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(us)); \
        if (OP == 13) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[i])); \
        if (OP == 14) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(s)); \
        if (OP == 15) asm volatile("v_cmp_u_f64 vcc, %0, %1" : : "v"(d[i]), "v"(s) : "vcc"); \
        if (OP == 16) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\tv_cndmask_b32_e32 %4, %4, %3, vcc" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us), "v"(u[(i + 1) & 7]) : "vcc"); \
        if (OP == 17) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(us)); \
        if (OP == 18) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(us)); \
        if (OP == 19) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(us)); \
        if (OP == 20) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(us) : "vcc"); \
        if (OP == 21) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(u[i])); \
        if (OP == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(us)); \
        if (OP == 23) asm volatile("ds_write_b32 %1, %0" : : "v"(u[i]), "v"(addr)); \
        if (OP == 40) asm volatile("v_add_f64 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(d[i]), "+s"(sr4[i & 3]) : "v"(s) : "scc"); \
        if (OP == 41) asm volatile("v_add_f64 %0, %0, %3\n\ts_add_u32 %1, %1, 3\n\ts_lshl_b32 %2, %2, 1" : "+v"(d[i]), "+s"(sr4[i & 3]), "+s"(sr4[(i + 1) & 3]) : "v"(s) : "scc"); \
        if (OP == 42) asm volatile("v_add_f64 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(d[i]) : "v"(s)); \
        if (OP == 43) asm volatile("v_add_f64 %0, %0, %1\n\ts_nop 0" : "+v"(d[i]) : "v"(s)); \
        if (OP == 44) asm volatile("v_add_f64 %0, %0, %1\n\ts_and_saveexec_b64 s[20:21], exec\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(d[i]) : "v"(s) : "s20", "s21", "scc"); \
        if (OP == 45) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(u[i]), "+s"(sr4[i & 3]) : "v"(us) : "scc"); \
        if (OP == 46) asm volatile("v_add_f64 %0, %0, %2\n\tv_add_u32 %1, %1, %3" : "+v"(d[i]), "+v"(u[i]) : "v"(s), "v"(us)); \
        if (OP == 47) asm volatile("s_cmp_eq_u32 %0, 77\n\ts_cbranch_scc1 1f\n\tv_add_f64 %1, %1, %2\n1:" : : "s"(sr4[0]), "v"(d[i]), "v"(s) : "scc"); \
        if (OP == 48) asm volatile("v_add_f64 %0, %0, %2\n\tds_read_b32 %1, %3" : "+v"(d[i]), "=v"(u[i]) : "v"(s), "v"(addr)); \
        if (OP == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(us)); \
        if (OP == 25) asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %2, %2, %3, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %3, s[20:21]" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us), "v"(u[(i + 1) & 7]) : "s20", "s21"); \
        if (OP == 26) asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1\n\tv_cmp_u_f64_e64 s[22:23], %0, %0\n\ts_or_b64 s[20:21], s[20:21], s[22:23]\n\tv_cndmask_b32_e64 %2, %2, %3, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %3, s[20:21]" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us), "v"(u[(i + 1) & 7]) : "s20", "s21", "s22", "s23", "scc"); \
        if (OP == 27) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cmp_u_f64_e64 s[22:23], %0, %0\n\ts_or_b64 vcc, vcc, s[22:23]\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\tv_cndmask_b32_e32 %4, %4, %3, vcc" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us), "v"(u[(i + 1) & 7]) : "vcc", "s22", "s23", "scc"); \
        if (OP == 28) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32_e64 %2, %2, %3, vcc\n\tv_cndmask_b32_e64 %4, %4, %3, vcc" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us), "v"(u[(i + 1) & 7]) : "vcc"); \
        if (OP == 29) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(s)); \
        if (OP == 30) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(us)); \
        if (OP == 31) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\ts_mov_b64 s[20:21], vcc\n\tv_cndmask_b32_e64 %2, %2, %3, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %3, s[20:21]" : : "v"(d[i]), "v"(s), "v"(u[i]), "v"(us), "v"(u[(i + 1) & 7]) : "vcc", "s20", "s21");
        R8(ONE) R8(ONE)
        if (OP == 7 || OP == 8 || OP == 48) asm volatile("s_waitcnt lgkmcnt(0)");
        if (OP >= 40) asm volatile("" :: "s"(sr4[0]), "s"(sr4[1]), "s"(sr4[2]), "s"(sr4[3]));
    }
    const long long t1 = clock64();
    double acc = 0;
    for (int i = 0; i < 8; i++) acc += d[i] + u[i];
    if (acc == 12345.678) sink[0] = acc + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
static void run(const char *name, long long *d_out, double *d_sink) {
    const int iters = 256;
    for (int threads : {256, 1024}) {
        hipLaunchKernelGGL((k_rate<OP>), dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
        (void)hipDeviceSynchronize();
        const int nw = 256 * threads / 64;
        std::vector<long long> o(nw);
        (void)hipMemcpy(o.data(), d_out, nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nw; i++) s += (double)o[i];
        const double per_wave = s / nw / (iters * 16.0);
        printf("%-44s %d wave(s)/SIMD: %6.2f cycles per element and wave -> %6.2f per element and SIMD\n", name, threads / 256, per_wave, per_wave / (threads / 256));
    }
}
int main() {
    long long *d_out; double *d_sink;
    (void)hipMalloc(&d_out, 256 * 16 * 8); (void)hipMalloc(&d_sink, 8);
    run<0>("v_cndmask_b32_e32 (vcc set)", d_out, d_sink); run<1>("v_cndmask_b32_e64 (sgpr pair)", d_out, d_sink); run<21>("v_cndmask_b32_e32 0, v, vcc", d_out, d_sink);
    run<2>("v_cmp_lt_f64 + v_cndmask", d_out, d_sink); run<16>("v_cmp_lt_f64 + 2 x v_cndmask", d_out, d_sink); run<3>("v_cmp_lt_u32", d_out, d_sink); run<15>("v_cmp_u_f64", d_out, d_sink);
    run<14>("v_max_f64", d_out, d_sink); run<17>("v_bfi_b32", d_out, d_sink); run<20>("v_addc_co_u32", d_out, d_sink);
    run<4>("v_mov_b32_dpp row_shr", d_out, d_sink); run<5>("v_add_u32_dpp row_shr", d_out, d_sink); run<6>("v_readfirstlane_b32", d_out, d_sink);
    run<7>("ds_read_b32", d_out, d_sink); run<8>("ds_read_b64", d_out, d_sink); run<23>("ds_write_b32", d_out, d_sink);
    run<9>("v_lshlrev_b32", d_out, d_sink); run<10>("v_or_b32", d_out, d_sink); run<22>("v_xor_b32", d_out, d_sink); run<11>("v_ffbl_b32", d_out, d_sink); run<12>("v_min_u32", d_out, d_sink);
    run<24>("v_cndmask_b32_e64 with vcc operand", d_out, d_sink); run<25>("v_cmp_e64 -> s pair + 2 x v_cndmask_e64", d_out, d_sink);
    run<26>("2 v_cmp_e64 + s_or + 2 x v_cndmask_e64 (sgpr)", d_out, d_sink); run<27>("2 v_cmp + s_or vcc + 2 x v_cndmask_e32 (vcc)", d_out, d_sink);
    run<28>("v_cmp vcc + 2 x v_cndmask_e64 vcc", d_out, d_sink); run<31>("v_cmp vcc + s_mov to s pair + 2 x cndmask_e64", d_out, d_sink); run<29>("v_div_fmas_f64", d_out, d_sink); run<30>("v_cndmask_b32_dpp vcc", d_out, d_sink);
    run<40>("v_add_f64 + s_add_u32", d_out, d_sink); run<41>("v_add_f64 + 2 SALU", d_out, d_sink); run<45>("v_add_u32 + s_add_u32", d_out, d_sink); run<42>("v_add_f64 + s_waitcnt", d_out, d_sink);
    run<43>("v_add_f64 + s_nop", d_out, d_sink); run<44>("v_add_f64 + saveexec + s_or exec", d_out, d_sink); run<46>("v_add_f64 + v_add_u32", d_out, d_sink);
    run<47>("s_cmp + s_cbranch (not taken) + v_add_f64", d_out, d_sink); run<48>("v_add_f64 + ds_read_b32", d_out, d_sink);
    run<13>("v_cvt_f32_u32", d_out, d_sink); run<18>("v_mul_u32_u24", d_out, d_sink); run<19>("v_sub_u32", d_out, d_sink);
    return 0;
}
