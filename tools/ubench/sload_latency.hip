// micro-benchmark (GPU box): latency of dependent scalar loads (s_load through the scalar cache)
// vs dependent LDS reads, measured with s_memtime.  Build + run: see tools/ubench/run.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_chase_scalar(const unsigned *__restrict__ next, int steps, long long *out, unsigned *sink) {
    unsigned p = 0;
    // warm the scalar cache
    for (int i = 0; i < 64; i++) p = __builtin_amdgcn_readfirstlane(next[p]);
    long long t0 = clock64();
    for (int i = 0; i < steps; i++) p = __builtin_amdgcn_readfirstlane(next[p]);
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = p; }
}

__global__ void k_chase_lds(const unsigned *__restrict__ next, int n, int steps, long long *out, unsigned *sink) {
    __shared__ unsigned tab[1024];
    for (int i = threadIdx.x; i < n; i += blockDim.x) tab[i] = next[i];
    __syncthreads();
    unsigned p = 0;
    long long t0 = clock64();
    for (int i = 0; i < steps; i++) p = __builtin_amdgcn_readfirstlane(tab[p]);
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = p; }
}

int main() {
    const int n = 256, steps = 4096;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; i++) h[i] = (i * 37 + 11) % n;
    unsigned *d_next, *d_sink; long long *d_out;
    hipMalloc(&d_next, n * 4); hipMalloc(&d_sink, 1024 * 4); hipMalloc(&d_out, 1024 * 8);
    hipMemcpy(d_next, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int blocks : {1, 256, 1024}) {
        for (int threads : {64, 1024}) {
            long long o[1024];
            hipLaunchKernelGGL(k_chase_scalar, dim3(blocks), dim3(threads), 0, 0, d_next, steps, d_out, d_sink);
            hipMemcpy(o, d_out, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < blocks; i++) s += o[i];
            printf("scalar chase: blocks %4d threads %4d -> %.1f cycles per dependent s_load\n", blocks, threads, s / blocks / steps);
            hipLaunchKernelGGL(k_chase_lds, dim3(blocks), dim3(threads), 0, 0, d_next, n, steps, d_out, d_sink);
            hipMemcpy(o, d_out, blocks * 8, hipMemcpyDeviceToHost);
            s = 0; for (int i = 0; i < blocks; i++) s += o[i];
            printf("lds chase:    blocks %4d threads %4d -> %.1f cycles per dependent ds_read + readfirstlane\n", blocks, threads, s / blocks / steps);
        }
    }
    return 0;
}
