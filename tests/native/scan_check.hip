#include <hip/hip_runtime.h>
#include <cstdio>
#include "sdf_device.h"
using namespace sdfk;
__global__ void k(const int *in, int *out, int *tot) {
    __shared__ int ws[16];
    int t;
    out[threadIdx.x] = block_exclusive_scan<1024>(in[threadIdx.x], ws, t);
    if (threadIdx.x == 0) *tot = t;
}
int main() {
    int h[1024], o[1024], *di, *dout, *dt, t;
    for (int i = 0; i < 1024; i++) h[i] = (i * 7919) % 13;
    hipMalloc(&di, 4096); hipMalloc(&dout, 4096); hipMalloc(&dt, 4);
    hipMemcpy(di, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 1024>>>(di, dout, dt);
    hipMemcpy(o, dout, 4096, hipMemcpyDeviceToHost); hipMemcpy(&t, dt, 4, hipMemcpyDeviceToHost);
    int acc = 0, bad = 0;
    for (int i = 0; i < 1024; i++) { if (o[i] != acc) bad++; acc += h[i]; }
    printf("scan bad=%d total %d vs %d\n", bad, t, acc);
    return bad || t != acc;
}
