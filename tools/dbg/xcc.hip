// prints how the blocks of a launch spread over HW_REG_XCC_ID values
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* hist, unsigned* raw)
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) {
        atomicAdd(&hist[v & 15u], 1u);
        if (blockIdx.x < 32) raw[blockIdx.x] = v;
    }
}
int main()
{
    unsigned *h, *r;
    hipMalloc(&h, 64);
    hipMalloc(&r, 128);
    hipMemset(h, 0, 64);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, h, r);
    unsigned hh[16], rr[32];
    hipMemcpy(hh, h, 64, hipMemcpyDeviceToHost);
    hipMemcpy(rr, r, 128, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf("xcc %d: %u blocks\n", i, hh[i]);
    for (int i = 0; i < 32; i++) printf("%08x ", rr[i]);
    printf("\n");
    return 0;
}
