// L1 line-size probe: every lane reads a dword at a random 128-B aligned address, then (variant 1) the dword
// 64 bytes further.  If TCP_TCC_READ_REQ does not grow in variant 1, a vector-L1 miss fetches 128 B.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const unsigned* __restrict__ buf, unsigned long long nlines, int variant, unsigned* out)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = i * 0x9E3779B97F4A7C15ull;
    unsigned acc = 0;
    for (int it = 0; it < 16; it++) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        unsigned long long line = h % nlines;
        acc += buf[line * 32];
        if (variant == 1) acc += buf[line * 32 + 16];
        if (variant == 2) acc += buf[line * 32 + 4];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv)
{
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    unsigned long long nlines = 1ull << 21; // 256 MB
    unsigned *buf, *out;
    if (hipMalloc(&buf, nlines * 128) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, nlines * 128);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, buf, nlines, variant, out);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("variant %d done\n", variant);
    return 0;
}
