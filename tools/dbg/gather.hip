// Gather-rate probe for the vector memory pipeline (TA/TCP) of one CU: how many lane-level 16-byte gathers
// per clock, for different address patterns and table sizes.  Prints Glane-loads/s and per-CU rate per clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ buf, unsigned nlines, int iters, float* out)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned lane = threadIdx.x & 63;
    unsigned long long h = (unsigned long long)(V == 1 ? (i >> 2) : V == 4 ? (i >> 6) : i) * 0x9E3779B97F4A7C15ull + 12345;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        unsigned line = (unsigned)(h % nlines);           // 64-byte lines, 4 float4 each
        if (V == 0) acc += buf[line * 4ull].x;                                    // one 16-B gather per lane
        if (V == 1) acc += buf[line * 4ull + (lane & 3)].x;                       // quad shares a line, consecutive pieces
        if (V == 2) { const float4* p = buf + line * 4ull; acc += p[0].x + p[1].y + ((const float*)(p + 2))[0]; } // 36 B, 3 loads
        if (V == 3) acc += ((const float*)buf)[line * 16ull];                     // 4-B gather
        if (V == 4) acc += buf[line * 4ull].x;                                    // whole wave same address
        if (V == 5) { const float4* p = buf + line * 4ull; float4 a = p[0]; asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x)::"memory"); acc += a.x + p[1].y + ((const float*)(p + 2))[0]; }
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int V>
int run(const float4* buf, unsigned nlines, float* out, const char* name, int loads)
{
    const int iters = 256, blocks = 256 * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out);
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double lanes = (double)blocks * 256 * iters * loads;
    printf("  %-34s %8.3f ms  %7.1f Glane-loads/s  %.3f lane-loads/clk/CU (2.4 GHz, 256 CUs)\n", name, ms, lanes / ms / 1e6, lanes / (ms * 1e-3) / 2.4e9 / 256);
    return 0;
}

int main()
{
    float* out; CHECK(hipMalloc(&out, 4));
    const size_t sizes[] = {16u << 10, 2u << 20, 64u << 20, 1024u << 20};
    for (size_t sz : sizes) {
        float4* buf; CHECK(hipMalloc(&buf, sz)); CHECK(hipMemset(buf, 0, sz));
        unsigned nlines = (unsigned)(sz / 64);
        printf("table %zu KB\n", sz >> 10);
        if (run<0>(buf, nlines, out, "x4 gather, 64 lines/wave", 1)) return 1;
        if (run<1>(buf, nlines, out, "x4, quad shares a 64-B line", 1)) return 1;
        if (run<2>(buf, nlines, out, "36 B as x4,x4,x1 (3 loads)", 3)) return 1;
        if (run<5>(buf, nlines, out, "36 B, wait after first load", 3)) return 1;
        if (run<3>(buf, nlines, out, "4-B gather", 1)) return 1;
        if (run<4>(buf, nlines, out, "x4, wave-uniform address", 1)) return 1;
        CHECK(hipFree(buf));
    }
    return 0;
}
