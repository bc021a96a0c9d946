// fma_latency.hip -- issue rate of DEPENDENT v_fma_f64 chains on gfx950: cycles per FMA for C interleaved chains, one wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fma_latency.hip -o /tmp/fma_latency
#include <hip/hip_runtime.h>

#include <cstdio>

template < int C >
__global__ __launch_bounds__(64) void chains(double* out, long long* cyc, double a, double b)
{
    double x[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
        x[c] = threadIdx.x * 1e-3 + c;
    const long long t0 = __builtin_readcyclecounter();
    const long long s0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 4096; ++it)
    {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < C; ++c)
                x[c] = __builtin_fma(x[c], a, b);
    }
    const long long s1 = __builtin_amdgcn_s_memtime();
    const long long t1 = __builtin_readcyclecounter();
    double          s  = 0.;
#pragma unroll
    for (int c = 0; c < C; ++c)
        s += x[c];
    out[threadIdx.x + 64 * blockIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        cyc[0] = t1 - t0;
        cyc[1] = s1 - s0;
    }
}

template < int C >
void run(int waves_per_simd)
{
    double*    out;
    long long* cyc;
    (void)hipMalloc(&out, 64 * 4096 * sizeof(double));
    (void)hipMalloc(&cyc, 2 * sizeof(long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(chains< C >, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(chains< C >, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double fmas = 4096. * 16 * C;
    std::printf("chains %2d, %d wave(s)/SIMD: %.2f ms, %.1f TFLOP/s, wave 0: %.2f counter ticks / %.2f memtime ticks per FMA\n", C,
                waves_per_simd, ms, fmas * 128. * blocks / (ms * 1e-3) / 1e12, double(h[0]) / fmas, double(h[1]) / fmas);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main()
{
    for (int w : {1, 2})
    {
        run< 1 >(w);
        run< 2 >(w);
        run< 3 >(w);
        run< 4 >(w);
        run< 8 >(w);
    }
    return 0;
}
