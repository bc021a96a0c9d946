// fp64_peak.hip -- measures the FP64 issue rates this kernel family is bound by (vector FMA, vector MUL+ADD, MFMA f64)
// hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o /tmp/fp64_peak && /tmp/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template < int CHAINS >
__global__ __launch_bounds__(256) void fmaKernel(double* out, int iters, double a, double b)
{
    double acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        acc[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
            acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template < int CHAINS >
__global__ __launch_bounds__(256) void mulAddKernel(double* out, int iters, double a, double b)
{
    double acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        acc[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
        {
            double t;
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(acc[c]), "v"(a));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(acc[c]) : "v"(t), "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
using d4 = __attribute__((ext_vector_type(4))) double;
template < int CHAINS >
__global__ __launch_bounds__(256) void mfmaKernel(double* out, int iters, double a, double b)
{
    d4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        acc[c] = d4{0., 0., 0., 0.};
    const double av = a + threadIdx.x * 1e-6, bv = b;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template < typename F >
double timeIt(F&& f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i)
        f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e-3;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 4096;
    double*   out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    constexpr int CH = 8;
    for (int wpb : {64, 128, 256})
    {
        double t = timeIt([&] { hipLaunchKernelGGL(fmaKernel< CH >, dim3(blocks), dim3(wpb), 0, 0, out, iters, 1.0000001, 1e-9); });
        printf("v_fma_f64  block=%3d: %.2f TFLOP/s (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", wpb,
               2.0 * CH * iters * blocks * wpb / t / 1e12, t * 2.4e9 / (double(CH) * iters * (blocks * wpb / 64) / (prop.multiProcessorCount * 4)));
    }
    {
        double t = timeIt([&] { hipLaunchKernelGGL(mulAddKernel< CH >, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9); });
        printf("v_mul_f64 + v_add_f64: %.2f TFLOP/s (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n",
               2.0 * CH * iters * blocks * threads / t / 1e12, t * 2.4e9 / (2.0 * CH * iters * (blocks * threads / 64) / (prop.multiProcessorCount * 4)));
    }
    {
        double t = timeIt([&] { hipLaunchKernelGGL(mfmaKernel< 4 >, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9); });
        const double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * (blocks * threads / 64);
        printf("v_mfma_f64_16x16x4: %.2f TFLOP/s (%.2f cycles per MFMA per SIMD at 2.4 GHz)\n", flops / t / 1e12,
               t * 2.4e9 / (4.0 * iters * (blocks * threads / 64) / (prop.multiProcessorCount * 4)));
    }
    return 0;
}
