// fp64_coissue.hip -- does the FP64 matrix pipe (v_mfma_f64_*) run beside the FP64 vector pipe (v_fma_f64) on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_coissue.hip -o tools/fp64_coissue && tools/fp64_coissue
// Cases, each on every CU with one 512-thread workgroup (8 waves: waves 0-3 and 4-7 land on the four SIMDs once each):
//   V    waves 0-3 run a v_fma_f64 loop, waves 4-7 exit            (one vector wave per SIMD)
//   VV   all eight waves run the v_fma_f64 loop                      (two vector waves per SIMD)
//   M    waves 4-7 run a v_mfma_f64_16x16x4 loop, waves 0-3 exit   (one matrix wave per SIMD)
//   V|M  waves 0-3 vector, waves 4-7 matrix                          (one of each per SIMD: the co-execution question)
//   m4   as M with v_mfma_f64_4x4x4_4b
//   V|m4 as V|M with the 4x4x4 form
//   I<n> every wave: n independent v_fma_f64 between two MFMAs (one wave per SIMD; in-wave interleave)
// Reported: wall time per case, TFLOP/s of each part, and (V|M) / max(V, M): 1.0 = the pipes run side by side, 2.0 = they
// serialise.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int CH = 8; // independent chains per wave

__device__ __forceinline__ double fmaLoop(int iters, double a, double b)
{
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
        acc[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        s += acc[c];
    return s;
}
__device__ __forceinline__ double mfma16Loop(int iters, double a, double b)
{
    d4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        acc[c] = d4{0., 0., 0., 0.};
    const double av = a + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[c], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    return s;
}
__device__ __forceinline__ double mfma4Loop(int iters, double a, double b)
{
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        acc[c] = 0.;
    const double av = a + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, acc[c], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        s += acc[c];
    return s;
}
// mode bits: 1 = waves 0-3 vector, 2 = waves 4-7 vector, 4 = waves 4-7 mfma16, 8 = waves 4-7 mfma4
__global__ __launch_bounds__(512) void mixKernel(double* out, int mode, int it_v, int it_m, double a, double b)
{
    const int wave = threadIdx.x >> 6;
    double    s    = 0.;
    if (wave < 4)
    {
        if (mode & 1)
            s = fmaLoop(it_v, a, b);
    }
    else
    {
        if (mode & 2)
            s = fmaLoop(it_v, a, b);
        else if (mode & 4)
            s = mfma16Loop(it_m, a, b);
        else if (mode & 8)
            s = mfma4Loop(it_m, a, b);
    }
    if (s == 1.2345e300)
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// in-wave interleave: NF independent FMAs after every MFMA (16x16x4), one wave per SIMD
template < int NF, bool SMALL >
__global__ __launch_bounds__(256) void interleaveKernel(double* out, int iters, double a, double b)
{
    d4     macc[2] = {d4{0., 0., 0., 0.}, d4{0., 0., 0., 0.}};
    double sacc[2] = {0., 0.};
    double acc[NF > 0 ? NF : 1];
#pragma unroll
    for (int c = 0; c < NF; ++c)
        acc[c] = threadIdx.x * 1e-3 + c;
    const double av = a + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int m = 0; m < 2; ++m)
        {
            if constexpr (SMALL)
                sacc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, sacc[m], 0, 0, 0);
            else
                macc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, macc[m], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NF; ++c)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
        }
    }
    double s = macc[0][0] + macc[1][1] + sacc[0] + sacc[1];
#pragma unroll
    for (int c = 0; c < NF; ++c)
        s += acc[c];
    if (s == 1.2345e300)
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// in-kernel clock and issue interval of a pure v_fma_f64 loop: W waves per SIMD (blockDim = 256 * W), `lanes` active lanes per wave.
// Every wave stamps its start and end (s_memtime, shader cycles; s_memrealtime, 100 MHz): the SIMD's arbiter favours the
// oldest wave, so one wave's own duration says nothing about the others; the block's span max(end) - min(start) does.
__global__ __launch_bounds__(1024) void clockKernel(double* out, long long* clk, int iters, int lanes, double a, double b)
{
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double          s  = 0.;
    if ((threadIdx.x & 63) < lanes)
        s = fmaLoop(iters, a, b);
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0)
    {
        long long* c = clk + 4 * (blockIdx.x * 16 + (threadIdx.x >> 6));
        c[0] = t0, c[1] = t1, c[2] = r0, c[3] = r1;
    }
    if (s == 1.2345e300)
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
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    double* out;
    hipMalloc(&out, sizeof(double) * cus * 512);
    const int    it_v = 8192, it_m = 4096;
    const double fl_v = 2.0 * 64 * CH * it_v * 4.0 * cus;             // flops of four vector waves per CU
    const double fl_m16 = 2.0 * 16 * 16 * 4 * 4.0 * it_m * 4.0 * cus; // four matrix waves per CU, 16x16x4
    const double fl_m4  = 2.0 * 4 * 4 * 4 * 4 * 8.0 * it_m * 4.0 * cus;
    auto run = [&](int mode) { return timeIt([&] { hipLaunchKernelGGL(mixKernel, dim3(cus), dim3(512), 0, 0, out, mode, it_v, it_m, 1.0000001, 1e-9); }); };
    const double tV = run(1), tVV = run(1 | 2), tM = run(4), tVM = run(1 | 4), tm4 = run(8), tVm4 = run(1 | 8);
    printf("V    %.3f ms  vector %.1f TFLOP/s\n", tV * 1e3, fl_v / tV / 1e12);
    printf("VV   %.3f ms  vector %.1f TFLOP/s\n", tVV * 1e3, 2 * fl_v / tVV / 1e12);
    printf("M    %.3f ms  matrix(16x16x4) %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", tM * 1e3, fl_m16 / tM / 1e12, tM * 2.4e9 / (4.0 * it_m));
    printf("V|M  %.3f ms  vector %.1f + matrix %.1f = %.1f TFLOP/s;  (V|M) / max(V, M) = %.2f, / (V + M) = %.2f\n", tVM * 1e3, fl_v / tVM / 1e12,
           fl_m16 / tVM / 1e12, (fl_v + fl_m16) / tVM / 1e12, tVM / (tV > tM ? tV : tM), tVM / (tV + tM));
    printf("m4   %.3f ms  matrix(4x4x4_4b) %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", tm4 * 1e3, fl_m4 / tm4 / 1e12, tm4 * 2.4e9 / (8.0 * it_m));
    printf("V|m4 %.3f ms  vector %.1f + matrix %.1f = %.1f TFLOP/s;  (V|m4) / max = %.2f, / sum = %.2f\n", tVm4 * 1e3, fl_v / tVm4 / 1e12,
           fl_m4 / tVm4 / 1e12, (fl_v + fl_m4) / tVm4 / 1e12, tVm4 / (tV > tm4 ? tV : tm4), tVm4 / (tV + tm4));
    {
        long long* clk;
        hipMalloc(&clk, sizeof(long long) * 4 * 16 * cus);
        std::vector< long long > h(4 * 16 * cus);
        for (int lanes : {64, 49})
            for (int W : {1, 2, 3, 4})
            {
                const int it = 16384, reps = 40;
                auto      go = [&] { hipLaunchKernelGGL(clockKernel, dim3(cus), dim3(256 * W), 0, 0, out, clk, it, lanes, 1.0000001, 1e-9); };
                for (int rep = 0; rep < reps; ++rep) // ~0.1 s of back-to-back launches so that the clock settles under load
                    go();
                hipDeviceSynchronize();
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                hipEventRecord(e0);
                for (int rep = 0; rep < reps; ++rep)
                    go();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h.data(), clk, sizeof(long long) * 4 * 16 * cus, hipMemcpyDeviceToHost);
                std::vector< double > ghz, cyc, first;
                for (int i = 0; i < cus; ++i)
                {
                    long long t0 = h[4 * 16 * i], t1 = h[4 * 16 * i + 1];
                    for (int w = 1; w < 4 * W; ++w)
                    {
                        t0 = std::min(t0, h[4 * (16 * i + w)]);
                        t1 = std::max(t1, h[4 * (16 * i + w) + 1]);
                    }
                    const long long* c = &h[4 * 16 * i];
                    ghz.push_back(double(c[1] - c[0]) / double(c[3] - c[2]) * 0.1);
                    cyc.push_back(double(t1 - t0) / (double(CH) * it * W));       // shader cycles per FMA wave-instruction per SIMD
                    first.push_back(double(c[1] - c[0]) / (double(CH) * it));   // wave 0 alone: its own issue interval
                }
                std::sort(ghz.begin(), ghz.end());
                std::sort(cyc.begin(), cyc.end());
                std::sort(first.begin(), first.end());
                const double flops = 2.0 * lanes * CH * it * 4.0 * W * cus;
                printf("clock: %d waves/SIMD, %d lanes: in-kernel clock %.2f GHz; block span: %.2f shader cycles per v_fma_f64 per SIMD (wave 0 alone issues one per %.2f); "
                       "events: %.3f ms per launch = %.1f TFLOP/s (%.1f at 64 lanes)\n",
                       W, lanes, ghz[cus / 2], cyc[cus / 2], first[cus / 2], ms / reps, flops / (ms / reps * 1e-3) / 1e12, flops * 64 / lanes / (ms / reps * 1e-3) / 1e12);
            }
    }
    auto inter = [&](auto kern, int nf, bool small) {
        const int    it = 4096;
        const double t  = timeIt([&] { hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 0, 0, out, it, 1.0000001, 1e-9); });
        const double fm = (small ? 2.0 * 4 * 4 * 4 * 4 : 2.0 * 16 * 16 * 4) * 2.0 * it * 4.0 * cus, fv = 2.0 * 64 * nf * 2.0 * it * 4.0 * cus;
        printf("I%-2d%s %.3f ms  %.1f cycles per (MFMA + %d FMA) at 2.4 GHz;  matrix %.1f + vector %.1f = %.1f TFLOP/s\n", nf, small ? "s" : " ", t * 1e3,
               t * 2.4e9 / (2.0 * it), nf, fm / t / 1e12, fv / t / 1e12, (fm + fv) / t / 1e12);
    };
    inter(interleaveKernel< 0, false >, 0, false);
    inter(interleaveKernel< 4, false >, 4, false);
    inter(interleaveKernel< 8, false >, 8, false);
    inter(interleaveKernel< 12, false >, 12, false);
    inter(interleaveKernel< 16, false >, 16, false);
    inter(interleaveKernel< 0, true >, 0, true);
    inter(interleaveKernel< 2, true >, 2, true);
    inter(interleaveKernel< 4, true >, 4, true);
    inter(interleaveKernel< 8, true >, 8, true);
    return 0;
}
