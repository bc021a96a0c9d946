// v_fmac_f64 with a DPP row_newbcast source on gfx950: acc += src0[lane k of this lane's row of 16] * src1 -- a value held by ONE lane of
// every 16-lane row is an operand of all lanes of the row, without LDS traffic, scalar loads or extra instructions.
// (1) semantics: src0 = lane id, src1 = 1 -> acc = 16 * (lane / 16) + k;  (2) rate against the plain v_fmac_f64 (8 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA_DPP(acc, a, b, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b))
__global__ void sem(double* out)
{
    const double a = double(threadIdx.x), b = 1.;
    double       c0 = 0., c5 = 0., c15 = 0.;
    FMA_DPP(c0, a, b, 0);
    FMA_DPP(c5, a, b, 5);
    FMA_DPP(c15, a, b, 15);
    out[threadIdx.x]       = c0;
    out[64 + threadIdx.x]  = c5;
    out[128 + threadIdx.x] = c15;
}
// (3) a source lane that is switched off by EXEC: lanes 8..15 of every row are inactive, the active lanes ask for lane 12 and lane 3
__global__ void semMasked(double* out)
{
    const double a = double(threadIdx.x), b = 1.;
    double       c12 = -1., c3 = -1.;
    if ((threadIdx.x & 15) < 8)
    {
        FMA_DPP(c12, a, b, 12);
        FMA_DPP(c3, a, b, 3);
    }
    out[threadIdx.x]      = c12;
    out[64 + threadIdx.x] = c3;
}
template < bool DPP >
__global__ __launch_bounds__(64) void rate(double* p, int n)
{
    double       acc[8];
    const double a = p[threadIdx.x & 7], b = p[8 + (threadIdx.x & 7)];
    for (int j = 0; j < 8; ++j)
        acc[j] = double(j);
    for (int i = 0; i < n; ++i)
    {
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            if constexpr (DPP)
                FMA_DPP(acc[j], a, b, 7);
            else
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int j = 0; j < 8; ++j)
        s += acc[j];
    if (s == 1.2345)
        p[0] = s;
}
int main()
{
    double* d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    sem<<< 1, 64 >>>(d);
    double h[192];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; ++l)
        ok = ok && h[l] == 16 * (l / 16) + 0 && h[64 + l] == 16 * (l / 16) + 5 && h[128 + l] == 16 * (l / 16) + 15;
    std::printf("semantics (acc = src0 of lane k of the row * src1): %s   e.g. lane 37: k=0 -> %g, k=5 -> %g, k=15 -> %g\n", ok ? "as expected" : "NOT as expected", h[37], h[64 + 37], h[128 + 37]);
    semMasked<<< 1, 64 >>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::printf("source lane off (EXEC): lane 37 asks lane 12 of its row (= 44, inactive): acc -1 -> %g (%s); asks lane 3 (= 35, active): acc -1 -> %g\n", h[37],
                h[37] == 43. ? "the inactive lane's register IS read" : (h[37] == -1. ? "the FMA does not execute / adds 0" : "something else"), h[64 + 37]);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    hipMemset(d, 0, 4096);
    for (int waves_per_simd : {1, 2})
    {
        const int grid = pr.multiProcessorCount * 4 * waves_per_simd, n = 20000;
        float     t[2];
        for (int v = 0; v < 2; ++v)
        {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            if (v == 0)
                rate< false ><<< grid, 64 >>>(d, n);
            else
                rate< true ><<< grid, 64 >>>(d, n);
            hipEventRecord(e0, 0);
            if (v == 0)
                rate< false ><<< grid, 64 >>>(d, n);
            else
                rate< true ><<< grid, 64 >>>(d, n);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t[v], e0, e1);
        }
        const double inst = double(n) * 8 * waves_per_simd; // per SIMD
        std::printf("%d wave(s) per SIMD: plain v_fmac_f64 %.3f ms = %.2f ns per instruction and SIMD, with row_newbcast %.3f ms = %.2f ns  (ratio %.3f)\n",
                    waves_per_simd, t[0], t[0] * 1e6 / inst, t[1], t[1] * 1e6 / inst, t[1] / t[0]);
    }
    return 0;
}
