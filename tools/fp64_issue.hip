// fp64_issue.hip -- issue interval of FP64 vector instructions for ONE wave per SIMD and for W waves per SIMD, by operand
// form (all-VGPR, SGPR coefficient, inline constant), measured in shader cycles (s_memtime) over the block's span.
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_issue.hip -o tools/fp64_issue && tools/fp64_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int CH = 8, UNR = 16;
enum Form { FMA_VVV, FMA_VSV, FMA_VSS, FMAC_VS, FMAC_VV, MUL_VV, MUL_VS, ADD_VV, FMA_F32, MOV_B32, N_FORMS };
const char* const names[N_FORMS] = {"v_fma_f64 v,v,v,v", "v_fma_f64 v,v,s,v", "v_fma_f64 v,v,s,s", "v_fmac_f64 v,s,v (acc += s*v)", "v_fmac_f64 v,v,v",
                                    "v_mul_f64 v,v,v", "v_mul_f64 v,s,v", "v_add_f64 v,v,v", "v_fma_f32 v,v,v,v", "v_mov_b32 v,v"};

template < int FORM >
__global__ __launch_bounds__(1024) void issueKernel(double* out, long long* clk, int iters, double a_, double b_)
{
    double acc[CH], x[CH];
    float  facc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
    {
        acc[c]  = threadIdx.x * 1e-3 + c;
        x[c]    = 1.0 + 1e-9 * (threadIdx.x + c);
        facc[c] = float(acc[c]);
    }
    double a = a_, b = b_;
    asm volatile("" : "+v"(a), "+v"(b));
    const float fa = float(a_), fb = float(b_);
    const long long t0 = __builtin_amdgcn_s_memtime();
    // UNR x CH instructions between two branches (a taken branch costs ~28 cycles of a lone wave's issue: with 8
    // instructions per branch every form read 7.5 cycles per instruction)
    for (int i = 0; i < iters; i += UNR)
    {
#pragma unroll
        for (int cc = 0; cc < CH * UNR; ++cc)
        {
            const int c = cc % CH;
            if constexpr (FORM == FMA_VVV)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
            else if constexpr (FORM == FMA_VSV)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "s"(a_), "v"(b));
            else if constexpr (FORM == FMA_VSS)
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(acc[c]) : "s"(a_));
            else if constexpr (FORM == FMAC_VS)
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[c]) : "s"(b_), "v"(x[c]));
            else if constexpr (FORM == FMAC_VV)
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[c]) : "v"(b), "v"(x[c]));
            else if constexpr (FORM == MUL_VV)
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(a));
            else if constexpr (FORM == MUL_VS)
                asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[c]) : "s"(a_));
            else if constexpr (FORM == ADD_VV)
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
            else if constexpr (FORM == FMA_F32)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(facc[c]) : "v"(fa), "v"(fb));
            else
                asm volatile("v_mov_b32 %0, %1" : "+v"(facc[c]) : "v"(fa));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
    {
        long long* c = clk + 2 * (blockIdx.x * 16 + (threadIdx.x >> 6));
        c[0] = t0, c[1] = t1;
    }
    double s = 0.;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        s += acc[c] + facc[c] + x[c];
    if (s == 1.2345e300)
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template < int FORM >
void run(double* out, long long* clk, int cus)
{
    std::vector< long long > h(2 * 16 * cus);
    printf("%-32s", names[FORM]);
    for (int W : {1, 2, 3, 4})
    {
        const int it = 8192;
        for (int rep = 0; rep < 20; ++rep)
            hipLaunchKernelGGL(issueKernel< FORM >, dim3(cus), dim3(256 * W), 0, 0, out, clk, it, 1.0000001, 1e-9);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), clk, sizeof(long long) * 2 * 16 * cus, hipMemcpyDeviceToHost);
        std::vector< double > cyc;
        for (int i = 0; i < cus; ++i)
        {
            long long t0 = h[2 * 16 * i], t1 = h[2 * 16 * i + 1];
            for (int w = 1; w < 4 * W; ++w)
            {
                t0 = std::min(t0, h[2 * (16 * i + w)]);
                t1 = std::max(t1, h[2 * (16 * i + w) + 1]);
            }
            cyc.push_back(double(t1 - t0) / (double(CH) * it * W));
        }
        std::sort(cyc.begin(), cyc.end());
        printf("  W=%d: %5.2f", W, cyc[cus / 2]);
    }
    printf("   (shader cycles per wave-instruction per SIMD)\n");
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double*    out;
    long long* clk;
    (void)hipMalloc(&out, sizeof(double) * cus * 1024);
    (void)hipMalloc(&clk, sizeof(long long) * 2 * 16 * cus);
    run< FMA_VVV >(out, clk, cus);
    run< FMA_VSV >(out, clk, cus);
    run< FMA_VSS >(out, clk, cus);
    run< FMAC_VS >(out, clk, cus);
    run< FMAC_VV >(out, clk, cus);
    run< MUL_VV >(out, clk, cus);
    run< MUL_VS >(out, clk, cus);
    run< ADD_VV >(out, clk, cus);
    run< FMA_F32 >(out, clk, cus);
    run< MOV_B32 >(out, clk, cus);
    return 0;
}
