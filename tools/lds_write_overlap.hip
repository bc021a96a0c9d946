// lds_write_overlap.hip -- does a wave's VALU work hide its own ds_write_b128 issue?  Per iteration 70 v_fma_f64 and 7
// ds_write_b128 (one sweep stage of the element kernel), as a burst after the FMAs or interleaved 10 : 1.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_write_overlap.hip -o tools/lds_write_overlap && tools/lds_write_overlap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

// MODE 0: FMAs only, 1: writes only, 2: 70 FMAs then 7 writes, 3: (10 FMAs, 1 write) x 7, 4: reads: 7 ds_read_b128 + wait + 70 FMAs
template < int MODE >
__global__ __launch_bounds__(512) void k(double* out, long long* clk, int iters, double a, double b)
{
    __shared__ double2 lds[8 * 64 * 8];
    double             acc[10];
#pragma unroll
    for (int c = 0; c < 10; ++c)
        acc[c] = threadIdx.x * 1e-3 + c;
    using d2 = double __attribute__((ext_vector_type(2)));
    const unsigned  p  = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192; // byte address in LDS (the only __shared__ object starts at 0): conflict-free, 1 KiB per wave-instruction
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
    {
        if constexpr (MODE == 4)
        {
            d2 r[7];
#pragma unroll
            for (int w = 0; w < 7; ++w)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[w]) : "v"(p), "n"(w * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int w = 0; w < 7; ++w)
                acc[w] += r[w].x;
        }
#pragma unroll
        for (int w = 0; w < 7; ++w)
        {
            if constexpr (MODE != 1 && MODE < 5)
            {
#pragma unroll
                for (int c = 0; c < 10; ++c)
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
            }
            if constexpr (MODE == 3)
                asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(p), "v"(d2{acc[w], acc[w + 1]}), "n"(w * 1024) : "memory");
        }
        if constexpr (MODE == 5) // the same bytes as 7 ds_write_b128 in 14 ds_write_b64 (conflict-free: 512 B per wave-instruction)
        {
            const unsigned p8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 8192;
#pragma unroll
            for (int w = 0; w < 14; ++w)
                asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(p8), "v"(acc[w % 8]), "n"(w * 512) : "memory");
        }
        if constexpr (MODE == 6) // ... in 7 ds_write2_b64 (two 8-byte values per lane at two offsets 512 B apart)
        {
            const unsigned p8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 8192;
#pragma unroll
            for (int w = 0; w < 7; ++w)
                asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64" : : "v"(p8 + w * 1024), "v"(acc[w]), "v"(acc[w + 1]) : "memory");
        }
        if constexpr (MODE == 7) // 7 ds_write_b128 by 49 of the 64 lanes (the element kernel's sweeps)
        {
            if ((threadIdx.x & 63) < 49)
            {
#pragma unroll
                for (int w = 0; w < 7; ++w)
                    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(p), "v"(d2{acc[w], acc[w + 1]}), "n"(w * 1024) : "memory");
            }
        }
        if constexpr (MODE == 8) // 14 ds_read_b64 + wait
        {
            const unsigned p8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 8192;
            double         r[14];
#pragma unroll
            for (int w = 0; w < 14; ++w)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[w]) : "v"(p8), "n"(w * 512));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int w = 0; w < 7; ++w)
                acc[w] += r[w] + r[w + 7];
        }
        if constexpr (MODE == 9) // 7 ds_read_b128 + wait, no FMAs
        {
            d2 r[7];
#pragma unroll
            for (int w = 0; w < 7; ++w)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[w]) : "v"(p), "n"(w * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int w = 0; w < 7; ++w)
                acc[w] += r[w].x;
        }
        if constexpr (MODE == 1 || MODE == 2)
        {
#pragma unroll
            for (int w = 0; w < 7; ++w)
                asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(p), "v"(d2{acc[w], acc[w + 1]}), "n"(w * 1024) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
    {
        long long* c = clk + 2 * (blockIdx.x * 8 + (threadIdx.x >> 6));
        c[0] = t0, c[1] = t1;
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 10; ++c)
        s += acc[c];
    if (s == 1.2345e300)
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x].x;
}

template < int MODE >
void run(const char* name, double* out, long long* clk, int cus)
{
    std::vector< long long > h(2 * 8 * cus);
    printf("%-44s", name);
    for (int waves : {1, 4, 8})
    {
        const int it = 2048;
        for (int rep = 0; rep < 10; ++rep)
            hipLaunchKernelGGL(k< MODE >, dim3(cus), dim3(64 * waves), 0, 0, out, clk, it, 1.0000001, 1e-9);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), clk, sizeof(long long) * 2 * 8 * cus, hipMemcpyDeviceToHost);
        std::vector< double > cyc;
        for (int i = 0; i < cus; ++i)
        {
            long long t0 = h[2 * 8 * i], t1 = h[2 * 8 * i + 1];
            for (int w = 1; w < waves; ++w)
            {
                t0 = std::min(t0, h[2 * (8 * i + w)]);
                t1 = std::max(t1, h[2 * (8 * i + w) + 1]);
            }
            cyc.push_back(double(t1 - t0) / it);
        }
        std::sort(cyc.begin(), cyc.end());
        printf("  %d waves/CU: %7.1f", waves, cyc[cus / 2]);
    }
    printf("   cycles per iteration (block span)\n");
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int  cus = prop.multiProcessorCount;
    double*    out;
    long long* clk;
    (void)hipMalloc(&out, sizeof(double) * cus * 512);
    (void)hipMalloc(&clk, sizeof(long long) * 2 * 8 * cus);
    run< 0 >("70 v_fma_f64", out, clk, cus);
    run< 1 >("7 ds_write_b128", out, clk, cus);
    run< 2 >("70 v_fma_f64, then 7 ds_write_b128", out, clk, cus);
    run< 3 >("(10 v_fma_f64, 1 ds_write_b128) x 7", out, clk, cus);
    run< 5 >("14 ds_write_b64 (same bytes)", out, clk, cus);
    run< 6 >("7 ds_write2_b64 (same bytes)", out, clk, cus);
    run< 7 >("7 ds_write_b128, 49 lanes", out, clk, cus);
    run< 9 >("7 ds_read_b128, wait", out, clk, cus);
    run< 8 >("14 ds_read_b64, wait", out, clk, cus);
    run< 4 >("7 ds_read_b128, wait, 77 VALU", out, clk, cus);
    return 0;
}
