// Are two 384-thread workgroups with ~80 KB of LDS each resident on one CU?  (hipOccupancyMaxActiveBlocksPerMultiprocessor says 2.)
// Each workgroup spins for a fixed number of dependent FMAs: grid = #CUs takes T; grid = 2 #CUs takes T if two are co-resident, 2T if not.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(384, 3) void spin(double* p, int n)
{
    extern __shared__ double l[];
    double x = p[threadIdx.x & 7];
    for (int i = 0; i < n; ++i)
        x = x * 1.0000001 + 1e-9;
    l[threadIdx.x] = x;
    __syncthreads();
    if (l[(threadIdx.x + 1) % 384] == 1.2345)
        p[0] = x;
}
// the same with ~160 VGPRs live (72 doubles in flight): 6 waves per workgroup sit 2,2,1,1 on the four SIMDs, a second workgroup
// fits only if its waves go 1,1,2,2 (3 x 160 <= 512 registers per SIMD lane, 4 x 160 not)
__global__ __launch_bounds__(384, 3) void spinFat(double* p, int n)
{
    extern __shared__ double l[];
    double x[72];
#pragma unroll
    for (int j = 0; j < 72; ++j)
        x[j] = p[(threadIdx.x + j) & 7] + j;
    for (int i = 0; i < n; ++i)
    {
#pragma unroll
        for (int j = 0; j < 72; ++j)
            x[j] = x[j] * 1.0000001 + 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 72; ++j)
        s += x[j];
    l[threadIdx.x] = s;
    __syncthreads();
    if (l[(threadIdx.x + 1) % 384] == 1.2345)
        p[0] = s;
}
// Who sits where: every wave records its HW_ID (wave slot, SIMD, CU, SH, SE), XCC_ID and the wall clock at start and end.
__global__ __launch_bounds__(384, 3) void spinTrace(double* p, int n, unsigned long long* rec)
{
    extern __shared__ double l[];
    const unsigned long long t0 = wall_clock64();
    double x[72];
#pragma unroll
    for (int j = 0; j < 72; ++j)
        x[j] = p[(threadIdx.x + j) & 7] + j;
    for (int i = 0; i < n; ++i)
    {
#pragma unroll
        for (int j = 0; j < 72; ++j)
            x[j] = x[j] * 1.0000001 + 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 72; ++j)
        s += x[j];
    l[threadIdx.x] = s;
    __syncthreads();
    if (l[(threadIdx.x + 1) % blockDim.x] == 1.2345)
        p[0] = s;
    if ((threadIdx.x & 63) == 0)
    {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long* r = rec + 4 * (size_t(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64);
        r[0] = hw;
        r[1] = xcc;
        r[2] = t0;
        r[3] = wall_clock64();
    }
}
#include <algorithm>
#include <map>
#include <string>
#include <vector>
void trace(int threads, int cus, double* d, int lds = 79576)
{
    const int grid = 16 * cus * 64 / threads > 4 * cus ? 16 * cus * 64 / threads : 4 * cus, wpb = threads / 64;
    hipFuncSetAttribute(reinterpret_cast< const void* >(spinTrace), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned long long* rec;
    hipMalloc(&rec, size_t(grid) * wpb * 32);
    hipLaunchKernelGGL(spinTrace, dim3(grid), dim3(threads), lds, 0, d, 3000, rec);
    hipLaunchKernelGGL(spinTrace, dim3(grid), dim3(threads), lds, 0, d, 3000, rec);
    hipDeviceSynchronize();
    std::vector< unsigned long long > h(size_t(grid) * wpb * 4);
    hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(rec);
    // per CU (xcc, se, sh, cu): the workgroups' [start, end) intervals -> the largest number resident at once
    std::map< unsigned, std::vector< std::pair< unsigned long long, int > > > ev;
    std::map< std::string, int >                                              placement;
    for (int b = 0; b < grid; ++b)
    {
        unsigned long long t0 = ~0ull, t1 = 0;
        int                per_simd[4] = {0, 0, 0, 0};
        unsigned           cu_key      = 0;
        for (int w = 0; w < wpb; ++w)
        {
            const unsigned long long* r  = &h[4 * (size_t(b) * wpb + w)];
            const unsigned            hw = unsigned(r[0]), xcc = unsigned(r[1]) & 0xf;
            ++per_simd[(hw >> 4) & 3];
            cu_key = (xcc << 16) | (hw & 0xff00); // cu_id [11:8], sh_id [12], se_id [15:13]
            t0     = std::min(t0, r[2]);
            t1     = std::max(t1, r[3]);
        }
        ev[cu_key].push_back({t0, +1});
        ev[cu_key].push_back({t1, -1});
        char buf[32];
        std::snprintf(buf, sizeof buf, "%d,%d,%d,%d", per_simd[0], per_simd[1], per_simd[2], per_simd[3]);
        ++placement[buf];
    }
    std::map< int, int > hist;
    for (auto& [k, v] : ev)
    {
        std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first < b.first : a.second < b.second; });
        int cur = 0, mx = 0;
        for (auto& e : v)
            mx = std::max(mx, cur += e.second);
        ++hist[mx];
    }
    std::printf("trace, %d waves per workgroup, ~150 VGPRs, LDS %d B, grid of %d: %zu distinct CUs seen;", wpb, lds, grid, ev.size());
    for (auto& [m, c] : hist)
        std::printf(" %d CUs with at most %d workgroups resident at once;", c, m);
    std::printf(" waves per SIMD of a workgroup:");
    for (auto& [s, c] : placement)
        std::printf(" [%s] x %d", s.c_str(), c);
    std::printf("\n");
}
template < typename Kern >
void fat(Kern k, const char* name, int threads, int cus, double* d)
{
    const int lds = 79576;
    hipFuncSetAttribute(reinterpret_cast< const void* >(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncAttributes at;
    hipFuncGetAttributes(&at, reinterpret_cast< const void* >(k));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, threads, lds);
    float t[3];
    for (int g = 1; g <= 3; ++g)
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(cus * g), dim3(threads), lds, 0, d, 3000);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(cus * g), dim3(threads), lds, 0, d, 3000);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t[g - 1], e0, e1);
    }
    std::printf("%s: %d VGPRs, %d threads, LDS %d B, occupancy API says %d workgroups per CU: grid = CUs %.2f ms, 2 x CUs %.2f ms, 3 x CUs %.2f ms\n",
                name, at.numRegs, threads, lds, occ, t[0], t[1], t[2]);
}
int main()
{
    double* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (int lds : {32768, 65536, 73728, 79576, 81920})
    {
        hipFuncSetAttribute(reinterpret_cast< const void* >(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float t[3];
        for (int g = 1; g <= 3; ++g)
        {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(spin, dim3(cus * g), dim3(384), lds, 0, d, 200000);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin, dim3(cus * g), dim3(384), lds, 0, d, 200000);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t[g - 1], e0, e1);
        }
        std::printf("LDS %6d B per workgroup of 384 threads: grid = CUs %.2f ms, 2 x CUs %.2f ms, 3 x CUs %.2f ms\n", lds, t[0], t[1], t[2]);
    }
    fat(spinFat, "fat, 6 waves per workgroup", 384, cus, d);
    fat(spinFat, "fat, 4 waves per workgroup", 256, cus, d);
    fat(spinFat, "fat, 5 waves per workgroup", 320, cus, d);
    trace(384, cus, d);
    trace(256, cus, d);
    trace(320, cus, d);
    // small workgroups, LDS not limiting: is the register accounting per SIMD?
    trace(64, cus, d, 4096);
    trace(128, cus, d, 4096);
    trace(192, cus, d, 4096);
    trace(256, cus, d, 4096);
    trace(320, cus, d, 4096);
    trace(384, cus, d, 4096);
    return 0;
}
