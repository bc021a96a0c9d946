// lds_bank.hip -- which lane->address patterns does ds_read_b128 / ds_write_b128 serve without bank conflicts on gfx950?
// One wave per workgroup, addresses in 16-byte units supplied per lane; prints cycles per instruction (s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_bank.hip -o /tmp/lds_bank
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(const int* __restrict__ addr16, int pat, int write, double* sink)
{
    extern __shared__ double2 lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1280; i += 64)
        lds[i] = make_double2(i, -i);
    __syncthreads();
    const int a = addr16[pat * 64 + lane];
    if (a < 0)
        return;
    double2 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        acc[k] = make_double2(0., 0.);
#pragma unroll 1
    for (int it = 0; it < 2048; ++it)
    {
        if (write)
        {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                lds[a + ((it + k) & 1)] = acc[k];
        }
        else
        {
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                const double2 v = lds[a + ((it + k) & 1)];
                acc[k].x += v.x;
            }
        }
    }
    double s = 0.;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        s += acc[k].x + acc[k].y;
    if (s == 1.2345)
        sink[lane] = s;
}

int main()
{
    std::vector< std::vector< int > > pats;
    std::vector< const char* >        names;
    auto add = [&](const char* name, auto f) {
        std::vector< int > v(64);
        for (int l = 0; l < 64; ++l)
            v[l] = f(l);
        pats.push_back(v);
        names.push_back(name);
    };
    add("consecutive (l)", [](int l) { return l; });
    add("49 lanes consecutive", [](int l) { return l < 49 ? l : -1; });
    add("stride 2 (2l)", [](int l) { return 2 * l; });
    add("stride 4", [](int l) { return 4 * l; });
    add("stride 7", [](int l) { return 7 * l; });
    add("stride 8", [](int l) { return 8 * l; });
    add("stride 16", [](int l) { return 16 * l; });
    add("y-pencil natural: (l/7)*49 + l%7", [](int l) { return l < 49 ? (l / 7) * 49 + l % 7 : -1; });
    add("x-pencil natural: (l/7)*49 + (l%7)*7", [](int l) { return l < 49 ? (l / 7) * 49 + (l % 7) * 7 : -1; });
    add("z-pencil natural: l", [](int l) { return l < 49 ? l : -1; });
    // permuted maps: lane L = 8g + r holds the pencil (c1, c2) with (c1 + c2) % 8 == r (sum family) / (c1 - c2) % 8 == r
    auto permuted = [](bool diff, auto addr) {
        return [=](int L) {
            int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c2 = 0; c2 < 7; ++c2)
                for (int c1 = 0; c1 < 7; ++c1)
                {
                    const int r = ((diff ? c1 - c2 : c1 + c2) % 8 + 8) % 8;
                    if (8 * cnt[r] + r == L)
                        return addr(c1, c2);
                    ++cnt[r];
                }
            return -1;
        };
    };
    add("y-pencil permuted (sum): a=c2, c=c1", permuted(false, [](int c1, int c2) { return c2 * 49 + c1; }));
    add("x-pencil permuted (diff): b=c1, a=c2", permuted(true, [](int c1, int c2) { return c2 * 49 + c1 * 7; }));
    add("z-pencil permuted (diff): c=c1, b=c2", permuted(true, [](int c1, int c2) { return c2 * 7 + c1; }));
    add("half-wave pairs: (l%32)*1 + (l/32)*512", [](int l) { return (l % 32) + (l / 32) * 512; });
    add("conflict by 16 lanes: l%16", [](int l) { return l % 16; });
    add("l%8 + 8*(l/8)*8", [](int l) { return l % 8 + 64 * (l / 8); });
    add("lanes l and l+8 same bank: l%8 + 8*(l/8)", [](int l) { return l; });
    add("group of 4: (l%4) + 8*(l/4)", [](int l) { return (l % 4) + 8 * (l / 4); });
    add("group of 8 rotated: ((l%8)*3)%8 + 8*(l/8)", [](int l) { return ((l % 8) * 3) % 8 + 8 * (l / 8); });
    add("even lanes bank0-3, odd +8: (l/2)%8... l*9", [](int l) { return l * 9; });
    add("l*5", [](int l) { return l * 5; });
    add("l*3", [](int l) { return l * 3; });

    const int n = int(pats.size());
    int*      d_addr;
    double*   d_sink;
    (void)hipMalloc(&d_addr, n * 64 * sizeof(int));
    (void)hipMalloc(&d_sink, 64 * sizeof(double));
    std::vector< int > flat;
    for (auto& p : pats)
        for (int v : p)
            flat.push_back(v < 0 ? -1 : v % 1200); // stay inside the 20 KB of one wave; preserves address mod 8/16 structure only if 1200 % 16 == 0
    (void)hipMemcpy(d_addr, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int   waves_per_cu = 7, blocks = prop.multiProcessorCount * waves_per_cu;
    hipEvent_t  e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double clk_ghz = prop.clockRate * 1e-6;
    std::printf("%d CUs, %d waves/CU, clock %.2f GHz\n", prop.multiProcessorCount, waves_per_cu, clk_ghz);
    for (int p = 0; p < n; ++p)
    {
        float ms[2];
        for (int w = 0; w < 2; ++w)
        {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 22144, 0, d_addr, p, w, d_sink);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 22144, 0, d_addr, p, w, d_sink);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms[w], e0, e1);
        }
        const double instr_per_cu = 2048. * 8 * waves_per_cu;
        std::printf("%-48s read %5.2f  write %5.2f  clocks per b128 wave-instruction per CU\n", names[p],
                    ms[0] * 1e-3 * clk_ghz * 1e9 / instr_per_cu, ms[1] * 1e-3 * clk_ghz * 1e9 / instr_per_cu);
    }
    return 0;
}
