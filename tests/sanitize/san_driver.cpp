// Sanitizer leg (the reference's CI runs its test binary under ASan and UBSan, .github/workflows/ci.yml:39-134): the HOST-side product
// code (1-D tables, cube-mesh partitioner, native mesh / results files) and the CPU oracle in one executable built with
// -fsanitize=address,undefined -fno-sanitize-recover=all by tests/test_sanitizers_cpu.py.  No GPU code is involved (GPU AddressSanitizer
// is not available on the pool); every check below is also a functional one, so a wrong answer fails the run as a sanitizer report does.
#include "l3k.h"
#include "host/tables.hpp"
#include "oracle.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace l3k::dev
{
void setError(const char* fmt, ...) // (registry.cpp pulls in the device headers: the host units only need this)
{
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    std::fputc('\n', stderr);
    va_end(ap);
}
} // namespace l3k::dev

#define CHECK(c)                                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        if (!(c))                                                                                                      \
        {                                                                                                              \
            std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c);                                 \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

static void tables()
{
    for (int p = 1; p <= 8; ++p)
        for (int nq : {p + 1, 2 * p})
        {
            const auto       blk = l3k::host::deviceTableBlock(p, nq);
            CHECK(!blk.empty());
            std::vector< double > I, D, x, w;
            l3k::host::basis1d(p, nq, I, D);
            l3k::host::glRule(nq, x, w);
            double sw = 0.;
            for (double v : w)
                sw += v;
            CHECK(std::fabs(sw - 2.) < 1e-13);
            for (int q = 0; q < nq; ++q) // partition of unity, derivatives sum to zero (tests/MappingTests.cpp)
            {
                double s = 0., d = 0.;
                for (int b = 0; b <= p; ++b)
                {
                    s += I[b * nq + q];
                    d += D[b * nq + q];
                }
                CHECK(std::fabs(s - 1.) < 1e-12 && std::fabs(d) < 1e-10);
            }
        }
}

struct Part
{
    l3k_hostmesh*     hm = nullptr;
    l3k_hostmesh_view v{};
    Part(const int (&ne)[3], int order, const int (&parts)[3], int rank, double perturb)
    {
        CHECK(l3k_cube_partition_create(ne, order, parts, rank, perturb, &hm) == 0);
        CHECK(l3k_hostmesh_view_get(hm, &v) == 0);
    }
    ~Part() { l3k_hostmesh_destroy(hm); }
};

static void partitions()
{
    const int cases[][7] = {{3, 2, 2, 2, 2, 1, 1}, {2, 2, 2, 4, 1, 1, 1}, {4, 4, 4, 1, 2, 2, 2}, {5, 3, 2, 3, 2, 3, 1}, {1, 1, 1, 6, 1, 1, 1}};
    for (const auto& c : cases)
    {
        const int ne[3] = {c[0], c[1], c[2]}, parts[3] = {c[4], c[5], c[6]}, order = c[3], n_ranks = parts[0] * parts[1] * parts[2];
        int64_t   owned = 0, elems = 0, n_global = -1;
        for (int r = 0; r < n_ranks; ++r)
        {
            Part P(ne, order, parts, r, 0.1);
            owned += P.v.n_owned_nodes;
            elems += P.v.n_elems;
            n_global = P.v.n_global_nodes;
            const int     N     = (order + 1) * (order + 1) * (order + 1);
            const int64_t n_loc = P.v.n_owned_nodes + P.v.n_ghost_nodes;
            for (int64_t i = 0; i < P.v.n_elems * N; ++i)
                CHECK(P.v.elem_nodes[i] < n_loc);
            for (int k = 0; k < P.v.n_nbrs; ++k)
            {
                CHECK(P.v.nbr_rank[k] >= 0 && P.v.nbr_rank[k] < n_ranks && P.v.nbr_rank[k] != r);
                for (int64_t s = P.v.send_offsets[k]; s < P.v.send_offsets[k + 1]; ++s)
                    CHECK(P.v.send_nodes[s] >= 0 && P.v.send_nodes[s] < P.v.n_owned_nodes);
            }
            if (P.v.n_nbrs)
                CHECK(P.v.ghost_offsets[P.v.n_nbrs] == P.v.n_ghost_nodes);
        }
        CHECK(owned == n_global && elems == int64_t(ne[0]) * ne[1] * ne[2]);
        CHECK(n_global == int64_t(ne[0] * order + 1) * (ne[1] * order + 1) * (ne[2] * order + 1));
    }
    const int ne[3] = {2, 2, 2}, bad_parts[3] = {3, 1, 1}; // more parts than elements along x: an error, not a crash
    l3k_hostmesh* hm = nullptr;
    if (l3k_cube_partition_create(ne, 2, bad_parts, 0, 0., &hm) == 0)
        l3k_hostmesh_destroy(hm);
}

// the oracle on a partition: sum-factorised mesh apply == sum over the elements of K_e x_e (local assembly), diag == diag K
static void oracle(int kernel_id, int p)
{
    int dim, E, U, F;
    CHECK(orc_kernel_params(kernel_id, &dim, &E, &U, &F) == 0 && dim == 3);
    const int ne[3] = {2, 1, 2}, parts[3] = {1, 1, 1};
    Part      P(ne, p, parts, 0, 0.15);
    const int N = (p + 1) * (p + 1) * (p + 1), Nd = N * U, nq = orc_n_qps1d(p, 1, 0);
    const int64_t          n = P.v.n_owned_nodes, nd = n * U;
    std::vector< int >     finds(U);
    for (int u = 0; u < U; ++u)
        finds[u] = u;
    std::vector< double > fields(size_t(F > 0 ? F : 1) * n);
    for (size_t i = 0; i < fields.size(); ++i)
        fields[i] = std::sin(0.37 * double(i) + 0.1);
    orc_mesh m{3, p, nq, P.v.n_elems, P.v.elem_nodes, P.v.elem_verts, n, U, finds.data(), nullptr, F ? fields.data() : nullptr};
    std::vector< double > x(nd), y(nd, 0.), yref(nd, 0.), diag(nd, 0.), rhs(nd, 0.), dref(nd, 0.);
    for (int64_t i = 0; i < nd; ++i)
        x[i] = std::cos(0.11 * double(i));
    for (int threads : {1, 3})
    {
        std::fill(y.begin(), y.end(), 0.);
        CHECK(orc_mf_apply(&m, kernel_id, nullptr, 0.25, 1, 1, x.data(), nd, y.data(), nd, 1., 0., 0, P.v.n_elems, 1, 1, nd, threads) == 0);
    }
    CHECK(orc_mf_diag_rhs(&m, kernel_id, nullptr, 0.25, 1, nullptr, 0, diag.data(), rhs.data(), nd, 0, P.v.n_elems, 1, nd, 1) == 0);
    std::vector< double > K(size_t(Nd) * Nd), Fe(Nd), nf(size_t(N) * (F > 0 ? F : 1));
    for (int64_t e = 0; e < P.v.n_elems; ++e)
    {
        const uint32_t* en = P.v.elem_nodes + e * N;
        for (int i = 0; i < N; ++i)
            for (int f = 0; f < F; ++f)
                nf[size_t(i) * F + f] = fields[size_t(f) * n + en[i]];
        CHECK(orc_assemble_local(kernel_id, p, nq, 1, P.v.elem_verts + e * 24, F ? nf.data() : nullptr, nullptr, 0.25, K.data(), Fe.data()) == 0);
        for (int i = 0; i < Nd; ++i)
        {
            double s = 0.;
            for (int j = 0; j < Nd; ++j)
                s += K[size_t(i) * Nd + j] * x[size_t(en[j / U]) * U + j % U];
            yref[size_t(en[i / U]) * U + i % U] += s;
            dref[size_t(en[i / U]) * U + i % U] += K[size_t(i) * Nd + i];
        }
    }
    double err = 0., nrm = 0., derr = 0.;
    for (int64_t i = 0; i < nd; ++i)
    {
        err += (y[i] - yref[i]) * (y[i] - yref[i]);
        nrm += yref[i] * yref[i];
        derr = std::fmax(derr, std::fabs(diag[i] - dref[i]) / (1. + std::fabs(dref[i])));
    }
    CHECK(std::sqrt(err) < 1e-11 * std::sqrt(nrm) && derr < 1e-11);
}

static void files(const char* dir)
{
    // results file: two writers (node ranges), one reader
    const std::string     rp = std::string(dir) + "/san_results.bin";
    const int64_t         n_global = 37, split = 20;
    const size_t          n_fields = 3;
    std::vector< double > a(n_fields * split), b(n_fields * (n_global - split));
    for (size_t f = 0; f < n_fields; ++f)
    {
        for (int64_t i = 0; i < split; ++i)
            a[f * split + i] = double(100 * f + i);
        for (int64_t i = split; i < n_global; ++i)
            b[f * (n_global - split) + (i - split)] = double(100 * f + i);
    }
    CHECK(l3k_results_save(rp.c_str(), "sanitizer run", n_fields, n_global, 0, split, a.data(), split, 1) == 0);
    CHECK(l3k_results_save(rp.c_str(), "sanitizer run", n_fields, n_global, split, n_global - split, b.data(), n_global - split, 0) == 0);
    size_t nf = 0, nn = 0;
    CHECK(l3k_results_info(rp.c_str(), &nf, &nn) == 0 && nf == n_fields && nn == size_t(n_global));
    const int64_t ids[4] = {36, 0, 19, 20};
    double        out[4];
    CHECK(l3k_results_load(rp.c_str(), 2, 4, ids, 0, out) == 0);
    for (int k = 0; k < 4; ++k)
        CHECK(out[k] == double(200 + ids[k]));
    const int64_t bad[1] = {37};
    CHECK(l3k_results_load(rp.c_str(), 0, 1, bad, 0, out) != 0); // out of range: an error, not a read past the end
    std::remove(rp.c_str());

    // mesh file: two parts (order 2: one hex domain of two elements + a boundary domain of quads; an empty part), read back
    const std::string       mp = std::string(dir) + "/san_mesh.bin";
    std::vector< uint64_t > hn(2 * 27), qn(3 * 9), hid = {0, 1}, qid = {2, 3, 4};
    std::vector< double >   hv(2 * 8 * 3), qv(3 * 4 * 3);
    for (size_t i = 0; i < hn.size(); ++i)
        hn[i] = 3 * i + 1;
    for (size_t i = 0; i < qn.size(); ++i)
        qn[i] = 5 * i;
    for (size_t i = 0; i < hv.size(); ++i)
        hv[i] = 0.25 * double(i);
    for (size_t i = 0; i < qv.size(); ++i)
        qv[i] = -0.5 * double(i);
    l3k_meshfile_domain doms[2]{};
    doms[0].id  = 0;
    doms[0].hex = {2, hn.data(), hv.data(), hid.data()};
    doms[1].id   = 7;
    doms[1].quad = {3, qn.data(), qv.data(), qid.data()};
    const uint16_t         bids[1] = {7};
    l3k_meshfile_part_desc d0{2, 2, doms, 11, 40, 1, bids}, d1{2, 0, nullptr, 0, 0, 0, nullptr};
    size_t                 bytes[2];
    CHECK(l3k_meshfile_part_bytes(&d0, &bytes[0]) == 0 && l3k_meshfile_part_bytes(&d1, &bytes[1]) == 0);
    CHECK(l3k_meshfile_save(mp.c_str(), "two parts", 2, bytes, 1, &d1, 0) == 0); // (any order of the ranks' calls)
    CHECK(l3k_meshfile_save(mp.c_str(), "two parts", 2, bytes, 0, &d0, 1) == 0);
    size_t np = 0, pb[2] = {0, 0};
    CHECK(l3k_meshfile_info(mp.c_str(), &np, pb, 2) == 0 && np == 2 && pb[0] == bytes[0] && pb[1] == bytes[1]);
    l3k_meshfile_part* part = nullptr;
    CHECK(l3k_meshfile_load(mp.c_str(), 0, 2, &part) == 0);
    l3k_meshfile_part_desc r{};
    CHECK(l3k_meshfile_part_get(part, &r) == 0);
    CHECK(r.n_domains == 2 && r.nodes_begin == 11 && r.n_owned_nodes == 40 && r.n_boundary_ids == 1 && r.boundary_ids[0] == 7);
    CHECK(r.domains[0].hex.n == 2 && r.domains[1].quad.n == 3);
    CHECK(std::memcmp(r.domains[0].hex.nodes, hn.data(), hn.size() * 8) == 0 && std::memcmp(r.domains[1].quad.verts, qv.data(), qv.size() * 8) == 0);
    CHECK(l3k_meshfile_part_destroy(part) == 0);
    CHECK(l3k_meshfile_load(mp.c_str(), 1, 2, &part) == 0);
    CHECK(l3k_meshfile_part_get(part, &r) == 0 && r.n_domains == 0);
    CHECK(l3k_meshfile_part_destroy(part) == 0);
    CHECK(l3k_meshfile_load(mp.c_str(), 2, 2, &part) != 0); // no such part
    CHECK(l3k_meshfile_load(mp.c_str(), 0, 3, &part) != 0); // the sizes do not parse as an order-3 mesh: refused
    // a truncated file is refused, not read past its end
    {
        FILE* f = std::fopen(mp.c_str(), "rb");
        CHECK(f != nullptr);
        std::vector< char > all(1 << 16);
        const size_t        got = std::fread(all.data(), 1, all.size(), f);
        std::fclose(f);
        const std::string tp = std::string(dir) + "/san_mesh_truncated.bin";
        f                    = std::fopen(tp.c_str(), "wb");
        std::fwrite(all.data(), 1, got - 40, f);
        std::fclose(f);
        CHECK(l3k_meshfile_load(tp.c_str(), 1, 2, &part) != 0 || l3k_meshfile_part_destroy(part) == 0);
        CHECK(l3k_meshfile_load(tp.c_str(), 0, 2, &part) != 0 || l3k_meshfile_part_destroy(part) == 0);
        std::remove(tp.c_str());
    }
    std::remove(mp.c_str());
}

int main(int argc, char** argv)
{
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    tables();
    partitions();
    oracle(0, 2);  // Diffusion3D
    oracle(0, 3);
    oracle(1, 2);  // variable coefficient from an external field
    oracle(10, 2); // operators and rhs read the point and the time
    files(dir);
    std::puts("sanitizer driver: ok");
    return 0;
}
