// diffusion3d_mf.cpp -- the C++ host shim (include/l3k/operator.hpp) used the way the reference's own tests use its
// API: (1) single distorted element: matrix-free apply == K_e * x with K_e from LocalAssembly (the property
// tests/LocalOperatorTests.cpp:3-95 checks, < 1e-8); (2) a Dirichlet-constrained mesh operator is symmetric,
// <A x, z> = <x, A z> (Operator::apply ignores `mode`, algsys/MatrixFreeSystem.hpp:34-41); (3) a boundary term attached
// with assembleProblem keeps the operator symmetric, and the surface of the unit cube integrates to 6, its volume to 1
// (the side-area check of tests/MappingTests.cpp:555-610 on a mesh).
// Build: hipcc -std=c++20 -Iinclude tests/cpp/diffusion3d_mf.cpp -Ll3ster_amd/lib -ll3k -o /tmp/diffusion3d_mf
#include "l3k/operator.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <map>
#include <random>
#include <thread>
#include <vector>

#define HIP_CHECK(x)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if ((x) != hipSuccess)                                                                                         \
        {                                                                                                              \
            std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__);                                          \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)

struct DevVec
{
    double* p{};
    size_t  n{};
    explicit DevVec(size_t n_) : n{n_} { (void)hipMalloc(reinterpret_cast< void** >(&p), n * sizeof(double)); }
    ~DevVec() { (void)hipFree(p); }
    void up(const std::vector< double >& h) { (void)hipMemcpy(p, h.data(), n * sizeof(double), hipMemcpyHostToDevice); }
    std::vector< double > down() const
    {
        std::vector< double > h(n);
        (void)hipMemcpy(h.data(), p, n * sizeof(double), hipMemcpyDeviceToHost);
        return h;
    }
};

int main()
{
    struct DiffusionParams
    {
        double k, s;
    } params{0.7, 1.0};
    l3k::Context ctx{0};
    auto         prng = std::mt19937_64{42};
    auto         dist = std::uniform_real_distribution< double >{-1., 1.};
    int          failures = 0;

    { // (1) one element, order 3: apply == K_e x
        constexpr int         p = 3, U = 4, N = (p + 1) * (p + 1) * (p + 1), Nd = N * U;
        l3k::CubeMesh         mesh{{1, 1, 1}, p, {1, 1, 1}, 0, 0.};
        l3k::DeviceMesh       dmesh{ctx, mesh, U};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
        std::vector< double > x(Nd);
        for (auto& v : x)
            v = dist(prng);
        DevVec dx{Nd}, dy{Nd}, dK{size_t(Nd) * Nd}, dF{Nd};
        dx.up(x);
        sys.apply(dx.p, Nd, dy.p, Nd);
        sys.assembleLocal(0, 1, dK.p, dF.p);
        ctx.synchronize();
        const auto y = dy.down();
        const auto K = dK.down();
        double     err = 0., nrm = 0.;
        for (int i = 0; i < Nd; ++i)
        {
            double kx = 0.;
            for (int j = 0; j < Nd; ++j)
                kx += K[size_t(i) * Nd + j] * x[mesh.view().elem_nodes[j / U] * U + j % U]; // element-local -> global dof
            const double yi = y[mesh.view().elem_nodes[i / U] * U + i % U];
            err += (yi - kx) * (yi - kx);
            nrm += kx * kx;
        }
        std::printf("single element: |y - K_e x| / |K_e x| = %.3e\n", std::sqrt(err / nrm));
        failures += !(std::sqrt(err / nrm) < 1e-12);
        // assembleGlobalSystem in one call: with one element the global matrix over a dense graph in global dof order is K_e
        // with its rows and columns renumbered
        std::vector< int64_t > row_ptr(Nd + 1);
        std::vector< int32_t > col_ind(size_t(Nd) * Nd);
        for (int i = 0; i <= Nd; ++i)
            row_ptr[size_t(i)] = int64_t(i) * Nd;
        for (int i = 0; i < Nd; ++i)
            for (int j = 0; j < Nd; ++j)
                col_ind[size_t(i) * Nd + j] = j;
        int64_t* d_rp = nullptr;
        int32_t* d_ci = nullptr;
        HIP_CHECK(hipMalloc(reinterpret_cast< void** >(&d_rp), row_ptr.size() * sizeof(int64_t)));
        HIP_CHECK(hipMalloc(reinterpret_cast< void** >(&d_ci), col_ind.size() * sizeof(int32_t)));
        HIP_CHECK(hipMemcpy(d_rp, row_ptr.data(), row_ptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_ci, col_ind.data(), col_ind.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        DevVec dV{size_t(Nd) * Nd}, dR{Nd};
        HIP_CHECK(hipMemset(dV.p, 0, sizeof(double) * size_t(Nd) * Nd));
        HIP_CHECK(hipMemset(dR.p, 0, sizeof(double) * Nd));
        const int64_t missing = sys.assembleGlobal(0, 1, d_rp, d_ci, dV.p, dR.p, Nd);
        ctx.synchronize();
        const auto V = dV.down();
        double     gerr = 0., gmax = 0.;
        for (int i = 0; i < Nd; ++i)
            for (int j = 0; j < Nd; ++j)
            {
                const size_t gi = mesh.view().elem_nodes[i / U] * U + i % U, gj = mesh.view().elem_nodes[j / U] * U + j % U;
                gerr = std::max(gerr, std::fabs(V[gi * Nd + gj] - K[size_t(i) * Nd + j]));
                gmax = std::max(gmax, std::fabs(K[size_t(i) * Nd + j]));
            }
        std::printf("assembleGlobal (one call): max |A - K_e| / |K_e|_max = %.3e, %lld entries outside the graph\n", gerr / gmax, (long long)missing);
        failures += !(gerr < 1e-13 * gmax && missing == 0);
        (void)hipFree(d_rp);
        (void)hipFree(d_ci);
    }
    { // (2) 4^3 elements, order 4, Dirichlet on unknown 0: symmetry
        constexpr int         p = 4, U = 4;
        l3k::CubeMesh         mesh{{4, 4, 4}, p, {1, 1, 1}, 0, 0.1};
        const int             unknowns[] = {0};
        const auto            mask = mesh.dirichletMask(U, unknowns);
        l3k::DeviceMesh       dmesh{ctx, mesh, U, mask.data()};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
        const size_t          n = static_cast< size_t >(dmesh.nOwnedDofs());
        std::vector< double > x(n), z(n);
        for (size_t i = 0; i < n; ++i)
        {
            x[i] = dist(prng);
            z[i] = dist(prng);
        }
        DevVec dx{n}, dz{n}, dax{n}, daz{n};
        dx.up(x);
        dz.up(z);
        sys.apply(dx.p, n, dax.p, n);
        sys.apply(dz.p, n, daz.p, n);
        ctx.synchronize();
        const auto ax = dax.down(), az = daz.down();
        double     s1 = 0., s2 = 0.;
        for (size_t i = 0; i < n; ++i)
        {
            s1 += ax[i] * z[i];
            s2 += x[i] * az[i];
        }
        std::printf("symmetry: <Ax,z> = %.15e, <x,Az> = %.15e\n", s1, s2);
        failures += !(std::fabs(s1 - s2) < 1e-11 * std::fabs(s1));
    }
    { // (3) boundary term + integrals
        constexpr int         p = 2, U = 4;
        l3k::CubeMesh         mesh{{3, 3, 3}, p, {1, 1, 1}, 0, 0.};
        const int             unknowns[] = {0};
        const auto            mask = mesh.dirichletMask(U, unknowns, 0x30); // x- and x+ sides
        l3k::DeviceMesh       dmesh{ctx, mesh, U, mask.data()};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
        struct RobinParams
        {
            double h, t_inf;
        } robin{2., 0.5};
        const auto        walls = mesh.boundarySides(0x0f);
        l3k::BoundaryTerm term{dmesh, L3K_KERNEL_ROBIN3D, robin, walls};
        sys.assembleProblem(term);
        const size_t          n = static_cast< size_t >(dmesh.nOwnedDofs());
        std::vector< double > x(n), z(n);
        for (size_t i = 0; i < n; ++i)
        {
            x[i] = dist(prng);
            z[i] = dist(prng);
        }
        DevVec dx{n}, dz{n}, dax{n}, daz{n};
        dx.up(x);
        dz.up(z);
        sys.apply(dx.p, n, dax.p, n);
        sys.apply(dz.p, n, daz.p, n);
        ctx.synchronize();
        const auto ax = dax.down(), az = daz.down();
        double     s1 = 0., s2 = 0.;
        for (size_t i = 0; i < n; ++i)
        {
            s1 += ax[i] * z[i];
            s2 += x[i] * az[i];
        }
        std::printf("symmetry with boundary term: <Ax,z> = %.15e, <x,Az> = %.15e\n", s1, s2);
        failures += !(std::fabs(s1 - s2) < 1e-11 * std::fabs(s1));
        const auto all  = mesh.boundarySides();
        const auto area = l3k::computeIntegral< RobinParams >(dmesh, L3K_RESIDUAL_UNIT3D, nullptr, nullptr, 0, {}, &all);
        const auto vol  = l3k::computeIntegral< RobinParams >(dmesh, L3K_RESIDUAL_UNIT3D, nullptr, nullptr, 0);
        std::printf("surface = %.15f, volume = %.15f\n", area[0], vol[0]);
        failures += !(std::fabs(area[0] - 6.) < 1e-12 && std::fabs(vol[0] - 1.) < 1e-13);
        // endAssembly + solve: diag / rhs, Jacobi, PCG; then |A x - b| / |b| through a fresh apply
        DevVec diag{n}, rhs{n}, minv{n}, sol{n}, check{n};
        (void)hipMemset(diag.p, 0, n * sizeof(double));
        (void)hipMemset(rhs.p, 0, n * sizeof(double));
        (void)hipMemset(sol.p, 0, n * sizeof(double));
        sys.diagAndRhs(nullptr, 0, diag.p, rhs.p, n);
        l3k::check(l3k_jacobi_inverse(ctx.get(), diag.p, int64_t(n), 1., 0., minv.p));
        const auto res = sys.solve(rhs.p, sol.p, minv.p, {1e-10, 5000, 2, 1});
        sys.apply(sol.p, n, check.p, n);
        ctx.synchronize();
        const auto ax2 = check.down(), b = rhs.down();
        double     e2 = 0., b2 = 0.;
        for (size_t i = 0; i < n; ++i)
        {
            e2 += (ax2[i] - b[i]) * (ax2[i] - b[i]);
            b2 += b[i] * b[i];
        }
        std::printf("PCG: %d iterations, achieved %.2e, |Ax-b|/|b| = %.2e\n", res.iterations, res.achieved_tol, std::sqrt(e2 / b2));
        failures += !(std::sqrt(e2 / b2) < 1e-9);
    }
    { // (4) the point a kernel sees, the launch route as text, Dirichlet values at nodes, a multivector solve
        constexpr int   p = 2, U = 4, ne = 3;
        l3k::CubeMesh   mesh{{ne, ne, ne}, p, {1, 1, 1}, 0, 0.1};
        const int       unk0[1] = {0};
        const auto      mask    = mesh.dirichletMask(U, unk0, 0x3f);
        l3k::DeviceMesh dmesh{ctx, mesh, U, mask.data()};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params, {}, {}, 2};
        const std::string     route = sys.route();
        std::printf("route (generic_below = %lld): %s\n", (long long)ctx.tuning().generic_below, route.c_str());
        failures += route.empty();
        auto tune = ctx.tuning();
        tune.generic_below = 1 << 30; // every launch on the generic kernel
        ctx.setTuning(tune);
        failures += sys.route().find("generic") == std::string::npos;
        tune.generic_below = 1500;
        ctx.setTuning(tune);
        ctx.setReferenceZ0(true); // (Diffusion3D does not read the point: the results must not change)
        const size_t          n = size_t(dmesh.nOwnedDofs());
        std::vector< double > x(n);
        for (auto& v : x)
            v = dist(prng);
        DevVec dx{n}, dy0{n}, dy1{n};
        dx.up(x);
        sys.apply(dx.p, n, dy1.p, n);
        ctx.setReferenceZ0(false);
        sys.apply(dx.p, n, dy0.p, n);
        ctx.synchronize();
        {
            const auto y0 = dy0.down(), y1 = dy1.down(); // (atomic adds: equal to rounding, not bitwise)
            double     d = 0., s = 0.;
            for (size_t i = 0; i < n; ++i)
            {
                d = std::fmax(d, std::fabs(y0[i] - y1[i]));
                s = std::fmax(s, std::fabs(y0[i]));
            }
            failures += !(d < 1e-13 * s);
        }
        // Dirichlet values T = x on the boundary nodes through computeValuesAtNodes (the Dirichlet value kernel of
        // tests/Diffusion2D.hpp:49-50 in 3-D), then two right-hand sides solved as one multivector
        const auto all = mesh.boundarySides();
        DevVec     g{2 * n}, work{2 * n};
        (void)hipMemset(g.p, 0, 2 * n * sizeof(double));
        (void)hipMemset(work.p, 0, 2 * n * sizeof(double));
        l3k::computeValuesAtNodes< DiffusionParams >(dmesh, L3K_RESIDUAL_COORDX3D, nullptr, unk0, &all, nullptr, 0, 0., int64_t(n), g.p, work.p);
        ctx.synchronize();
        {
            const auto gv = g.down();
            double     worst = 0.;
            for (int64_t i = 0; i < mesh.view().n_owned_nodes; ++i)
                if (mesh.view().node_boundary[i])
                {
                    // (node coordinates from the element vertices would need the map; the cube is perturbed in its interior only:
                    // boundary nodes of an unperturbed boundary lie on the regular grid)
                    const int64_t gid = mesh.view().node_grid_id[i], G = ne * p + 1;
                    const double  X = double(gid % G) / (G - 1);
                    worst = std::fmax(worst, std::fabs(gv[size_t(i) * U] - X));
                }
            std::printf("Dirichlet values at the boundary nodes: max error %.2e\n", worst);
            failures += !(worst < 1e-12);
        }
        DevVec diag{n}, rhs{2 * n}, minv{n}, sol{2 * n};
        (void)hipMemset(diag.p, 0, n * sizeof(double));
        (void)hipMemset(rhs.p, 0, 2 * n * sizeof(double));
        (void)hipMemset(sol.p, 0, 2 * n * sizeof(double));
        sys.diagAndRhs(g.p, n, diag.p, rhs.p, n);
        l3k::jacobiInverse(dmesh, diag.p, int64_t(n), minv.p);
        const auto res = sys.solve(rhs.p, n, sol.p, n, 2, minv.p, {1e-10, 5000, 2, 1});
        ctx.synchronize();
        std::printf("multivector solve: %d and %d iterations\n", res[0].iterations, res[1].iterations);
        failures += !(res.size() == 2 && res[0].converged && res[1].converged && res[1].iterations <= 1); // (column 1: zero data)
    }
    { // (5) <x, A x> from the operator's own pass, order elevation on the device, results file round trip
        constexpr int   p = 4, U = 4, ne = 3;
        l3k::CubeMesh   mesh{{ne, ne, ne}, p, {1, 1, 1}, 0, 0.1};
        const int       unk0[1] = {0};
        const auto      mask    = mesh.dirichletMask(U, unk0, 0x3f);
        l3k::DeviceMesh dmesh{ctx, mesh, U, mask.data()};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
        const size_t          n = size_t(dmesh.nOwnedDofs());
        std::vector< double > x(n);
        for (auto& v : x)
            v = dist(prng);
        DevVec dx{n}, dy{n}, ds{8};
        dx.up(x);
        sys.applyEnergy(dx.p, dy.p, ds.p);
        ctx.synchronize();
        const auto y = dy.down(), sblock = ds.down();
        double     xy = 0.;
        for (size_t i = 0; i < n; ++i)
            xy += x[i] * y[i];
        std::printf("<x, A x>: from the apply %.15e, dot product %.15e\n", sblock[1], xy);
        failures += !(std::fabs(sblock[1] - xy) < 1e-11 * std::fabs(xy));

        // the same cube as an order-1 connectivity, elevated on the device: (ne*p + 1)^3 nodes
        std::vector< uint32_t > conn;
        const auto              vid = [&](int i, int j, int k) { return uint32_t(i + (ne + 1) * (j + (ne + 1) * k)); };
        for (int k = 0; k < ne; ++k)
            for (int j = 0; j < ne; ++j)
                for (int i = 0; i < ne; ++i)
                    for (int v = 0; v < 8; ++v)
                        conn.push_back(vid(i + (v & 1), j + ((v >> 1) & 1), k + (v >> 2)));
        int64_t    n_nodes = 0;
        const auto en      = l3k::elevateOrder(ctx, conn, (ne + 1) * (ne + 1) * (ne + 1), p, n_nodes);
        std::printf("elevated %d^3 hexes to order %d: %lld nodes\n", ne, p, (long long)n_nodes);
        failures += !(n_nodes == int64_t(ne * p + 1) * (ne * p + 1) * (ne * p + 1) && en.size() == size_t(ne * ne * ne) * 125);

        // results file: two fields of the solution vector, saved as two ranks' slices, loaded back by node id
        const int64_t         nn = int64_t(n) / U, half = nn / 2;
        std::vector< double > f(2 * size_t(nn));
        for (int64_t i = 0; i < nn; ++i)
            f[size_t(i)] = y[size_t(i) * U], f[size_t(nn + i)] = y[size_t(i) * U + 2];
        const char* path = "/tmp/l3k_cpp_test.res";
        l3k::saveResults(path, "from the C++ shim", 2, nn, half, nn - half, f.data() + half, size_t(nn), false);
        l3k::saveResults(path, "from the C++ shim", 2, nn, 0, half, f.data(), size_t(nn), true);
        const std::vector< int64_t > ids{nn - 1, 0, half, half - 1};
        const auto                   back = l3k::loadResults(path, 1, ids);
        bool                         same = true;
        for (size_t i = 0; i < ids.size(); ++i)
            same = same && back[i] == y[size_t(ids[i]) * U + 2];
        std::printf("results file round trip: %s\n", same ? "ok" : "MISMATCH");
        failures += !same;
    }
    { // partitioned apply through the library's RCCL halo (l3k::Halo): a world of one rank without neighbours runs the whole
      // schedule (scale, pack, import || interior, border, export || interior, unpack, Dirichlet rows) and must equal the
      // plain apply; deterministic mode gives the same numbers to rounding, twice bitwise the same
        constexpr int         p = 4, U = 4;
        l3k::CubeMesh         mesh{{4, 3, 3}, p, {1, 1, 1}, 0, 0.1};
        const int             unknown0[] = {0};
        const auto            mask = mesh.dirichletMask(U, unknown0);
        l3k::DeviceMesh       dmesh{ctx, mesh, U, mask.data()};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
        l3k::Halo             halo{ctx, mesh, U, l3k::Halo::uniqueId(), 0, 1};
        const size_t          n = size_t(dmesh.nOwnedDofs());
        std::vector< double > x(n);
        for (auto& v : x)
            v = dist(prng);
        DevVec dx{n}, dy{n}, dz{n};
        dx.up(x);
        sys.apply(dx.p, n, dy.p, n, 1, 1.5, 0.);
        sys.apply(halo, dx.p, n, dz.p, n, 1, 1.5, 0.);
        ctx.synchronize();
        const auto y = dy.down(), z = dz.down();
        double     err = 0., nrm = 0.;
        for (size_t i = 0; i < n; ++i)
            err += (y[i] - z[i]) * (y[i] - z[i]), nrm += y[i] * y[i];
        std::printf("partitioned apply (one rank, RCCL communicator of size 1): |dy| / |y| = %.3e, %lld ghost dofs\n", std::sqrt(err / nrm),
                    (long long)halo.nGhostDofs());
        failures += !(std::sqrt(err / nrm) < 1e-13);
        l3k::Context dctx{0};
        dctx.setDeterministic();
        l3k::DeviceMesh       dmesh2{dctx, mesh, U, mask.data()};
        l3k::MatrixFreeSystem sys2{dmesh2, L3K_KERNEL_DIFFUSION3D, params};
        DevVec                d1{n}, d2{n};
        sys2.apply(dx.p, n, d1.p, n, 1, 1.5, 0.);
        sys2.apply(dx.p, n, d2.p, n, 1, 1.5, 0.);
        dctx.synchronize();
        const auto a1 = d1.down(), a2 = d2.down();
        bool       same = true;
        double     e2 = 0.;
        for (size_t i = 0; i < n; ++i)
            same = same && a1[i] == a2[i], e2 += (a1[i] - y[i]) * (a1[i] - y[i]);
        std::printf("deterministic mode: two applies bitwise %s, vs atomic mode %.3e\n", same ? "equal" : "DIFFERENT", std::sqrt(e2 / nrm));
        failures += !(same && std::sqrt(e2 / nrm) < 1e-13);
    }
    { // two ranks as two threads of this process (each its own Context and stream), the exchange through the library's
      // in-process transport (l3k::InprocGroup): l3k_mf_apply_dist of both ranks equals the one-rank apply of the whole mesh
        constexpr int         p = 2, U = 4, world = 2;
        const int             unknown0[] = {0};
        l3k::CubeMesh         whole{{4, 2, 2}, p, {1, 1, 1}, 0, 0.1};
        const auto            wmask = whole.dirichletMask(U, unknown0);
        l3k::DeviceMesh       wmesh{ctx, whole, U, wmask.data()};
        l3k::MatrixFreeSystem wsys{wmesh, L3K_KERNEL_DIFFUSION3D, params};
        const auto&           wv = whole.view();
        const size_t          nw = size_t(wmesh.nOwnedDofs());
        const auto            value_of = [](int64_t grid_id, int u) { return std::sin(0.37 * double(grid_id) + u); };
        std::vector< double > xw(nw);
        for (size_t i = 0; i < nw; ++i)
            xw[i] = value_of(wv.node_grid_id[i / U], int(i % U));
        DevVec dxw{nw}, dyw{nw};
        dxw.up(xw);
        wsys.apply(dxw.p, nw, dyw.p, nw, 1, 1.25, 0.);
        ctx.synchronize();
        const auto                    yw = dyw.down();
        std::map< int64_t, size_t >   row_of;
        for (int64_t i = 0; i < wv.n_owned_nodes; ++i)
            row_of[wv.node_grid_id[i]] = size_t(i);
        l3k::InprocGroup      group{world};
        std::vector< double > err2(world, 0.), nrm2(world, 0.);
        std::vector< int >    bad(world, 0);
        const auto            body = [&](int rank) {
            try
            {
                l3k::Context          rctx{0};
                l3k::CubeMesh         mesh{{4, 2, 2}, p, {2, 1, 1}, rank, 0.1};
                const auto            mask = mesh.dirichletMask(U, unknown0);
                l3k::DeviceMesh       dmesh{rctx, mesh, U, mask.data()};
                l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
                l3k::Halo             halo{rctx, mesh, U, group.transport(rank), rank, world};
                const auto&           v = mesh.view();
                const size_t          n = size_t(dmesh.nOwnedDofs());
                std::vector< double > x(n);
                for (size_t i = 0; i < n; ++i)
                    x[i] = value_of(v.node_grid_id[i / U], int(i % U));
                DevVec dx{n}, dy{n};
                dx.up(x);
                for (int rep = 0; rep < 3; ++rep)
                    sys.apply(halo, dx.p, n, dy.p, n, 1, 1.25, 0.);
                rctx.synchronize();
                const auto y = dy.down();
                for (size_t i = 0; i < n; ++i)
                {
                    const double ref = yw[row_of.at(v.node_grid_id[i / U]) * U + i % U];
                    err2[rank] += (y[i] - ref) * (y[i] - ref), nrm2[rank] += ref * ref;
                }
            }
            catch (const std::exception& e)
            {
                std::printf("rank %d: %s\n", rank, e.what());
                bad[rank] = 1;
            }
        };
        std::thread t0{body, 0}, t1{body, 1};
        t0.join();
        t1.join();
        const double rel = std::sqrt((err2[0] + err2[1]) / (nrm2[0] + nrm2[1]));
        std::printf("two thread ranks, in-process transport: |dy| / |y| = %.3e\n", rel);
        failures += bad[0] + bad[1] + !(rel < 1e-12);
    }
    try
    { // error behaviour: too many columns -> exception (algsys/MatrixFreeSystem.hpp:1035-1037)
        l3k::CubeMesh         mesh{{1, 1, 1}, 2};
        l3k::DeviceMesh       dmesh{ctx, mesh, 4};
        l3k::MatrixFreeSystem sys{dmesh, L3K_KERNEL_DIFFUSION3D, params};
        DevVec                a{size_t(dmesh.nOwnedDofs()) * 2}, b{size_t(dmesh.nOwnedDofs()) * 2};
        sys.apply(a.p, dmesh.nOwnedDofs(), b.p, dmesh.nOwnedDofs(), 2);
        std::printf("expected an exception\n");
        ++failures;
    }
    catch (const std::runtime_error& e)
    {
        std::printf("error reported as exception: %s\n", e.what());
    }
    std::printf(failures ? "FAILED\n" : "OK\n");
    return failures;
}
