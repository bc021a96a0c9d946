// user_kernels.hpp -- the registered operator kernels (the "user code" of the reference, common/KernelInterface.hpp).
//
// Each kernel is a POD functor with the reference's (in, out) signature plus a `params` member and an optional
// parameter block as data members.  To add a kernel: define the functor, add it to L3K_FOR_EACH_KERNEL with a fresh id
// and list the (order, nq, ncols) shapes to instantiate in L3K_FOR_EACH_INSTANCE; rebuild (python -m l3ster_amd.build).
#ifndef L3K_USER_KERNELS_HPP
#define L3K_USER_KERNELS_HPP

#include "l3k/kernel_interface.hpp"

#include <cmath>

namespace l3k::kernels
{
// benchmarks/Diffusion3D.hpp:50-79 (== benchmarks/Kernels.hpp:85-113, tests/Kernels.hpp:55-81 with s = 0)
struct Diffusion3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 7, .n_unknowns = 4};
    double                        k = 1., s = 1.; // diffusivity, source

    template < typename In, typename Out >
    L3K_HD void operator()(const In&, Out& out) const
    {
        auto& [operators, rhs] = out;
        auto& [A0, Ax, Ay, Az] = operators;
        // -k * div q = s
        Ax(0, 1) = -k;
        Ay(0, 2) = -k;
        Az(0, 3) = -k;
        rhs[0]   = s;
        // grad T = q
        A0(1, 1) = -1.;
        Ax(1, 0) = 1.;
        A0(2, 2) = -1.;
        Ay(2, 0) = 1.;
        A0(3, 3) = -1.;
        Az(3, 0) = 1.;
        // rot q = 0
        Ay(4, 3) = 1.;
        Az(4, 2) = -1.;
        Ax(5, 3) = -1.;
        Az(5, 1) = 1.;
        Ax(6, 2) = 1.;
        Ay(6, 1) = -1.;
    }
};

// tests/Kernels.hpp:84-118: variable diffusivity passed as an external field
struct Diffusion3DVar
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 7, .n_unknowns = 4, .n_fields = 1};

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [field_vals, field_ders, _] = in;
        const auto lambda                       = field_vals[0];
        const auto& [dx, dy, dz]                = field_ders;
        const auto dl_dx                        = dx[0];
        const auto dl_dy                        = dy[0];
        const auto dl_dz                        = dz[0];

        auto& [operators, rhs] = out;
        auto& [A0, Ax, Ay, Az] = operators;
        // -grad k * q - k * div q = s
        A0(0, 1) = -dl_dx;
        A0(0, 2) = -dl_dy;
        A0(0, 3) = -dl_dz;
        Ax(0, 1) = -lambda;
        Ay(0, 2) = -lambda;
        Az(0, 3) = -lambda;
        // grad T = q
        A0(1, 1) = -1.;
        Ax(1, 0) = 1.;
        A0(2, 2) = -1.;
        Ay(2, 0) = 1.;
        A0(3, 3) = -1.;
        Az(3, 0) = 1.;
        // curl q = 0
        Ay(4, 3) = 1.;
        Az(4, 2) = -1.;
        Ax(5, 3) = -1.;
        Az(5, 1) = 1.;
        Ax(6, 2) = 1.;
        Ay(6, 1) = -1.;
    }
};

// Config-5 synthetic: advection-diffusion-reaction in first-order form, unknowns (c, q),
// velocity u = 3 interpolated fields (the karman-style "kernel reads interpolated field values"):
//   sigma c + u . grad c - k div q = s,   grad c = q,   rot q = 0
struct AdvDiff3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 7, .n_unknowns = 4, .n_fields = 3};
    // the velocity enters by value only: lets the device kernels skip the derivative sweeps of the three fields
    static constexpr bool         field_derivatives = false;
    double                        k = 1., sigma = 1., s = 1.;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [u, u_ders, point] = in;
        auto& [operators, rhs]         = out;
        auto& [A0, Ax, Ay, Az]         = operators;
        A0(0, 0) = sigma;
        Ax(0, 0) = u[0];
        Ay(0, 0) = u[1];
        Az(0, 0) = u[2];
        Ax(0, 1) = -k;
        Ay(0, 2) = -k;
        Az(0, 3) = -k;
        rhs[0]   = s;
        A0(1, 1) = -1.;
        Ax(1, 0) = 1.;
        A0(2, 2) = -1.;
        Ay(2, 0) = 1.;
        A0(3, 3) = -1.;
        Az(3, 0) = 1.;
        Ay(4, 3) = 1.;
        Az(4, 2) = -1.;
        Ax(5, 3) = -1.;
        Az(5, 1) = 1.;
        Ax(6, 2) = 1.;
        Ay(6, 1) = -1.;
    }
};

// Domain kernel whose operators AND rhs read the space-time point (synthetic; the reference's examples do:
// examples/03-advection-2D/source.cpp:52-66 takes the velocity from point.space.y(), examples/04-periodic-bc/source.cpp:88-89
// reads point.time): Diffusion3D with k(x,t) = k (1 + 0.3 sin(x + 2y + 3z + t)), -(1 + 0.2 cos(z - t)) instead of -1 on the
// flux rows and the source s (1 + x y - z t / 2)
struct Diffusion3DPoint
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 7, .n_unknowns = 4};
    double                        k = 1., s = 1.;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [field_vals, field_ders, point] = in;
        const double x = point.space.x(), y = point.space.y(), z = point.space.z(), t = point.time;
        const double kv = k * (1. + 0.3 * sin(x + 2. * y + 3. * z + t));
        const double c  = -(1. + 0.2 * cos(z - t));

        auto& [operators, rhs] = out;
        auto& [A0, Ax, Ay, Az] = operators;
        Ax(0, 1) = -kv;
        Ay(0, 2) = -kv;
        Az(0, 3) = -kv;
        rhs[0]   = s * (1. + x * y - 0.5 * z * t);
        A0(1, 1) = c;
        Ax(1, 0) = 1.;
        A0(2, 2) = c;
        Ay(2, 0) = 1.;
        A0(3, 3) = c;
        Az(3, 0) = 1.;
        Ay(4, 3) = 1.;
        Az(4, 2) = -1.;
        Ax(5, 3) = -1.;
        Az(5, 1) = 1.;
        Ax(6, 2) = 1.;
        Ay(6, 1) = -1.;
    }
};

// Scalar advection, one unknown: the 3-D analogue of examples/04-periodic-bc/source.cpp:60-75 (BDF3: the three fields are the
// solution at the previous time steps) with the point-dependent velocity of examples/03-advection-2D/source.cpp:52-66
struct Advection3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1, .n_unknowns = 1, .n_fields = 3};
    static constexpr bool         field_derivatives = false; // the history enters by value only
    double                        dt = .02;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [field_vals, field_ders, point] = in;
        const double y_scaled = point.space.y() * 2. - 1., z_scaled = point.space.z() * 2. - 1.;
        const double vx = (1. - y_scaled * y_scaled) * (1. - .5 * z_scaled * z_scaled); // parabolic profiles
        const double vy = .25 * point.space.x();
        const double vz = -.125;

        constexpr double bdf_leading_coef = 11. / 6.;
        constexpr double bdf_coefs[3]     = {3., -1.5, 1. / 3.};

        auto& [operators, rhs] = out;
        auto& [A0, A1, A2, A3] = operators;
        A0(0, 0) = bdf_leading_coef;
        A1(0, 0) = vx * dt;
        A2(0, 0) = vy * dt;
        A3(0, 0) = vz * dt;
        rhs[0]   = field_vals[0] * bdf_coefs[0] + field_vals[1] * bdf_coefs[1] + field_vals[2] * bdf_coefs[2];
    }
};

// Div-curl system (synthetic; an odd number of unknowns): a(x) div u = f, curl u = omega, a zeroth-order coupling on row 1
struct DivCurl3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 4, .n_unknowns = 3};
    double                        f = 1.;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const double a         = 1. + .5 * in.point.space.x() * in.point.space.z();
        auto& [operators, rhs] = out;
        auto& [A0, Ax, Ay, Az] = operators;
        Ax(0, 0) = a;
        Ay(0, 1) = a;
        Az(0, 2) = a;
        rhs[0]   = f;
        Ay(1, 2) = 1.;
        Az(1, 1) = -1.;
        Az(2, 0) = 1.;
        Ax(2, 2) = -1.;
        Ax(3, 1) = 1.;
        Ay(3, 0) = -1.;
        A0(1, 0) = .1 * in.point.space.y();
        rhs[1]   = .5;
    }
};

// Mass-type kernel, A0 = I (no derivative terms), rhs = (1, 2): the known answers that pin the quadrature weight times
// detJ in the domain path -- sum_ij K_e[(i,u),(j,u)] = volume (partition of unity), sum_i F_e[(i,u)] = rhs_u * volume.
// Not in the reference: its least-squares solve tests hold for any positive weight per quadrature point.
struct Mass3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 2, .n_unknowns = 2};

    template < typename In, typename Out >
    L3K_HD void operator()(const In&, Out& out) const
    {
        auto& [operators, rhs] = out;
        auto& [A0, Ax, Ay, Az] = operators;
        A0(0, 0) = 1.;
        A0(1, 1) = 1.;
        rhs[0]   = 1.;
        rhs[1]   = 2.;
    }
};

// ---- boundary equation kernels: the input additionally carries the outward unit normal ----------------------------
// 3-D twin of tests/Kernels.hpp:120-128 (adiabatic wall of the first-order diffusion system): q . n = 0
struct Adiabatic3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1, .n_unknowns = 4};

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [vals, ders, point, normal] = in;
        auto& [operators, rhs]                  = out;
        auto& [A0, A1, A2, A3]                  = operators;
        A0(0, 1) = normal[0];
        A0(0, 2) = normal[1];
        A0(0, 3) = normal[2];
    }
};
// Robin condition q . n + h T = h T_inf (synthetic: A0 on the primary unknown and a non-zero boundary rhs)
struct Robin3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1, .n_unknowns = 4};
    double                        h = 1., t_inf = 0.;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [vals, ders, point, normal] = in;
        auto& [operators, rhs]                  = out;
        auto& [A0, A1, A2, A3]                  = operators;
        A0(0, 0) = h;
        A0(0, 1) = normal[0];
        A0(0, 2) = normal[1];
        A0(0, 3) = normal[2];
        rhs[0]   = h * t_inf;
    }
};

// Robin condition with coefficients read from the point and the time (synthetic): q . n + h(x, t) T = h(x, t) T_inf(x)
struct RobinPoint3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1, .n_unknowns = 4};
    double                        h0 = 1., t0 = 0.;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [vals, ders, point, normal] = in;
        const double h         = h0 * (1. + .5 * sin(point.space.x() - point.space.y() + point.time));
        auto& [operators, rhs] = out;
        auto& [A0, A1, A2, A3] = operators;
        A0(0, 0) = h;
        A0(0, 1) = normal[0];
        A0(0, 2) = normal[1];
        A0(0, 3) = normal[2];
        rhs[0]   = h * t0 * (1. + point.space.z());
    }
};

// Boundary kernel with derivative operators (synthetic): n . grad T + c d(q_x)/dx + h T = g -- fills A1..A3, so the side
// kernel's normal-derivative path (every node of the element takes part) is exercised
struct NormalFlux3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1, .n_unknowns = 4};
    double                        h = 1., g = 0., c = 0.5;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        const auto& [vals, ders, point, normal] = in;
        auto& [operators, rhs]                  = out;
        auto& [A0, A1, A2, A3]                  = operators;
        A0(0, 0) = h;
        A1(0, 0) = normal[0];
        A2(0, 0) = normal[1];
        A3(0, 0) = normal[2];
        A1(0, 1) = c;
        rhs[0]   = g;
    }
};

// ---- residual kernels (integrals, L2 norms): out[n_equations] from the interpolated fields ------------------------
// benchmarks/Diffusion3D.hpp:81-103: residuals of the first-order diffusion system for the fields (T, qx, qy, qz)
struct Diffusion3DError
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 4, .n_fields = 4};
    double                        k = 1., s = 1.;

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& error) const
    {
        const auto& vals                     = in.field_vals;
        const auto& ders                     = in.field_ders;
        const auto& [x_ders, y_ders, z_ders] = ders;
        error[0] = k * (x_ders[1] + y_ders[2] + z_ders[3]) + s;
        error[1] = x_ders[0] - vals[1];
        error[2] = y_ders[0] - vals[2];
        error[3] = z_ders[0] - vals[3];
    }
};
// 3-D twin of tests/Diffusion2D.hpp:84-92: error against the exact solution T = x, q = (1, 0, 0)
struct Linear3DError
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 4, .n_fields = 4};

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& error) const
    {
        const auto& vals = in.field_vals;
        error[0]         = vals[0] - in.point.space.x();
        error[1]         = vals[1] - 1.;
        error[2]         = vals[2];
        error[3]         = vals[3];
    }
};
// 3-D twin of the Dirichlet value kernel of tests/Diffusion2D.hpp:49-50: out[0] = x
struct CoordX3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1};

    template < typename In, typename Out >
    L3K_HD void operator()(const In& in, Out& out) const
    {
        out[0] = in.point.space.x();
    }
};
// tests/MappingTests.cpp:567-569: integrand 1 (volume / area)
struct Unit3D
{
    static constexpr KernelParams params{.dimension = 3, .n_equations = 1};

    template < typename In, typename Out >
    L3K_HD void operator()(const In&, Out& out) const
    {
        out[0] = 1.;
    }
};
} // namespace l3k::kernels

// id, functor type, name
#define L3K_FOR_EACH_KERNEL(X)                                                                                         \
    X(0, ::l3k::kernels::Diffusion3D, "diffusion3d")                                                                   \
    X(1, ::l3k::kernels::Diffusion3DVar, "diffusion3d_var")                                                            \
    X(4, ::l3k::kernels::AdvDiff3D, "advdiff3d")                                                                       \
    X(8, ::l3k::kernels::Mass3D, "mass3d")                                                                             \
    X(10, ::l3k::kernels::Diffusion3DPoint, "diffusion3d_point")                                                       \
    X(11, ::l3k::kernels::Advection3D, "advection3d")                                                                  \
    X(12, ::l3k::kernels::DivCurl3D, "divcurl3d")

// boundary equation kernels (ids continue the numbering above; 5 is the 2-D adiabatic kernel of the CPU oracle)
#define L3K_FOR_EACH_BOUNDARY_KERNEL(X)                                                                                \
    X(6, ::l3k::kernels::Adiabatic3D, "adiabatic3d")                                                                   \
    X(7, ::l3k::kernels::Robin3D, "robin3d")                                                                           \
    X(9, ::l3k::kernels::NormalFlux3D, "normalflux3d")                                                                 \
    X(14, ::l3k::kernels::RobinPoint3D, "robinpoint3d")

// residual kernels (own id space; 1 and 3 are the 2-D kernels of the CPU oracle)
#define L3K_FOR_EACH_RESIDUAL_KERNEL(X)                                                                                \
    X(0, ::l3k::kernels::Diffusion3DError, "diffusion3d_error")                                                        \
    X(2, ::l3k::kernels::Linear3DError, "linear3d_error")                                                              \
    X(4, ::l3k::kernels::Unit3D, "unit3d")                                                                             \
    X(6, ::l3k::kernels::CoordX3D, "coordx3d")

// Shapes instantiated on the device: (functor, order p, quadrature points per direction nq, columns R).
// nq = value_order*p + derivative_order*(p-1) + 1 (algsys/AssembleLocalSystem.hpp:32-35).
#define L3K_FOR_EACH_INSTANCE(X)                                                                                       \
    X(::l3k::kernels::Diffusion3D, 1, 2, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 2, 3, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 3, 4, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 4, 5, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 5, 6, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 6, 7, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 7, 8, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 8, 9, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 3, 7, 1)                                                                            \
    X(::l3k::kernels::Diffusion3D, 3, 7, 3)                                                                            \
    X(::l3k::kernels::Diffusion3D, 2, 3, 2)                                                                            \
    X(::l3k::kernels::Diffusion3DVar, 3, 7, 1)                                                                         \
    X(::l3k::kernels::Diffusion3DVar, 3, 7, 2)                                                                         \
    X(::l3k::kernels::Diffusion3DVar, 4, 5, 1)                                                                         \
    X(::l3k::kernels::AdvDiff3D, 2, 3, 1)                                                                              \
    X(::l3k::kernels::AdvDiff3D, 2, 3, 2)                                                                              \
    X(::l3k::kernels::AdvDiff3D, 4, 5, 1)                                                                              \
    X(::l3k::kernels::Mass3D, 3, 7, 1)                                                                                 \
    X(::l3k::kernels::Mass3D, 2, 3, 1)                                                                                 \
    X(::l3k::kernels::Diffusion3DPoint, 2, 3, 1)                                                                       \
    X(::l3k::kernels::Diffusion3DPoint, 2, 5, 1)                                                                       \
    X(::l3k::kernels::Diffusion3DPoint, 4, 5, 1)                                                                       \
    X(::l3k::kernels::Diffusion3DPoint, 6, 7, 1)                                                                       \
    X(::l3k::kernels::Advection3D, 2, 3, 1)                                                                            \
    X(::l3k::kernels::Advection3D, 4, 5, 1)                                                                            \
    X(::l3k::kernels::Advection3D, 6, 7, 1)                                                                            \
    X(::l3k::kernels::DivCurl3D, 2, 3, 1)                                                                              \
    X(::l3k::kernels::DivCurl3D, 4, 5, 1)                                                                              \
    X(::l3k::kernels::DivCurl3D, 6, 7, 1)

#define L3K_FOR_EACH_BOUNDARY_INSTANCE(X)                                                                              \
    X(::l3k::kernels::Adiabatic3D, 2, 3, 1)                                                                            \
    X(::l3k::kernels::Adiabatic3D, 4, 5, 1)                                                                            \
    X(::l3k::kernels::Adiabatic3D, 6, 7, 1)                                                                            \
    X(::l3k::kernels::Robin3D, 2, 3, 1)                                                                                \
    X(::l3k::kernels::Robin3D, 2, 3, 2)                                                                                \
    X(::l3k::kernels::Robin3D, 3, 7, 1)                                                                                \
    X(::l3k::kernels::Robin3D, 4, 5, 1)                                                                                \
    X(::l3k::kernels::NormalFlux3D, 2, 3, 1)                                                                           \
    X(::l3k::kernels::NormalFlux3D, 2, 3, 2)                                                                           \
    X(::l3k::kernels::NormalFlux3D, 3, 7, 1)                                                                           \
    X(::l3k::kernels::NormalFlux3D, 4, 5, 1)                                                                           \
    X(::l3k::kernels::RobinPoint3D, 2, 3, 1)                                                                           \
    X(::l3k::kernels::RobinPoint3D, 2, 3, 2)                                                                           \
    X(::l3k::kernels::RobinPoint3D, 3, 7, 1)                                                                           \
    X(::l3k::kernels::RobinPoint3D, 4, 5, 1)

// (functor, order p, nq); computeNormL2 doubles the quadrature orders: nq = 2p+1 for the default options
#define L3K_FOR_EACH_RESIDUAL_INSTANCE(X)                                                                              \
    X(::l3k::kernels::Diffusion3DError, 2, 5)                                                                          \
    X(::l3k::kernels::Diffusion3DError, 4, 9)                                                                          \
    X(::l3k::kernels::Diffusion3DError, 6, 13)                                                                         \
    X(::l3k::kernels::Diffusion3DError, 6, 7)                                                                          \
    X(::l3k::kernels::Linear3DError, 2, 5)                                                                             \
    X(::l3k::kernels::Linear3DError, 4, 9)                                                                             \
    X(::l3k::kernels::Unit3D, 1, 6)                                                                                    \
    X(::l3k::kernels::Unit3D, 2, 3)                                                                                    \
    X(::l3k::kernels::Unit3D, 2, 5)                                                                                    \
    X(::l3k::kernels::CoordX3D, 2, 3)                                                                                  \
    X(::l3k::kernels::CoordX3D, 4, 5)                                                                                  \
    X(::l3k::kernels::CoordX3D, 6, 7)

#endif
