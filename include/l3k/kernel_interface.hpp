// kernel_interface.hpp -- the kernel-definition API of the reference, usable in host AND device code.
//
// Mirrors common/KernelInterface.hpp:13-57,102-119,178-183 of the reference: a kernel is a callable
//     (const DomainInput& in, Result& out)
// that fills out.operators[0..D] (each E x U) and out.rhs (E x R) at one point, given the interpolated external fields
// in.field_vals[F], their physical derivatives in.field_ders[D][F] and the space-time point in.point.  The reference
// uses Eigen fixed-size matrices; here a minimal fixed-size matrix gives exactly the operations the reference's kernels
// use (A(i,j), rhs[i], rhs(i,0), structured bindings over `operators`, `field_ders`, and the input/result structs),
// so the lambdas of benchmarks/Kernels.hpp, tests/Kernels.hpp and examples/* port by adding __host__ __device__.
#ifndef L3K_KERNEL_INTERFACE_HPP
#define L3K_KERNEL_INTERFACE_HPP

#include <cstddef>
#include <utility>

#if defined(__HIPCC__)
#define L3K_HD __host__ __device__ __attribute__((always_inline)) inline
#else
#define L3K_HD inline
#endif

namespace l3k
{
// common/KernelInterface.hpp:13-20
struct KernelParams
{
    int dimension;
    int n_equations;
    int n_unknowns = 1;
    int n_fields   = 0;
    int n_rhs      = 1;
};

// fixed-size dense matrix, zero-initialised (detail::initKernelResult, common/KernelInterface.hpp:61-68)
template < int Rows, int Cols >
struct Matrix
{
    double v[Rows * Cols > 0 ? Rows * Cols : 1];
    L3K_HD Matrix()
    {
        for (int i = 0; i < Rows * Cols; ++i)
            v[i] = 0.;
    }
    L3K_HD double&       operator()(int i, int j) { return v[i * Cols + j]; }
    L3K_HD const double& operator()(int i, int j) const { return v[i * Cols + j]; }
    L3K_HD double&       operator[](int i) { return v[i * Cols]; } // Eigen vector-style access: rhs[i] == rhs(i, 0)
    L3K_HD const double& operator[](int i) const { return v[i * Cols]; }
    static constexpr int rows() { return Rows; }
    static constexpr int cols() { return Cols; }
};

// field values / one derivative direction of the F external fields.  Like the reference's std::array it decomposes with
// structured bindings (benchmarks/Kernels.hpp:5-9: `const auto& [u, v, w, p, ox, oy, oz] = vals;`) and iterates
// (examples/04-periodic-bc/source.cpp:74: std::transform_reduce over field_vals)
template < int F >
struct FieldArray
{
    double               v[F > 0 ? F : 1];
    L3K_HD double&       operator[](int i) { return v[i]; }
    L3K_HD const double& operator[](int i) const { return v[i]; }
    static constexpr int size() { return F; }
    L3K_HD const double* begin() const { return v; }
    L3K_HD const double* end() const { return v + F; }
    template < std::size_t I >
    L3K_HD const double& get() const
    {
        static_assert(I < std::size_t(F > 0 ? F : 1));
        return v[I];
    }
    template < std::size_t I >
    L3K_HD double& get()
    {
        static_assert(I < std::size_t(F > 0 ? F : 1));
        return v[I];
    }
};

// common/Structs.hpp: Point<3>, SpaceTimePoint
struct Point3
{
    double        c[3];
    L3K_HD double x() const { return c[0]; }
    L3K_HD double y() const { return c[1]; }
    L3K_HD double z() const { return c[2]; }
    L3K_HD double operator[](int i) const { return c[i]; }
};
struct SpaceTimePoint
{
    Point3 space;
    double time;
};

// common/KernelInterface.hpp:29-57
template < KernelParams params >
struct KernelInterface
{
    using Operator = Matrix< params.n_equations, params.n_unknowns >;
    using Rhs      = Matrix< params.n_equations, params.n_rhs >;
    struct Result
    {
        Operator operators[params.dimension + 1];
        Rhs      rhs;
    };
    using FieldVals = FieldArray< params.n_fields >;
    struct DomainInput
    {
        FieldVals      field_vals;
        FieldVals      field_ders[params.dimension];
        SpaceTimePoint point;
    };
    // boundary kernels additionally see the outward unit normal (common/KernelInterface.hpp:48-57)
    struct Normal
    {
        double               c[params.dimension];
        L3K_HD double        operator[](int i) const { return c[i]; }
        static constexpr int size() { return params.dimension; }
    };
    struct BoundaryInput
    {
        FieldVals      field_vals;
        FieldVals      field_ders[params.dimension];
        SpaceTimePoint point;
        Normal         normal;
    };
};

// common/KernelInterface.hpp:102-119: zero-initialise the result, then invoke the user callable
template < typename Kernel, KernelParams params >
struct DomainEquationKernel
{
    static constexpr KernelParams parameters = params;
    Kernel                        kernel;

    L3K_HD typename KernelInterface< params >::Result
    operator()(const typename KernelInterface< params >::DomainInput& in) const
    {
        typename KernelInterface< params >::Result out{};
        kernel(in, out);
        return out;
    }
};

// common/KernelInterface.hpp:178-183
template < KernelParams params, typename Kernel >
constexpr auto wrapDomainEquationKernel(Kernel kernel)
{
    return DomainEquationKernel< Kernel, params >{kernel};
}

// Boundary equation kernel: same result type, BoundaryInput (common/KernelInterface.hpp:132-153,185-190)
template < typename Kernel, KernelParams params >
struct BoundaryEquationKernel
{
    static constexpr KernelParams parameters = params;
    Kernel                        kernel;

    L3K_HD typename KernelInterface< params >::Result
    operator()(const typename KernelInterface< params >::BoundaryInput& in) const
    {
        typename KernelInterface< params >::Result out{};
        kernel(in, out);
        return out;
    }
};
template < KernelParams params, typename Kernel >
constexpr auto wrapBoundaryEquationKernel(Kernel kernel)
{
    return BoundaryEquationKernel< Kernel, params >{kernel};
}

// Residual kernels (common/KernelInterface.hpp:121-130,155-176,192-204): (in, out) with out = Rhs (n_equations x 1,
// zero-initialised); the same callable may serve as domain and as boundary residual (tests/Diffusion2D.hpp:84-95).
template < typename Kernel, KernelParams params >
struct ResidualDomainKernel
{
    static constexpr KernelParams parameters = params;
    Kernel                        kernel;

    L3K_HD typename KernelInterface< params >::Rhs operator()(const typename KernelInterface< params >::DomainInput& in) const
    {
        typename KernelInterface< params >::Rhs out{};
        kernel(in, out);
        return out;
    }
};
template < typename Kernel, KernelParams params >
struct ResidualBoundaryKernel
{
    static constexpr KernelParams parameters = params;
    Kernel                        kernel;

    L3K_HD typename KernelInterface< params >::Rhs operator()(const typename KernelInterface< params >::BoundaryInput& in) const
    {
        typename KernelInterface< params >::Rhs out{};
        kernel(in, out);
        return out;
    }
};
template < KernelParams params, typename Kernel >
constexpr auto wrapDomainResidualKernel(Kernel kernel)
{
    return ResidualDomainKernel< Kernel, params >{kernel};
}
template < KernelParams params, typename Kernel >
constexpr auto wrapBoundaryResidualKernel(Kernel kernel)
{
    return ResidualBoundaryKernel< Kernel, params >{kernel};
}
} // namespace l3k

// tuple-like protocol of FieldArray (structured bindings)
template < int F >
struct std::tuple_size< l3k::FieldArray< F > > : std::integral_constant< std::size_t, std::size_t(F) >
{};
template < std::size_t I, int F >
struct std::tuple_element< I, l3k::FieldArray< F > >
{
    using type = double;
};
#endif
