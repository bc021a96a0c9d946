// instantiate.hpp -- included by the generated per-TU instance files (l3ster_amd/build.py writes them from
// L3K_FOR_EACH_INSTANCE in user_kernels.hpp so that the heavy templates compile in parallel).
#ifndef L3K_DEVICE_INSTANTIATE_HPP
#define L3K_DEVICE_INSTANTIATE_HPP

#include <cstring>
#include <type_traits>

#include "../user_kernels.hpp"
#include "sumfact_apply.hpp"
#include "sumfact_fast.hpp"
#include "diag.hpp"
#include "assemble.hpp"
#include "boundary.hpp"
#include "integral.hpp"

namespace l3k::dev
{
#define L3K_X(id, T, name)                                                                                             \
    template <>                                                                                                        \
    struct KernelId< T >                                                                                               \
    {                                                                                                                  \
        static constexpr int value = id;                                                                               \
    };
L3K_FOR_EACH_KERNEL(L3K_X)
L3K_FOR_EACH_BOUNDARY_KERNEL(L3K_X)
#undef L3K_X
#define L3K_X(id, T, name)                                                                                             \
    template <>                                                                                                        \
    struct ResidualId< T >                                                                                             \
    {                                                                                                                  \
        static constexpr int value = id;                                                                               \
    };
L3K_FOR_EACH_RESIDUAL_KERNEL(L3K_X)
#undef L3K_X
} // namespace l3k::dev

namespace l3k::dev
{
// R columns through the one-wave-per-element kernel, one column per launch: at order 6 it is ~3x faster per column than
// the generic LDS kernel, so R launches beat one R-column launch (the apply never reads the kernel's rhs, so the
// single-column instantiation of the functor gives the same operator); unknowns on a subset of the node's dofs: column by
// column through the strided-dof variant
template < typename T, int P, int NQ, int R >
int launchColumnsFast(const ElemArgs& a, const void* kparam_blob, hipStream_t stream)
{
    if constexpr (FastCfg< T, P, NQ >::multi_column)
        if (a.dense)
        {
            ElemArgs ac = a; // all R columns in one pass over the elements (the multi-column variant of the single-wave kernel)
            ac.n_cols   = R;
            return launchSumfactFastCols< T, P, NQ >(ac, kparam_blob, stream);
        }
    {
        for (int c = 0; c < R; ++c)
        {
            ElemArgs ac = a;
            ac.x        = a.x + a.ldx * c;
            ac.xg       = a.xg ? a.xg + a.ldxg * c : nullptr;
            ac.y        = a.y + a.ldy * c;
            ac.yg       = a.yg ? a.yg + a.ldyg * c : nullptr;
            if (int rc = launchSumfactFast< T, P, NQ >(ac, kparam_blob, stream))
                return rc;
        }
        return 0;
    }
}
// the column-loop entry of a single-column instance (Instance::apply_cols), or nullptr where the single-wave kernel does not fit
template < typename T, int P, int NQ, int R >
constexpr LaunchFn selectApplyCols()
{
    if constexpr (R == 1 && FastCfg< T, P, NQ >::feasible && FastCfg< T, P, NQ >::multi_column)
        return &launchSumfactFastCols< T, P, NQ >;
    else
        return nullptr;
}
// applies use the register-resident pipelined kernel when its working set fits
template < typename T, int P, int NQ, int R >
constexpr LaunchFn selectApply()
{
    if constexpr (FastCfg< T, P, NQ >::feasible)
        return R == 1 ? &launchSumfactFast< T, P, NQ > : &launchColumnsFast< T, P, NQ, R >;
    else
        return &launchSumfactApply< T, P, NQ, R, false >;
}
// the route text of an instance's apply (l3k_mf_route)
template < typename T, int P, int NQ, int R >
constexpr RouteFn selectRoute()
{
    if constexpr (FastCfg< T, P, NQ >::feasible && R == 1)
        return &describeSumfactFast< T, P, NQ >;
    else if constexpr (FastCfg< T, P, NQ >::feasible)
        return +[](const ElemArgs& a, char* buf, size_t n) {
            ElemArgs ac = a; // launchColumnsFast: one multi-column pass, or R launches of the single-column kernel
            ac.n_cols   = a.dense ? R : 1;
            const int rc = describeSumfactFast< T, P, NQ >(ac, buf, n);
            if (rc == 0 && (!FastCfg< T, P, NQ >::multi_column || !a.dense))
                std::snprintf(buf + std::strlen(buf), n - std::strlen(buf), "; %d launches, one per column", R);
            return rc;
        };
    else
        return &describeSumfactApply< T, P, NQ, R >;
}
} // namespace l3k::dev

#define L3K_CAT2(a, b) a##b
#define L3K_CAT(a, b) L3K_CAT2(a, b)
#define L3K_INSTANTIATE(T, P, NQ, R)                                                                                   \
    namespace                                                                                                          \
    {                                                                                                                  \
    const struct L3K_CAT(Registrar_, __LINE__)                                                                         \
    {                                                                                                                  \
        L3K_CAT(Registrar_, __LINE__)()                                                                                \
        {                                                                                                              \
            ::l3k::dev::registerInstance({::l3k::dev::KernelId< T >::value, P, NQ, R,                                  \
                                          ::l3k::dev::selectApply< T, P, NQ, R >(),                                  \
                                          &::l3k::dev::launchDiagRhs< T, P, NQ, R >,                               \
                                          &::l3k::dev::launchAssemble< T, P, NQ >,                                 \
                                          ::l3k::dev::assembleWorkspaceDoublesPerElem< T, P, NQ >(),               \
                                          ::l3k::dev::selectApplyCols< T, P, NQ, R >(),                              \
                                          ::l3k::dev::SfAsmCfg< P, NQ >::feasible,                                   \
                                          ::l3k::dev::selectRoute< T, P, NQ, R >()});                                \
        }                                                                                                              \
    } L3K_CAT(registrar_, __LINE__);                                                                                   \
    }
// A kernel plugin announces its functor: id, kind (0 domain, 1 boundary, 2 residual), name
#define L3K_PLUGIN_KERNEL(ID, KIND, T, NAME)                                                                           \
    namespace                                                                                                          \
    {                                                                                                                  \
    const struct L3K_CAT(PRegistrar_, __LINE__)                                                                        \
    {                                                                                                                  \
        L3K_CAT(PRegistrar_, __LINE__)()                                                                               \
        {                                                                                                              \
            ::l3k::dev::registerPluginKernel({ID, KIND, T::params.dimension, T::params.n_equations, T::params.n_unknowns, \
                                              T::params.n_fields, T::params.n_rhs, NAME,                              \
                                              std::is_empty_v< T > ? size_t{0} : sizeof(T)});                         \
        }                                                                                                              \
    } L3K_CAT(pregistrar_, __LINE__);                                                                                  \
    }
#define L3K_INSTANTIATE_BOUNDARY(T, P, NQ, R)                                                                          \
    namespace                                                                                                          \
    {                                                                                                                  \
    const struct L3K_CAT(BRegistrar_, __LINE__)                                                                        \
    {                                                                                                                  \
        L3K_CAT(BRegistrar_, __LINE__)()                                                                               \
        {                                                                                                              \
            ::l3k::dev::registerBoundaryInstance({::l3k::dev::KernelId< T >::value, P, NQ, R,                          \
                                                  &::l3k::dev::launchFace< T, P, NQ, R, false >,                       \
                                                  &::l3k::dev::launchFace< T, P, NQ, R, true >});                      \
        }                                                                                                              \
    } L3K_CAT(bregistrar_, __LINE__);                                                                                  \
    }
#define L3K_INSTANTIATE_RESIDUAL(T, P, NQ)                                                                             \
    namespace                                                                                                          \
    {                                                                                                                  \
    const struct L3K_CAT(RRegistrar_, __LINE__)                                                                        \
    {                                                                                                                  \
        L3K_CAT(RRegistrar_, __LINE__)()                                                                               \
        {                                                                                                              \
            ::l3k::dev::registerIntegralInstance({::l3k::dev::ResidualId< T >::value, P, NQ,                           \
                                                  &::l3k::dev::launchIntegral< T, P, NQ, false >,                      \
                                                  &::l3k::dev::launchIntegral< T, P, NQ, true >,                       \
                                                  &::l3k::dev::launchValuesAtNodesAny< T, P, NQ >});                   \
        }                                                                                                              \
    } L3K_CAT(rregistrar_, __LINE__);                                                                                  \
    }
#endif
