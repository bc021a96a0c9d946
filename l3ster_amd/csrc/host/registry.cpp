// registry.cpp -- instance registry and thread-local error text of libl3k.
#include "../device/common.hpp"

#include <cstdarg>
#include <mutex>
#include <cstdio>
#include <string>
#include <vector>

namespace l3k::dev
{
namespace
{
std::vector< Instance >& table()
{
    static std::vector< Instance > t;
    return t;
}
thread_local std::string g_error;
std::vector< BoundaryInstance >& boundaryTable()
{
    static std::vector< BoundaryInstance > t;
    return t;
}
std::vector< IntegralInstance >& integralTable()
{
    static std::vector< IntegralInstance > t;
    return t;
}
} // namespace

namespace
{
std::vector< PluginKernel >& pluginKernels()
{
    static std::vector< PluginKernel > t;
    return t;
}
} // namespace
int deviceComputeUnits()
{
    static int        cus[64] = {};
    static std::mutex m;
    int               dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 256;
    std::lock_guard< std::mutex > lock{m};
    if (cus[dev] == 0)
    {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}
const l3k_tuning& defaultTuning()
{
    static const l3k_tuning t{.generic_below = 1500, .static_deal = 0, .waves_per_cu = 0, .no_affine = 0, .column_by_column = 0,
                              .assemble_dense = 0, .assemble_two_launches = 0, .scatter_per_entry = 0, .assemble_direct_store = 0, .assemble_sub_batch = 0, .assemble_no_symmetrise = 0};
    return t;
}
void registerPluginKernel(const PluginKernel& k)
{
    pluginKernels().push_back(k);
}
const PluginKernel* findPluginKernel(int id, bool residual)
{
    for (const auto& k : pluginKernels())
        if (k.id == id && (k.kind == 2) == residual)
            return &k;
    return nullptr;
}

void registerBoundaryInstance(const BoundaryInstance& inst)
{
    boundaryTable().push_back(inst);
}
const BoundaryInstance* findBoundaryInstance(int kernel_id, int order, int nq, int ncols)
{
    for (const auto& i : boundaryTable())
        if (i.kernel_id == kernel_id && i.order == order && i.nq == nq && i.ncols == ncols)
            return &i;
    return nullptr;
}
void registerIntegralInstance(const IntegralInstance& inst)
{
    integralTable().push_back(inst);
}
const IntegralInstance* findIntegralInstance(int residual_id, int order, int nq)
{
    for (const auto& i : integralTable())
        if (i.residual_id == residual_id && i.order == order && (nq < 0 || i.nq == nq)) // nq < 0: any (values at nodes)
            return &i;
    return nullptr;
}

void registerInstance(const Instance& inst)
{
    table().push_back(inst);
}
const Instance* findInstance(int kernel_id, int order, int nq, int ncols)
{
    for (const auto& i : table())
        if (i.kernel_id == kernel_id && i.order == order && i.nq == nq && i.ncols == ncols)
            return &i;
    return nullptr;
}
int instanceCount()
{
    return static_cast< int >(table().size());
}
const Instance* instanceAt(int i)
{
    return i >= 0 && i < instanceCount() ? &table()[i] : nullptr;
}
void setError(const char* fmt, ...)
{
    char    buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}
const char* lastError()
{
    return g_error.c_str();
}
} // namespace l3k::dev
