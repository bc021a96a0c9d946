// registry.cpp -- instance registry and thread-local error text of libl3k.
#include "../device/common.hpp"

#include <cstdarg>
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
} // namespace

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
