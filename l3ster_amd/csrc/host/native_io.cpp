// native_io.cpp -- the reference's native results file (post/NativeIO.hpp:15-60 writer, :115-146 reader), host only.
//
// Layout: three text lines "L3STER results file\n" "v1.0\n" "// <comment, newlines replaced by spaces>\n", then two raw
// size_t (n_fields, n_nodes_global; util/Serialization.hpp:20-30: the object representation), then n_fields contiguous
// arrays of n_nodes_global doubles indexed by GLOBAL node id.  Every rank writes its owned slice [node_begin,
// node_begin + n_local) of every field at its offset (the reference does this with MPI-IO writeAtAsync, :45-57); rank 0
// also writes the header.  Here: POSIX pwrite into a file created / sized by whichever rank comes first -- no ordering
// between ranks is needed.
#include "l3k.h"

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace l3k::dev
{
void setError(const char* fmt, ...);
}

namespace
{
using l3k::dev::setError;
constexpr char magic[] = "L3STER results file\nv1.0\n// ";

std::string makeHeader(const char* comment, size_t n_fields, size_t n_nodes)
{
    std::string c = comment ? comment : "";
    for (auto& ch : c)
        if (ch == '\n')
            ch = ' '; // (:38 std::ranges::replace(comment, '\n', ' '))
    std::string h = std::string(magic) + c + "\n";
    h.append(reinterpret_cast< const char* >(&n_fields), sizeof n_fields);
    h.append(reinterpret_cast< const char* >(&n_nodes), sizeof n_nodes);
    return h;
}

bool writeAll(int fd, const void* buf, size_t n, off_t off)
{
    const char* p = static_cast< const char* >(buf);
    while (n > 0)
    {
        const ssize_t w = pwrite(fd, p, n, off);
        if (w < 0)
        {
            if (errno == EINTR)
                continue;
            return false;
        }
        p += w;
        off += w;
        n -= size_t(w);
    }
    return true;
}
bool readAll(int fd, void* buf, size_t n, off_t off)
{
    char* p = static_cast< char* >(buf);
    while (n > 0)
    {
        const ssize_t r = pread(fd, p, n, off);
        if (r < 0 && errno == EINTR)
            continue;
        if (r <= 0)
            return false;
        p += r;
        off += r;
        n -= size_t(r);
    }
    return true;
}

// parses the three text lines and the two sizes; returns the offset of the first double or -1
off_t parseHeader(int fd, size_t& n_fields, size_t& n_nodes, const char* path)
{
    char          buf[4096];
    const ssize_t got = pread(fd, buf, sizeof buf, 0);
    if (got <= 0)
    {
        setError("Error parsing results file: %s (empty)", path);
        return -1;
    }
    off_t pos = 0;
    for (int line = 0; line < 3; ++line) // LoadedResults skips the first 3 lines (:120-121)
    {
        const void* nl = memchr(buf + pos, '\n', size_t(got - pos));
        if (!nl)
        {
            setError("Error parsing results file: %s (header lines)", path);
            return -1;
        }
        pos = static_cast< const char* >(nl) - buf + 1;
    }
    if (pos + off_t(2 * sizeof(size_t)) > got)
    {
        setError("Error parsing results file: %s (sizes)", path);
        return -1;
    }
    memcpy(&n_fields, buf + pos, sizeof(size_t));
    memcpy(&n_nodes, buf + pos + sizeof(size_t), sizeof(size_t));
    return pos + off_t(2 * sizeof(size_t));
}
} // namespace

extern "C" {

int l3k_results_save(const char* path, const char* comment, size_t n_fields, int64_t n_global_nodes, int64_t node_begin,
                     int64_t n_local_nodes, const double* fields, size_t ld, int write_header)
{
    if (!path || n_fields == 0 || n_global_nodes < 0 || node_begin < 0 || n_local_nodes < 0 ||
        node_begin + n_local_nodes > n_global_nodes || (n_local_nodes > 0 && (!fields || ld < size_t(n_local_nodes))))
    {
        setError("l3k_results_save: bad arguments"); // (util::throwingAssert(not inds.empty()) :22)
        return -1;
    }
    const std::string header = makeHeader(comment, n_fields, size_t(n_global_nodes));
    const off_t       total  = off_t(header.size()) + off_t(n_fields) * off_t(sizeof(double)) * off_t(n_global_nodes);
    const int         fd     = open(path, O_CREAT | O_RDWR, 0644);
    if (fd < 0)
    {
        setError("l3k_results_save: cannot open %s: %s", path, strerror(errno));
        return -4;
    }
    bool ok = ftruncate(fd, total) == 0; // (file.preallocate :44; same size from every rank)
    if (ok && write_header)
        ok = writeAll(fd, header.data(), header.size(), 0);
    for (size_t f = 0; ok && f < n_fields; ++f) // dest_offset = header + 8 * (n_global * f + node_begin)  (:50-52)
        ok = writeAll(fd, fields + f * ld, sizeof(double) * size_t(n_local_nodes),
                      off_t(header.size()) + off_t(sizeof(double)) * (off_t(n_global_nodes) * off_t(f) + off_t(node_begin)));
    if (!ok)
        setError("l3k_results_save: write to %s failed: %s", path, strerror(errno));
    close(fd);
    return ok ? 0 : -4;
}

int l3k_results_info(const char* path, size_t* n_fields, size_t* n_nodes)
{
    if (!path || !n_fields || !n_nodes)
    {
        setError("l3k_results_info: null argument");
        return -1;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
    {
        setError("Error parsing results file: %s (%s)", path, strerror(errno));
        return -4;
    }
    const off_t data = parseHeader(fd, *n_fields, *n_nodes, path);
    struct stat st;
    const bool  sized = data >= 0 && fstat(fd, &st) == 0 &&
                       st.st_size >= data + off_t(*n_fields) * off_t(*n_nodes) * off_t(sizeof(double));
    close(fd);
    if (data >= 0 && !sized)
        setError("Error parsing results file: %s (truncated)", path);
    return sized ? 0 : -4;
}

int l3k_results_load(const char* path, size_t field, int64_t n, const int64_t* node_ids, int64_t node_begin, double* out)
{
    if (!path || n < 0 || (n > 0 && !out))
    {
        setError("l3k_results_load: bad arguments");
        return -1;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0)
    {
        setError("Error parsing results file: %s (%s)", path, strerror(errno));
        return -4;
    }
    size_t      n_fields = 0, n_nodes = 0;
    const off_t data = parseHeader(fd, n_fields, n_nodes, path);
    int         rc   = data < 0 ? -4 : 0;
    if (rc == 0 && field >= n_fields) // (throwingAssert(max(src_inds) < results.fields()) :287)
    {
        setError("l3k_results_load: field %zu of %zu", field, n_fields);
        rc = -1;
    }
    const off_t base = data + off_t(sizeof(double)) * off_t(n_nodes) * off_t(field); // operator()(node, field) :133-139
    if (rc == 0 && !node_ids)
    {
        if (node_begin < 0 || size_t(node_begin + n) > n_nodes)
        {
            setError("l3k_results_load: nodes [%lld, %lld) of %zu", (long long)node_begin, (long long)(node_begin + n), n_nodes);
            rc = -1;
        }
        else if (!readAll(fd, out, sizeof(double) * size_t(n), base + off_t(sizeof(double)) * off_t(node_begin)))
        {
            setError("Error parsing results file: %s (truncated)", path);
            rc = -4;
        }
    }
    else if (rc == 0)
    {
        // dest(i) = results(old_node[i], field)  (Loader::loadResultsImpl :291-294); runs of consecutive ids in one read
        for (int64_t i = 0; rc == 0 && i < n;)
        {
            if (node_ids[i] < 0 || size_t(node_ids[i]) >= n_nodes) // (throwingAssert(m_max_old_node < results.nodes()) :289)
            {
                setError("l3k_results_load: node id %lld of %zu", (long long)node_ids[i], n_nodes);
                rc = -1;
                break;
            }
            int64_t j = i + 1;
            while (j < n && node_ids[j] == node_ids[j - 1] + 1 && size_t(node_ids[j]) < n_nodes)
                ++j;
            if (!readAll(fd, out + i, sizeof(double) * size_t(j - i), base + off_t(sizeof(double)) * off_t(node_ids[i])))
            {
                setError("Error parsing results file: %s (truncated)", path);
                rc = -4;
            }
            i = j;
        }
    }
    close(fd);
    return rc;
}
} // extern "C"
