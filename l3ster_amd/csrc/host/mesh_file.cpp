// mesh_file.cpp -- the reference's native mesh file (post/NativeIO.hpp:75-108 writer, :161-232 readers;
// mesh/MeshUtils.hpp:318-360 serializeMesh / deserializeMesh; util/Serialization.hpp:20-66), host only.
//
// Layout: three text lines "L3STER mesh file\n" "v1.0\n" "// <comment, newlines replaced by spaces>\n", a raw size_t
// n_parts, n_parts raw size_t part sizes, then the parts back to back.  One part is the serialisation of
//   tuple(range of pair(d_id_t id, tuple(span<Element<Hex,p>>, span<Element<Quad,p>>, span<Element<Line,p>>)),
//         n_id_t nodes_begin, size_t num_owned_nodes, span<d_id_t> boundary_ids)
// with the rules of util::Serializer: a trivially copyable non-range non-tuple object is its object representation, a
// range is a size_t count followed by its values, a tuple-like is its members in order (so the 2-byte domain id of the
// pair is NOT padded).  The element types of one order come in the order of mesh/ElementType.hpp:11-16 (Hex, Quad,
// Line); domains in ascending id (MeshPartition::domain_map_t is a std::map, MeshPartition.hpp:49).  An element
// (mesh/Element.hpp:29-31, ElementData.hpp:29) is { uint64 nodes[(p+1)^d]; double vertices[2^d][3]; uint64 id } -- all
// members 8-byte aligned, no padding.  The boundary here takes and returns structure-of-arrays; the interleaving happens
// in this file.
#include "l3k.h"

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace l3k::dev
{
void setError(const char* fmt, ...);
}

struct l3k_meshfile_part
{
    struct Dom
    {
        uint16_t                id;
        size_t                  n[3];
        std::vector< uint64_t > nodes[3], ids[3];
        std::vector< double >   verts[3];
    };
    int                             order = 0;
    std::vector< Dom >              doms;
    std::vector< l3k_meshfile_domain > view;
    uint64_t                        nodes_begin = 0;
    size_t                          n_owned     = 0;
    std::vector< uint16_t >         bnd;
};

namespace
{
using l3k::dev::setError;
constexpr char magic[] = "L3STER mesh file\nv1.0\n// ";

size_t nodesPer(int order, int type) // type 0 hex, 1 quad, 2 line (mesh/ElementTraits.hpp: (p+1)^dim)
{
    const size_t n = size_t(order) + 1;
    return type == 0 ? n * n * n : type == 1 ? n * n : n;
}
size_t vertsPer(int type)
{
    return type == 0 ? 8 : type == 1 ? 4 : 2;
}
size_t elemBytes(int order, int type)
{
    return 8 * nodesPer(order, type) + 24 * vertsPer(type) + 8;
}
const l3k_meshfile_elems& elemsOf(const l3k_meshfile_domain& d, int type)
{
    return type == 0 ? d.hex : type == 1 ? d.quad : d.line;
}
l3k_meshfile_elems& elemsOf(l3k_meshfile_domain& d, int type)
{
    return type == 0 ? d.hex : type == 1 ? d.quad : d.line;
}

bool checkDesc(const l3k_meshfile_part_desc* d, const char* who)
{
    if (!d || d->order < 1 || (d->n_domains && !d->domains) || (d->n_boundary_ids && !d->boundary_ids))
    {
        setError("%s: bad part description", who);
        return false;
    }
    for (size_t i = 0; i < d->n_domains; ++i)
        for (int t = 0; t < 3; ++t)
        {
            const auto& e = elemsOf(d->domains[i], t);
            if (e.n && (!e.nodes || !e.verts || !e.ids))
            {
                setError("%s: domain %u has a null element array", who, unsigned(d->domains[i].id));
                return false;
            }
        }
    for (size_t i = 0; i < d->n_domains; ++i)
        for (size_t j = i + 1; j < d->n_domains; ++j)
            if (d->domains[i].id == d->domains[j].id)
            {
                setError("%s: domain id %u given twice", who, unsigned(d->domains[i].id));
                return false;
            }
    return true;
}

size_t partBytes(const l3k_meshfile_part_desc& d)
{
    size_t b = 8; // number of domains
    for (size_t i = 0; i < d.n_domains; ++i)
    {
        b += 2; // d_id_t
        for (int t = 0; t < 3; ++t)
            b += 8 + elemsOf(d.domains[i], t).n * elemBytes(d.order, t);
    }
    return b + 8 + 8 + 8 + 2 * d.n_boundary_ids;
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

template < typename T >
void put(std::string& s, const T& v)
{
    s.append(reinterpret_cast< const char* >(&v), sizeof v);
}

std::string makeHeader(const char* comment, size_t n_parts, const size_t* part_bytes)
{
    std::string c = comment ? comment : "";
    std::replace(c.begin(), c.end(), '\n', ' '); // (NativeIO.hpp:91)
    std::string h = std::string(magic) + c + "\n";
    put(h, n_parts);
    for (size_t i = 0; i < n_parts; ++i)
        put(h, part_bytes[i]);
    return h;
}

// a bounds-checked cursor over the mapped file (util::Deserializer: throwingAssert(serial_data.size() >= sizeof(T)))
struct Cursor
{
    const char* p;
    size_t      left;
    bool        ok = true;
    bool take(void* dst, size_t n)
    {
        if (!ok || n > left)
            return ok = false;
        if (n)
            memcpy(dst, p, n);
        p += n;
        left -= n;
        return true;
    }
    template < typename T >
    T get()
    {
        T v{};
        take(&v, sizeof v);
        return v;
    }
};

struct Mapped
{
    int         fd   = -1;
    const char* data = nullptr;
    size_t      size = 0;
    ~Mapped()
    {
        if (data && size)
            munmap(const_cast< char* >(data), size);
        if (fd >= 0)
            close(fd);
    }
    bool open(const char* path)
    {
        fd = ::open(path, O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0)
        {
            setError("Error parsing results file: %s (%s)", path, strerror(errno)); // (the reference's asserter text, :113)
            return false;
        }
        size = size_t(st.st_size);
        if (size == 0)
        {
            setError("Error parsing results file: %s (empty)", path);
            return false;
        }
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED)
        {
            setError("Error parsing results file: %s (mmap: %s)", path, strerror(errno));
            size = 0;
            return false;
        }
        data = static_cast< const char* >(m);
        return true;
    }
};

// extractSavedPartitionInfo (:161-182): skips 3 lines, reads n_parts and the sizes; `c` is left at the first part
bool parseHeader(Cursor& c, std::vector< size_t >& sizes, const char* path)
{
    for (int line = 0; line < 3; ++line)
    {
        const void* nl = c.left ? memchr(c.p, '\n', c.left) : nullptr;
        if (!nl)
        {
            setError("Error parsing results file: %s (header lines)", path);
            return false;
        }
        const size_t adv = size_t(static_cast< const char* >(nl) - c.p) + 1;
        c.p += adv;
        c.left -= adv;
    }
    const size_t n_parts = c.get< size_t >();
    if (!c.ok || n_parts > c.left / sizeof(size_t))
    {
        setError("Error parsing results file: %s (partition table)", path);
        return false;
    }
    sizes.resize(n_parts);
    c.take(sizes.data(), n_parts * sizeof(size_t));
    size_t total = 0;
    for (size_t s : sizes)
    {
        if (s > c.left - total)
        {
            setError("Error parsing results file: %s (truncated: the parts need more bytes than the file holds)", path);
            return false;
        }
        total += s;
    }
    return true;
}
} // namespace

extern "C" {

int l3k_meshfile_part_bytes(const l3k_meshfile_part_desc* desc, size_t* bytes)
{
    if (!bytes || !checkDesc(desc, "l3k_meshfile_part_bytes"))
    {
        if (!bytes)
            setError("l3k_meshfile_part_bytes: null argument");
        return -1;
    }
    *bytes = partBytes(*desc);
    return 0;
}

int l3k_meshfile_save(const char* path, const char* comment, size_t n_parts, const size_t* part_bytes, size_t part,
                      const l3k_meshfile_part_desc* desc, int write_header)
{
    if (!path || !part_bytes || part >= n_parts || !checkDesc(desc, "l3k_meshfile_save"))
    {
        if (!path || !part_bytes || part >= n_parts)
            setError("l3k_meshfile_save: bad arguments");
        return -1;
    }
    const size_t mine = partBytes(*desc);
    if (mine != part_bytes[part])
    {
        setError("l3k_meshfile_save: part %zu serialises to %zu bytes, the size table says %zu", part, mine, part_bytes[part]);
        return -1;
    }
    // serializeMesh (MeshUtils.hpp:318-332)
    std::string blob;
    blob.reserve(mine);
    std::vector< size_t > dom_order(desc->n_domains);
    std::iota(dom_order.begin(), dom_order.end(), size_t{0});
    std::sort(dom_order.begin(), dom_order.end(),
              [&](size_t a, size_t b) { return desc->domains[a].id < desc->domains[b].id; }); // std::map order
    put(blob, size_t(desc->n_domains));
    for (size_t di : dom_order)
    {
        const auto& d = desc->domains[di];
        put(blob, d.id);
        for (int t = 0; t < 3; ++t)
        {
            const auto&  e  = elemsOf(d, t);
            const size_t nn = nodesPer(desc->order, t), nv = vertsPer(t);
            put(blob, size_t(e.n));
            for (size_t i = 0; i < e.n; ++i)
            {
                blob.append(reinterpret_cast< const char* >(e.nodes + i * nn), 8 * nn);
                blob.append(reinterpret_cast< const char* >(e.verts + i * nv * 3), 24 * nv);
                put(blob, e.ids[i]);
            }
        }
    }
    put(blob, uint64_t(desc->nodes_begin));
    put(blob, size_t(desc->n_owned_nodes));
    put(blob, size_t(desc->n_boundary_ids));
    for (size_t i = 0; i < desc->n_boundary_ids; ++i)
        put(blob, desc->boundary_ids[i]);
    if (blob.size() != mine)
    {
        setError("l3k_meshfile_save: internal size mismatch");
        return -1;
    }
    const std::string header = makeHeader(comment, n_parts, part_bytes);
    off_t             offset = off_t(header.size()), total = off_t(header.size()); // exclusive_scan(sizes, header.size()) :100-101
    for (size_t i = 0; i < n_parts; ++i)
    {
        if (i < part)
            offset += off_t(part_bytes[i]);
        total += off_t(part_bytes[i]);
    }
    const int fd = open(path, O_CREAT | O_RDWR, 0644);
    if (fd < 0)
    {
        setError("l3k_meshfile_save: cannot open %s: %s", path, strerror(errno));
        return -4;
    }
    bool ok = ftruncate(fd, total) == 0;
    if (ok && write_header)
        ok = writeAll(fd, header.data(), header.size(), 0);
    ok = ok && writeAll(fd, blob.data(), blob.size(), offset);
    if (!ok)
        setError("l3k_meshfile_save: write to %s failed: %s", path, strerror(errno));
    close(fd);
    return ok ? 0 : -4;
}

int l3k_meshfile_info(const char* path, size_t* n_parts, size_t* part_bytes, size_t capacity)
{
    if (!path || !n_parts)
    {
        setError("l3k_meshfile_info: null argument");
        return -1;
    }
    Mapped f;
    if (!f.open(path))
        return -4;
    Cursor                c{f.data, f.size};
    std::vector< size_t > sizes;
    if (!parseHeader(c, sizes, path))
        return -4;
    *n_parts = sizes.size();
    if (part_bytes)
        for (size_t i = 0; i < sizes.size() && i < capacity; ++i)
            part_bytes[i] = sizes[i];
    return 0;
}

int l3k_meshfile_load(const char* path, size_t part, int order, l3k_meshfile_part** out)
{
    if (!path || !out || order < 1)
    {
        setError("l3k_meshfile_load: bad arguments");
        return -1;
    }
    *out = nullptr;
    Mapped f;
    if (!f.open(path))
        return -4;
    Cursor                c{f.data, f.size};
    std::vector< size_t > sizes;
    if (!parseHeader(c, sizes, path))
        return -4;
    if (part >= sizes.size()) // (part_offsets.at(comm.getRank()) :229)
    {
        setError("l3k_meshfile_load: part %zu of %zu", part, sizes.size());
        return -1;
    }
    size_t ofs = 0;
    for (size_t i = 0; i < part; ++i)
        ofs += sizes[i];
    Cursor s{c.p + ofs, sizes[part]};
    auto   p = new l3k_meshfile_part;
    p->order = order;
    // deserializeMesh (MeshUtils.hpp:334-360); an element order other than the one the file was written with shows up as a
    // part that does not end where its size says (the reference has the same blind spot: the order is a template argument)
    const size_t n_dom = s.get< size_t >();
    bool         ok    = s.ok && n_dom <= s.left / 2;
    if (ok)
        p->doms.resize(n_dom);
    for (size_t d = 0; ok && d < n_dom; ++d)
    {
        auto& dom = p->doms[d];
        dom.id    = s.get< uint16_t >();
        for (int t = 0; ok && t < 3; ++t)
        {
            const size_t n  = s.get< size_t >();
            const size_t nn = nodesPer(order, t), nv = vertsPer(t);
            ok = s.ok && n <= s.left / elemBytes(order, t);
            if (!ok)
                break;
            dom.n[t] = n;
            dom.nodes[t].resize(n * nn);
            dom.verts[t].resize(n * nv * 3);
            dom.ids[t].resize(n);
            for (size_t i = 0; i < n; ++i)
            {
                s.take(dom.nodes[t].data() + i * nn, 8 * nn);
                s.take(dom.verts[t].data() + i * nv * 3, 24 * nv);
                s.take(dom.ids[t].data() + i, 8);
            }
            ok = s.ok;
        }
    }
    if (ok)
    {
        p->nodes_begin   = s.get< uint64_t >();
        p->n_owned       = s.get< size_t >();
        const size_t nb  = s.get< size_t >();
        ok               = s.ok && nb <= s.left / 2;
        if (ok)
        {
            p->bnd.resize(nb);
            s.take(p->bnd.data(), 2 * nb);
            ok = s.ok && s.left == 0;
        }
    }
    if (!ok)
    {
        setError("Error parsing results file: %s (part %zu does not parse as a mesh of order %d)", path, part, order);
        delete p;
        return -4;
    }
    p->view.resize(p->doms.size());
    for (size_t d = 0; d < p->doms.size(); ++d)
    {
        p->view[d].id = p->doms[d].id;
        for (int t = 0; t < 3; ++t)
        {
            auto& e = elemsOf(p->view[d], t);
            e.n     = p->doms[d].n[t];
            e.nodes = p->doms[d].nodes[t].data();
            e.verts = p->doms[d].verts[t].data();
            e.ids   = p->doms[d].ids[t].data();
        }
    }
    *out = p;
    return 0;
}

int l3k_meshfile_part_get(const l3k_meshfile_part* part, l3k_meshfile_part_desc* out)
{
    if (!part || !out)
    {
        setError("l3k_meshfile_part_get: null argument");
        return -1;
    }
    out->order          = part->order;
    out->n_domains      = part->view.size();
    out->domains        = part->view.data();
    out->nodes_begin    = part->nodes_begin;
    out->n_owned_nodes  = part->n_owned;
    out->n_boundary_ids = part->bnd.size();
    out->boundary_ids   = part->bnd.data();
    return 0;
}

int l3k_meshfile_part_destroy(l3k_meshfile_part* part)
{
    delete part;
    return 0;
}
} // extern "C"
