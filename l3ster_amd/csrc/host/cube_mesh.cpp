// cube_mesh.cpp -- synthetic structured hex mesh of order p on [0,1]^3 with a block partition, in the numbering
// conventions of the reference.  Host only.
//
// Stands in (for structured cubes) for: mesh::makeCubeMesh (mesh/primitives/CubeMesh.hpp:16-138, vertex order
// v = i + 2j + 4k :46-61), convertMeshToOrder (mesh/ConvertMeshToOrder.hpp:51-104), the METIS partition
// (mesh/PartitionMesh.hpp:142-183; here px*py*pz blocks), the ownership rule "lowest part touching a node owns it" with
// a contiguous global range per rank (util/SegmentedOwnership.hpp:11-45, dofs/NodeToDofMap.hpp:242-247), the local
// numbering [owned non-internal | element-internal, contiguous per element | ghosts sorted by global id]
// (mesh/LocalMeshView.hpp:425-458, util/SegmentedOwnership.hpp:21-41), the interior/border element split
// (algsys/MatrixFreeSystem.hpp:969-981) and the neighbour lists of comm::ImportExportContext
// (comm/ImportExport.hpp:29-72).
//
// Locality choices (free in the reference, which takes whatever the mesh file / METIS gives): elements are traversed in
// bricks (edge chosen by brickEdge below); every owned non-internal node is numbered with its "home" element (the element that has it on a
// high face), faces first so that each face's (p-1)^2 nodes are one contiguous run, then edges, then vertices.
#include "l3k.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <unordered_map>
#include <vector>

namespace l3k::dev
{
void setError(const char* fmt, ...);
}

namespace
{
using i64 = int64_t;
// Elements are traversed in brick x brick x brick blocks.  The brick edge is a free choice of this implementation (the
// reference takes whatever order the mesh generator / METIS gives); measured on MI355X (profiles/r01_kbench_brick_sweep.log,
// 64^3 elements): power-of-two edges alias -- the node rows that the ~1800 resident waves touch at the same time are then
// a power-of-two apart in HBM -- and cost 5 % at order 6 (edge 4: 15.9 ns per element, 6: 15.1) and 27 % at order 4
// (edge 4: 6.9, 12: 5.0).  Hence 6 for orders >= 5, 12 below (more elements per wave there).  (The sweep was run with builds
// that define L3K_MESH_BRICK: tools/brick_sweep.sh.)
int brickEdge(int order)
{
#ifdef L3K_MESH_BRICK
    return L3K_MESH_BRICK;
#else
    return order >= 5 ? 6 : 12;
#endif
}

struct TypeTable // homed non-internal nodes of an element whose low-boundary flags are (fx, fy, fz)
{
    std::vector< int32_t > index; // [N] -> position among the homed nodes or -1
    int                    count = 0;
};

struct Layout
{
    int                        p, n, N, n_internal;
    std::array< int, 3 >       ne, parts, G; // elements, parts, grid nodes per direction
    std::array< TypeTable, 8 > types;
    std::vector< int32_t >     internal_index; // [N] -> lexicographic index among internal nodes or -1

    Layout(const int ne_[3], int p_, const int parts_[3]) : p{p_}, n{p_ + 1}
    {
        N          = n * n * n;
        n_internal = (p - 1) * (p - 1) * (p - 1);
        for (int d = 0; d < 3; ++d)
        {
            ne[d]    = ne_[d];
            parts[d] = parts_[d];
            G[d]     = p * ne[d] + 1;
        }
        internal_index.assign(N, -1);
        int c = 0;
        for (int i = 0; i < N; ++i)
        {
            const int ix = i % n, iy = (i / n) % n, iz = i / (n * n);
            if (ix > 0 && ix < p && iy > 0 && iy < p && iz > 0 && iz < p)
                internal_index[i] = c++;
        }
        for (int t = 0; t < 8; ++t)
        {
            const bool f[3] = {(t & 1) != 0, (t & 2) != 0, (t & 4) != 0};
            struct Key
            {
                int n_int, entity, lex;
            };
            std::vector< Key > homed;
            for (int i = 0; i < N; ++i)
            {
                const int idx[3] = {i % n, (i / n) % n, i / (n * n)};
                bool      ok = true, internal = true;
                int       entity = 0, n_int = 0, mul = 1;
                for (int d = 0; d < 3; ++d)
                {
                    if (idx[d] == 0 && !f[d])
                        ok = false;
                    const int cls = idx[d] == 0 ? 0 : (idx[d] == p ? 2 : 1);
                    internal &= cls == 1;
                    n_int += cls == 1;
                    entity += cls * mul;
                    mul *= 3;
                }
                if (ok && !internal)
                    homed.push_back({n_int, entity, i});
            }
            std::sort(homed.begin(), homed.end(), [](const Key& a, const Key& b) {
                if (a.n_int != b.n_int)
                    return a.n_int > b.n_int;
                if (a.entity != b.entity)
                    return a.entity < b.entity;
                return a.lex < b.lex;
            });
            types[t].index.assign(N, -1);
            types[t].count = static_cast< int >(homed.size());
            for (int k = 0; k < types[t].count; ++k)
                types[t].index[homed[k].lex] = k;
        }
    }
    int  elemBegin(int d, int b) const { return static_cast< int >(i64(ne[d]) * b / parts[d]); }
    i64  gridId(i64 gx, i64 gy, i64 gz) const { return gx + G[0] * (gy + G[1] * gz); }
    int  nParts() const { return parts[0] * parts[1] * parts[2]; }
    void partCoords(int r, int b[3]) const
    {
        b[0] = r % parts[0];
        b[1] = (r / parts[0]) % parts[1];
        b[2] = r / (parts[0] * parts[1]);
    }
    int partRank(const int b[3]) const { return b[0] + parts[0] * (b[1] + parts[1] * b[2]); }
};

// Numbering of the nodes OWNED by one part.
struct PartNumbering
{
    const Layout&          L;
    int                    b[3], eb[3], ee[3], ext[3]; // block coords, element box, extents
    i64                    lo[3], hi[3];               // owned node range (inclusive)
    std::vector< int32_t > trav_pos;                   // local lexicographic element index -> traversal position
    std::vector< int32_t > trav_elem;                  // traversal position -> local lexicographic element index
    std::vector< i64 >     elem_offset;                // traversal position -> first homed non-internal node id
    i64                    n_nonint = 0, n_owned = 0, n_elems = 0;

    PartNumbering(const Layout& L_, int rank) : L{L_}
    {
        L.partCoords(rank, b);
        for (int d = 0; d < 3; ++d)
        {
            eb[d]  = L.elemBegin(d, b[d]);
            ee[d]  = L.elemBegin(d, b[d] + 1);
            ext[d] = ee[d] - eb[d];
            lo[d]  = b[d] == 0 ? 0 : i64(L.p) * eb[d] + 1;
            hi[d]  = i64(L.p) * ee[d];
        }
        n_elems = i64(ext[0]) * ext[1] * ext[2];
        trav_pos.assign(n_elems, -1);
        trav_elem.reserve(n_elems);
        const int brick = brickEdge(L.p);
        for (int bz = 0; bz < ext[2]; bz += brick)
            for (int by = 0; by < ext[1]; by += brick)
                for (int bx = 0; bx < ext[0]; bx += brick)
                    for (int ez = bz; ez < std::min(bz + brick, ext[2]); ++ez)
                        for (int ey = by; ey < std::min(by + brick, ext[1]); ++ey)
                            for (int ex = bx; ex < std::min(bx + brick, ext[0]); ++ex)
                            {
                                const int32_t lex = ex + ext[0] * (ey + ext[1] * ez);
                                trav_pos[lex]     = static_cast< int32_t >(trav_elem.size());
                                trav_elem.push_back(lex);
                            }
        elem_offset.assign(n_elems + 1, 0);
        for (i64 t = 0; t < n_elems; ++t)
            elem_offset[t + 1] = elem_offset[t] + L.types[elemType(trav_elem[t])].count;
        n_nonint = elem_offset[n_elems];
        n_owned  = n_nonint + n_elems * L.n_internal;
    }
    int elemType(int32_t lex) const
    {
        const int ex = lex % ext[0], ey = (lex / ext[0]) % ext[1], ez = lex / (ext[0] * ext[1]);
        return (eb[0] + ex == 0 ? 1 : 0) | (eb[1] + ey == 0 ? 2 : 0) | (eb[2] + ez == 0 ? 4 : 0);
    }
    bool owns(const i64 g[3]) const
    {
        return g[0] >= lo[0] && g[0] <= hi[0] && g[1] >= lo[1] && g[1] <= hi[1] && g[2] >= lo[2] && g[2] <= hi[2];
    }
    // local id of an owned grid node
    i64 localId(const i64 g[3]) const
    {
        int h[3], il[3]; // home element (global coords), local index in it
        for (int d = 0; d < 3; ++d)
        {
            h[d]  = g[d] == 0 ? 0 : static_cast< int >((g[d] - 1) / L.p);
            il[d] = static_cast< int >(g[d] - i64(L.p) * h[d]);
        }
        const int32_t lex  = (h[0] - eb[0]) + ext[0] * ((h[1] - eb[1]) + ext[1] * (h[2] - eb[2]));
        const int32_t t    = trav_pos[lex];
        const int     i    = il[0] + L.n * (il[1] + L.n * il[2]);
        const int32_t intl = L.internal_index[i];
        if (intl >= 0)
            return n_nonint + i64(t) * L.n_internal + intl;
        return elem_offset[t] + L.types[elemType(lex)].index[i];
    }
    i64 ownedCountClosedForm() const { return (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1); }
};

i64 ownedCount(const Layout& L, int rank)
{
    int b[3];
    L.partCoords(rank, b);
    i64 c = 1;
    for (int d = 0; d < 3; ++d)
    {
        const i64 lo = b[d] == 0 ? 0 : i64(L.p) * L.elemBegin(d, b[d]) + 1;
        const i64 hi = i64(L.p) * L.elemBegin(d, b[d] + 1);
        c *= hi - lo + 1;
    }
    return c;
}
} // namespace

struct l3k_hostmesh
{
    int                     order;
    i64                     n_elems = 0, n_interior = 0, n_owned = 0, n_ghost = 0, base = 0, n_global = 0;
    std::vector< uint32_t > elem_nodes;
    std::vector< double >   elem_verts;
    std::vector< i64 >      node_grid_id, ghost_global_id;
    std::vector< uint8_t >  node_boundary;
    std::vector< uint8_t >  elem_boundary;
    std::vector< int >      nbr_rank;
    std::vector< i64 >      send_offsets, ghost_offsets;
    std::vector< int32_t >  send_nodes;
};

extern "C" int l3k_cube_partition_create(const int ne[3], int order, const int parts[3], int rank, double perturb,
                                         l3k_hostmesh** out)
{
    using l3k::dev::setError;
    if (!ne || !parts || !out || order < 1 || order > 15)
    {
        setError("l3k_cube_partition_create: bad arguments");
        return -1;
    }
    for (int d = 0; d < 3; ++d)
        if (ne[d] < 1 || parts[d] < 1 || parts[d] > ne[d])
        {
            setError("l3k_cube_partition_create: need 1 <= parts[d] <= ne[d]");
            return -1;
        }
    const Layout L{ne, order, parts};
    if (rank < 0 || rank >= L.nParts())
    {
        setError("l3k_cube_partition_create: rank %d outside [0,%d)", rank, L.nParts());
        return -1;
    }
    const int p = order, n = L.n, N = L.N;
    auto      hm = std::make_unique< l3k_hostmesh >();
    hm->order    = order;
    const PartNumbering me{L, rank};
    if (me.n_owned != me.ownedCountClosedForm())
    {
        setError("internal error: owned-node count mismatch");
        return -9;
    }
    for (int r = 0; r < rank; ++r)
        hm->base += ownedCount(L, r);
    hm->n_global = i64(L.G[0]) * L.G[1] * L.G[2];
    hm->n_elems  = me.n_elems;
    hm->n_owned  = me.n_owned;

    // ---- ghosts: nodes of my node box on a low plane shared with a lower part; owner = me - (on-low flags)
    std::vector< std::unique_ptr< PartNumbering > > lower(8); // index = flags
    std::vector< i64 >                              lower_base(8, 0);
    struct Ghost
    {
        i64 global_id, grid_id;
        int owner;
    };
    std::vector< Ghost > ghosts;
    {
        const i64 blo[3] = {i64(p) * me.eb[0], i64(p) * me.eb[1], i64(p) * me.eb[2]};
        auto      visit  = [&](i64 gx, i64 gy, i64 gz) {
            const i64 g[3]  = {gx, gy, gz};
            int       flags = 0, ob[3];
            for (int d = 0; d < 3; ++d)
            {
                const bool on_low = me.b[d] > 0 && g[d] == blo[d];
                flags |= on_low << d;
                ob[d] = me.b[d] - on_low;
            }
            const int owner = L.partRank(ob);
            if (!lower[flags])
            {
                lower[flags] = std::make_unique< PartNumbering >(L, owner);
                for (int r = 0; r < owner; ++r)
                    lower_base[flags] += ownedCount(L, r);
            }
            ghosts.push_back({lower_base[flags] + lower[flags]->localId(g), L.gridId(gx, gy, gz), owner});
        };
        const i64 bhi[3] = {i64(p) * me.ee[0], i64(p) * me.ee[1], i64(p) * me.ee[2]};
        // enumerate each ghost node exactly once: x-low plane; y-low plane minus x-low; z-low plane minus x/y-low
        if (me.b[0] > 0)
            for (i64 gz = blo[2]; gz <= bhi[2]; ++gz)
                for (i64 gy = blo[1]; gy <= bhi[1]; ++gy)
                    visit(blo[0], gy, gz);
        if (me.b[1] > 0)
            for (i64 gz = blo[2]; gz <= bhi[2]; ++gz)
                for (i64 gx = blo[0] + (me.b[0] > 0); gx <= bhi[0]; ++gx)
                    visit(gx, blo[1], gz);
        if (me.b[2] > 0)
            for (i64 gy = blo[1] + (me.b[1] > 0); gy <= bhi[1]; ++gy)
                for (i64 gx = blo[0] + (me.b[0] > 0); gx <= bhi[0]; ++gx)
                    visit(gx, gy, blo[2]);
    }
    std::sort(ghosts.begin(), ghosts.end(), [](const Ghost& a, const Ghost& b) { return a.global_id < b.global_id; });
    hm->n_ghost = static_cast< i64 >(ghosts.size());
    hm->ghost_global_id.reserve(ghosts.size());
    for (const auto& g : ghosts)
        hm->ghost_global_id.push_back(g.global_id);
    if (hm->n_owned + hm->n_ghost >= (i64(1) << 32))
    {
        setError("partition has more than 2^32 local nodes");
        return -1;
    }
    std::unordered_map< i64, uint32_t > ghost_index;
    ghost_index.reserve(ghosts.size() * 2);
    for (size_t i = 0; i < ghosts.size(); ++i)
        ghost_index.emplace(ghosts[i].grid_id, static_cast< uint32_t >(hm->n_owned + i));
    lower.clear();

    // ---- elements: interior first, then border, both in traversal order
    std::vector< int32_t > order_list;
    order_list.reserve(me.n_elems);
    auto isBorder = [&](int32_t lex) {
        const int ex = lex % me.ext[0], ey = (lex / me.ext[0]) % me.ext[1], ez = lex / (me.ext[0] * me.ext[1]);
        return (ex == 0 && me.b[0] > 0) || (ey == 0 && me.b[1] > 0) || (ez == 0 && me.b[2] > 0);
    };
    for (int pass = 0; pass < 2; ++pass)
    {
        for (i64 t = 0; t < me.n_elems; ++t)
            if (isBorder(me.trav_elem[t]) == (pass == 1))
                order_list.push_back(me.trav_elem[t]);
        if (pass == 0)
            hm->n_interior = static_cast< i64 >(order_list.size());
    }
    const i64 n_local = hm->n_owned + hm->n_ghost;
    hm->elem_nodes.resize(size_t(me.n_elems) * N);
    hm->elem_verts.resize(size_t(me.n_elems) * 24);
    hm->node_grid_id.assign(n_local, -1);
    hm->node_boundary.assign(n_local, 0);
    hm->elem_boundary.assign(size_t(me.n_elems), 0);
    const double h[3]  = {1. / ne[0], 1. / ne[1], 1. / ne[2]};
    const double hmin  = std::min({h[0], h[1], h[2]});
    const double twopi = 6.283185307179586476925286766559;
    for (i64 k = 0; k < me.n_elems; ++k)
    {
        const int32_t lex = order_list[k];
        const int     E[3] = {me.eb[0] + lex % me.ext[0], me.eb[1] + (lex / me.ext[0]) % me.ext[1],
                              me.eb[2] + lex / (me.ext[0] * me.ext[1])};
        uint32_t*     en   = hm->elem_nodes.data() + size_t(k) * N;
        // element sides on the cube boundary (the BoundaryViews of makeCubeMesh, mesh/primitives/CubeMesh.hpp:66-138)
        hm->elem_boundary[k] = static_cast< uint8_t >((E[2] == 0) | ((E[2] == ne[2] - 1) << 1) | ((E[1] == 0) << 2) |
                                                      ((E[1] == ne[1] - 1) << 3) | ((E[0] == 0) << 4) |
                                                      ((E[0] == ne[0] - 1) << 5));
        for (int i = 0; i < N; ++i)
        {
            const i64 g[3] = {i64(p) * E[0] + i % n, i64(p) * E[1] + (i / n) % n, i64(p) * E[2] + i / (n * n)};
            uint32_t  id;
            if (me.owns(g))
                id = static_cast< uint32_t >(me.localId(g));
            else
                id = ghost_index.at(L.gridId(g[0], g[1], g[2]));
            en[i] = id;
            if (hm->node_grid_id[id] < 0)
            {
                hm->node_grid_id[id]  = L.gridId(g[0], g[1], g[2]);
                hm->node_boundary[id] = static_cast< uint8_t >(
                    (g[2] == 0) | ((g[2] == L.G[2] - 1) << 1) | ((g[1] == 0) << 2) | ((g[1] == L.G[1] - 1) << 3) |
                    ((g[0] == 0) << 4) | ((g[0] == L.G[0] - 1) << 5));
            }
        }
        double* ev = hm->elem_verts.data() + size_t(k) * 24;
        for (int v = 0; v < 8; ++v)
        {
            double x[3] = {(E[0] + (v & 1)) * h[0], (E[1] + ((v >> 1) & 1)) * h[1], (E[2] + (v >> 2)) * h[2]};
            if (perturb != 0.)
            {
                const double d = perturb * hmin * std::sin(twopi * x[0]) * std::sin(twopi * x[1]) * std::sin(twopi * x[2]);
                for (double& c : x)
                    c += d;
            }
            for (int s = 0; s < 3; ++s)
                ev[v * 3 + s] = x[s];
        }
    }

    // ---- neighbours: lower parts own my ghosts, upper parts share my owned nodes
    struct Nbr
    {
        int                    rank;
        std::vector< int32_t > send;
        i64                    ghost_begin = 0, ghost_end = 0;
    };
    std::vector< Nbr > nbrs;
    {
        size_t i = 0;
        while (i < ghosts.size())
        {
            size_t j = i;
            while (j < ghosts.size() && ghosts[j].owner == ghosts[i].owner)
                ++j;
            Nbr nb;
            nb.rank        = ghosts[i].owner;
            nb.ghost_begin = static_cast< i64 >(i);
            nb.ghost_end   = static_cast< i64 >(j);
            nbrs.push_back(std::move(nb));
            i = j;
        }
    }
    for (int off = 1; off < 8; ++off)
    {
        int  sb[3];
        bool exists = true;
        for (int d = 0; d < 3; ++d)
        {
            sb[d] = me.b[d] + ((off >> d) & 1);
            exists &= sb[d] < L.parts[d];
        }
        if (!exists)
            continue;
        // nodes I own inside the sharer's node box
        i64 rlo[3], rhi[3];
        for (int d = 0; d < 3; ++d)
        {
            const i64 slo = i64(p) * L.elemBegin(d, sb[d]), shi = i64(p) * L.elemBegin(d, sb[d] + 1);
            rlo[d] = std::max(me.lo[d], slo);
            rhi[d] = std::min(me.hi[d], shi);
        }
        Nbr nb;
        nb.rank = L.partRank(sb);
        for (i64 gz = rlo[2]; gz <= rhi[2]; ++gz)
            for (i64 gy = rlo[1]; gy <= rhi[1]; ++gy)
                for (i64 gx = rlo[0]; gx <= rhi[0]; ++gx)
                {
                    const i64 g[3] = {gx, gy, gz};
                    nb.send.push_back(static_cast< int32_t >(me.localId(g)));
                }
        if (nb.send.empty())
            continue;
        std::sort(nb.send.begin(), nb.send.end());
        nbrs.push_back(std::move(nb));
    }
    std::sort(nbrs.begin(), nbrs.end(), [](const Nbr& a, const Nbr& b) { return a.rank < b.rank; });
    hm->send_offsets.push_back(0);
    hm->ghost_offsets.push_back(0);
    i64 ghost_cursor = 0;
    for (const auto& nb : nbrs)
    {
        hm->nbr_rank.push_back(nb.rank);
        hm->send_nodes.insert(hm->send_nodes.end(), nb.send.begin(), nb.send.end());
        hm->send_offsets.push_back(static_cast< i64 >(hm->send_nodes.size()));
        if (nb.ghost_end > nb.ghost_begin)
        {
            if (nb.ghost_begin != ghost_cursor)
            {
                setError("internal error: ghost slabs are not contiguous per owner");
                return -9;
            }
            ghost_cursor = nb.ghost_end;
        }
        hm->ghost_offsets.push_back(ghost_cursor);
    }
    for (size_t i = 0; i < ghosts.size(); ++i)
        if (hm->node_grid_id[hm->n_owned + i] != ghosts[i].grid_id)
        {
            setError("internal error: ghost node %zu not referenced by any element", i);
            return -9;
        }
    *out = hm.release();
    return 0;
}

extern "C" int l3k_hostmesh_destroy(l3k_hostmesh* hm)
{
    delete hm;
    return 0;
}

extern "C" int l3k_hostmesh_view_get(const l3k_hostmesh* hm, l3k_hostmesh_view* v)
{
    if (!hm || !v)
    {
        l3k::dev::setError("l3k_hostmesh_view_get: null argument");
        return -1;
    }
    v->dim              = 3;
    v->order            = hm->order;
    v->n_elems          = hm->n_elems;
    v->n_interior_elems = hm->n_interior;
    v->n_owned_nodes    = hm->n_owned;
    v->n_ghost_nodes    = hm->n_ghost;
    v->global_node_base = hm->base;
    v->n_global_nodes   = hm->n_global;
    v->elem_nodes       = hm->elem_nodes.data();
    v->elem_verts       = hm->elem_verts.data();
    v->node_grid_id     = hm->node_grid_id.data();
    v->node_boundary    = hm->node_boundary.data();
    v->elem_boundary    = hm->elem_boundary.data();
    v->n_nbrs           = static_cast< int >(hm->nbr_rank.size());
    v->nbr_rank         = hm->nbr_rank.data();
    v->send_offsets     = hm->send_offsets.data();
    v->send_nodes       = hm->send_nodes.data();
    v->ghost_offsets    = hm->ghost_offsets.data();
    v->ghost_global_id  = hm->ghost_global_id.data();
    return 0;
}
