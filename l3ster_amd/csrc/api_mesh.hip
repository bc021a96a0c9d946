// api_mesh.hip -- order elevation of an order-1 hex mesh on the device.
//
// Stands in for mesh::convertMeshToOrder (mesh/ConvertMeshToOrder.hpp:51-104), which walks the elements serially and
// matches every new element against its already converted neighbours through the METIS dual graph, followed by the local
// renumbering of mesh/LocalMeshView.hpp:425-458 (non-internal nodes first, element-internal nodes contiguous per
// element).  Here all elements are processed at once: every element emits the keys of its 12 edges (sorted vertex
// pair) and 6 faces (sorted vertex quadruple), the keys are sorted and run-length numbered on the device (rocPRIM merge
// sort + scan), and every (element, local node) then computes its id from the entity it lies on:
//     [ vertices | (p-1) per unique edge | (p-1)^2 per unique face | (p-1)^3 per element ]
// i.e. the numbering class l3k_mesh_create wants (internal nodes last, contiguous per element).  Entities are numbered in
// lexicographic key order, positions inside an edge / face in the entity's canonical frame (from the lower vertex id;
// face: origin at the lowest vertex id, first axis towards the lower of its two neighbours), so elements that see a shared
// entity in different orientations agree.  The numbering is NOT the reference's (which depends on its traversal order);
// what is reproduced is the resulting topology: which nodes are shared (node count, tests/MeshTests.cpp:244-279).
// HBM-bound integer work: no attempt to reach the matrix cores.
#include "objects.hpp"

#include <rocprim/rocprim.hpp>

namespace
{
using l3k::dev::setError;

struct FaceKey
{
    uint32_t v[4];
};
struct EdgeKey
{
    uint32_t a, b;
};
struct FaceLess
{
    __host__ __device__ bool operator()(const FaceKey& x, const FaceKey& y) const
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (x.v[i] != y.v[i])
                return x.v[i] < y.v[i];
        return false;
    }
};
struct EdgeLess
{
    __host__ __device__ bool operator()(const EdgeKey& x, const EdgeKey& y) const { return x.a != y.a ? x.a < y.a : x.b < y.b; }
};

// local vertex v = i + 2j + 4k (mesh/primitives/CubeMesh.hpp:46-61).  Edges: 0-3 along x at (j,k) = (e&1, e>>1), 4-7
// along y at (i,k), 8-11 along z at (i,j).  Faces = sides 0 z-, 1 z+, 2 y-, 3 y+, 4 x-, 5 x+ (ElementTraits.hpp:84-95)
// with in-face axes (s,t) = (x,y), (x,z), (y,z).
__host__ __device__ inline void edgeVerts(int e, int& va, int& vb)
{
    const int d = e >> 2, q = e & 3, c0 = q & 1, c1 = q >> 1;
    if (d == 0)
        va = 2 * c0 + 4 * c1, vb = va + 1;
    else if (d == 1)
        va = c0 + 4 * c1, vb = va + 2;
    else
        va = c0 + 2 * c1, vb = va + 4;
}
__host__ __device__ inline int faceVert(int f, int s, int t) // local vertex at in-face corner (s,t)
{
    const int hi = f & 1;
    switch (f >> 1)
    {
    case 0: return s + 2 * t + 4 * hi;
    case 1: return s + 4 * t + 2 * hi;
    default: return 2 * s + 4 * t + hi;
    }
}

__global__ void emitKeys(const uint32_t* conn, int64_t n_elems, EdgeKey* ek, FaceKey* fk, uint32_t* ev, uint32_t* fv)
{
    const int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (t >= n_elems * 18)
        return;
    const int64_t   e = t / 18;
    const int       r = int(t - e * 18);
    const uint32_t* c = conn + e * 8;
    if (r < 12)
    {
        int va, vb;
        edgeVerts(r, va, vb);
        const uint32_t a = c[va], b = c[vb];
        ek[e * 12 + r] = EdgeKey{a < b ? a : b, a < b ? b : a};
        ev[e * 12 + r] = uint32_t(e * 12 + r);
    }
    else
    {
        const int f = r - 12;
        uint32_t  v[4] = {c[faceVert(f, 0, 0)], c[faceVert(f, 1, 0)], c[faceVert(f, 0, 1)], c[faceVert(f, 1, 1)]};
        // sorting network for 4
        auto cs = [&](int i, int j) {
            if (v[j] < v[i])
            {
                const uint32_t x = v[i];
                v[i] = v[j];
                v[j] = x;
            }
        };
        cs(0, 1), cs(2, 3), cs(0, 2), cs(1, 3), cs(1, 2);
        fk[e * 6 + f] = FaceKey{{v[0], v[1], v[2], v[3]}};
        fv[e * 6 + f] = uint32_t(e * 6 + f);
    }
}

template < typename Key, typename Less >
__global__ void markHeads(const Key* sorted, int64_t n, uint32_t* head, Less less)
{
    const int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (t < n)
        head[t] = (t == 0 || less(sorted[t - 1], sorted[t])) ? 1u : 0u; // (sorted: "different" = "previous is less")
}
// id of the run each sorted slot belongs to -> back to the (element, local entity) that emitted it
__global__ void scatterIds(const uint32_t* incl, const uint32_t* src, int64_t n, uint32_t* id_of)
{
    const int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (t < n)
        id_of[src[t]] = incl[t] - 1u;
}

__global__ void numberNodes(const uint32_t* conn, int64_t n_elems, int p, uint32_t n_vertices, uint32_t n_edges, uint32_t n_faces,
                            const uint32_t* edge_id, const uint32_t* face_id, uint32_t* out)
{
    const int     n1 = p + 1, N = n1 * n1 * n1, m = p - 1;
    const int64_t t  = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (t >= n_elems * N)
        return;
    const int64_t   e  = t / N;
    const int       ln = int(t - e * N), i = ln % n1, j = (ln / n1) % n1, k = ln / (n1 * n1);
    const uint32_t* c  = conn + e * 8;
    const bool      bi = i == 0 || i == p, bj = j == 0 || j == p, bk = k == 0 || k == p;
    const int       nb = int(bi) + int(bj) + int(bk);
    const uint32_t  edge_base = n_vertices, face_base = edge_base + n_edges * uint32_t(m),
                   int_base = face_base + n_faces * uint32_t(m * m);
    uint32_t id;
    if (nb == 3)
        id = c[(i ? 1 : 0) + 2 * (j ? 1 : 0) + 4 * (k ? 1 : 0)];
    else if (nb == 2)
    {
        int le, tpar; // local edge, parameter 1..p-1 from the edge's first local vertex
        if (!bi)
            le = (j ? 1 : 0) + 2 * (k ? 1 : 0), tpar = i;
        else if (!bj)
            le = 4 + (i ? 1 : 0) + 2 * (k ? 1 : 0), tpar = j;
        else
            le = 8 + (i ? 1 : 0) + 2 * (j ? 1 : 0), tpar = k;
        int va, vb;
        edgeVerts(le, va, vb);
        const int pos = c[va] < c[vb] ? tpar - 1 : p - 1 - tpar;
        id            = edge_base + edge_id[e * 12 + le] * uint32_t(m) + uint32_t(pos);
    }
    else if (nb == 1)
    {
        int f, s, tt;
        if (bk)
            f = k ? 1 : 0, s = i, tt = j;
        else if (bj)
            f = 2 + (j ? 1 : 0), s = i, tt = k;
        else
            f = 4 + (i ? 1 : 0), s = j, tt = k;
        // canonical frame: origin = corner with the lowest vertex id, first axis towards the lower of its two neighbours
        int      os = 0, ot = 0;
        uint32_t best = c[faceVert(f, 0, 0)];
        for (int q = 1; q < 4; ++q)
        {
            const uint32_t v = c[faceVert(f, q & 1, q >> 1)];
            if (v < best)
                best = v, os = q & 1, ot = q >> 1;
        }
        const uint32_t ns = c[faceVert(f, 1 - os, ot)], nt = c[faceVert(f, os, 1 - ot)];
        const int      ds = os ? p - s : s, dt = ot ? p - tt : tt; // distances from the origin corner, 1..p-1
        const int      a = ns < nt ? ds : dt, b = ns < nt ? dt : ds;
        id               = face_base + face_id[e * 6 + f] * uint32_t(m * m) + uint32_t((a - 1) + m * (b - 1));
    }
    else
        id = int_base + uint32_t(e) * uint32_t(m * m * m) + uint32_t((i - 1) + m * ((j - 1) + m * (k - 1)));
    out[t] = id;
}

template < typename Key, typename Less >
int uniqueIds(Key* keys, uint32_t* vals, int64_t n, Less less, uint32_t* id_of, uint32_t& n_unique, hipStream_t stream)
{
    if (n == 0)
    {
        n_unique = 0;
        return 0;
    }
    DevBuf< Key >      keys_out;
    DevBuf< uint32_t > vals_out, head;
    if (int rc = keys_out.alloc(size_t(n)))
        return rc;
    if (int rc = vals_out.alloc(size_t(n)))
        return rc;
    if (int rc = head.alloc(size_t(n)))
        return rc;
    size_t tmp_bytes = 0;
    L3K_HIP(rocprim::merge_sort(nullptr, tmp_bytes, keys, keys_out.ptr, vals, vals_out.ptr, size_t(n), less, stream));
    DevBuf< char > tmp;
    if (int rc = tmp.alloc(tmp_bytes ? tmp_bytes : 1))
        return rc;
    L3K_HIP(rocprim::merge_sort(tmp.ptr, tmp_bytes, keys, keys_out.ptr, vals, vals_out.ptr, size_t(n), less, stream));
    const unsigned blocks = unsigned((n + 255) / 256);
    hipLaunchKernelGGL((markHeads< Key, Less >), dim3(blocks), dim3(256), 0, stream, keys_out.ptr, n, head.ptr, less);
    size_t scan_bytes = 0;
    L3K_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, head.ptr, head.ptr, size_t(n), rocprim::plus< uint32_t >(), stream));
    DevBuf< char > tmp2;
    if (int rc = tmp2.alloc(scan_bytes ? scan_bytes : 1))
        return rc;
    L3K_HIP(rocprim::inclusive_scan(tmp2.ptr, scan_bytes, head.ptr, head.ptr, size_t(n), rocprim::plus< uint32_t >(), stream));
    hipLaunchKernelGGL(scatterIds, dim3(blocks), dim3(256), 0, stream, head.ptr, vals_out.ptr, n, id_of);
    L3K_HIP(hipMemcpyAsync(&n_unique, head.ptr + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    L3K_HIP(hipStreamSynchronize(stream));
    return 0;
}
} // namespace

extern "C" int l3k_elevate_order(l3k_ctx* ctx, int64_t n_elems, const uint32_t* conn, int64_t n_vertices, int order,
                                 uint32_t* elem_nodes, int64_t* n_nodes, int64_t* n_noninternal)
{
    if (!ctx || n_elems < 0 || (n_elems > 0 && (!conn || !elem_nodes)) || n_vertices < 0 || order < 1 || order > 15 || !n_nodes ||
        !n_noninternal)
    {
        setError("l3k_elevate_order: bad arguments");
        return -1;
    }
    for (int64_t i = 0; i < n_elems * 8; ++i)
        if (conn[i] >= uint64_t(n_vertices))
        {
            setError("l3k_elevate_order: conn[%lld] = %u is outside [0,%lld)", (long long)i, conn[i], (long long)n_vertices);
            return -1;
        }
    const int64_t N = int64_t(order + 1) * (order + 1) * (order + 1), m = order - 1;
    // ids are 32-bit (Typedefs.h:14: u32 local node ids): bound with the worst case of no shared entity
    if (n_vertices + n_elems * (12 * m + 6 * m * m + m * m * m) >= (int64_t(1) << 32))
    {
        setError("l3k_elevate_order: the elevated mesh may exceed 2^32 nodes");
        return -1;
    }
    L3K_HIP(hipSetDevice(ctx->device));
    hipStream_t             s = ctx->stream;
    DevBuf< uint32_t > d_conn, ev, fv, edge_id, face_id, d_out;
    DevBuf< EdgeKey >  ek;
    DevBuf< FaceKey >  fk;
    uint32_t                n_edges = 0, n_faces = 0;
    if (n_elems > 0)
    {
        if (int rc = d_conn.upload(conn, size_t(n_elems * 8), s))
            return rc;
        if (ek.alloc(size_t(n_elems * 12)) || fk.alloc(size_t(n_elems * 6)) || ev.alloc(size_t(n_elems * 12)) ||
            fv.alloc(size_t(n_elems * 6)) || edge_id.alloc(size_t(n_elems * 12)) || face_id.alloc(size_t(n_elems * 6)) ||
            d_out.alloc(size_t(n_elems * N)))
            return -3;
        hipLaunchKernelGGL(emitKeys, dim3(unsigned((n_elems * 18 + 255) / 256)), dim3(256), 0, s, d_conn.ptr, n_elems, ek.ptr, fk.ptr,
                           ev.ptr, fv.ptr);
        if (int rc = uniqueIds(ek.ptr, ev.ptr, n_elems * 12, EdgeLess{}, edge_id.ptr, n_edges, s))
            return rc;
        if (int rc = uniqueIds(fk.ptr, fv.ptr, n_elems * 6, FaceLess{}, face_id.ptr, n_faces, s))
            return rc;
        hipLaunchKernelGGL(numberNodes, dim3(unsigned((n_elems * N + 255) / 256)), dim3(256), 0, s, d_conn.ptr, n_elems, order,
                           uint32_t(n_vertices), n_edges, n_faces, edge_id.ptr, face_id.ptr, d_out.ptr);
        L3K_HIP(hipGetLastError());
        L3K_HIP(hipMemcpyAsync(elem_nodes, d_out.ptr, sizeof(uint32_t) * size_t(n_elems * N), hipMemcpyDeviceToHost, s));
        L3K_HIP(hipStreamSynchronize(s));
    }
    *n_noninternal = n_vertices + int64_t(n_edges) * m + int64_t(n_faces) * m * m;
    *n_nodes       = *n_noninternal + n_elems * m * m * m;
    return 0;
}
