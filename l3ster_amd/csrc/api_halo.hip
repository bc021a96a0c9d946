// api_halo.hip -- the ghost exchange of a partitioned system behind the C ABI: RCCL neighbour send / receive over xGMI on a
// stream of its own, overlapped with the interior element launches.
//
// Reference: comm::Import / comm::Export (comm/ImportExport.hpp:130-215: owner -> sharer copy, sharer -> owner add, one
// message per neighbour and direction, :295-372, :402-470) and the schedule of MatrixFreeSystem::applyImpl
// (algsys/MatrixFreeSystem.hpp:1020-1140: scale, post import, interior elements, border elements, export, Dirichlet rows).
// The reference goes through MPI on host memory; here a C++ host hands over device pointers and the library issues
//   ncclGroupStart; ncclSend / ncclRecv per neighbour; ncclGroupEnd
// itself ("the multi-GPU variant takes a communicator handle and does the RCCL exchange internally").
// RCCL is loaded at run time (dlopen librccl.so.1): a process that never creates a halo does not need it, and a process
// that also runs torch.distributed shares torch's copy (same SONAME).
#include "objects.hpp"

#include <algorithm>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace
{
struct Rccl
{
    void* lib = nullptr;
    decltype(&ncclGetUniqueId)    getUniqueId    = nullptr;
    decltype(&ncclCommInitRank)   commInitRank   = nullptr;
    decltype(&ncclCommDestroy)    commDestroy    = nullptr;
    decltype(&ncclGroupStart)     groupStart     = nullptr;
    decltype(&ncclGroupEnd)       groupEnd       = nullptr;
    decltype(&ncclSend)           send           = nullptr;
    decltype(&ncclRecv)           recv           = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
};
const Rccl* rccl()
{
    static Rccl       r;
    static std::mutex mtx;
    std::lock_guard   lock{mtx};
    if (r.lib)
        return &r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!r.lib)
    {
        setError("cannot load RCCL (librccl.so.1): %s", dlerror());
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](auto& fn, const char* name) {
        fn = reinterpret_cast< std::remove_reference_t< decltype(fn) > >(dlsym(r.lib, name));
        ok = ok && fn != nullptr;
    };
    sym(r.getUniqueId, "ncclGetUniqueId");
    sym(r.commInitRank, "ncclCommInitRank");
    sym(r.commDestroy, "ncclCommDestroy");
    sym(r.groupStart, "ncclGroupStart");
    sym(r.groupEnd, "ncclGroupEnd");
    sym(r.send, "ncclSend");
    sym(r.recv, "ncclRecv");
    sym(r.getErrorString, "ncclGetErrorString");
    if (!ok)
    {
        setError("librccl.so.1 lacks an expected symbol");
        dlclose(r.lib);
        r.lib = nullptr;
        return nullptr;
    }
    return &r;
}
#define L3K_NCCL(R, call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        const ncclResult_t res_ = (call);                                                                              \
        if (res_ != ncclSuccess)                                                                                       \
        {                                                                                                              \
            setError("%s failed: %s", #call, (R)->getErrorString(res_));                                               \
            return -3;                                                                                                 \
        }                                                                                                              \
    } while (0)
} // namespace

// one rank's exchange lists (ImportExportContext, comm/ImportExport.hpp:29-72) + communicator, stream, events, buffers
// the RCCL implementation of the transport table (the default): user = this
struct RcclTransport
{
    const Rccl* r;
    ncclComm_t  comm = nullptr;
    ~RcclTransport()
    {
        if (comm && r)
            (void)r->commDestroy(comm);
    }
    static int groupBegin(void* u)
    {
        auto* t = static_cast< RcclTransport* >(u);
        L3K_NCCL(t->r, t->r->groupStart());
        return 0;
    }
    static int send(void* u, const double* buf, size_t n, int peer, void* stream)
    {
        auto* t = static_cast< RcclTransport* >(u);
        L3K_NCCL(t->r, t->r->send(buf, n, ncclDouble, peer, t->comm, static_cast< hipStream_t >(stream)));
        return 0;
    }
    static int recv(void* u, double* buf, size_t n, int peer, void* stream)
    {
        auto* t = static_cast< RcclTransport* >(u);
        L3K_NCCL(t->r, t->r->recv(buf, n, ncclDouble, peer, t->comm, static_cast< hipStream_t >(stream)));
        return 0;
    }
    static int groupEnd(void* u, void*)
    {
        auto* t = static_cast< RcclTransport* >(u);
        L3K_NCCL(t->r, t->r->groupEnd());
        return 0;
    }
    static void destroy(void* u) { delete static_cast< RcclTransport* >(u); }
};

struct l3k_halo
{
    l3k_ctx*           ctx;
    l3k_halo_transport tp{}; // group begin / send / recv / group end: RCCL by default, or the caller's (l3k_halo_create_transport)
    bool               owns_tp = false; // set once creation has succeeded: until then tp.user stays the creator's
    int                rank, world, dpn;
    hipStream_t comm_stream = nullptr;
    hipEvent_t  ev_main = nullptr, ev_import = nullptr, ev_export = nullptr;
    struct Nbr
    {
        int                rank;
        int64_t            n_send = 0; // dof rows this rank owns and the neighbour reads (import send / export receive)
        DevBuf< int32_t >  send_rows;
        int64_t            send_off = 0;   // position of this neighbour's block in the packed buffers (in rows)
        int64_t            g0 = 0, g1 = 0; // ghost dof range owned by the neighbour (import receive / export send)
    };
    std::vector< Nbr > nbrs;
    int64_t            n_ghost_dofs = 0, n_send_total = 0;
    DevBuf< double >   xg, yg, sendbuf, recvbuf; // [cols][n_ghost_dofs], [cols][...] per neighbour block
    int                cols = 0;
    // optional timing of the three element launches of the next applies (bench.py's roofline at N > 1): 6 events per apply
    std::vector< hipEvent_t > timing;
    int                       timing_cap = 0, timing_n = 0;
    ~l3k_halo()
    {
        if (owns_tp && tp.destroy)
            tp.destroy(tp.user);
        if (ev_main)
            (void)hipEventDestroy(ev_main);
        if (ev_import)
            (void)hipEventDestroy(ev_import);
        if (ev_export)
            (void)hipEventDestroy(ev_export);
        if (comm_stream)
            (void)hipStreamDestroy(comm_stream);
        for (hipEvent_t e : timing)
            (void)hipEventDestroy(e);
    }
};

namespace
{
int ensureBuffers(l3k_halo* h, int ncols)
{
    if (ncols <= h->cols)
        return 0;
    const auto grow = [&](DevBuf< double >& b, int64_t rows) {
        b = DevBuf< double >{};
        return b.alloc(size_t(std::max< int64_t >(rows, 1)) * ncols);
    };
    if (int rc = grow(h->xg, h->n_ghost_dofs))
        return rc;
    if (int rc = grow(h->yg, h->n_ghost_dofs))
        return rc;
    if (int rc = grow(h->sendbuf, h->n_send_total))
        return rc;
    if (int rc = grow(h->recvbuf, h->n_send_total))
        return rc;
    h->cols = ncols;
    return 0;
}
// owner -> sharer: the packed owned rows go out, the ghost slabs come in.  Issued on the communication stream behind
// `after` (an event of the main stream); `done` is recorded behind the group.
int postImport(l3k_halo* h, int ncols, double* ghost, size_t ldg, hipEvent_t after, hipEvent_t done)
{
    if (h->n_send_total == 0 && h->n_ghost_dofs == 0) // nothing to exchange (a world of one rank): no group call
        return hipEventRecord(done, h->ctx->stream) == hipSuccess ? 0 : -3;
    L3K_HIP(hipStreamWaitEvent(h->comm_stream, after, 0));
    const auto& tp = h->tp;
    if (int rc = tp.group_begin(tp.user))
        return rc;
    // (a failed send / recv must not leave the transport's group open -- an open RCCL group swallows every later call: the group
    // is closed first, the FIRST error is what the caller sees)
    int posted = 0;
    for (const auto& nb : h->nbrs)
        for (int c = 0; c < ncols && posted == 0; ++c)
        {
            if (nb.n_send > 0) // block of neighbour nb: [ncols][n_send]
                posted = tp.send(tp.user, h->sendbuf.ptr + nb.send_off * ncols + nb.n_send * c, size_t(nb.n_send), nb.rank, h->comm_stream);
            if (posted == 0 && nb.g1 > nb.g0)
                posted = tp.recv(tp.user, ghost + ldg * c + nb.g0, size_t(nb.g1 - nb.g0), nb.rank, h->comm_stream);
        }
    if (posted != 0)
    {
        const std::string first = l3k::dev::lastError();
        (void)tp.group_end(tp.user, h->comm_stream);
        setError("%s", first.c_str());
        return posted;
    }
    if (int rc = tp.group_end(tp.user, h->comm_stream))
        return rc;
    L3K_HIP(hipEventRecord(done, h->comm_stream));
    return 0;
}
// sharer -> owner: the ghost slabs go out, the packed contributions to the owned rows come in
int postExport(l3k_halo* h, int ncols, const double* ghost, size_t ldg, hipEvent_t after, hipEvent_t done)
{
    if (h->n_send_total == 0 && h->n_ghost_dofs == 0)
        return hipEventRecord(done, h->ctx->stream) == hipSuccess ? 0 : -3;
    L3K_HIP(hipStreamWaitEvent(h->comm_stream, after, 0));
    const auto& tp = h->tp;
    if (int rc = tp.group_begin(tp.user))
        return rc;
    int posted = 0; // (as in postImport: the group is closed before an error is reported)
    for (const auto& nb : h->nbrs)
        for (int c = 0; c < ncols && posted == 0; ++c)
        {
            if (nb.g1 > nb.g0)
                posted = tp.send(tp.user, ghost + ldg * c + nb.g0, size_t(nb.g1 - nb.g0), nb.rank, h->comm_stream);
            if (posted == 0 && nb.n_send > 0)
                posted = tp.recv(tp.user, h->recvbuf.ptr + nb.send_off * ncols + nb.n_send * c, size_t(nb.n_send), nb.rank, h->comm_stream);
        }
    if (posted != 0)
    {
        const std::string first = l3k::dev::lastError();
        (void)tp.group_end(tp.user, h->comm_stream);
        setError("%s", first.c_str());
        return posted;
    }
    if (int rc = tp.group_end(tp.user, h->comm_stream))
        return rc;
    L3K_HIP(hipEventRecord(done, h->comm_stream));
    return 0;
}
int packAll(l3k_halo* h, const double* d_owned, size_t ld, int ncols)
{
    for (const auto& nb : h->nbrs)
        if (nb.n_send > 0)
            if (int rc = l3k_pack_rows(h->ctx, d_owned, ld, ncols, nb.send_rows.ptr, nb.n_send, h->sendbuf.ptr + nb.send_off * ncols))
                return rc;
    return 0;
}
int unpackAll(l3k_halo* h, double* d_owned, size_t ld, int ncols)
{
    // one launch per neighbour, in the order of the neighbour list: rows shared with several neighbours receive their
    // contributions in a fixed order (comm/ImportExport.hpp:448-470 adds them as the messages arrive)
    for (const auto& nb : h->nbrs)
        if (nb.n_send > 0)
            if (int rc = l3k_unpack_add_rows(h->ctx, h->recvbuf.ptr + nb.send_off * ncols, nb.n_send, nb.send_rows.ptr, d_owned, ld, ncols))
                return rc;
    return 0;
}
} // namespace

extern "C" {
int l3k_halo_unique_id(char* id128)
{
    const Rccl* r = rccl();
    if (!r || !id128)
    {
        if (r)
            setError("l3k_halo_unique_id: null argument");
        return -1;
    }
    ncclUniqueId id;
    L3K_NCCL(r, r->getUniqueId(&id));
    static_assert(sizeof id == 128);
    __builtin_memcpy(id128, &id, sizeof id);
    return 0;
}

} // extern "C"

namespace
{
// the exchange lists, stream, events of one rank around a transport table (takes ownership of tp.user on success only)
int createHalo(l3k_ctx* ctx, const l3k_halo_transport& tp, int rank, int world, int dofs_per_node, int n_nbrs, const int* nbr_rank,
               const int64_t* send_offsets, const int32_t* send_nodes, const int64_t* ghost_offsets, l3k_halo** out)
{
    auto h   = std::make_unique< l3k_halo >();
    h->ctx   = ctx;
    h->tp    = tp;
    h->rank  = rank;
    h->world = world;
    h->dpn   = dofs_per_node;
    L3K_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < n_nbrs; ++i)
    {
        if (nbr_rank[i] < 0 || nbr_rank[i] >= world)
        {
            setError("neighbour rank %d outside [0,%d)", nbr_rank[i], world);
            return -1;
        }
        l3k_halo::Nbr nb;
        nb.rank               = nbr_rank[i];
        const int64_t n_nodes = send_offsets[i + 1] - send_offsets[i];
        nb.n_send             = n_nodes * dofs_per_node;
        nb.send_off           = h->n_send_total;
        if (n_nodes > 0)
        {
            if (!send_nodes)
            {
                setError("l3k_halo_create: send_nodes is null");
                return -1;
            }
            std::vector< int32_t > rows(static_cast< size_t >(nb.n_send));
            for (int64_t k = 0; k < n_nodes; ++k)
                for (int u = 0; u < dofs_per_node; ++u)
                    rows[k * dofs_per_node + u] = send_nodes[send_offsets[i] + k] * dofs_per_node + u;
            if (int rc = nb.send_rows.upload(rows.data(), rows.size(), ctx->stream))
                return rc;
            L3K_HIP(hipStreamSynchronize(ctx->stream)); // (rows is a local)
        }
        nb.g0 = ghost_offsets[i] * dofs_per_node;
        nb.g1 = ghost_offsets[i + 1] * dofs_per_node;
        h->n_send_total += nb.n_send;
        h->n_ghost_dofs = std::max(h->n_ghost_dofs, nb.g1);
        h->nbrs.push_back(std::move(nb));
    }
    L3K_HIP(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    L3K_HIP(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    L3K_HIP(hipEventCreateWithFlags(&h->ev_import, hipEventDisableTiming));
    L3K_HIP(hipEventCreateWithFlags(&h->ev_export, hipEventDisableTiming));
    h->owns_tp = true;
    *out       = h.release();
    return 0;
}
bool haloArgsOk(l3k_ctx* ctx, l3k_halo** out, int rank, int world, int dofs_per_node, int n_nbrs, const int* nbr_rank,
                const int64_t* send_offsets, const int64_t* ghost_offsets)
{
    return ctx && out && rank >= 0 && rank < world && dofs_per_node >= 1 && n_nbrs >= 0 &&
           (n_nbrs == 0 || (nbr_rank && send_offsets && ghost_offsets));
}

// ---- the in-process transport: ranks are threads of ONE process (each with its own context and stream, on one GPU or on
// several), messages are device-to-device copies ordered by events.  It is what a single-process multi-GPU host uses
// instead of RCCL, and the seam through which the tests drive l3k_mf_apply_dist with more than one rank on a one-GPU box.
struct InprocMsg
{
    const double* buf;
    size_t        n;
    hipEvent_t    ready; // recorded by the sender: the payload is complete
    hipEvent_t    done  = nullptr; // recorded by the receiver behind its copy
    bool          acked = false;
    bool          failed = false; // the receiver rejected the message, or the sender's group was abandoned: no `done` event
};
} // namespace
struct l3k_inproc_group
{
    int                                                            world;
    std::mutex                                                     m;
    std::condition_variable                                        cv;
    std::map< std::pair< int, int >, std::deque< std::shared_ptr< InprocMsg > > > channel; // (src, dst) -> FIFO
    struct Endpoint
    {
        l3k_inproc_group* g;
        int               rank;
        struct Recv
        {
            double* buf;
            size_t  n;
            int     peer;
        };
        std::vector< std::shared_ptr< InprocMsg > > sends;
        std::vector< Recv >                         recvs;
        // events in two pools used by alternate groups: an event of group k is recorded again in group k + 2 at the
        // earliest, when every peer has long called hipStreamWaitEvent on its group-k record
        std::vector< hipEvent_t > pool[2];
        size_t                    used = 0;
        unsigned                  gen  = 0;
        int takeEvent(hipEvent_t* e)
        {
            auto& p = pool[gen & 1u];
            if (used == p.size())
            {
                hipEvent_t ev;
                L3K_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                p.push_back(ev);
            }
            *e = p[used++];
            return 0;
        }
        ~Endpoint()
        {
            for (auto& p : pool)
                for (hipEvent_t e : p)
                    (void)hipEventDestroy(e);
        }
    };
    std::vector< std::unique_ptr< Endpoint > > endpoints;
    static constexpr auto                      timeout = std::chrono::seconds(120);

    static int groupBegin(void* u)
    {
        auto* ep = static_cast< Endpoint* >(u);
        ep->sends.clear();
        ep->recvs.clear();
        ep->used = 0;
        ++ep->gen;
        return 0;
    }
    static int send(void* u, const double* buf, size_t n, int peer, void* stream)
    {
        auto*      ep = static_cast< Endpoint* >(u);
        hipEvent_t ready;
        if (int rc = ep->takeEvent(&ready))
            return rc;
        L3K_HIP(hipEventRecord(ready, static_cast< hipStream_t >(stream)));
        auto msg = std::make_shared< InprocMsg >(InprocMsg{buf, n, ready});
        {
            std::lock_guard lock{ep->g->m};
            ep->g->channel[{ep->rank, peer}].push_back(msg);
        }
        ep->g->cv.notify_all();
        ep->sends.push_back(std::move(msg));
        return 0;
    }
    static int recv(void* u, double* buf, size_t n, int peer, void*)
    {
        static_cast< Endpoint* >(u)->recvs.push_back({buf, n, peer}); // matched at the end of the group
        return 0;
    }
    // a group that failed: this endpoint's messages still queued are withdrawn (a later group would pair them with the wrong
    // receive) and marked, so that nobody waits for them; its own receive list is dropped
    static void abandon(Endpoint* ep)
    {
        l3k_inproc_group* g = ep->g;
        {
            std::lock_guard lock{g->m};
            for (const auto& msg : ep->sends)
            {
                for (auto& [key, q] : g->channel)
                    if (key.first == ep->rank)
                        q.erase(std::remove(q.begin(), q.end(), msg), q.end());
                if (!msg->acked)
                    msg->acked = msg->failed = true;
            }
        }
        g->cv.notify_all();
        ep->sends.clear();
        ep->recvs.clear();
    }
    static int groupEnd(void* u, void* stream_)
    {
        const int rc = groupEndImpl(u, stream_);
        if (rc != 0)
            abandon(static_cast< Endpoint* >(u));
        return rc;
    }
    static int groupEndImpl(void* u, void* stream_)
    {
        auto*             ep     = static_cast< Endpoint* >(u);
        l3k_inproc_group* g      = ep->g;
        hipStream_t       stream = static_cast< hipStream_t >(stream_);
        for (const auto& r : ep->recvs)
        {
            std::shared_ptr< InprocMsg > msg;
            {
                std::unique_lock lock{g->m};
                auto&            q = g->channel[{r.peer, ep->rank}];
                if (!g->cv.wait_for(lock, timeout, [&] { return !q.empty(); }))
                {
                    setError("in-process transport: rank %d waited %d s for a message of rank %d", ep->rank, int(timeout.count()), r.peer);
                    return -3;
                }
                msg = q.front();
                q.pop_front();
            }
            if (msg->n != r.n)
            {
                {
                    std::lock_guard lock{g->m}; // (the sender must not wait for a copy that will not happen)
                    msg->acked = msg->failed = true;
                }
                g->cv.notify_all();
                setError("in-process transport: rank %d expects %zu doubles from rank %d, which sent %zu", ep->rank, r.n, r.peer, msg->n);
                return -1;
            }
            hipEvent_t done;
            if (int rc = ep->takeEvent(&done))
                return rc;
            L3K_HIP(hipStreamWaitEvent(stream, msg->ready, 0));
            L3K_HIP(hipMemcpyAsync(r.buf, msg->buf, sizeof(double) * r.n, hipMemcpyDefault, stream));
            L3K_HIP(hipEventRecord(done, stream));
            {
                std::lock_guard lock{g->m};
                msg->done  = done;
                msg->acked = true;
            }
            g->cv.notify_all();
        }
        // a send is complete (its buffer may be rewritten) when the receiver's copy is: the stream waits for it
        for (const auto& msg : ep->sends)
        {
            {
                std::unique_lock lock{g->m};
                if (!g->cv.wait_for(lock, timeout, [&] { return msg->acked; }))
                {
                    setError("in-process transport: rank %d waited %d s for a receiver", ep->rank, int(timeout.count()));
                    return -3;
                }
                if (msg->failed)
                {
                    setError("in-process transport: a message of rank %d was rejected by its receiver", ep->rank);
                    return -1;
                }
            }
            L3K_HIP(hipStreamWaitEvent(stream, msg->done, 0));
        }
        ep->sends.clear();
        ep->recvs.clear();
        return 0;
    }
};

extern "C" {
int l3k_halo_create(l3k_ctx* ctx, const char* id128, int rank, int world, int dofs_per_node, int n_nbrs, const int* nbr_rank,
                    const int64_t* send_offsets, const int32_t* send_nodes, const int64_t* ghost_offsets, l3k_halo** out)
{
    if (!id128 || !haloArgsOk(ctx, out, rank, world, dofs_per_node, n_nbrs, nbr_rank, send_offsets, ghost_offsets))
    {
        setError("l3k_halo_create: inconsistent arguments");
        return -1;
    }
    const Rccl* r = rccl();
    if (!r)
        return -3;
    L3K_HIP(hipSetDevice(ctx->device));
    auto t = std::make_unique< RcclTransport >();
    t->r   = r;
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof id);
    L3K_NCCL(r, r->commInitRank(&t->comm, world, id, rank));
    l3k_halo_transport tp{t.get(), &RcclTransport::groupBegin, &RcclTransport::send, &RcclTransport::recv, &RcclTransport::groupEnd,
                          &RcclTransport::destroy};
    const int          rc = createHalo(ctx, tp, rank, world, dofs_per_node, n_nbrs, nbr_rank, send_offsets, send_nodes, ghost_offsets, out);
    if (rc == 0)
        (void)t.release(); // (the halo owns it now; on failure the unique_ptr destroys the communicator)
    return rc;
}
int l3k_halo_create_transport(l3k_ctx* ctx, const l3k_halo_transport* transport, int rank, int world, int dofs_per_node, int n_nbrs,
                              const int* nbr_rank, const int64_t* send_offsets, const int32_t* send_nodes, const int64_t* ghost_offsets,
                              l3k_halo** out)
{
    if (!transport || !transport->group_begin || !transport->send || !transport->recv || !transport->group_end ||
        !haloArgsOk(ctx, out, rank, world, dofs_per_node, n_nbrs, nbr_rank, send_offsets, ghost_offsets))
    {
        setError("l3k_halo_create_transport: inconsistent arguments");
        return -1;
    }
    return createHalo(ctx, *transport, rank, world, dofs_per_node, n_nbrs, nbr_rank, send_offsets, send_nodes, ghost_offsets, out);
}
int l3k_inproc_group_create(int world, l3k_inproc_group** out)
{
    if (world < 1 || !out)
    {
        setError("l3k_inproc_group_create: bad argument");
        return -1;
    }
    auto g   = std::make_unique< l3k_inproc_group >();
    g->world = world;
    for (int r = 0; r < world; ++r)
    {
        g->endpoints.push_back(std::make_unique< l3k_inproc_group::Endpoint >());
        g->endpoints.back()->g    = g.get();
        g->endpoints.back()->rank = r;
    }
    *out = g.release();
    return 0;
}
int l3k_inproc_group_destroy(l3k_inproc_group* group)
{
    delete group;
    return 0;
}
int l3k_inproc_transport(l3k_inproc_group* group, int rank, l3k_halo_transport* out)
{
    if (!group || !out || rank < 0 || rank >= group->world)
    {
        setError("l3k_inproc_transport: bad argument");
        return -1;
    }
    // (the endpoint belongs to the group: no destroy callback; the group must outlive the halos that use it)
    *out = l3k_halo_transport{group->endpoints[size_t(rank)].get(), &l3k_inproc_group::groupBegin, &l3k_inproc_group::send,
                              &l3k_inproc_group::recv,              &l3k_inproc_group::groupEnd,   nullptr};
    return 0;
}
int l3k_halo_destroy(l3k_halo* halo)
{
    delete halo;
    return 0;
}
int64_t l3k_halo_n_ghost_dofs(const l3k_halo* halo)
{
    return halo ? halo->n_ghost_dofs : -1;
}

// comm::Import of a multivector over the owned rows (comm/ImportExport.hpp:295-372)
int l3k_halo_import(l3k_halo* h, const double* d_owned, size_t ld, int ncols, double* d_ghost, size_t ldg)
{
    if (!h || !d_owned || !d_ghost || ncols < 1)
    {
        setError("l3k_halo_import: bad argument");
        return -1;
    }
    L3K_HIP(hipSetDevice(h->ctx->device));
    if (int rc = ensureBuffers(h, ncols))
        return rc;
    if (int rc = packAll(h, d_owned, ld, ncols))
        return rc;
    L3K_HIP(hipEventRecord(h->ev_main, h->ctx->stream));
    if (int rc = postImport(h, ncols, d_ghost, ldg, h->ev_main, h->ev_import))
        return rc;
    L3K_HIP(hipStreamWaitEvent(h->ctx->stream, h->ev_import, 0));
    return 0;
}
// comm::Export: every ghost row added into its owner's row (comm/ImportExport.hpp:402-470)
int l3k_halo_export_add(l3k_halo* h, const double* d_ghost, size_t ldg, int ncols, double* d_owned, size_t ld)
{
    if (!h || !d_owned || !d_ghost || ncols < 1)
    {
        setError("l3k_halo_export_add: bad argument");
        return -1;
    }
    L3K_HIP(hipSetDevice(h->ctx->device));
    if (int rc = ensureBuffers(h, ncols))
        return rc;
    L3K_HIP(hipEventRecord(h->ev_main, h->ctx->stream));
    if (int rc = postExport(h, ncols, d_ghost, ldg, h->ev_main, h->ev_export))
        return rc;
    L3K_HIP(hipStreamWaitEvent(h->ctx->stream, h->ev_export, 0));
    return unpackAll(h, d_owned, ld, ncols);
}

// HIP events around the three element launches (first interior half, border, second interior half) of the next `n_applies`
// calls of l3k_mf_apply_dist, on the stream they are launched on; l3k_halo_timing_get waits for the apply and returns the
// three durations in milliseconds
int l3k_halo_timing_begin(l3k_halo* h, int n_applies)
{
    if (!h || n_applies < 0)
    {
        setError("l3k_halo_timing_begin: bad argument");
        return -1;
    }
    L3K_HIP(hipSetDevice(h->ctx->device));
    while (int(h->timing.size()) < 6 * n_applies)
    {
        hipEvent_t e;
        L3K_HIP(hipEventCreate(&e));
        h->timing.push_back(e);
    }
    h->timing_cap = n_applies;
    h->timing_n   = 0;
    return 0;
}
int l3k_halo_timing_get(l3k_halo* h, int apply, double ms[3])
{
    if (!h || !ms || apply < 0 || apply >= h->timing_n)
    {
        setError("l3k_halo_timing_get: apply %d was not timed", apply);
        return -1;
    }
    hipEvent_t* e = &h->timing[size_t(6) * apply];
    L3K_HIP(hipEventSynchronize(e[5]));
    for (int k = 0; k < 3; ++k)
    {
        float t = 0.f;
        L3K_HIP(hipEventElapsedTime(&t, e[2 * k], e[2 * k + 1]));
        ms[k] = t;
    }
    return 0;
}

// y <- alpha A x + beta y on the owned rows of a partitioned system: MatrixFreeSystem::applyImpl
// (algsys/MatrixFreeSystem.hpp:1020-1140) with both exchanges hidden behind interior element launches
int l3k_mf_apply_dist(l3k_mf* mf, l3k_halo* h, const double* d_x, size_t ldx, double* d_y, size_t ldy, int ncols, double alpha,
                      double beta)
{
    if (!mf || !h || ncols < 1)
    {
        setError("l3k_mf_apply_dist: bad argument");
        return -1;
    }
    // a rank that owns nothing (no element, no node, no neighbour: tests/EmptyPartitionTest.cpp) takes part and returns
    if (mf->mesh->n_elems == 0 && mf->mesh->n_owned_nodes == 0 && mf->mesh->n_ghost_nodes == 0 && h->nbrs.empty())
    {
        // (its timing slot is filled like everybody's -- three launches of no duration -- so that l3k_halo_timing_get(i) means
        // the same apply on every rank)
        if (h->timing_n < h->timing_cap)
        {
            L3K_HIP(hipSetDevice(h->ctx->device));
            hipEvent_t* tev = &h->timing[size_t(6) * h->timing_n++];
            for (int k = 0; k < 6; ++k)
                L3K_HIP(hipEventRecord(tev[k], mf->ctx->stream));
        }
        return 0;
    }
    if (!d_x || !d_y)
    {
        setError("l3k_mf_apply_dist: null vector");
        return -1;
    }
    if (h->n_ghost_dofs != mf->mesh->n_ghost_nodes * mf->mesh->dofs_per_node || h->dpn != mf->mesh->dofs_per_node)
    {
        setError("l3k_mf_apply_dist: the halo's ghost range (%lld dofs) does not match the mesh (%lld)", (long long)h->n_ghost_dofs,
                 (long long)(mf->mesh->n_ghost_nodes * mf->mesh->dofs_per_node));
        return -1;
    }
    if (h->ctx != mf->ctx) // (pack / unpack run on the halo's context, the element launches on the system's: one stream orders both)
    {
        setError("l3k_mf_apply_dist: the halo and the system belong to different contexts");
        return -1;
    }
    L3K_HIP(hipSetDevice(h->ctx->device));
    if (int rc = ensureBuffers(h, ncols))
        return rc;
    hipStream_t  s   = mf->ctx->stream;
    const size_t ldg = size_t(std::max< int64_t >(h->n_ghost_dofs, 1));
    double *     xg = h->xg.ptr, *yg = h->yg.ptr;
    if (int rc = l3k_mf_scale(mf, d_y, ldy, ncols, beta)) // (:1038)
        return rc;
    L3K_HIP(hipMemsetAsync(yg, 0, sizeof(double) * ldg * ncols, s)); // export buffer <- 0 (:1048)
    // ---- import: pack, post; the first half of the interior elements runs meanwhile (:1055, :1073-1105)
    if (int rc = packAll(h, d_x, ldx, ncols))
        return rc;
    L3K_HIP(hipEventRecord(h->ev_main, s));
    if (int rc = postImport(h, ncols, xg, ldg, h->ev_main, h->ev_import))
        return rc;
    hipEvent_t* tev = h->timing_n < h->timing_cap ? &h->timing[size_t(6) * h->timing_n++] : nullptr;
    const auto  stamp = [&](int k) { return tev ? hipEventRecord(tev[k], s) : hipSuccess; };
    L3K_HIP(stamp(0));
    if (int rc = l3k_mf_apply_elems(mf, 3, d_x, ldx, nullptr, 0, d_y, ldy, nullptr, 0, ncols, alpha, beta))
        return rc;
    L3K_HIP(stamp(1));
    // ---- border elements read the imported ghosts and add into the export buffer (:1058-1072)
    L3K_HIP(hipStreamWaitEvent(s, h->ev_import, 0));
    L3K_HIP(stamp(2));
    if (int rc = l3k_mf_apply_elems(mf, 1, d_x, ldx, xg, ldg, d_y, ldy, yg, ldg, ncols, alpha, beta))
        return rc;
    L3K_HIP(stamp(3));
    // ---- export: post; the second half of the interior elements runs meanwhile (:1071, ImportExport.hpp:402-433)
    L3K_HIP(hipEventRecord(h->ev_main, s));
    if (int rc = postExport(h, ncols, yg, ldg, h->ev_main, h->ev_export))
        return rc;
    L3K_HIP(stamp(4));
    if (int rc = l3k_mf_apply_elems(mf, 4, d_x, ldx, nullptr, 0, d_y, ldy, nullptr, 0, ncols, alpha, beta))
        return rc;
    L3K_HIP(stamp(5));
    L3K_HIP(hipStreamWaitEvent(s, h->ev_export, 0));
    if (int rc = unpackAll(h, d_y, ldy, ncols)) // (:1107, ImportExport.hpp:448-470)
        return rc;
    return l3k_mf_dirichlet_rows(mf, d_x, ldx, d_y, ldy, ncols, alpha); // (:1087-1098)
}
} // extern "C"
