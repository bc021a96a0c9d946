"""Multi-rank matrix-free operator: one process per GPU, ghost exchange by neighbour send/recv over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Replaces comm::Import / comm::Export (comm/ImportExport.hpp:130-215) and the schedule of
MatrixFreeSystem::applyImpl (algsys/MatrixFreeSystem.hpp:1020-1140):

    Y <- beta*Y, export buffer <- 0                      (:1038, :1048)
    pack owned x rows shared with each upper neighbour, post import sends/receives     (:1055, ImportExport.hpp:295-372)
    interior elements (owned dofs only) -- overlaps the import                         (:1073-1105)
    wait for the import; border elements (read ghost x, add into the ghost export buffer)   (:1058-1072)
    post export sends (ghost slabs are contiguous per owner) / receives                 (:1071, ImportExport.hpp:402-433)
    unpack-add received contributions into owned rows                                   (:1107, ImportExport.hpp:448-470)
    y[d] += alpha*x[d] on owned Dirichlet rows                                          (:1087-1098)

The exchange is a neighbour all-to-all: one message per neighbour per direction, no collective on the apply path.
`backend` supplies the local pieces (scale / pack_rows / apply_elems / unpack_add_rows / dirichlet_rows); in the product
it is l3ster_amd.system.MatrixFreeSystem (HIP kernels through the C ABI).
"""
import numpy as np
import torch
import torch.distributed as dist


class HaloPlan:
    """DOF-level exchange lists of one rank (ImportExportContext, comm/ImportExport.hpp:29-72)."""

    def __init__(self, part, dofs_per_node, device):
        dpn = dofs_per_node
        self.n_ghost_dofs = part.n_ghost_nodes * dpn
        self.sharers = []  # (rank, int32 device tensor of owned dof rows)  -- import send / export receive
        self.owners = []   # (rank, ghost dof begin, ghost dof end)          -- import receive / export send
        for i, nb in enumerate(part.nbr_rank):
            nodes = part.send_nodes[i]
            if len(nodes):
                rows = (nodes.astype(np.int64)[:, None] * dpn + np.arange(dpn)[None, :]).reshape(-1)
                self.sharers.append((nb, torch.as_tensor(rows.astype(np.int32), device=device)))
            g0, g1 = part.ghost_ranges[i]
            if g1 > g0:
                self.owners.append((nb, g0 * dpn, g1 * dpn))


class TorchDistTransport:
    """Neighbour exchange over torch.distributed point-to-point ops (RCCL on GPUs, gloo on CPU): every message of one
    phase is posted in a single batch_isend_irecv (one ncclGroupStart/End)."""

    def __init__(self, group=None):
        self.group = group

    def post(self, sends, recvs):
        ops = [dist.P2POp(dist.isend, t, peer, self.group) for peer, t in sends]
        ops += [dist.P2POp(dist.irecv, t, peer, self.group) for peer, t in recvs]
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def wait(reqs):
        for r in reqs:
            r.wait()


class HostStagedTransport:
    """The same exchange for device tensors over a backend that only moves host memory (gloo): every message goes through a
    host copy.  For rehearsals of the multi-rank schedule with several ranks on ONE GPU (bench.py L3K_BENCH_REHEARSAL=1; RCCL
    refuses two ranks on one device) -- never the measured path."""

    def __init__(self, group=None):
        self.group = group

    def post(self, sends, recvs):
        if torch.cuda.is_available():
            torch.cuda.synchronize()  # the packed rows are on the stream
        host_s = [(peer, t.detach().to("cpu").contiguous()) for peer, t in sends]
        host_r = [(peer, t, torch.empty(t.shape, dtype=t.dtype)) for peer, t in recvs]
        ops = [dist.P2POp(dist.isend, h, peer, self.group) for peer, h in host_s]
        ops += [dist.P2POp(dist.irecv, h, peer, self.group) for peer, _, h in host_r]
        return (dist.batch_isend_irecv(ops) if ops else [], host_s, host_r)

    @staticmethod
    def wait(handle):
        reqs, _keep, host_r = handle
        for r in reqs:
            r.wait()
        for _, t, h in host_r:
            t.copy_(h)


class DistributedOperator:
    def __init__(self, backend, plan, group=None, transport=None):
        self.backend, self.plan = backend, plan
        self.transport = transport or TorchDistTransport(group)
        self._bufs = {}

    def _buffers(self, ncols, like):
        key = (ncols, like.device)
        if key not in self._bufs:
            mk = lambda n: torch.zeros((ncols, max(n, 1)), dtype=torch.float64, device=like.device)
            self._bufs[key] = dict(
                xg=mk(self.plan.n_ghost_dofs), yg=mk(self.plan.n_ghost_dofs),
                send=[mk(idx.numel()) for _, idx in self.plan.sharers],
                recv=[mk(idx.numel()) for _, idx in self.plan.sharers],
                stage=[mk(e - b) for _, b, e in self.plan.owners])
        return self._bufs[key]

    def apply(self, X, Y, alpha=1.0, beta=0.0, events=None, energy=None):
        """events: optional three (start, stop) torch.cuda.Event pairs recorded around the element launches (first interior
        half, border, second interior half: bench.py); a single pair times the first launch only.
        energy: the PCG's device scalar block S; if the backend can, S[1] receives this rank's share of <X, A X> from the
        element kernels (alpha = 1, beta = 0, one column) and self.energy_fused says whether it did."""
        be, plan = self.backend, self.plan
        nc = X.shape[0]
        self.energy_fused = False
        arm = energy is not None and nc == 1 and alpha == 1.0 and beta == 0.0 and hasattr(be, "energy_begin")
        if arm:
            be.energy_begin(energy)
        b = self._buffers(nc, X)
        xg, yg = b["xg"], b["yg"]
        be.scale(Y, beta)
        yg.zero_()
        # ---- import: owner -> sharer
        sends, recvs = [], []
        for (nb, idx), sbuf in zip(plan.sharers, b["send"]):
            be.pack_rows(X, idx, sbuf)
            sends.append((nb, sbuf))
        for (nb, g0, g1), stage in zip(plan.owners, b["stage"]):
            recvs.append((nb, xg[:, g0:g1] if nc == 1 else stage))  # one column: straight into the ghost slab
        reqs = self.transport.post(sends, recvs)
        ev = None if events is None else (events if isinstance(events[0], (tuple, list)) else [events])
        if ev:
            ev[0][0].record()
        be.apply_elems(3, X, None, Y, None, alpha, beta)  # first half of the interior: overlaps the import
        if ev:
            ev[0][1].record()
        self.transport.wait(reqs)
        if nc > 1:
            for (nb, g0, g1), stage in zip(plan.owners, b["stage"]):
                xg[:, g0:g1].copy_(stage)
        # ---- border elements, then export: sharer -> owner
        if ev and len(ev) > 1:
            ev[1][0].record()
        be.apply_elems(1, X, xg, Y, yg, alpha, beta)
        if ev and len(ev) > 1:
            ev[1][1].record()
        sends, recvs = [], []
        for (nb, g0, g1), stage in zip(plan.owners, b["stage"]):
            if nc == 1:
                src = yg[:, g0:g1]
            else:
                stage.copy_(yg[:, g0:g1])
                src = stage
            sends.append((nb, src))
        for (nb, idx), rbuf in zip(plan.sharers, b["recv"]):
            recvs.append((nb, rbuf))
        reqs = self.transport.post(sends, recvs)
        if ev and len(ev) > 2:
            ev[2][0].record()
        be.apply_elems(4, X, None, Y, None, alpha, beta)  # second half of the interior: overlaps the export
        if ev and len(ev) > 2:
            ev[2][1].record()
        self.transport.wait(reqs)
        for (nb, idx), rbuf in zip(plan.sharers, b["recv"]):
            be.unpack_add_rows(rbuf, idx, Y)
        be.dirichlet_rows(X, Y, alpha)
        if arm:
            self.energy_fused = be.energy_end(X)
        return Y

    def import_ghosts(self, V):
        """comm::Import of a multivector over the owned rows: returns the (ncols, n_ghost_dofs) ghost rows (owner ->
        sharer copy, comm/ImportExport.hpp:295-372)."""
        be, plan = self.backend, self.plan
        nc = V.shape[0]
        ghost = torch.zeros((nc, max(plan.n_ghost_dofs, 1)), dtype=torch.float64, device=V.device)
        sends, recvs, stages = [], [], []
        for nb, idx in plan.sharers:
            sbuf = torch.empty((nc, idx.numel()), dtype=torch.float64, device=V.device)
            be.pack_rows(V, idx, sbuf)
            sends.append((nb, sbuf))
        for nb, g0, g1 in plan.owners:
            stage = torch.empty((nc, g1 - g0), dtype=torch.float64, device=V.device)
            stages.append((g0, g1, stage))
            recvs.append((nb, stage))
        self.transport.wait(self.transport.post(sends, recvs))
        for g0, g1, stage in stages:
            ghost[:, g0:g1].copy_(stage)
        return ghost

    def export_add(self, ghost, owned):
        """comm::Export: adds every ghost row into its owner's row (sharer -> owner, comm/ImportExport.hpp:402-470)."""
        be, plan = self.backend, self.plan
        nc = owned.shape[0]
        sends, recvs = [], []
        for nb, g0, g1 in plan.owners:
            sends.append((nb, ghost[:, g0:g1].contiguous()))
        for nb, idx in plan.sharers:
            recvs.append((nb, torch.empty((nc, idx.numel()), dtype=torch.float64, device=owned.device)))
        self.transport.wait(self.transport.post(sends, recvs))
        for (nb, idx), (_, rbuf) in zip(plan.sharers, recvs):
            be.unpack_add_rows(rbuf, idx, owned)

    def diag_rhs(self, dirichlet_vals_owned=None):
        """computeDiagAndRhs of a partitioned system (algsys/MatrixFreeSystem.hpp:888-941): Dirichlet values imported to
        the ghost rows, local diag / rhs of interior and border elements, ghost rows exported and added to their
        owners, Dirichlet rows finalised.  dirichlet_vals_owned: (n_rhs, n_owned_dofs) or None.
        Returns (diag [n_owned], rhs (n_rhs, n_owned))."""
        be = self.backend
        n_owned, n_ghost = be.mesh.n_owned_dofs, be.mesh.n_ghost_dofs
        g_all = None
        if dirichlet_vals_owned is not None:
            g_all = torch.cat([dirichlet_vals_owned, self.import_ghosts(dirichlet_vals_owned)[:, :n_ghost]], dim=1).contiguous()
        dev = "cuda"
        diag = torch.zeros(n_owned, dtype=torch.float64, device=dev)
        rhs = torch.zeros((be.n_rhs, n_owned), dtype=torch.float64, device=dev)
        dg = torch.zeros(max(n_ghost, 1), dtype=torch.float64, device=dev)
        rg = torch.zeros((be.n_rhs, max(n_ghost, 1)), dtype=torch.float64, device=dev)
        be.diag_rhs(g_all, which=2, diag=diag, rhs=rhs, diag_ghost=dg, rhs_ghost=rg, finalize=False)
        both_g = torch.cat([dg[None, :], rg], dim=0).contiguous()   # one exchange for diag and rhs
        both_o = torch.cat([diag[None, :], rhs], dim=0).contiguous()
        self.export_add(both_g, both_o)
        diag, rhs = both_o[0].contiguous(), both_o[1:].contiguous()
        be.dirichlet_finalize(dirichlet_vals_owned, diag, rhs)
        return diag, rhs


class InprocGroup:
    """The mailboxes of the library's in-process transport (l3k_inproc_group_*): the ranks are threads of this process, each
    with its own context and stream, on one GPU or on several."""

    def __init__(self, world):
        import ctypes as C
        from . import capi
        self._g = C.c_void_p()
        capi.check(capi.load().l3k_inproc_group_create(world, C.byref(self._g)))
        self.world = world

    def table(self, rank):
        import ctypes as C
        from . import capi
        t = capi.HaloTransport()
        capi.check(capi.load().l3k_inproc_transport(self._g, rank, C.byref(t)))
        return t

    def __del__(self):
        try:
            from . import capi
            if getattr(self, "_g", None):
                capi.load().l3k_inproc_group_destroy(self._g)
                self._g = None
        except Exception:  # interpreter shutdown
            pass


class NativeHalo:
    """The exchange lists of one rank handed to the library, which runs the neighbour exchange itself through RCCL on a
    stream of its own (l3k_halo_*, include/l3k.h): what a C++ host of the reference would use instead of its MPI
    Import / Export.  unique_id: the 128 bytes of l3k_halo_unique_id from one rank (None: rank 0 draws it and it is
    broadcast over torch.distributed, which must then be initialised when world > 1)."""

    @staticmethod
    def check_local(part):
        """What can fail on ONE rank before any collective of the set-up runs, checked without communication: RCCL loads
        (l3k_halo_unique_id dlopens it) and the ghost ranges are contiguous in neighbour order.  Callers agree on the result
        (all-reduce MIN) before any of them constructs a NativeHalo, so that no rank falls out of the collectives."""
        import ctypes as C
        from . import capi
        capi.check(capi.load().l3k_halo_unique_id(C.create_string_buffer(128)))
        for (g0, _), prev in zip(part.ghost_ranges[1:], part.ghost_ranges[:-1]):
            if g0 != prev[1]:
                raise ValueError("ghost ranges must be contiguous in neighbour order")

    def __init__(self, ctx, part, dofs_per_node, rank=0, world=1, unique_id=None, group=None, transport=None):
        """transport: None (RCCL), an InprocGroup (ranks are threads of this process), or a capi.HaloTransport table."""
        import ctypes as C
        from . import capi
        lib = capi.load()
        if transport is None and unique_id is None:
            buf = C.create_string_buffer(128)
            if rank == 0:
                capi.check(lib.l3k_halo_unique_id(buf))
            if world > 1:
                dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
                t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(dev)
                dist.broadcast(t, 0, group=group)
                buf = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
            unique_id = buf.raw
        nn = len(part.nbr_rank)
        so = np.concatenate([[0], np.cumsum([len(x) for x in part.send_nodes])]).astype(np.int64)
        sn = np.concatenate([np.asarray(x, np.int32) for x in part.send_nodes]) if nn else np.zeros(0, np.int32)
        go = np.array([part.ghost_ranges[0][0]] + [g1 for _, g1 in part.ghost_ranges], dtype=np.int64) if nn else np.zeros(1, np.int64)
        for (g0, _), prev in zip(part.ghost_ranges[1:], part.ghost_ranges[:-1]):
            if g0 != prev[1]:
                raise ValueError("ghost ranges must be contiguous in neighbour order")
        nr = np.asarray(part.nbr_rank, dtype=np.int32)
        self._h = C.c_void_p()
        lists = (nn, nr.ctypes.data_as(capi.c_int_p), so.ctypes.data_as(capi.c_int64_p), sn.ctypes.data_as(capi.c_int32_p),
                 go.ctypes.data_as(capi.c_int64_p), C.byref(self._h))
        if transport is None:
            capi.check(lib.l3k_halo_create(ctx._h, unique_id, rank, world, dofs_per_node, *lists))
        else:
            table = transport.table(rank) if isinstance(transport, InprocGroup) else transport
            self._transport = transport  # (the group / the callbacks must outlive the halo)
            capi.check(lib.l3k_halo_create_transport(ctx._h, C.byref(table), rank, world, dofs_per_node, *lists))
        self.ctx = ctx

    def __del__(self):
        try:
            from . import capi
            if getattr(self, "_h", None):
                capi.load().l3k_halo_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass


class NativeDistributedOperator:
    """y <- alpha A x + beta y of a partitioned system with the exchange inside the library (l3k_mf_apply_dist)."""

    def __init__(self, mf, halo):
        self.mf, self.halo = mf, halo

    def apply(self, X, Y, alpha=1.0, beta=0.0):
        import ctypes as C
        from . import capi
        vp = lambda t: C.c_void_p(t.data_ptr())
        capi.check(capi.load().l3k_mf_apply_dist(self.mf._h, self.halo._h, vp(X), X.stride(0), vp(Y), Y.stride(0), X.shape[0], alpha, beta))
        return Y

    def timing_begin(self, n_applies):
        from . import capi
        capi.check(capi.load().l3k_halo_timing_begin(self.halo._h, n_applies))

    def timing_get(self, apply):
        """(ms of the first interior half, the border elements, the second interior half) of a timed apply"""
        import ctypes as C
        from . import capi
        ms = (C.c_double * 3)()
        capi.check(capi.load().l3k_halo_timing_get(self.halo._h, apply, ms))
        return tuple(ms)

    def import_ghosts(self, V):
        import ctypes as C
        from . import capi
        lib = capi.load()
        ng = max(int(lib.l3k_halo_n_ghost_dofs(self.halo._h)), 1)
        ghost = torch.zeros((V.shape[0], ng), dtype=torch.float64, device=V.device)
        capi.check(lib.l3k_halo_import(self.halo._h, C.c_void_p(V.data_ptr()), V.stride(0), V.shape[0], C.c_void_p(ghost.data_ptr()), ng))
        return ghost

    def export_add(self, ghost, owned):
        import ctypes as C
        from . import capi
        capi.check(capi.load().l3k_halo_export_add(self.halo._h, C.c_void_p(ghost.data_ptr()), ghost.stride(0), owned.shape[0],
                                                   C.c_void_p(owned.data_ptr()), owned.stride(0)))
