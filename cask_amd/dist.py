"""Row-sharded SpMV / CG across the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Plumbing only -- the arithmetic is the C-ABI engine.

The reference has no multi-device code at all; its only
precedent is the per-pipe contiguous row split of Spmv::preprocess
(src/runtime/Spmv.cpp:334-364), which splits by ROW COUNT.  Here blocks are
nnz-balanced, each rank owns rows [r_g, r_{g+1}) with global column indices,
the matching slice of every vector, and one exchange per product:

    x_full = all_gather(x_local)          (RCCL; 8*n bytes in total)
    y_local = A_g @ x_full                (local HIP kernel)

Dot products are a local two-stage reduction plus a one-element all_reduce.

``exchange="p2p"`` replaces the all-gather by the peer-to-peer halo pull of
``cask_amd/p2p.py`` (remote loads over xGMI from shared x slices, only the
entries the block references): no collective on the data path.
"""
from __future__ import annotations

import numpy as np


def partition_rows_by_nnz(row_ptr, world: int):
    """Contiguous row blocks with ~equal nnz (+1 per row so empty rows spread too).
    Returns world+1 boundaries; deterministic, identical on every rank."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n = row_ptr.size - 1
    work = row_ptr + np.arange(n + 1, dtype=np.int64)       # merge-path diagonal: nnz + rows
    total = work[-1]
    bounds = [0]
    for g in range(1, world):
        target = total * g // world
        r = int(np.searchsorted(work, target, side="left"))
        bounds.append(min(max(r, bounds[-1]), n))
    bounds.append(n)
    return bounds


def partition_rows_even(n: int, world: int):
    """Equal row counts (the reference's rule, remainder to the last block; Spmv.cpp:334,353-364)."""
    per = n // world
    return [g * per for g in range(world)] + [n]


def slice_rows(row_ptr, col_ind, values, r0: int, r1: int):
    row_ptr = np.asarray(row_ptr)
    k0, k1 = int(row_ptr[r0]), int(row_ptr[r1])
    return (row_ptr[r0:r1 + 1] - k0).astype(np.int32), np.asarray(col_ind[k0:k1], dtype=np.int32), \
        np.asarray(values[k0:k1], dtype=np.float64)


class ShardedSpmv:
    """y_local = A[rows of this rank, :] @ all_gather(x_local).

    ``local_product(x_full_tensor, y_local_tensor)`` performs the local block
    product; on GPUs it is ``capi.CsrMatrix.spmv_device`` (built by
    ``from_global``), in the gloo CPU tests the test injects a checker.
    """

    def __init__(self, bounds, rank: int, world: int, local_product, device, group=None, exchange=None):
        import torch
        self.torch = torch
        self.exchange = exchange              # p2p.PeerExchange or None (= all-gather)
        self.bounds = list(bounds)
        self.rank, self.world = rank, world
        self.local_product = local_product
        self.device = device
        self.group = group
        self.n = self.bounds[-1]
        self.sizes = [self.bounds[g + 1] - self.bounds[g] for g in range(world)]
        self.n_local = self.sizes[rank]
        self.max_local = max(self.sizes) if self.sizes else 0
        self.even = all(s == self.max_local for s in self.sizes)
        self.x_full = torch.zeros(self.n, dtype=torch.float64, device=device)
        if not self.even:
            self._pad_in = torch.zeros(self.max_local, dtype=torch.float64, device=device)
            self._pad_out = torch.zeros(self.max_local * world, dtype=torch.float64, device=device)

    @classmethod
    def from_global(cls, row_ptr, col_ind, values, n_cols, rank, world, params=None, balance="nnz", group=None,
                    exchange="all_gather", fence=None, fused_halo=False):
        """Build this rank's block of a globally known CSR matrix on the current GPU.

        exchange="all_gather": block with global columns, x all-gathered per product.
        exchange="p2p": block with extended columns [own | halo], halo pulled from the peers' shared
        slices per product (collective construction; raises on every rank if any rank cannot map a peer).
        ``fence`` orders device work across ranks for the p2p path (default: a 1-element all-reduce).
        ``fused_halo`` (p2p only): no pull step -- the product kernel loads the halo from the peers."""
        import torch
        from . import capi
        n = len(row_ptr) - 1
        bounds = partition_rows_by_nnz(row_ptr, world) if balance == "nnz" else partition_rows_even(n, world)
        if n != n_cols:
            raise ValueError("row sharding of x needs a square matrix")
        rp, ci, va = slice_rows(row_ptr, col_ind, values, bounds[rank], bounds[rank + 1])
        dev = torch.device("cuda", torch.cuda.current_device())
        n_local = bounds[rank + 1] - bounds[rank]
        if exchange == "p2p":
            import torch.distributed as dist
            from . import p2p
            ci_ext, halo_cols, halo_owner, halo_index = p2p.plan_halo(ci, bounds, rank)

            def gather_objects(obj):
                out = [None] * world
                dist.all_gather_object(out, obj, group=group)
                return out

            if fence is None:
                token = torch.zeros(1, device=dev)

                def fence():
                    dist.all_reduce(token, group=group)
            ex = p2p.PeerExchange(bounds, rank, world, halo_owner, halo_index, dev, gather_objects, fence)
            mat = capi.CsrMatrix.from_host(n_local, n_local + ex.n_halo, rp, ci_ext, va, params)
            obj = cls(bounds, rank, world, lambda xe, yl: mat.spmv_device(xe, yl), dev, group, exchange=ex)
            obj.fused_halo = False
            if fused_halo and mat.params.as_dict()["variant"] == "merge" and mat.nnz >= 2:
                ex.attach(mat)                  # the product kernel reads the halo from the peers itself
                obj.fused_halo = True
        elif exchange == "all_gather":
            mat = capi.CsrMatrix.from_host(n_local, n_cols, rp, ci, va, params)
            obj = cls(bounds, rank, world, lambda xf, yl: mat.spmv_device(xf, yl), dev, group)
        else:
            raise ValueError(f"unknown exchange {exchange!r}")
        obj.matrix = mat
        return obj

    # -- exchange ---------------------------------------------------------------
    def gather_x(self, x_local):
        """all-gather of the x slices into self.x_full (uneven slices are padded to the longest)."""
        import torch.distributed as dist
        if self.world == 1:
            self.x_full.copy_(x_local)
            return self.x_full
        if self.even:
            dist.all_gather_into_tensor(self.x_full, x_local.contiguous(), group=self.group)
            return self.x_full
        self._pad_in[: self.n_local].copy_(x_local)
        dist.all_gather_into_tensor(self._pad_out, self._pad_in, group=self.group)
        for g in range(self.world):
            self.x_full[self.bounds[g]: self.bounds[g + 1]].copy_(
                self._pad_out[g * self.max_local: g * self.max_local + self.sizes[g]])
        return self.x_full

    def spmv(self, x_local, y_local=None, fence_before=True, fence_after=True):
        """y_local = A_g x.  With a peer-to-peer exchange two orderings across ranks are needed: every slice
        is final before anyone reads it (``fence_before``) and nobody overwrites its slice while a peer still
        reads it (``fence_after``).  A caller whose next step is a collective anyway (the solvers' dot
        products: an all-reduce completes only after every rank's product has run) passes
        ``fence_after=False`` and saves one latency-bound collective per product."""
        if y_local is None:
            y_local = self.torch.empty(self.n_local, dtype=self.torch.float64, device=self.device)
        ex = self.exchange
        if ex is not None:
            if x_local.data_ptr() != ex.x_local.data_ptr():
                ex.x_local.copy_(x_local)
            if fence_before:
                ex.fence()          # every slice is final ...
            if getattr(self, "fused_halo", False):
                self.local_product(ex.x_ext, y_local)    # remote loads happen inside the product kernel
            else:
                ex.pull()
            if fence_after:
                ex.fence()          # ... and nobody overwrites its slice while a peer still reads from it
            if not getattr(self, "fused_halo", False):
                self.local_product(ex.x_ext, y_local)
            return y_local
        xf = self.gather_x(x_local)
        self.local_product(xf, y_local)
        return y_local

    def close(self):
        """Collective: unmap the peers' slices, then free the own one."""
        ex, self.exchange = self.exchange, None
        if ex is not None:
            for g, p in list(ex.peers.items()):
                from . import p2p
                p2p.close_peer(p)
            ex.peers = {}
            ex.fence()
            ex.close()

    def dot(self, a_local, b_local):
        """Global dot product as a 1-element tensor on the device (no host sync)."""
        import torch.distributed as dist
        s = (a_local * b_local).sum().reshape(1)
        if self.world > 1:
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        return s

    def cg(self, b_local, x_local=None, maxiters=2000, tol=1e-5):
        """Distributed un-preconditioned CG with the recurrence, stopping rule and
        `iterations` convention of pcg (src/runtime/SparseLinearSolvers.hpp:162-239).
        Every rank sees the same all-reduced scalars, so all ranks stop in the same
        pass.  Returns (x_local, iterations, converged)."""
        torch = self.torch
        x = torch.zeros_like(b_local) if x_local is None else x_local.clone()
        r = b_local - self.spmv(x)                               # :189-190
        p = r.clone()
        rsold = self.dot(r, r)                                   # :198
        iterations, tol2 = 0, tol * tol
        for i in range(maxiters):
            Ap = self.spmv(p, fence_after=False)                 # :206 (the all-reduce of p.Ap orders the ranks)
            alpha = rsold / self.dot(p, Ap)                      # :208
            x = x + alpha * p                                    # :210
            r = r - alpha * Ap                                   # :212
            rsnew = self.dot(r, r)                               # :218
            if float(rsnew) <= tol2:                             # :220
                return x, iterations, True
            p = r + (rsnew / rsold) * p                          # :229
            rsold = rsnew
            iterations = i                                       # :231
        return x, iterations, False

    def bicg(self, transposed: "ShardedSpmv", b_local, x_local=None, maxiters=2000, tol=1e-5):
        """Distributed classical BiCG (BASELINE config 5): products with A (this operator) and A^T
        (``transposed``: the row-sharded transpose, see ``transpose_csr``), all-reduced dot products.
        Same recurrence, stopping rule (r.r <= tol^2) and `iterations` convention as the single-GPU
        ``cask_hip_bicg`` and the oracle; the reference only declares this solver
        (DfeBiCgSolver, src/runtime/SparseLinearSolvers.hpp:56-61).  Returns (x_local, iterations, converged)."""
        torch = self.torch
        x = torch.zeros_like(b_local) if x_local is None else x_local.clone()
        r = b_local - self.spmv(x)
        rt, p, pt = r.clone(), r.clone(), r.clone()
        rho = self.dot(rt, r)
        iterations, tol2 = 0, tol * tol
        for i in range(maxiters):
            q = self.spmv(p, fence_after=False)                  # each product publishes its operand and fences
            qt = transposed.spmv(pt, fence_after=False)          # before reading; the all-reduce of pt.q orders
            alpha = rho / self.dot(pt, q)                        # the reads against the updates below
            x.add_(alpha * p)
            r.sub_(alpha * q)
            rt.sub_(alpha * qt)
            rr = self.dot(r, r)
            if float(rr) <= tol2:
                return x, iterations, True
            rho_new = self.dot(rt, r)
            beta = rho_new / rho
            p = r + beta * p
            pt = rt + beta * pt
            rho = rho_new
            iterations = i
        return x, iterations, False


def transpose_csr(n_rows, n_cols, row_ptr, col_ind, values):
    """CSR of A^T by counting sort (rows of A^T come out with ascending columns); host arrays.
    The engine does the same for a single GPU (ensure_transpose in cask_hip.hip); a sharded BiCG needs
    the transpose BEFORE the rows are dealt to the ranks."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    col_ind = np.asarray(col_ind, dtype=np.int64)
    values = np.asarray(values, dtype=np.float64)
    counts = np.bincount(col_ind, minlength=n_cols)
    trp = np.zeros(n_cols + 1, dtype=np.int64)
    np.cumsum(counts, out=trp[1:])
    order = np.argsort(col_ind, kind="stable")                 # stable: original row order within a column
    rows_of = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(row_ptr))
    return trp.astype(np.int32), rows_of[order].astype(np.int32), values[order]

