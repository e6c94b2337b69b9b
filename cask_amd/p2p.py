"""Peer-to-peer exchange step of the row-sharded SpMV (``include/cask_hip_p2p.h``).

One process per GPU.  Rank g owns rows/columns [b_g, b_{g+1}) and keeps its
slice of x at the front of an *extended* x vector that lives in a shared
device allocation:

    x_ext = [ x_local (n_local, readable by every peer) | halo (n_halo, private) ]

The halo holds the remote x entries this rank's block references (sorted by
global column).  The local block is stored with EXTENDED column indices (own
column c -> c - b_g, remote column -> n_local + rank in the halo), so the
ordinary single-GPU kernel computes y_local = A_g @ x_ext unchanged.  Per
product the halo is refreshed by ``cask_hip_halo_pull_device``: remote loads
over xGMI straight from the owners' slices -- no collective and no host on the
data path, and legal inside a HIP graph.  For banded / stencil matrices the
halo is a few hundred entries instead of the 8*n bytes of an all-gather.
``PeerExchange.attach(matrix)`` goes one step further and hands the address
table to the product kernel (``cask_hip_csr_set_halo_sources``): the workgroups
at a seam stage their x window with remote loads themselves, and a sharded
product is a single launch.

The caller orders accesses across ranks (a peer must have finished writing its
slice before it is pulled, and must not overwrite it while someone pulls):
``PeerExchange.fence()`` is a 1-element all-reduce; solvers get the same
ordering from their dot-product all-reduces.

The reference has no multi-device code; the closest notions are
the per-pipe row split (src/runtime/Spmv.cpp:334-364) and the per-controller
memory banks that ``Spmv_<id>_dramWrite`` addresses (Spmv.cpp:109-140).
"""
from __future__ import annotations

import ctypes
from ctypes import c_int64, c_void_p

import numpy as np

from . import capi

HANDLE_BYTES = 64
P2P_SYMBOLS = ("cask_hip_shared_alloc", "cask_hip_shared_free", "cask_hip_shared_open", "cask_hip_shared_close",
               "cask_hip_copy_to_device", "cask_hip_copy_to_host", "cask_hip_halo_pull_device",
               "cask_hip_csr_set_halo_sources", "cask_hip_push_create", "cask_hip_push_destroy",
               "cask_hip_push_allgather", "cask_hip_push_check", "cask_hip_push_allreduce", "cask_hip_push_own_slot")


def _lib():
    L = capi.load()
    if not getattr(L, "_p2p_bound", False):
        vp, i64 = c_void_p, c_int64
        L.cask_hip_shared_alloc.argtypes = [i64, ctypes.POINTER(vp), vp]
        L.cask_hip_shared_free.argtypes = [vp]
        L.cask_hip_shared_open.argtypes = [vp, ctypes.POINTER(vp)]
        L.cask_hip_shared_close.argtypes = [vp]
        L.cask_hip_copy_to_device.argtypes = [vp, vp, i64]
        L.cask_hip_copy_to_host.argtypes = [vp, vp, i64]
        L.cask_hip_halo_pull_device.argtypes = [i64, vp, vp, vp]
        L.cask_hip_csr_set_halo_sources.argtypes = [vp, ctypes.c_int32, vp]
        L.cask_hip_push_create.argtypes = [ctypes.c_int32, ctypes.c_int32, i64, vp, vp, ctypes.POINTER(vp)]
        L.cask_hip_push_destroy.argtypes = [vp]
        L.cask_hip_push_allgather.argtypes = [vp, vp, ctypes.POINTER(vp), vp]
        L.cask_hip_push_check.argtypes = [vp]
        L.cask_hip_push_allreduce.argtypes = [vp, ctypes.c_int32, vp, vp]
        L.cask_hip_push_own_slot.argtypes = [vp, ctypes.POINTER(vp)]
        for s in P2P_SYMBOLS:
            getattr(L, s).restype = ctypes.c_int
        L._p2p_bound = True
    return L


# ------------------------------------------------------------------ host-side plan (pure numpy)
def plan_halo(col_ind, bounds, rank: int):
    """Split a row block's GLOBAL column indices into own and remote columns.

    Returns (ci_ext, halo_cols, halo_owner, halo_index):
      ci_ext      int32  extended indices (own: c - b_rank; remote: n_local + position in halo_cols)
      halo_cols   int64  sorted distinct remote global columns
      halo_owner  int32  owning rank of each halo column
      halo_index  int64  index of each halo column inside its owner's slice
    """
    col_ind = np.asarray(col_ind)
    bounds = np.asarray(bounds, dtype=np.int64)
    b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
    n_local = b1 - b0
    own = (col_ind >= b0) & (col_ind < b1)
    halo_cols = np.unique(col_ind[~own]).astype(np.int64)
    ci_ext = np.empty(col_ind.size, dtype=np.int32)
    ci_ext[own] = (col_ind[own] - b0).astype(np.int32)
    ci_ext[~own] = (n_local + np.searchsorted(halo_cols, col_ind[~own])).astype(np.int32)
    halo_owner = (np.searchsorted(bounds, halo_cols, side="right") - 1).astype(np.int32)
    if halo_cols.size and (halo_owner.min() < 0 or halo_owner.max() >= bounds.size - 1):
        raise ValueError("column index outside the partition")
    halo_index = halo_cols - bounds[halo_owner]
    return ci_ext, halo_cols, halo_owner, halo_index


def halo_addresses(halo_owner, halo_index, bases):
    """Absolute device address of every halo entry: bases[owner] + 8*index (uint64 values in an int64 array)."""
    bases = np.asarray([int(b) for b in bases], dtype=np.uint64)
    addr = bases[np.asarray(halo_owner, dtype=np.int64)] + np.asarray(halo_index, dtype=np.uint64) * np.uint64(8)
    return addr.view(np.int64)


# ------------------------------------------------------------------ device objects
class _RawDeviceArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor."""

    def __init__(self, ptr: int, n: int, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class SharedVector:
    """n doubles in a shared device allocation on the current device (zero-initialised)."""

    def __init__(self, n: int):
        self.n = int(n)
        p = c_void_p()
        h = (ctypes.c_ubyte * HANDLE_BYTES)()
        capi._check(_lib().cask_hip_shared_alloc(max(self.n, 1) * 8, ctypes.byref(p), h))
        self.ptr = int(p.value)
        self.handle = bytes(h)
        self._tensor = None

    def tensor(self, device):
        """torch view of the allocation (no copy)."""
        if self._tensor is None:
            import torch
            self._tensor = torch.as_tensor(_RawDeviceArray(self.ptr, max(self.n, 1)), device=device)[: self.n]
            if self._tensor.data_ptr() != self.ptr:
                raise capi.CaskHipError("torch copied the shared allocation instead of viewing it")
        return self._tensor

    def write(self, values, offset=0):
        a = np.ascontiguousarray(values, dtype=np.float64)
        if offset < 0 or offset + a.size > self.n:
            raise ValueError("write outside the shared vector")
        capi._check(_lib().cask_hip_copy_to_device(c_void_p(self.ptr + 8 * offset), a.ctypes.data_as(c_void_p), a.size * 8))

    def read(self, offset=0, count=None):
        count = self.n - offset if count is None else count
        out = np.empty(count, dtype=np.float64)
        capi._check(_lib().cask_hip_copy_to_host(out.ctypes.data_as(c_void_p), c_void_p(self.ptr + 8 * offset), count * 8))
        return out

    def free(self):
        if self.ptr:
            self._tensor = None
            _lib().cask_hip_shared_free(c_void_p(self.ptr))
            self.ptr = 0


def open_peer(handle: bytes) -> int:
    p = c_void_p()
    buf = (ctypes.c_ubyte * HANDLE_BYTES).from_buffer_copy(handle)
    capi._check(_lib().cask_hip_shared_open(buf, ctypes.byref(p)))
    return int(p.value)


def close_peer(ptr: int):
    if ptr:
        _lib().cask_hip_shared_close(c_void_p(ptr))


def peek(ptr: int, index: int = 0) -> float:
    """One double of a (peer) allocation read through the runtime, not a kernel."""
    out = np.empty(1, dtype=np.float64)
    capi._check(_lib().cask_hip_copy_to_host(out.ctypes.data_as(c_void_p), c_void_p(ptr + 8 * index), 8))
    return float(out[0])


class PeerExchange:
    """Shared x slice + halo of one rank, and the pull that refreshes the halo.

    ``exchange_objects(obj) -> list`` is the control-plane all-gather of python objects
    (``torch.distributed.all_gather_object`` bound to a group); ``fence_fn()`` orders
    device work across ranks.  Construction is collective: every rank must call it.
    Raises on every rank or on none (the outcome is agreed through ``exchange_objects``).
    """

    def __init__(self, bounds, rank, world, halo_owner, halo_index, device, exchange_objects, fence_fn=None,
                 n_slots=1):
        """``n_slots`` > 1 (row-sharded solvers, ``cask_hip_solve_device``): the allocation holds that many
        vectors, slot k at ``k * stride`` doubles, with the SAME stride on every rank (the longest slice,
        rounded up to 32 doubles); the address table points at slot 0 and a product on the vector in slot
        k adds ``k * stride * 8`` bytes.  The private halo copy behind the slots exists only for
        ``n_slots == 1`` (the pull mode needs [own | halo] contiguous)."""
        import torch
        self.rank, self.world, self.device = rank, world, device
        self.n_local = int(bounds[rank + 1] - bounds[rank])
        self.n_halo = int(len(halo_owner))
        self.fence_fn = fence_fn
        self.peers = {}
        self.shared = None
        self.n_slots = int(n_slots)
        longest = max(int(bounds[g + 1] - bounds[g]) for g in range(world))
        self.stride = self.n_local if self.n_slots == 1 else max(32, -(-longest // 32) * 32)
        err = None
        try:
            self.shared = SharedVector(self.n_local + self.n_halo if self.n_slots == 1
                                       else self.n_slots * self.stride)
            # signature in element 0 so that a peer can tell a good mapping from a stale one
            self.shared.write([rank + 0.5])
        except Exception as e:  # noqa: BLE001 - reported to every rank below
            err = repr(e)
        infos = exchange_objects({"handle": self.shared.handle if self.shared else None, "error": err})
        self._handles = [i["handle"] for i in infos]
        err = next((f"rank {g}: {i['error']}" for g, i in enumerate(infos) if i["error"]), None)
        if err is None:
            try:
                self._open_owners(halo_owner, check_signature=True)
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        oks = exchange_objects(err)                      # also: nobody clears its signature before all have looked
        bad = [f"rank {g}: {e}" for g, e in enumerate(oks) if e]
        if bad:
            self.close()
            raise capi.CaskHipError("peer-to-peer setup failed (" + "; ".join(bad) + ")")
        self.shared.write([0.0])
        self.addr = self.address_table(halo_owner, halo_index)
        self.x_ext = self.shared.tensor(device)
        self.x_local = self.x_ext[: self.n_local]
        self._halo_ptr = self.shared.ptr + 8 * self.n_local

    def _open_owners(self, halo_owner, check_signature=False):
        for g in sorted(set(int(o) for o in np.unique(halo_owner))):
            if g == self.rank:
                raise ValueError("halo column owned by this rank")
            if g in self.peers:
                continue
            self.peers[g] = open_peer(self._handles[g])
            if check_signature and peek(self.peers[g]) != g + 0.5:
                raise capi.CaskHipError(f"mapping of rank {g}'s slice does not show its signature")

    def address_table(self, halo_owner, halo_index):
        """Device table of the absolute addresses (slot 0 of the owners' allocations) of a halo list; maps
        owners that are not mapped yet.  A second operator over the same vectors (the A^T block of a sharded
        BiCG) gets its table here -- not collective: the handles were exchanged at construction."""
        import torch
        self._open_owners(halo_owner)
        bases = [self.peers.get(g, 0) for g in range(self.world)]
        addr = halo_addresses(halo_owner, halo_index, bases) if len(halo_owner) else np.zeros(0, dtype=np.int64)
        return torch.from_numpy(addr).to(self.device)

    def slot(self, k):
        """torch view of the vector in slot k (n_local entries)."""
        full = self.shared.tensor(self.device)
        return full[k * self.stride: k * self.stride + self.n_local]

    def pull(self, stream=None):
        """Refresh the halo from the owners' slices (asynchronous on ``stream``)."""
        if self.n_slots != 1:
            raise capi.CaskHipError("the halo pull needs the [own | halo] layout (n_slots == 1)")
        if self.n_halo:
            capi._check(_lib().cask_hip_halo_pull_device(self.n_halo, c_void_p(self.addr.data_ptr()),
                                                         c_void_p(self._halo_ptr), c_void_p(capi._stream_ptr(stream))))

    def attach(self, matrix):
        """Fold the exchange into ``matrix``'s product kernel (``cask_hip_csr_set_halo_sources``): its halo
        columns are then read straight from the owners' slices by the kernel, and ``pull`` is not needed
        for products of that handle.  ``matrix`` is a ``capi.CsrMatrix`` with the extended column layout."""
        if self.n_halo:
            matrix.set_halo_sources(self.n_local, self.addr)

    def fence(self):
        if self.fence_fn is not None:
            self.fence_fn()

    def close(self):
        for g, p in list(self.peers.items()):
            close_peer(p)
        self.peers = {}
        self.x_ext = self.x_local = None
        if self.shared is not None:
            self.shared.free()
            self.shared = None


class PushExchange:
    """The push all-gather of ``include/cask_hip_p2p.h``: every rank stores its slice into every peer's gathered
    vector over xGMI (one launch per exchange, no collective library on the data path).  Collective construction:
    each rank allocates one shared region -- two gathered vectors of ``world * stride`` doubles and a flag array --
    and opens every peer's.  Raises on every rank or on none (agreed through ``exchange_objects``).

        ex = PushExchange(rank, world, stride, device, exchange_objects)
        ex.x_slot[:n_local] = ...            # this rank's slice (stride doubles, tail zero)
        x_full = ex.allgather()              # the tensor the product that follows must read (alternates)
    """

    FLAG_DOUBLES = (2 * 64 * 4 + 2 * 64 * 4 * 16) // 8    # CASK_HIP_PUSH_FLAG_BYTES behind the vectors

    def __init__(self, rank, world, stride, device, exchange_objects):
        import torch
        self.rank, self.world, self.stride, self.device = rank, world, int(stride), device
        self.peers, self.shared, self._h = {}, None, None
        n_vec = world * self.stride
        err = None
        try:
            self.shared = SharedVector(2 * n_vec + self.FLAG_DOUBLES)
            self.shared.write([rank + 0.5])
        except Exception as e:  # noqa: BLE001 - reported to every rank below
            err = repr(e)
        infos = exchange_objects({"handle": self.shared.handle if self.shared else None, "error": err})
        err = next((f"rank {g}: {i['error']}" for g, i in enumerate(infos) if i["error"]), None)
        if err is None:
            try:
                for g in range(world):
                    if g == rank:
                        continue
                    self.peers[g] = open_peer(infos[g]["handle"])
                    if peek(self.peers[g]) != g + 0.5:
                        raise capi.CaskHipError(f"mapping of rank {g}'s region does not show its signature")
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        oks = exchange_objects(err)
        bad = [f"rank {g}: {e}" for g, e in enumerate(oks) if e]
        if bad:
            self.close()
            raise capi.CaskHipError("push all-gather setup failed (" + "; ".join(bad) + ")")
        # The signature sits in element 0 of this rank's region = rank 0's x[0] slot of gathered vector 0, a slot the
        # PEERS store to.  It is cleared here, and nobody may push before everybody has cleared (a fast rank
        # 0's first exchange could otherwise be overwritten by a slow rank's reset): a third control-plane round.
        self.shared.write([0.0])
        exchange_objects(None)
        bases = [self.shared.ptr if g == rank else self.peers[g] for g in range(world)]
        full = np.array([b for b in bases] + [b + 8 * n_vec for b in bases], dtype=np.uint64)
        flags = np.array([b + 16 * n_vec for b in bases], dtype=np.uint64)
        h = c_void_p()
        capi._check(_lib().cask_hip_push_create(rank, world, self.stride, full.ctypes.data_as(c_void_p),
                                                flags.ctypes.data_as(c_void_p), ctypes.byref(h)))
        self._h = h
        region = self.shared.tensor(device)
        self._full = [region[:n_vec], region[n_vec: 2 * n_vec]]
        self._ptr = {int(self._full[0].data_ptr()): 0, int(self._full[1].data_ptr()): 1}
        self.x_slot = torch.zeros(self.stride, dtype=torch.float64, device=device)

    def allgather(self, x_slot=None, stream=None):
        """One exchange (asynchronous on ``stream``); returns the gathered vector this rank's next product reads."""
        src = self.x_slot if x_slot is None else x_slot
        out = c_void_p()
        capi._check(_lib().cask_hip_push_allgather(self._h, c_void_p(src.data_ptr()), ctypes.byref(out),
                                                   c_void_p(capi._stream_ptr(stream))))
        return self._full[self._ptr[int(out.value)]]

    def own_slot(self):
        """This rank's slice inside the gathered vector the NEXT exchange fills (``stride`` doubles; alternates): write
        the slice there and pass it to ``allgather`` to skip the own copy."""
        out = c_void_p()
        capi._check(_lib().cask_hip_push_own_slot(self._h, ctypes.byref(out)))
        full = self._full[self._ptr[int(out.value) - 8 * self.rank * self.stride]]
        return full[self.rank * self.stride: (self.rank + 1) * self.stride]

    def allreduce(self, t, stream=None):
        """In-place sum of a 1..4-element float64 CUDA tensor over the ranks (one launch, rank-order sum)."""
        capi._check(_lib().cask_hip_push_allreduce(c_void_p(t.data_ptr()), t.numel(), c_void_p(capi._stream_ptr(stream)),
                                                   self._h))

    @property
    def handle(self):
        return self._h

    def check(self):
        """Synchronous: raises if a poll of an earlier exchange timed out."""
        capi._check(_lib().cask_hip_push_check(self._h))

    def close(self):
        if self._h:
            _lib().cask_hip_push_destroy(self._h)
            self._h = None
        for g, p in list(self.peers.items()):
            close_peer(p)
        self.peers = {}
        self._full = None
        if self.shared is not None:
            self.shared.free()
            self.shared = None
