"""First-contact self-checks of the hand-written exchanges (in-kernel halo loads, push all-gather, peer-store
all-reduce; ``include/cask_hip_p2p.h``).

None of these paths has a collective library underneath: their ordering and visibility across xGMI rest on the
engine's own flags, scopes and fences.  Before a multi-GPU run trusts one of them it is exercised ``n`` (>= 50) times
with an operand that CHANGES EVERY EXCHANGE and whose value every rank can recompute from a formula, so a stale line,
a torn granule or a late flag shows up as a wrong value on some rank; the outcome is agreed collectively (every rank
falls back or none) and reported as ``{path: "ok" | "fell back: <why>"}``.  The caller owns the fallback (RCCL
all-gather / all-reduce, the halo pull).

Fault injection (tests, dry runs): ``CASK_FAULT_STALE_HALO=1`` makes a rank serve the PREVIOUS exchange's operand on
every odd exchange -- what a missed fence or a stale cache line looks like to its peers -- so the fallback can be
proven without broken hardware.  ``CASK_FAULT_STALE_HALO=<path>`` (``halo``, ``push``, ``allreduce``) restricts it to a
path, ``CASK_FAULT_RANK=<g>`` to ONE rank: the realistic case -- a halo fault is seen by the readers of one slice only,
so one rank fails while its peers pass.

Every check runs ALL ``n`` exchanges on every rank whatever it has already seen (r5): the fences inside the
loops are collectives, and a rank that left its loop at the first mismatch would enter the verdict's collectives while
its peers still issue the remaining barriers -- mismatched collectives exactly where the fallback should engage.  The
first failure is recorded, the protocol is completed, the verdict comes afterwards.

The reference has no multi-device code; BASELINE.json configs[3], [4] are this build's own.
"""
from __future__ import annotations

import os

N_EXCHANGES = max(4, int(os.environ.get("CASK_SELFCHECK_EXCHANGES", "50") or 50))   # (tests of the fallback PROTOCOL ask for fewer)


def _fault(path: str, rank=None) -> bool:
    v = os.environ.get("CASK_FAULT_STALE_HALO", "")
    if v not in ("1", "all", path):
        return False
    only = os.environ.get("CASK_FAULT_RANK")
    if only is None or only == "":
        return True
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    return int(only) == int(rank)


def operand(e: int, idx, n_global: int):
    """The value of x[idx] in exchange ``e`` (``idx``: int64 device tensor of GLOBAL indices): differs from one
    exchange to the next in every entry, identical bits on every rank (one elementwise expression)."""
    return (idx.double() * 0.25 + float((7 * e) % 13)) / float(max(n_global, 1)) + float(e)


def agree(ok: bool, why, all_reduce_min, gather_objects=None):
    """Collective verdict: (ok on every rank, first reason)."""
    all_ok = all_reduce_min(1.0 if ok else 0.0) > 0.5
    if all_ok:
        return True, None
    reason = why or "a peer failed"
    if gather_objects is not None:                           # every failing rank's reason: a diagnosis, not just a verdict
        reasons = gather_objects(None if ok else (why or "mismatch"))
        failing = [f"rank {g}: {r}" for g, r in enumerate(reasons) if r]
        if failing:
            reason = "; ".join(failing[:8])
    return False, reason


def check_fused_halo(torch, product_fused, product_plain_refs, x_shared_local, idx_own, fence, n_global,
                     n=N_EXCHANGES, rank=None):
    """In-kernel halo loads.  ``product_plain_refs(e) -> y_ref`` is the product computed from a private, formula-built
    operand (no remote access; computed by the caller BEFORE the halo sources were attached, or by an unattached
    handle); ``product_fused(y)`` runs the attached kernel on the shared slices.  ``x_shared_local`` is this rank's
    slice inside the shared allocation, ``idx_own`` its global indices.  ``fence`` is a collective: every rank calls it
    2 n times, whatever it has seen.  Returns (ok, why) -- the FIRST failure of this rank."""
    y = None
    prev = None
    why = None
    for e in range(n):
        val = operand(e, idx_own, n_global)
        if _fault("halo", rank) and e % 2 == 1 and prev is not None:
            val = prev                                       # injected fault: the peers see the previous operand
        x_shared_local.copy_(val)
        prev = val
        fence()                                              # every slice is final before anyone loads from it
        try:
            y_ref = product_plain_refs(e)
            if y is None:
                y = torch.empty_like(y_ref)
            product_fused(y)
            if not bool(torch.equal(y, y_ref)) and why is None:
                bad = int((y != y_ref).sum())
                why = f"in-kernel halo: {bad} rows differ from the formula-built product in exchange {e}"
        except Exception as exc:  # noqa: BLE001 - a failing product must not take this rank out of the fences
            if why is None:
                why = f"in-kernel halo: {exc!r} in exchange {e}"
        fence()                                              # ... and nobody overwrites while a peer still loads
    return why is None, why


def check_push_allgather(torch, push, n_local, idx_own, idx_all_padded, valid_padded, n_global, n=N_EXCHANGES, rank=None):
    """Push all-gather: the gathered vector of every exchange against the formula, every entry, every rank.
    ``idx_all_padded`` / ``valid_padded``: global index and validity of every entry of the padded gathered vector.
    All ``n`` exchanges are issued on every rank (a rank that stopped pushing would leave its peers polling until their
    limit on every remaining exchange)."""
    prev = None
    why = None
    for e in range(n):
        val = operand(e, idx_own, n_global)
        if _fault("push", rank) and e % 2 == 1 and prev is not None:
            val = prev
        prev = val
        try:
            slot = push.own_slot()
            slot[:n_local].copy_(val)
            xf = push.allgather(slot)
            want_all = torch.where(valid_padded, operand(e, idx_all_padded, n_global), torch.zeros_like(xf))
            got = torch.where(valid_padded, xf, torch.zeros_like(xf))
            if not bool(torch.equal(got, want_all)) and why is None:
                bad = int((got != want_all).sum())
                why = f"push all-gather: {bad} entries differ from the formula in exchange {e}"
        except Exception as exc:  # noqa: BLE001
            if why is None:
                why = f"push all-gather: {exc!r} in exchange {e}"
    try:
        push.check()
    except Exception as exc:  # noqa: BLE001 - a poll that timed out
        if why is None:
            why = f"push all-gather: {exc!r}"
    return why is None, why


def check_push_allreduce(torch, push, rank, world, device, n=N_EXCHANGES):
    """Peer-store all-reduce of 1..4 scalars: every reduction against the rank-order sum of the formula; all ``n``
    reductions on every rank."""
    def contrib(g, e, j):
        return (g + 1) * 0.125 + e * 1.0009765625 + j * 3.0 + ((5 * e + g) % 7) * 0.0625

    why = None
    for e in range(n):
        count = 1 + e % 4
        sent = [contrib(rank, e, j) for j in range(count)]
        if _fault("allreduce", rank) and e % 2 == 1:
            sent = [contrib(rank, e - 1, j) for j in range(count)]   # injected fault: the previous reduction's values
        try:
            t = torch.tensor(sent, dtype=torch.float64, device=device)
            push.allreduce(t)
            want = []
            for j in range(count):
                s = 0.0
                for g in range(world):                           # rank order: the order the kernel adds in
                    s += contrib(g, e, j)
                want.append(s)
            got = t.cpu().tolist()
            if got != want and why is None:
                why = f"peer-store all-reduce: {got} != {want} in reduction {e}"
        except Exception as exc:  # noqa: BLE001
            if why is None:
                why = f"peer-store all-reduce: {exc!r} in reduction {e}"
    try:
        push.check()
    except Exception as exc:  # noqa: BLE001
        if why is None:
            why = f"peer-store all-reduce: {exc!r}"
    return why is None, why
