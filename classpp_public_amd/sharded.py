"""Multi-GPU form of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the node; "gloo" on CPU for the tests).

  stage A  k-modes are independent units (pm.cpp:686-707): rank r integrates modes r, r+N, r+2N, ... (round-robin, so
           every rank gets the same mix of cheap low-k and expensive high-k modes).
  exchange 1  all_gather of the local source blocks [tp][ntau][nk_local] -> full sources on every rank.  This exchange
           is real: the transfer stage splines the sources across ALL k (tm.cpp:604-639).
  stage B  rank r computes Delta_l(q) for multipoles l[r::N] and every q (cost grows with l, round-robin balances it), and - the C_l integral
           over q needs every q of one l, which the rank holds - finishes the C_l rows of its multipoles (compute.cl).
  exchange 2  gather of the [nl_local][ct] C_l blocks on rank 0 (7 numbers per multipole), or - when a consumer wants the reference's
           transfer_ table (the C++ shim's TransferModule) - of the [tt][nl_local][nq] transfer blocks.
  rank 0   lensing and P(k) (delta_m(k, tau0) is complete on every rank after exchange 1).

Data volumes are tiny (sources 17 MB, C_l 6 KB for explanatory.ini), the collectives are latency-bound; exactly two per step, no ring
all-reduce.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n, rank, world):
    """round-robin shard of range(n)"""
    return np.arange(rank, n, world)


def densify_k(k, factor):
    """k grid with (factor-1) geometrically spaced points inserted in every interval: the weak-scaling workload
    (BASELINE config 3 asks for ~3000 modes k-sharded over 8 GPUs; the reference's cl_permille.pre that would
    produce such a grid does not exist)."""
    if factor <= 1:
        return np.ascontiguousarray(k, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    out = []
    for i in range(k.size - 1):
        r = (k[i + 1] / k[i]) ** (1.0 / factor)
        out.extend(k[i] * r ** np.arange(factor))
    out.append(k[-1])
    return np.ascontiguousarray(out, dtype=np.float64)


class _Plan:
    """index tensors and exchange buffers of one (nk, nl, world, shapes) geometry, built once and reused by every step"""

    def __init__(self, nk, nl, rank, world, device):
        self.key = (nk, nl, rank, world, str(device))
        self.k_idx = [torch.as_tensor(shard_indices(nk, r, world), device=device) for r in range(world)]
        self.l_idx = [torch.as_tensor(shard_indices(nl, r, world), device=device) for r in range(world)]
        self.bufs = {}

    def buf(self, name, shape, device, count=1):
        b = self.bufs.get(name)
        if b is None or b[0].shape != tuple(shape):
            b = [torch.zeros(shape, dtype=torch.float64, device=device) for _ in range(count)]
            self.bufs[name] = b
        return b


_plan = None


def sharded_step(compute, k_all, l_all, rank, world, device, k_size_cl=None, exchange_device=None, force_exchange=False, gather="transfer"):
    """One pass of the hot path over `world` ranks.
    gather = "transfer": rank 0 receives the full transfer table [tt][nl][nq]; "cl": every rank finishes the C_l rows of its multipoles
    (compute.cl(transfer_local) -> [nl_local][ct]) and rank 0 receives the C_l table [nl][ct].

    compute.perturb(k_subset) -> torch f64 [tp][ntau][len(k_subset)] on `device`
    compute.transfer(sources_full, k_all, l_subset, k_size_cl) -> torch f64 [tt][len(l_subset)][nq] on `device`
    Returns the full transfer table [tt][nl][nq] on rank 0 (None elsewhere) and the full sources.
    exchange_device: where the collectives run (default: `device`, i.e. RCCL on GPU tensors; torch.device("cpu") stages the two
    exchanges through host memory - the gloo rehearsal of bench.py --backend gloo on a box with fewer GPUs than ranks).
    force_exchange: run both collectives even in a group of one (the test of the RCCL code path that a one-GPU box can make).
    """
    global _plan
    xdev = device if exchange_device is None else exchange_device
    nk, nl = len(k_all), len(l_all)
    if _plan is None or _plan.key != (nk, nl, rank, world, str(device)):
        _plan = _Plan(nk, nl, rank, world, device)
    plan = _plan
    my_k = shard_indices(nk, rank, world)
    local = compute.perturb(k_all[my_k])
    ntp, ntau = local.shape[0], local.shape[1]
    if world == 1 and not force_exchange:
        full = local
    else:
        # ---- exchange 1: all_gather (pad to the largest shard so that every block has the same shape) ----
        nmax = (nk + world - 1) // world
        buf = plan.buf("src_send", (ntp, ntau, nmax), xdev)[0]
        buf[:, :, : my_k.size] = local
        blocks = plan.buf("src_recv", (ntp, ntau, nmax), xdev, world)
        dist.all_gather(blocks, buf)
        full = plan.buf("src_full", (ntp, ntau, nk), device)[0]
        for r in range(world):
            idx = plan.k_idx[r]
            full.index_copy_(2, idx, blocks[r][:, :, : idx.numel()].to(device))
    my_l = shard_indices(nl, rank, world)
    tr_local = compute.transfer(full, k_all, l_all[my_l], nk if k_size_cl is None else k_size_cl)
    if gather == "cl":
        tr_local = compute.cl(tr_local).unsqueeze(0)      # [1][nl_local][ct]: the same block shape as the transfer rows
    if world == 1 and not force_exchange:
        return (tr_local[0] if gather == "cl" else tr_local), full
    # ---- exchange 2: gather on rank 0 ----
    ntt, nq = tr_local.shape[0], tr_local.shape[2]
    lmax = (nl + world - 1) // world
    buf = plan.buf("tr_send", (ntt, lmax, nq), xdev)[0]
    buf[:, : my_l.size, :] = tr_local
    if rank == 0:
        blocks = plan.buf("tr_recv", (ntt, lmax, nq), xdev, world)
        dist.gather(buf, blocks, dst=0)
        out = plan.buf("tr_full", (ntt, nl, nq), device)[0]
        for r in range(world):
            idx = plan.l_idx[r]
            out.index_copy_(1, idx, blocks[r][:, : idx.numel(), :].to(device))
        return (out[0] if gather == "cl" else out), full
    dist.gather(buf, None, dst=0)
    return None, full


class GpuCompute:
    """compute object backed by the HIP library through the C ABI"""

    def __init__(self, backend):
        self.be = backend

    def perturb(self, k_subset):
        src, self.stats, status = self.be.perturb_solve(k=k_subset)
        return src

    def transfer(self, sources_full, k_all, l_subset, k_size_cl):
        return self.be.transfer(sources_full.contiguous(), k=k_all, l=l_subset, k_size_cl=k_size_cl)

    def cl(self, transfer_local):
        return self.be.cl(transfer_local.contiguous())


def sharded_step_cabi(be, k_all, l_all, rank, world, k_size_cl=None, gather="cl"):
    """The same pass with the two exchanges inside the library (include/cpt.h: cpt_allgather_sources, cpt_gather_cl / cpt_gather_transfer -
    RCCL over xGMI on the handle's stream): nothing but the shard bookkeeping is left to the host language.  `be` must have joined a
    communicator (Backend.comm_init).  Returns, on rank 0 (None elsewhere), the C_l table [nl][ct] (gather = "cl": every rank finishes the
    spectra of its own multipoles) or the full transfer table (gather = "transfer")."""
    nk, nl = len(k_all), len(l_all)
    my_k, my_l = shard_indices(nk, rank, world), shard_indices(nl, rank, world)
    _, stats, _ = be.perturb_solve(k=k_all[my_k], want_sources=False)
    be.allgather_sources(nk)
    tr_local = be.transfer(None, k=k_all, l=l_all[my_l], k_size_cl=nk if k_size_cl is None else k_size_cl)
    if gather == "cl":
        return be.gather_cl(be.cl(tr_local), nl), stats
    return be.gather_transfer(tr_local, nl), stats
