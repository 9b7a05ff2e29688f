"""Multi-GPU form of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the node; "gloo" on CPU for the tests).

  stage A  k-modes are independent units (pm.cpp:686-707): rank r integrates modes r, r+N, r+2N, ... (round-robin, so
           every rank gets the same mix of cheap low-k and expensive high-k modes).
  exchange 1  all_gather of the local source blocks [tp][ntau][nk_local] -> full sources on every rank.  This exchange
           is real: the transfer stage splines the sources across ALL k (tm.cpp:604-639).
  stage B  rank r computes Delta_l(q) for multipoles l[r::N] and every q (cost grows with l, round-robin balances it).
  exchange 2  gather of the [tt][nl_local][nq] blocks on rank 0 (the downstream C_l integral needs every q of a given l,
           which each rank already holds, so only results travel).

Data volumes are tiny (sources 17 MB, transfer 12 MB for explanatory.ini), the collectives are latency-bound; exactly
two collectives per step, no ring all-reduce.
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


def sharded_step(compute, k_all, l_all, rank, world, device, k_size_cl=None):
    """One pass of the hot path over `world` ranks.

    compute.perturb(k_subset) -> torch f64 [tp][ntau][len(k_subset)] on `device`
    compute.transfer(sources_full, k_all, l_subset, k_size_cl) -> torch f64 [tt][len(l_subset)][nq] on `device`
    Returns the full transfer table [tt][nl][nq] on rank 0 (None elsewhere) and the full sources.
    """
    nk, nl = len(k_all), len(l_all)
    my_k = shard_indices(nk, rank, world)
    local = compute.perturb(k_all[my_k])
    ntp, ntau = local.shape[0], local.shape[1]
    if world == 1:
        full = local
    else:
        # ---- exchange 1: all_gather (pad to the largest shard so that every block has the same shape) ----
        nmax = (nk + world - 1) // world
        buf = torch.zeros((ntp, ntau, nmax), dtype=torch.float64, device=device)
        buf[:, :, : my_k.size] = local
        blocks = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(blocks, buf)
        full = torch.empty((ntp, ntau, nk), dtype=torch.float64, device=device)
        for r in range(world):
            idx = shard_indices(nk, r, world)
            full[:, :, torch.as_tensor(idx, device=device)] = blocks[r][:, :, : idx.size]
    my_l = shard_indices(nl, rank, world)
    tr_local = compute.transfer(full, k_all, l_all[my_l], nk if k_size_cl is None else k_size_cl)
    if world == 1:
        return tr_local, full
    # ---- exchange 2: gather on rank 0 ----
    ntt, nq = tr_local.shape[0], tr_local.shape[2]
    lmax = (nl + world - 1) // world
    buf = torch.zeros((ntt, lmax, nq), dtype=torch.float64, device=device)
    buf[:, : my_l.size, :] = tr_local
    if rank == 0:
        blocks = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, blocks, dst=0)
        out = torch.empty((ntt, nl, nq), dtype=torch.float64, device=device)
        for r in range(world):
            idx = shard_indices(nl, r, world)
            out[:, torch.as_tensor(idx, device=device), :] = blocks[r][:, : idx.size, :]
        return out, full
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
