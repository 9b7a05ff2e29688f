"""world_size-2 test of the multi-process path (classpp_public_amd/sharded.py) on CPU with the gloo backend.
The compute object is the ORACLE here (tests may use it); on the GPU box the same plumbing runs over RCCL with the
HIP backend (bench.py --gpus N).  Checks: the sharded sources equal the single-process sources bit for bit and the
transfer functions to round-off (k-modes and multipoles are independent units; the exchanges only move data)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleCompute:
    def __init__(self, inp):
        self.inp = inp

    def perturb(self, k_subset):
        import oracle_lib
        src, self.stats, status, rc = oracle_lib.perturb(self.inp, k=k_subset, threads=2)
        assert rc == 0
        return torch.from_numpy(src)

    def transfer(self, sources_full, k_all, l_subset, k_size_cl):
        import ctypes as C
        import oracle_lib
        L = oracle_lib.lib()
        inp = self.inp
        src = np.ascontiguousarray(sources_full.numpy())
        l_subset = np.ascontiguousarray(l_subset, dtype=np.int32)
        out = np.zeros((inp.config.tt_size, l_subset.size, inp.q.size))
        k_all = np.ascontiguousarray(k_all)
        rc = L.orc_transfer(C.byref(inp.config), oracle_lib.dptr(src), oracle_lib.dptr(k_all), k_all.size, k_size_cl,
                            oracle_lib.dptr(inp.tau), inp.ntau, oracle_lib.dptr(inp.q), inp.q.size, oracle_lib.iptr(l_subset),
                            l_subset.size, oracle_lib.dptr(out), 2, None)
        assert rc == 0
        return torch.from_numpy(out)

    def cl(self, transfer_local):
        import oracle_lib
        return torch.from_numpy(oracle_lib.cl_table(self.inp, transfer_local.numpy()))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from classpp_public_amd.inputs import Inputs
    from classpp_public_amd.sharded import densify_k, sharded_step
    inp = Inputs("small")
    # keep the test light: every 4th k (35 modes), densified x2 -> 69 modes over 2 ranks
    k_all = densify_k(inp.k[::4], 2)
    comp = OracleCompute(inp)
    out, full = sharded_step(comp, k_all, inp.l, rank, world, torch.device("cpu"), k_all.size)
    if rank == 0:
        ret["sharded"] = out.numpy()
        ret["full_sources"] = full.numpy()
        ret["k_all"] = k_all
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from classpp_public_amd.inputs import Inputs
    from classpp_public_amd.sharded import sharded_step
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    inp = Inputs("small")
    comp = OracleCompute(inp)
    k_all = ret["k_all"]
    single, full1 = sharded_step(comp, k_all, inp.l, 0, 1, torch.device("cpu"), k_all.size)
    assert np.array_equal(ret["full_sources"], full1.numpy())
    # sources: bit for bit.  Transfer functions: each rank builds the Bessel table for ITS multipoles, so the
    # recurrences start from a different l_max (hyperspherical.c:517-603) => round-off level differences only.
    a, b = ret["sharded"], single.numpy()
    scale = np.max(np.abs(b), axis=-1, keepdims=True)
    scale[scale == 0] = 1
    assert np.array_equal(a == 0, b == 0)
    assert np.max(np.abs(a - b) / scale) < 1e-10
    assert ret["sharded"].shape == (inp.config.tt_size, inp.l.size, inp.q.size)


def test_densify_and_shards_partition_the_grid():
    from classpp_public_amd.sharded import densify_k, shard_indices
    k = np.geomspace(1e-4, 1.0, 17)
    d = densify_k(k, 4)
    assert d.size == 16 * 4 + 1 and np.all(np.diff(d) > 0) and np.allclose(d[::4], k)
    for world in (1, 2, 3, 8):
        idx = np.concatenate([shard_indices(d.size, r, world) for r in range(world)])
        assert sorted(idx.tolist()) == list(range(d.size))


# ---- the HIP-shaped interface under real torch.distributed: sharded_step drives classpp_public_amd.sharded.GpuCompute, the class that
#      wraps the HIP Backend on a GPU box, exactly as bench.py --gpus N does.  Here the Backend is a stand-in with the SAME methods and
#      argument conventions (perturb_solve(k=...) -> (tensor, stats, status), transfer(sources, k=, l=, k_size_cl=)) computing with the
#      oracle, so the plumbing between GpuCompute and Backend - keyword names, shapes, dtypes, contiguity - is what is under test. ----
class BackendStandIn:
    def __init__(self, inp):
        self.inp = inp
        self.calls = []

    def perturb_solve(self, k=None, tau=None, want_sources=True):
        import oracle_lib
        assert tau is None and want_sources
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        src, stats, status, rc = oracle_lib.perturb(self.inp, k=k, threads=2)
        assert rc == 0
        self.calls.append(("perturb_solve", k.size))
        return torch.from_numpy(src), stats, status

    def transfer(self, sources=None, k=None, tau=None, q=None, l=None, k_size_cl=None):
        assert sources is not None and sources.is_contiguous() and sources.dtype == torch.float64 and tau is None and q is None
        assert tuple(sources.shape) == (self.inp.config.tp_size, self.inp.ntau, len(k))
        self.calls.append(("transfer", len(l)))
        return OracleCompute(self.inp).transfer(sources, k, l, k_size_cl)


def _worker_gpu_shaped(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from classpp_public_amd.inputs import Inputs
    from classpp_public_amd.sharded import GpuCompute, sharded_step
    inp = Inputs("small")
    k_all = np.ascontiguousarray(inp.k[::4])          # a FIXED grid, sharded (strong scaling, as bench.py --gpus N does by default)
    be = BackendStandIn(inp)
    out, full = sharded_step(GpuCompute(be), k_all, inp.l, rank, world, torch.device("cpu"), k_all.size)
    ret["calls%d" % rank] = be.calls
    if rank == 0:
        ret["sharded"] = out.numpy()
        ret["k_all"] = k_all
    dist.destroy_process_group()


def test_gpu_compute_interface_two_ranks():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from classpp_public_amd.inputs import Inputs
    from classpp_public_amd.sharded import GpuCompute, shard_indices, sharded_step
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + ((os.getpid() + 7) % 1000)
    mp.spawn(_worker_gpu_shaped, args=(2, port, ret), nprocs=2, join=True)
    inp = Inputs("small")
    k_all = ret["k_all"]
    for r in range(2):   # each rank integrated its round-robin shard of the k grid and projected its shard of the multipoles
        assert ret["calls%d" % r] == [("perturb_solve", shard_indices(k_all.size, r, 2).size), ("transfer", shard_indices(inp.l.size, r, 2).size)]
    single, _ = sharded_step(GpuCompute(BackendStandIn(inp)), k_all, inp.l, 0, 1, torch.device("cpu"), k_all.size)
    a, b = ret["sharded"], single.numpy()
    scale = np.max(np.abs(b), axis=-1, keepdims=True)
    scale[scale == 0] = 1
    assert np.array_equal(a == 0, b == 0) and np.max(np.abs(a - b) / scale) < 1e-10


# ---- four ranks, uneven shards (35 k-modes and 46 multipoles do not divide by 4), spectra finished where the transfer functions are:
#      exchange 2 gathers the C_l rows (7 numbers per multipole) instead of the transfer table
def _worker_cl(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from classpp_public_amd.inputs import Inputs
    from classpp_public_amd.sharded import shard_indices, sharded_step
    inp = Inputs("small")
    k_all = np.ascontiguousarray(inp.k[::4])
    assert k_all.size % world != 0 and inp.l.size % world != 0
    comp = OracleCompute(inp)
    out, full = sharded_step(comp, k_all, inp.l, rank, world, torch.device("cpu"), k_all.size, gather="cl")
    ret["nk%d" % rank] = shard_indices(k_all.size, rank, world).size
    if rank == 0:
        ret["cl"] = out.numpy().copy()
        ret["full_sources"] = full.numpy().copy()
        ret["k_all"] = k_all
    else:
        assert out is None
    dist.destroy_process_group()


def test_four_ranks_uneven_shards_gather_cl():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from classpp_public_amd.inputs import Inputs
    from classpp_public_amd.sharded import sharded_step
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + ((os.getpid() + 13) % 1000)
    mp.spawn(_worker_cl, args=(4, port, ret), nprocs=4, join=True)
    inp = Inputs("small")
    k_all = ret["k_all"]
    assert sorted(ret["nk%d" % r] for r in range(4)) == [8, 9, 9, 9] and sum(ret["nk%d" % r] for r in range(4)) == k_all.size
    comp = OracleCompute(inp)
    cl1, full1 = sharded_step(comp, k_all, inp.l, 0, 1, torch.device("cpu"), k_all.size, gather="cl")
    assert np.array_equal(ret["full_sources"], full1.numpy())          # k-modes are independent units: bit for bit
    a, b = ret["cl"], cl1.numpy()
    assert a.shape == (inp.l.size, inp.spectra.ct_size)
    scale = np.max(np.abs(b), axis=0, keepdims=True)
    scale[scale == 0] = 1
    assert np.max(np.abs(a - b) / scale) < 1e-10                        # multipoles are independent units: round-off of the Bessel recurrences
    # and it IS the spectrum of the gathered transfer table
    tr1, _ = sharded_step(comp, k_all, inp.l, 0, 1, torch.device("cpu"), k_all.size)
    assert np.array_equal(b, oracle_lib.cl_table(inp, tr1.numpy()))
