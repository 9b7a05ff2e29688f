"""The multi-GPU exchanges behind the C ABI (include/cpt.h, classpp_public_amd/csrc/cpt_comm.hip) as far as ONE GPU can check them:
  * the packing / un-interleaving kernels of both exchanges against numpy, for 2, 3 and 8 simulated ranks;
  * the whole sharded pass through a real RCCL communicator of world size 1 (init, all-gather, gather, destroy) against the
    unsharded pass;
  * a simulated world of 2 on one device: both ranks' shards run in turn, the blocks an all-gather would deliver are assembled by hand,
    un-interleaved by the library's kernel, and must reproduce the unsharded sources and transfer functions.
Runs with more than one rank need one GPU per process (RCCL refuses two ranks on one device): that is the driver's scaling run."""
import numpy as np
import pytest
import torch

from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs
from classpp_public_amd.sharded import shard_indices, sharded_step_cabi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    b = Backend(Inputs("small"), "cuda:0")
    yield b
    b.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_pad_and_uninterleave_kernels(be, world):
    rng = np.random.default_rng(world)
    nb, n_all, ninner = 3, 37, 11
    full = rng.standard_normal((nb, n_all, ninner))
    n_max = (n_all + world - 1) // world
    blocks = []
    for r in range(world):
        idx = shard_indices(n_all, r, world)
        local = torch.from_numpy(np.ascontiguousarray(full[:, idx, :])).cuda()
        padded = be.dbg_pad_rows(local, n_max)
        want = np.zeros((nb, n_max, ninner)); want[:, : idx.size, :] = full[:, idx, :]
        assert np.array_equal(padded.cpu().numpy(), want)
        blocks.append(padded)
    got = be.dbg_uninterleave(torch.stack(blocks).contiguous(), n_all)
    assert np.array_equal(got.cpu().numpy(), full)


def test_world_of_one_through_rccl(be):
    inp = be.inp
    src, _, _ = be.perturb_solve()
    tr = be.transfer(None).cpu().numpy()
    be.comm_init(be.comm_unique_id(), 0, 1)
    try:
        out, stats = sharded_step_cabi(be, inp.k, inp.l, 0, 1, inp.k_size_cl, gather="transfer")
        assert np.array_equal(out.cpu().numpy(), tr)
        assert np.array_equal(be.get_sources(inp.ntau, inp.nk).cpu().numpy(), src.cpu().numpy())
        # the default exchange 2: every rank finishes the C_l rows of its multipoles, cpt_gather_cl collects them
        cl, _ = sharded_step_cabi(be, inp.k, inp.l, 0, 1, inp.k_size_cl)
        assert np.array_equal(cl.cpu().numpy(), be.cl(torch.from_numpy(tr).cuda()).cpu().numpy())
    finally:
        be.lib.cpt_comm_destroy(be.h)


def test_simulated_world_of_two_on_one_device(be):
    inp = be.inp
    world, nk, nl = 2, inp.nk, inp.l.size
    src_ref, _, _ = be.perturb_solve()
    tr_ref = be.transfer(None).cpu().numpy()
    # stage A on every "rank", exchange 1 assembled by hand (what ncclAllGather delivers: the padded k-major blocks, rank-major)
    n_max = (nk + world - 1) // world
    blocks = []
    for r in range(world):
        local, _, _ = be.perturb_solve(k=inp.k[shard_indices(nk, r, world)])          # [tp][ntau][nk_local]
        blocks.append(be.dbg_pad_rows(local.permute(0, 2, 1).contiguous(), n_max))   # k-major, padded
    full_kmajor = be.dbg_uninterleave(torch.stack(blocks).contiguous(), nk)           # [tp][nk][ntau]
    full = full_kmajor.permute(0, 2, 1).contiguous()
    assert np.array_equal(full.cpu().numpy(), src_ref.cpu().numpy())                 # modes are independent units: bit for bit
    # stage B on every "rank", exchange 2 by hand
    l_max = (nl + world - 1) // world
    tblocks = [be.dbg_pad_rows(be.transfer(full, l=inp.l[shard_indices(nl, r, world)]), l_max) for r in range(world)]
    tr = be.dbg_uninterleave(torch.stack(tblocks).contiguous(), nl).cpu().numpy()
    scale = np.max(np.abs(tr_ref), axis=-1, keepdims=True)
    scale[scale == 0] = 1
    assert np.array_equal(tr == 0, tr_ref == 0) and np.max(np.abs(tr - tr_ref) / scale) < 1e-10


def test_torch_distributed_exchanges_over_rccl_in_a_group_of_one(be):
    """classpp_public_amd/sharded.py::sharded_step - what bench.py --gpus N runs by default - with backend "nccl" (= RCCL) on GPU tensors:
    a process group of one rank, both collectives forced (all_gather of the padded source block, gather of the transfer block on rank 0,
    the index_copy_ scatter into the full tables).  The N > 1 runs are the driver's; this is the same code on the same library."""
    import os
    import torch.distributed as dist
    from classpp_public_amd.sharded import GpuCompute, sharded_step
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        inp = be.inp
        dev = torch.device("cuda:0")
        want, want_src = sharded_step(GpuCompute(be), inp.k, inp.l, 0, 1, dev, inp.k_size_cl)
        want, want_src = want.cpu().numpy().copy(), want_src.cpu().numpy().copy()
        got, got_src = sharded_step(GpuCompute(be), inp.k, inp.l, 0, 1, dev, inp.k_size_cl, force_exchange=True)
        assert np.array_equal(got_src.cpu().numpy(), want_src) and np.array_equal(got.cpu().numpy(), want)
    finally:
        dist.destroy_process_group()
