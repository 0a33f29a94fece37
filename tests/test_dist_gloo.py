"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel path: sample sharding, the single fused
SVI all-reduce (== lax.pmean of value and gradient, jax/inference.py:123-128) and the final gather.
The likelihood kernel itself needs a GPU, so a closed-form log-density stands in for it here: what is under
test is the sharding / collective logic that bench.py --gpus N and ModellingSequence.SVI use unchanged."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_log_prob(z):
    a = torch.linspace(0.5, 2.0, z.shape[1])
    return -0.5 * ((z - 0.3) ** 2 * a).sum(-1) + 0.1 * torch.sin(z).sum(-1)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gigalens_amd import dist as gdist
    from gigalens_amd import inference as inf
    r, lr, w = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and gdist.world_size() == world
    d, n_local = 5, 64
    mu = torch.linspace(-0.2, 0.2, d)
    lp = inf.tril_pack(torch.eye(d) * 0.3 + torch.tril(torch.full((d, d), 0.01), -1))
    gen = gdist.rank_generator(7, rank)
    loss, g_mu, g_lp = inf.svi_step(mu, lp, _toy_log_prob, n_local, gen)
    lo, hi = gdist.shard_bounds(10, rank, world)
    rows = gdist.gather_rows(torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.ones(1, 3))
    t = torch.tensor([float(rank)])
    gdist.allreduce_max_(t)
    out[rank] = dict(loss=loss.clone(), g_mu=g_mu.clone(), g_lp=g_lp.clone(), rows=rows, tmax=float(t))
    gdist.barrier()
    dist.destroy_process_group()


def test_svi_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    # identical on every rank after the collective
    assert torch.equal(a["loss"], b["loss"]) and torch.equal(a["g_mu"], b["g_mu"]) and torch.equal(a["g_lp"], b["g_lp"])
    # equals the mean of the two per-shard ELBO evaluations computed in one process
    from gigalens_amd import dist as gdist
    from gigalens_amd import inference as inf
    d, n_local = 5, 64
    mu = torch.linspace(-0.2, 0.2, d)
    lp = inf.tril_pack(torch.eye(d) * 0.3 + torch.tril(torch.full((d, d), 0.01), -1))
    parts = [inf.svi_step(mu, lp, _toy_log_prob, n_local, gdist.rank_generator(7, r)) for r in range(world)]
    loss = sum(p[0] for p in parts) / world
    g_mu = sum(p[1] for p in parts) / world
    g_lp = sum(p[2] for p in parts) / world
    assert torch.allclose(a["loss"], loss, rtol=1e-6) and torch.allclose(a["g_mu"], g_mu, rtol=1e-5, atol=1e-7)
    assert torch.allclose(a["g_lp"], g_lp, rtol=1e-5, atol=1e-7)
    # shard bounds + gather restore the global order; max-over-ranks timing reduction works
    assert torch.equal(a["rows"][:, 0], torch.arange(10, dtype=torch.float32)) and a["tmax"] == 1.0


def test_tril_pack_round_trip_and_gradient():
    from gigalens_amd import inference as inf
    d = 6
    L = torch.tril(torch.randn(d, d, generator=torch.Generator().manual_seed(0))) * 0.1 + torch.eye(d)
    L = L - torch.diag(torch.diagonal(L)) + torch.diag(torch.diagonal(L).abs() + 0.1)
    p = inf.tril_pack(L)
    assert p.numel() == d * (d + 1) // 2
    assert torch.allclose(inf.tril_unpack(p, d), L, atol=1e-6)
    # ELBO gradient of a Gaussian target is zero at the exact posterior
    target_mu, target_L = torch.zeros(d) + 0.3, torch.eye(d) * 0.5
    prec = torch.linalg.inv(target_L @ target_L.T)
    log_p = lambda z: -0.5 * (((z - target_mu) @ prec) * (z - target_mu)).sum(-1)
    g = torch.Generator().manual_seed(1)
    loss, g_mu, g_lp = inf.svi_step(target_mu, inf.tril_pack(target_L), log_p, 20000, g)
    assert g_mu.abs().max() < 0.05 and g_lp.abs().max() < 0.05


def test_adam_lr_zero_is_noop():
    """tests/tf/test_model.py:29-43: an optimiser with lr=0 leaves the parameters unchanged, lr>0 moves them."""
    from gigalens_amd.inference import Adam
    x = torch.ones(4)
    Adam(0.0).step(x, torch.ones(4))
    assert torch.equal(x, torch.ones(4))
    Adam(1e-3).step(x, torch.ones(4))
    assert not torch.equal(x, torch.ones(4))
