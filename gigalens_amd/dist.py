"""Data-parallel sharding of the sample batch: one process per GPU, ``torch.distributed`` (backend
``nccl`` == RCCL over xGMI on ROCm; ``gloo`` for CPU rehearsal).

What the path needs (reference: src/gigalens/jax/inference.py:32-80,91-144,157-208):
samples are independent, so MAP and HMC shard with NO data-path collective (only a final gather);
SVI needs exactly one all-reduce per step of ``[ELBO, d ELBO/d mu (d), d ELBO/d L_packed (d(d+1)/2)]``
-- the equivalent of ``jax.lax.pmean`` at jax/inference.py:126-128.  At d = 132 that is 8 911 floats
(35 KB): latency-bound, so value and gradient travel in ONE fused buffer and nothing is bucketed.
"""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # GIGALENS_DIST_BACKEND=gloo rehearses the multi-rank path on a single-GPU box
            backend = os.environ.get("GIGALENS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _decide_avg()
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    return rank, local_rank, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r owns samples [r*n/W, (r+1)*n/W) of a batch divisible by W (jax/inference.py:33-38 uses n // dev_cnt)."""
    per = n_total // world
    return rank * per, (rank + 1) * per


def rank_generator(seed: int, rank: int, device="cpu") -> torch.Generator:
    """Independent per-rank RNG stream (the JAX driver splits its key per device, jax/inference.py:92,136)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) * 1000003 + int(rank))
    return g


_AVG_OK = [False]  # ReduceOp.AVG (an RCCL feature): decided ONCE, by all ranks together, when the process group comes up


def _decide_avg():
    """Every rank tries ``ReduceOp.AVG`` on a one-element tensor and the ranks agree on the outcome with a MIN all-reduce, so
    that no rank can ever issue a different collective from the others.  (Until round 4 a ``RuntimeError`` from the AVG call
    itself switched THAT rank to SUM + division for the rest of the process: a failure on one rank only -- a timeout, an
    asynchronous communicator error surfacing at this call -- would have left the ranks issuing different collectives.)"""
    _AVG_OK[0] = False
    if not (dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl" and torch.cuda.is_available()):
        return
    ok = 1.0
    try:
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
        torch.cuda.synchronize()
        if abs(float(t.item()) - 1.0) > 1e-6:
            ok = 0.0
    except RuntimeError:  # "unsupported reduction": raised on every rank alike, before anything is enqueued
        ok = 0.0
    flag = torch.tensor([ok], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    _AVG_OK[0] = bool(flag.item() > 0.5)


def allreduce_mean_(buf: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks (== lax.pmean). One collective for the whole fused buffer.  Errors propagate: a failed
    collective is never papered over with a different one."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if buf.is_cuda and _AVG_OK[0]:
            dist.all_reduce(buf, op=dist.ReduceOp.AVG)  # RCCL averages inside the collective kernel: no second launch
            return buf
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.div_(dist.get_world_size())
    return buf


def allreduce_min_(buf: torch.Tensor) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.MIN)
    return buf


def broadcast_(buf: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast from rank `src` (the SVI driver re-synchronises the surrogate's parameters with it every few hundred
    steps, so that ranks cannot drift apart even if a collective implementation is not bitwise identical on every rank)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, src=src)
    return buf


def allreduce_max_(buf: torch.Tensor) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    return buf


def gather_rows(local: torch.Tensor) -> torch.Tensor:
    """Final gather of per-rank rows (MAP solutions, HMC chains): concatenation in rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local.contiguous())
    return torch.cat(out, dim=0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
