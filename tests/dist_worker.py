"""One rank of the two-rank, real-kernel test of the sharded drivers (started by tests/test_gpu_dist.py through
``python -m torch.distributed.run --nproc-per-node 2``; both ranks share GPU 0 and talk over gloo -- RCCL refuses two ranks on
one device, so the collective LOGIC is what runs here, with the HIP kernels underneath).  Rank 0 writes what it checked to the
JSON file named on the command line.  Reference: src/gigalens/jax/inference.py:32-80 (MAP), 91-144 (SVI pmean), 157-208 (HMC)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gigalens_amd import dist as gdist  # noqa: E402
from gigalens_amd import inference as ginf  # noqa: E402
from gigalens_amd import workloads  # noqa: E402
from gigalens_amd.model import ForwardProbModel  # noqa: E402
from gigalens_amd.simulator import LensSimulator  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, _, world = gdist.init_from_env(backend="gloo")
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    res = {}
    # ---- config 5's shard: the cluster model (8 NFW + 20 Sersic, d = 132), reduced field so that the test takes seconds ----
    n_local = 24
    wl = workloads.make("C5", num_pix=48, batch=n_local)
    obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=n_local)
    z0 = pm.bij.inverse(wl.prior.sample(4, generator=gdist.rank_generator(5, 0)))[0].to(dev).contiguous()  # same on both ranks
    d = z0.numel()
    mu, lpk = z0.clone(), ginf.tril_pack(torch.eye(d, device=dev) * 1e-3)

    def vg(zz):
        lp_, _, g_ = pm.log_prob_and_grad(sim, zz)
        return lp_, g_

    def eps_of(r):
        return torch.randn((n_local, d), generator=gdist.rank_generator(11, r, device=dev), device=dev)

    buf = ginf.svi_step_buffer(mu, lpk, None, n_local, None, value_and_grad_fn=vg, full_rank=True, eps=eps_of(rank))
    res["buffer_floats"] = int(buf.numel())
    assert "gl_clusterw_kernel" in sim._model.last_main_kernel(), sim._model.last_main_kernel()
    # (1) the all-reduced buffer == the mean of the two shards' buffers computed in ONE process without a collective
    shards = [ginf.svi_step_buffer(mu, lpk, None, n_local, None, value_and_grad_fn=vg, full_rank=True, eps=eps_of(r), reduce=False)
              for r in range(world)]
    mean = (shards[0] + shards[1]) / world
    res["allreduce_equals_mean_of_shards"] = bool(torch.equal(buf, mean))
    res["allreduce_max_abs_diff"] = float((buf - mean).abs().max())
    res["shards_differ"] = bool(not torch.equal(shards[0], shards[1]))  # different eps per rank (jax/inference.py:92,136)
    both = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(both, buf)
    res["buffer_identical_on_both_ranks"] = bool(torch.equal(both[0], both[1]))
    # (2) 20 Adam steps of the SVI driver: (mu, L) bitwise equal across ranks (no periodic broadcast: sync_every = 0)
    seq = ginf.ModellingSequence(wl.phys_model, pm, wl.sim_config)
    (q_mean, q_tril), losses = seq.SVI(ginf.Adam(1e-3), z0, n_vi=2 * n_local, num_steps=20, sync_every=0)
    for name, t in (("q_mean", q_mean), ("q_scale_tril", q_tril)):
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        res[f"svi_{name}_bitwise_equal"] = bool(torch.equal(parts[0], parts[1]))
    res["svi_losses_finite"] = bool(all(map(lambda v: v == v and abs(v) != float("inf"), losses)))
    res["svi_moved"] = bool(not torch.equal(q_mean, z0))
    # (2b) the guard broadcast path runs (sync_every = 5) and leaves the ranks equal
    (q2, _), _ = seq.SVI(ginf.Adam(1e-3), z0, n_vi=2 * n_local, num_steps=10, sync_every=5)
    parts = [torch.zeros_like(q2) for _ in range(world)]
    dist.all_gather(parts, q2.contiguous())
    res["svi_with_broadcast_equal"] = bool(torch.equal(parts[0], parts[1]))
    # (3) MAP: the final gather restores the global sample order (0 steps: the gathered rows are the start rows) and a short
    # run returns all rows, finite
    start = wl.prior.sample(16, seed=3)
    z_start = pm.bij.inverse(start).to(dev)
    sol0 = seq.MAP(ginf.Adam(1e-2), start, n_samples=16, num_steps=0)
    res["map_gather_restores_order"] = bool(torch.allclose(sol0, z_start, rtol=0, atol=0))
    sol = seq.MAP(ginf.Adam(1e-2), start, n_samples=16, num_steps=5)
    res["map_rows"] = int(sol.shape[0])
    res["map_finite_and_moved"] = bool(torch.isfinite(sol).all() and not torch.equal(sol, z_start))
    lo, hi = gdist.shard_bounds(16, rank, world)
    mine = [torch.zeros_like(sol) for _ in range(world)]
    dist.all_gather(mine, sol.contiguous())
    res["map_solution_identical_on_both_ranks"] = bool(torch.equal(mine[0], mine[1]))
    # (4) HMC: n_hmc chains come back (each rank ran n_hmc / 2, no collective but the final gather)
    samples, stats = seq.HMC((q_mean, q_tril), n_hmc=8, init_eps=0.05, init_l=2, max_leapfrog_steps=3, num_burnin_steps=3,
                             num_results=4)
    res["hmc_shape"] = list(samples.shape)
    res["hmc_finite"] = bool(torch.isfinite(samples).all())
    res["hmc_chains_differ_across_ranks"] = bool(not torch.equal(samples[:, :4], samples[:, 4:]))
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
    gdist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
