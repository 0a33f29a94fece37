"""The sharded drivers with the REAL kernels under two ranks (-m gpu): two fresh child processes started through
``python -m torch.distributed.run --nproc-per-node 2`` share GPU 0 and reduce over gloo (RCCL refuses two ranks on one device;
no multi-GPU box is available to the builder).  What is checked, on config 5's cluster model (d = 132,
the 8 911-float fused buffer): the all-reduced buffer equals the mean of the two shards computed in one process, the surrogate
stays bitwise identical across ranks over 20 Adam steps, MAP's gather restores the global order, HMC returns n_hmc chains."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_drivers_with_real_kernels(tmp_path):
    out = tmp_path / "two_rank.json"
    env = dict(os.environ, GIGALENS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["buffer_floats"] == 1 + 132 + 132 * 133 // 2 == 8911
    assert res["shards_differ"] and res["buffer_identical_on_both_ranks"]
    assert res["allreduce_equals_mean_of_shards"] or res["allreduce_max_abs_diff"] < 1e-6, res
    assert res["svi_q_mean_bitwise_equal"] and res["svi_q_scale_tril_bitwise_equal"] and res["svi_with_broadcast_equal"]
    assert res["svi_losses_finite"] and res["svi_moved"]
    assert res["map_gather_restores_order"] and res["map_rows"] == 16 and res["map_finite_and_moved"]
    assert res["map_solution_identical_on_both_ranks"]
    assert res["hmc_shape"] == [4, 8, 132] and res["hmc_finite"] and res["hmc_chains_differ_across_ranks"]
