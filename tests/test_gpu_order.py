"""Cost-ordered dispatch (heaviest EPL series first): the order is a scheduling matter only.  The wavefront-per-sample front end
carries the sort as one extra workgroup (it recomputes the trip counts from (e1, e2) itself, through the bijectors on the z path)
instead of a launch of gl_order_kernel.  Results must be bitwise those of the separate launch and of no ordering at all, for batch
sizes that are not multiples of the front end's four samples per workgroup, through the parameter-row entry and the z entry."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_parity import gl  # noqa: F401

pytestmark = pytest.mark.gpu


def _run(gl, wl, obs, err, packed, z):
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    out = [t.clone() for t in sim._model.loglike(packed, obs, err, None, wl.background_rms, wl.exp_time, True)]
    pm = gl.ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    zz = z.clone().requires_grad_(True)
    lp, _ = pm.log_prob(sim, zz)
    lp.sum().backward()
    torch.cuda.synchronize()
    return out + [lp.detach().clone(), zz.grad.clone()]


@pytest.mark.parametrize("batch", [2, 7, 64, 1023])
def test_front_end_sort_changes_no_bit(gl, monkeypatch, batch):
    wl = gl.workloads.make("C2", num_pix=32, batch=batch)
    obs, err, _ = gl.workloads.synthetic_observation(wl, gl.LensSimulator)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    packed = H.sample_packed(wl, sim, seed=3)
    pm = gl.ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    z = pm.bij.inverse(wl.prior.sample(batch, seed=3)).to("cuda")
    fused = _run(gl, wl, obs, err, packed, z)
    monkeypatch.setenv("GIGALENS_HIP_ORDER_FUSED", "0")  # read at model creation
    separate = _run(gl, wl, obs, err, packed, z)
    monkeypatch.delenv("GIGALENS_HIP_ORDER_FUSED")
    monkeypatch.setenv("GIGALENS_HIP_ORDER", "0")
    unordered = _run(gl, wl, obs, err, packed, z)
    assert len(fused) == len(separate) == len(unordered) == 5
    for a, b, c in zip(fused, separate, unordered):
        assert torch.equal(a, b) and torch.equal(a, c)
        assert torch.isfinite(a).all()


def test_the_extra_workgroup_sorts_by_the_same_trip_counts(gl):
    """The order the front end's extra workgroup writes is a permutation, sorted by the cost the sample workgroups wrote."""
    import ctypes
    from gigalens_amd import _native
    wl = gl.workloads.make("C2", num_pix=32, batch=257)
    obs, err, _ = gl.workloads.synthetic_observation(wl, gl.LensSimulator)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = H.sample_packed(wl, sim, seed=5)
    m = sim._model
    m.loglike(packed, obs, err, None, wl.background_rms, wl.exp_time, True)
    torch.cuda.synchronize()
    B = wl.batch
    ws = m._workspace(B)
    chunk, nc, row, off = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    _native._check(_native.lib().gl_model_launch_shape(m._h, B, ctypes.byref(chunk), ctypes.byref(nc), ctypes.byref(row), ctypes.byref(off)))
    al = lambda n: (n + 255) // 256 * 256
    o_order = off.value + al(B * nc.value * row.value * 4) + al(B * m.P * 4)   # carve(): partial | params | order | cost
    order = ws[o_order:o_order + 4 * B].view(torch.int32).cpu().numpy()
    cost = ws[o_order + al(4 * B):o_order + al(4 * B) + 4 * B].view(torch.int32).cpu().numpy()
    assert sorted(order.tolist()) == list(range(B))
    assert np.all(np.diff(cost[order]) <= 0), "not sorted heaviest first"
    assert cost.min() >= 0 and cost.max() > cost.min()
