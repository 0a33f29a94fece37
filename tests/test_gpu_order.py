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
    # (the tapered end of the dispatch sums a tail sample's pixels over other chunks than an unordered launch does: off here)
    monkeypatch.setenv("GIGALENS_HIP_TAIL_ROWS", "0")
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


@pytest.mark.parametrize("batch,num_pix", [(1023, 32), (600, 64)])
def test_tapered_end_of_the_dispatch(gl, monkeypatch, batch, num_pix):
    """The cheapest samples of a launch -- the remainder beyond whole resident rounds -- run as twice as many workgroups of half
    the pixels.  Which samples those are is decided by a deterministic split of the sort (bitwise reproducible results), their
    partial sums are those of a finer chunking (a few ulps of the pixel sum), and every other sample keeps its bits."""
    wl = gl.workloads.make("C2", num_pix=num_pix, batch=batch)
    obs, err, _ = gl.workloads.synthetic_observation(wl, gl.LensSimulator)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    packed = H.sample_packed(wl, sim, seed=11)
    pm = gl.ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    z = pm.bij.inverse(wl.prior.sample(batch, seed=11)).to("cuda")
    runs = [_run(gl, wl, obs, err, packed, z) for _ in range(3)]
    sim_t = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    sim_t._model.loglike(packed, obs, err, None, wl.background_rms, wl.exp_time, True)
    rows_tapered = sim_t._model.partial_rows(batch).shape[1]
    monkeypatch.setenv("GIGALENS_HIP_TAIL_ROWS", "0")
    plain = _run(gl, wl, obs, err, packed, z)
    sim_p = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    assert rows_tapered == 2 * sim_p._model.partial_rows(batch).shape[1], "the tapered end did not engage at this size"
    for a, b, c in zip(*runs):
        assert torch.equal(a, b) and torch.equal(a, c)
    n_same = 0
    for a, p in zip(runs[0], plain):
        assert torch.isfinite(a).all()
        scale = p.abs().amax(0, keepdim=True) if p.dim() > 1 else p.abs()
        assert ((a - p).abs() <= 2e-5 * scale.clamp_min(1e-30)).all()
        n_same += int((a == p).reshape(batch, -1).all(1).sum())
    assert n_same >= 5 * (batch // 2), "samples outside the tail changed"
