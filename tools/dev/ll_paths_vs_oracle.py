import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gigalens_amd.model import ForwardProbModel
from gigalens_amd.simulator import LensSimulator
class gl: pass
gl.ForwardProbModel, gl.LensSimulator = ForwardProbModel, LensSimulator
from gigalens_amd import workloads
import helpers as H
wl = workloads.make("C5")
obs, err, _ = workloads.synthetic_observation(wl, gl.LensSimulator)
res = {}
for flag in ("1", "0"):
    os.environ["GIGALENS_HIP_CLUSTER"] = flag
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = H.sample_packed(wl, sim, seed=5)
    pm = gl.ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    p = packed.clone().requires_grad_(True)
    ll, red = pm._pixel_stats_packed(sim, p)
    ll.sum().backward()
    res[flag] = (ll.detach().double().cpu().numpy(), p.grad.double().cpu().numpy())
im = sim.simulate(packed).double(); o = obs.double()
sig2 = wl.background_rms ** 2 + im / wl.exp_time
ll_img = (-0.5 * (((im - o) ** 2 / sig2).sum((-2, -1)) + torch.log(2 * math.pi * sig2).sum((-2, -1)))).cpu().numpy()
d = np.abs(res["1"][0] - ll_img) / np.abs(ll_img)
idx = np.argsort(-d)[:6]
print("worst fused-vs-image", d[idx], idx)
wl2 = workloads.make("C5", batch=6)
ll_o, red_o, g_o, img_o = H.oracle_loglike_and_grad(wl2, packed[idx].double().cpu(), obs.cpu().numpy(), None, 6)
for k, i in enumerate(idx):
    print(i, "oracle", ll_o[k], "table", res["1"][0][i], "interp", res["0"][0][i], "img", ll_img[i],
          "rel: table %.2e interp %.2e img %.2e" % tuple(abs(v - ll_o[k]) / abs(ll_o[k]) for v in (res["1"][0][i], res["0"][0][i], ll_img[i])))
