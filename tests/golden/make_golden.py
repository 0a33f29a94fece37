#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs and float64 oracle outputs for the hot path.

The reference (TensorFlow / TFP / lenstronomy) cannot be executed in the build container (the packages are
not installed and there is no network), so these vectors come from the line-by-line restatement in
``oracle/ref_torch.py`` (float64), which tests/test_oracle_pins.py pins against the reference's known-answer
test and independently restated published formulas.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gigalens_amd import workloads  # noqa: E402  (model / prior descriptions only -- no GPU involved)
from gigalens_amd.model import _Packing  # noqa: E402
from oracle import ref_torch as ref  # noqa: E402
from tests.helpers import struct_from_packed  # noqa: E402

CASES = {
    "c1_sie_sersic_64": ("C1", dict(num_pix=64, batch=1)),          # BASELINE.json configs[0]
    "c2_epl_shear_sersic_32": ("C2", dict(num_pix=32, batch=4)),
    "c3_shapelets_table_24": ("C3", dict(num_pix=24, batch=2, interpolate=True)),
    "c3_shapelets_direct_24": ("C3", dict(num_pix=24, batch=2, interpolate=False)),
    "c4_cluster_32": ("C4", dict(num_pix=32, batch=2, n_halos=3, n_sources=4)),
    "c6_cluster_members_28": ("C6", dict(num_pix=28, batch=2, n_galaxies=15, n_sources=2)),  # dPIE halo + catalogue
}


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (wname, kw) in CASES.items():
        if os.path.exists(os.path.join(out_dir, name + ".npz")) and "--all" not in sys.argv:
            continue  # committed fixtures are never silently regenerated
        wl = workloads.make(wname, **kw)
        B, n = wl.batch, wl.sim_config.num_pix
        pack = _Packing(wl.phys_model)
        truth = pack.pack(wl.prior.sample(1, seed=1), 1, "cpu").double()
        params = pack.pack(wl.prior.sample(B, seed=11), B, "cpu")  # float32 values, as the product receives them
        rs1 = ref.RefSimulator(wl.phys_model, wl.sim_config, 1, dtype=torch.float64)
        img_truth = rs1.simulate(struct_from_packed(wl.phys_model, truth)).reshape(n, n)
        g = torch.Generator().manual_seed(2)
        noise = torch.randn((n, n), generator=g, dtype=torch.float64)
        if wl.use_error_map:
            err = torch.full((n, n), float(0.05 * img_truth.abs().max() + wl.background_rms), dtype=torch.float64)
            obs = img_truth + err * noise
            err_np = err.numpy().astype(np.float32)
        else:
            obs = img_truth + torch.sqrt(wl.background_rms ** 2 + img_truth.clamp_min(0) / wl.exp_time) * noise
            err_np = None
        obs_np = obs.numpy().astype(np.float32)
        rs = ref.RefSimulator(wl.phys_model, wl.sim_config, B, dtype=torch.float64)
        p = params.double().clone().requires_grad_(True)
        ll, red = ref.stats_pixels(rs, struct_from_packed(wl.phys_model, p), obs_np, wl.background_rms, wl.exp_time,
                                   error_map=err_np)
        (grad,) = torch.autograd.grad(ll.sum(), p)
        img = rs.simulate(struct_from_packed(wl.phys_model, params.double())).reshape(B, n, n)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), workload=wname, kwargs=repr(kw), params=params.numpy(),
            observed=obs_np, error_map=(err_np if err_np is not None else np.zeros(0, np.float32)),
            background_rms=np.float32(wl.background_rms), exp_time=np.float32(wl.exp_time),
            image=img.detach().numpy(), loglike=ll.detach().reshape(B).numpy(), red_chi2=red.detach().reshape(B).numpy(),
            grad=grad.numpy())
        print(name, "P =", params.shape[1], "loglike[0] =", float(ll.reshape(B)[0]))


if __name__ == "__main__":
    main()
