"""What a model with user-written profiles costs: forward+gradient of the log-likelihood for SIS + Shear | Sersic at 128 x 128 px,
1024 samples -- the profiles as user-written hip_body classes (run-time compiled interpreter kernel, VJP from forward-mode duals)
against the built-in kinds (specialised pair kernel; interpreter with GIGALENS_HIP_STATIC=0).

    python tools/dev/user_model_time.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gigalens_amd.model import PhysicalModel  # noqa: E402
from gigalens_amd.profile import LightProfile, MassProfile  # noqa: E402
from gigalens_amd.profiles.light.sersic import Sersic  # noqa: E402
from gigalens_amd.profiles.mass.shear import Shear  # noqa: E402
from gigalens_amd.profiles.mass.sis import SIS  # noqa: E402
from gigalens_amd.simulator import LensSimulator, SimulatorConfig  # noqa: E402


class UserSIS(MassProfile):
    _name, _params = "USER_SIS", ["theta_E", "center_x", "center_y"]
    hip_body = """
    template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {
      R dx = x - p[1], dy = y - p[2];
      R r = sqrt(dx * dx + dy * dy);
      fx = p[0] * dx / r;
      fy = p[0] * dy / r;
    }"""


class UserSersic(LightProfile):
    _name, _params, _amp = "USER_SERSIC", ["R_sersic", "n_sersic", "center_x", "center_y"], "Ie"
    hip_body = """
    template <class R> __device__ R light(R x, R y, const R* p) {
      R dx = x - p[2], dy = y - p[3];
      R r = sqrt(dx * dx + dy * dy);
      R bn = 1.9992f * p[1] - 0.3271f;
      return p[4] * exp(-bn * (pow(r / p[0], 1.f / p[1]) - 1.f));
    }"""


def run(phys, label, B=1024, iters=50):
    cfg = SimulatorConfig(delta_pix=0.065, num_pix=128)
    t0 = time.time()
    sim = LensSimulator(phys, cfg, bs=B)
    build = time.time() - t0
    r = np.random.default_rng(0)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    params = {"lens_mass": [dict(theta_E=t(r.uniform(0.9, 1.3, B)), center_x=t(r.normal(0, 0.05, B)), center_y=t(r.normal(0, 0.05, B))),
                            dict(gamma1=t(r.normal(0, 0.03, B)), gamma2=t(r.normal(0, 0.03, B)))],
              "source_light": [dict(R_sersic=t(r.uniform(0.2, 0.4, B)), n_sersic=t(r.uniform(1.0, 3.0, B)), center_x=t(r.normal(0, 0.1, B)),
                                    center_y=t(r.normal(0, 0.1, B)), Ie=t(r.uniform(20, 60, B)))]}
    packed = sim.pack(params)
    obs = sim.simulate({g: [{k: v[:1].expand(B) for k, v in d.items()} for d in lst] for g, lst in params.items()})[0].contiguous()
    m = sim._model
    for _ in range(5):
        m.loglike(packed, obs, None, None, 0.5, 100.0, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        m.loglike(packed, obs, None, None, 0.5, 100.0, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{label}: {ms:.3f} ms per forward+gradient of {B} samples (model construction {build:.1f} s)", flush=True)


if __name__ == "__main__":
    run(PhysicalModel([UserSIS(), Shear()], [], [UserSersic()]), "user-written SIS + Shear | user-written Sersic (run-time compiled interpreter)")
    run(PhysicalModel([SIS(), Shear()], [], [Sersic()]), "built-in kinds (default dispatch)")
    os.environ["GIGALENS_HIP_STATIC"] = "0"
    run(PhysicalModel([SIS(), Shear()], [], [Sersic()]), "built-in kinds through the interpreter (GIGALENS_HIP_STATIC=0)")
