from typing import Dict, List

import numpy as np
import torch

from gigalens_amd import _native
from gigalens_amd.profile import MassProfile

_BASE_KINDS = (6, 7, 8)  # dPIS, dPIE, dPIEP

# Member bodies of built-in kinds, in the form of a user-written `hip_body` (profile.py): what a population of such members is
# compiled from when it sits inside a PhysicalModel (`_member_loop_body` below).  Restated from the reference's formulas:
# tf/profiles/mass/sis.py:13-18, sie.py:14-50 (s_scale = 0), nfw.py:15-51 (acosh(1/x) = log((1 + sqrt(1 - x^2)) / x),
# acos(1/x) = atan(sqrt(x^2 - 1))).
_MEMBER_BODIES = {
    "SIE": """
template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {
  const R e1 = p[1], e2 = p[2];
  const R phi = atan2(e2, e1) * 0.5f;
  R c = sqrt(e1 * e1 + e2 * e2);
  if (value(c) > 0.9999f) c = R(0.9999f);
  const R q = (1.f - c) / (1.f + c), q2 = q * q;
  const R b = p[0] / sqrt((1.f + q2) / (2.f * q)) * sqrt((1.f + q2) * 0.5f);
  const R cs = cos(phi), sn = sin(phi);
  const R dx = x - p[3], dy = y - p[4];
  const R xr = dx * cs + dy * sn, yr = dy * cs - dx * sn;
  const R psi = sqrt(q2 * xr * xr + yr * yr), sq = sqrt(1.f - q2);
  const R ax = b / sq * atan(sq * xr / psi), ay = b / sq * atanh(sq * yr / psi);
  fx = ax * cs - ay * sn;  fy = ax * sn + ay * cs;
}
""",
    "SIS": """
template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {
  const R dx = x - p[1], dy = y - p[2];
  const R r = sqrt(dx * dx + dy * dy);
  if (value(r) == 0.f) { fx = R(0.f); fy = R(0.f); return; }
  const R a = p[0] / r;
  fx = a * dx;  fy = a * dy;
}
""",
    "NFW": """
template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {
  const R rho0 = p[1] / (4.f * p[0] * p[0] * (1.f - 0.693147180559945f));
  const R dx = x - p[2], dy = y - p[3];
  R r = sqrt(dx * dx + dy * dy);
  if (value(r) < 1e-7f) r = R(1e-7f);
  R rs = p[0];
  if (value(rs) < 1e-7f) rs = R(1e-7f);
  const R X = r / rs;
  R xg = X;
  if (value(xg) < 1e-6f) xg = R(1e-6f);
  R g = R(1.f);
  if (value(xg) < 1.f) { const R s = sqrt(1.f - xg * xg); g = log(xg * 0.5f) + log((1.f + s) / xg) / s; }
  else if (value(xg) > 1.f) { const R s = sqrt(xg * xg - 1.f); g = log(xg * 0.5f) + atan(s) / s; }
  const R a = 4.f * rho0 * rs * g / (X * X);
  fx = a * dx;  fy = a * dy;
}
""",
}


def _float_literal(v) -> str:
    """A float32 as a C++ literal that reads back to the same bits."""
    if not np.isfinite(v):
        raise ValueError(f"galaxy catalogue holds a non-finite value ({v})")
    txt = "%.9g" % float(v)
    return txt + ("f" if ("." in txt or "e" in txt) else ".f")


class ScalingRelation(MassProfile):
    """Population of galaxies following luminosity scaling relations
    (reference: src/gigalens/tf/profiles/mass/scaling_relation.py:6-70).

    Every galaxy ``g`` of ``galaxy_catalogue`` is one ``profile`` whose scaling parameters are
    ``(lum_g / lum_star) ** power * scale`` and whose other parameters are catalogue columns; the sampled parameters
    are the scales (``params == scaling_params``).  The native kernels sum the population per pixel inside the fused
    ray-shooting pass (gl_dpie.h), so no ``(x, y, b, g)`` tensor is ever formed and ``chunk_size`` (the reference's
    memory bound, :33-36,46) is accepted and unused.

    The fused kernels are built for the dPIE family (``DPIS``, ``DPIE``, ``DPIEP``) with scaling parameters among the
    amplitude and the two radii -- what ``DPIESubhalo`` (dpie_subhalo.py) uses.  Any other base profile -- another built-in
    kind, other scaling parameters, a user-written ``hip_body`` -- is served at the plugin level the way the reference does it
    (:61-83): ``deriv`` / ``hessian`` evaluate the base profile on ``chunk_size`` galaxies at a time and sum.  Inside a
    ``PhysicalModel`` such a population becomes ONE run-time compiled lens (csrc/gl_user.hip): the member loop around the base
    profile's body -- its own ``hip_body``, or the restated body of a built-in kind (``_MEMBER_BODIES``: SIS, SIE, NFW) -- with the
    catalogue as constants of the program and the gradient with respect to the scales from forward-mode duals.
    """

    _kind = 9

    def __init__(self, profile: MassProfile, scaling_params: List, lum_star: float,
                 scaling_params_power: Dict[str, float], galaxy_catalogue: Dict[str, List], chunk_size=None, **kwargs):
        self.profile = profile
        self._name = f"Scaled-{profile.name}"
        self._params = list(scaling_params)
        self.scaling_params = list(scaling_params)
        super().__init__(**kwargs)
        if not self.scaling_params:
            raise ValueError("ScalingRelation needs at least one scaling parameter")
        unknown = [p for p in self.scaling_params if p not in profile.params]
        if unknown:
            raise ValueError(f"{profile.name} has no parameters {unknown}")
        slots = list(profile.params[:3])  # dPIE family: amplitude, inner radius, outer radius
        # fused kernels: dPIE family with scales among its first three parameters; everything else: the generic plugin-level sum
        self._generic = getattr(profile, "_kind", 0) not in _BASE_KINDS or any(p not in slots for p in self.scaling_params)
        self.lum_star = float(lum_star)
        self.power = {k: float(v) for k, v in scaling_params_power.items()}
        self.galaxy_cat = galaxy_catalogue
        self._luminosities = np.asarray(galaxy_catalogue["lum"], dtype=np.float32)
        self.n_galaxy = len(self._luminosities)
        self.chunk_size = self.n_galaxy if chunk_size is None else chunk_size
        constants = getattr(profile, "constants", [])
        self.not_scaling_params = [p for p in list(profile.params) + list(constants) if p not in self.scaling_params]
        missing = [p for p in self.not_scaling_params if p not in galaxy_catalogue]
        if missing:
            raise KeyError(f"galaxy catalogue lacks the columns {missing}")
        self._slots = slots
        self._dev_table = None
        if self._generic:
            # inside a PhysicalModel such a population is ONE run-time compiled lens: the member loop around the base profile's
            # body (its own `hip_body`, or the restated body of a built-in kind), catalogue and (L/L*)^power factors as constants
            # of the program.  No body: the population stays a plugin-level object (deriv / hessian on points).
            member = getattr(profile, "hip_body", "") or _MEMBER_BODIES.get(profile.name, "")
            self._kind = 0
            self.hip_body = self._member_loop_body(member) if member else ""

    def _member_loop_body(self, member: str) -> str:
        """`hip_body` of the population (scaling_relation.py:61-70 as one device function): for every galaxy the base profile's
        body on parameters ``scale * (L/L*)^power`` (float32 product, :52-55) or the catalogue column, summed."""
        import re
        names = list(self.profile.params)
        unscaled = self._unscaled()
        rows = []
        for g in range(self.n_galaxy):
            vals = [unscaled[n][g] if n in self.scaling_params else np.float32(np.asarray(self.galaxy_cat[n], dtype=np.float32)[g])
                    for n in names]
            rows.append("{" + ", ".join(_float_literal(v) for v in vals) + "}")
        q = "\n".join(
            f"    q[{k}] = p[{self.scaling_params.index(n)}] * sr_cat[g][{k}];" if n in self.scaling_params
            else f"    q[{k}] = R(sr_cat[g][{k}]);" for k, n in enumerate(names))
        return (re.sub(r"\bderiv\b", "sr_member_deriv", member)
                + f"\n__device__ const float sr_cat[{self.n_galaxy}][{len(names)}] = {{\n  " + ",\n  ".join(rows) + "\n};\n"
                + "template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {\n"
                + "  fx = R(0.f);  fy = R(0.f);\n#pragma unroll 1\n"
                + f"  for (int g = 0; g < {self.n_galaxy}; ++g) {{\n    R q[{len(names)}], ax, ay;\n" + q
                + "\n    sr_member_deriv<R>(x, y, q, ax, ay);\n    fx += ax;  fy += ay;\n  }\n}\n")

    # -- what the native library consumes (include/gigalens_hip.h: gl_model_set_catalogue) --------------------------
    def _component(self):
        if self._generic and not self.hip_body:
            raise _native.NativeLibraryError(
                f"ScalingRelation over {self.profile.name} with scales {self.scaling_params}: inside a PhysicalModel the fused kernels "
                "sum populations of dPIS, dPIE, dPIEP members scaled in amplitude and radii, and the run-time compiled member loop "
                f"serves bases that carry a `hip_body` (user-written) or one of {sorted(_MEMBER_BODIES)}; this population is served at "
                "the plugin level only (deriv / hessian on points)")
        if self._generic:
            return (0, 0, 0)  # -> _native.component_of: a GL_USER_MASS component compiled from self.hip_body
        return (self._kind, len(self.scaling_params), 0)

    def _unscaled(self):
        """``(L/L*)^power`` in float32 (scaling_relation.py:27-30,52-55)."""
        lum = torch.from_numpy(self._luminosities)
        return {k: ((lum / torch.tensor(self.lum_star, dtype=torch.float32))
                    ** torch.tensor(self.power[k], dtype=torch.float32)).numpy() for k in self.scaling_params}

    def _catalogue(self):
        """(base_kind, scale_col[3], table [G,7]) in the row layout of gl_dpie.h."""
        t = np.zeros((self.n_galaxy, 7), dtype=np.float32)
        unscaled = self._unscaled()
        cols = []
        for k, name in enumerate(self._slots):
            if name in self.scaling_params:
                t[:, k] = unscaled[name]
                cols.append(self.scaling_params.index(name))
            else:
                t[:, k] = np.asarray(self.galaxy_cat[name], dtype=np.float32)
                cols.append(-1)
        t[:, 3] = np.asarray(self.galaxy_cat["center_x"], dtype=np.float32)
        t[:, 4] = np.asarray(self.galaxy_cat["center_y"], dtype=np.float32)
        if "e1" in self.profile.params:
            t[:, 5] = np.asarray(self.galaxy_cat["e1"], dtype=np.float32)
            t[:, 6] = np.asarray(self.galaxy_cat["e2"], dtype=np.float32)
        return self.profile._kind, cols, t

    def deriv(self, x, y, **scales):
        """scaling_relation.py:61-70."""
        if self._generic:
            return self._sum_over_galaxies(self.profile.deriv, 2, x, y, scales)
        return _native.scaled_eval(self, x, y, scales)

    def hessian(self, x, y, **scales):
        """scaling_relation.py:72-83: the sum of the members' Hessians as the base profile resolves ``hessian``."""
        if self._generic:
            return self._sum_over_galaxies(self.profile.hessian, 4, x, y, scales)
        return _native.scaled_hessian(self, x, y, scales)

    # -- any base profile, at the plugin level: (x, y, b) -> (x, y, b, g) chunk by chunk like the reference (:61-70), with the
    # (b, g) pair folded into the base profile's batch axis (its parameters vary along the last axis only)
    def _sum_over_galaxies(self, fn, n_out, x, y, scales):
        dev = _native.device()
        missing = [k for k in self.scaling_params if k not in scales]
        if missing:
            raise TypeError(f"{self.name}: missing parameters {missing}")
        x = torch.as_tensor(x, dtype=torch.float32, device=dev)
        y = torch.as_tensor(y, dtype=torch.float32, device=dev)
        sc = {k: torch.as_tensor(scales[k], dtype=torch.float32, device=dev) for k in self.scaling_params}
        shape = torch.broadcast_shapes(x.shape, y.shape, *[v.shape for v in sc.values()])
        B = shape[-1] if len(shape) else 1
        xb, yb = x.expand(shape).reshape(-1, B), y.expand(shape).reshape(-1, B)
        unscaled = {k: torch.from_numpy(v).to(dev) for k, v in self._unscaled().items()}
        out = [torch.zeros_like(xb) for _ in range(n_out)]
        for pos in range(0, self.n_galaxy, max(1, int(self.chunk_size))):
            g = slice(pos, min(pos + max(1, int(self.chunk_size)), self.n_galaxy))
            ng = g.stop - g.start
            kw = {}
            for k in list(self.profile.params) + [c for c in getattr(self.profile, "constants", []) if c not in self.profile.params]:
                if k in self.scaling_params:  # scale_b * (L_g / L*)^power  ->  index b * ng + g
                    s_b = sc[k].reshape(-1)[-B:].expand(B) if sc[k].numel() > 1 else sc[k].reshape(()).expand(B)
                    kw[k] = (s_b[:, None] * unscaled[k][g][None, :]).reshape(B * ng)
                else:
                    col = torch.as_tensor(np.asarray(self.galaxy_cat[k], dtype=np.float32)[g], device=dev)
                    kw[k] = col[None, :].expand(B, ng).reshape(B * ng)
            res = fn(xb[:, :, None].expand(-1, B, ng).reshape(-1, B * ng), yb[:, :, None].expand(-1, B, ng).reshape(-1, B * ng), **kw)
            for o, r in zip(out, res):
                o += r.reshape(-1, B, ng).sum(dim=-1)
        return tuple(o.reshape(shape) for o in out)
