"""``SimulatorConfig`` / ``LensWCS`` / ``LensSimulator`` with the reference's surface
(src/gigalens/simulator.py:11-127, src/gigalens/tf/simulator.py:13-156), executing on HIP kernels.

Differences in HOW (not WHAT): the pixel grid is stored once as two ``(N,)`` device arrays instead of
``(N, bs)`` replicas (the reference's own ``# TODO: no need for batched grid``, tf/simulator.py:11);
ray-shooting, rendering, NaN->0 and the det(T) scale are one fused kernel; gradients come from
hand-written VJP kernels wrapped in a ``torch.autograd.Function``.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from gigalens_amd import _native


@dataclass
class SimulatorConfig:
    """src/gigalens/simulator.py:11-29 (field for field)."""

    delta_pix: float
    num_pix: int
    supersample: Optional[int] = 1
    kernel: Optional[Any] = None
    transform_pix2angle: Optional[np.array] = None
    pix_region: Optional[np.array] = None


class LensWCS:
    """Pixel <-> angle transform, src/gigalens/simulator.py:32-64, quirks included: ``pix2angle``
    applies T^T while the origin uses T (:50,53); ``transform_angle2pix`` inverts the un-supersampled
    T (:37-38); ``pixel_grid`` uses ``n_y`` for both axes (:62)."""

    def __init__(self, n, supersample=1, transform_pix2angle=None, pix_scale=1.0):
        if transform_pix2angle is None:
            transform_pix2angle = np.eye(2) * pix_scale
        transform_pix2angle = np.asarray(transform_pix2angle, dtype=np.float64)
        self.transform_pix2angle = transform_pix2angle / supersample
        self.transform_angle2pix = np.linalg.inv(transform_pix2angle)
        if isinstance(n, (int, np.integer)):
            self.n_x, self.n_y = int(n), int(n)
        else:
            self.n_x, self.n_y = n
        self.supersample = supersample
        low_x = -(self.n_x * self.supersample - 1) / 2
        low_y = -(self.n_y * self.supersample - 1) / 2
        self.radec_at_xy_0 = np.squeeze(self.transform_pix2angle @ np.array([[low_x], [low_y]]))

    def pix2angle(self, x, y):
        xy = np.stack([np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)])
        radec = np.tensordot(self.transform_pix2angle.T, xy, axes=1)
        ra = radec[0] + self.radec_at_xy_0[0]
        dec = radec[1] + self.radec_at_xy_0[1]
        return ra.astype(np.float32), dec.astype(np.float32)

    def angle2pix(self, ra, dec):
        radec = np.stack([np.asarray(ra, dtype=np.float64) - self.radec_at_xy_0[0],
                          np.asarray(dec, dtype=np.float64) - self.radec_at_xy_0[1]])
        return np.tensordot(self.transform_angle2pix.T, radec, axes=1).astype(np.float32)

    def pixel_grid(self):
        x = np.arange(self.n_y * self.supersample)
        X, Y = np.meshgrid(x, x)
        return self.pix2angle(X, Y)


class LensSimulatorInterface:
    """src/gigalens/simulator.py:67-127."""

    def __init__(self, phys_model, sim_config: SimulatorConfig, bs: int):
        self.phys_model = phys_model
        self.sim_config = sim_config
        self.bs = bs
        self.wcs = LensWCS(n=sim_config.num_pix, supersample=sim_config.supersample,
                           transform_pix2angle=sim_config.transform_pix2angle, pix_scale=sim_config.delta_pix)

    @staticmethod
    def get_coords(supersample: int, num_pix: int, transform_pix2angle):
        """src/gigalens/simulator.py:129-162 (a legacy helper no simulator of the reference calls): the grid with mean
        coordinate (0, 0).  The reference builds it with lenstronomy's ``PixelGrid`` (third party); its published rule is
        ``ra = ra_at_xy_0 + T00 ix + T01 iy``, ``dec = dec_at_xy_0 + T10 ix + T11 iy`` with ``ix`` along the columns,
        restated here.  Returns ``(ra_at_xy_0, dec_at_xy_0, img_x, img_y)`` with float32 ``(n, n)`` grids."""
        n = int(supersample) * int(num_pix)
        T = np.asarray(transform_pix2angle, dtype=np.float64)
        lo = np.arange(0, n, dtype=np.float32)
        lo = np.min(lo - np.mean(lo))
        ra0, dec0 = np.squeeze(T @ np.array([[lo], [lo]], dtype=np.float64))
        ix, iy = np.meshgrid(np.arange(n), np.arange(n))
        img_x = (ra0 + T[0, 0] * ix + T[0, 1] * iy).astype(np.float32)
        img_y = (dec0 + T[1, 0] * ix + T[1, 1] * iy).astype(np.float32)
        return ra0, dec0, img_x, img_y


class _SimulateFn(torch.autograd.Function):
    """autograd glue: forward = gl_simulate_fwd, backward = gl_simulate_bwd (both HIP)."""

    @staticmethod
    def forward(ctx, params, model):
        ctx.model = model
        ctx.save_for_backward(params)
        return model.simulate_fwd(params)

    @staticmethod
    def backward(ctx, grad_img):
        (params,) = ctx.saved_tensors
        return ctx.model.simulate_bwd(params, grad_img), None


class LensSimulator(LensSimulatorInterface):
    """Drop-in for ``gigalens.tf.simulator.LensSimulator`` (tf/simulator.py:13-156).

    Attributes kept from the reference: ``img_X``, ``img_Y`` (here ``(N,)``, not ``(N, bs)``), ``img_region``,
    ``region``, ``bs``, ``wcs``, ``supersample``, ``conversion_factor``.
    """

    def __init__(self, phys_model, sim_config: SimulatorConfig, bs: int, supersampled_kernel=None):
        """``sim_config.kernel`` with ``supersample > 1`` is brought to the supersampled grid like the reference does
        (tf/simulator.py:60-70: lenstronomy's ``subgrid_kernel(kernel, supersample, odd=True)``, restated in
        gigalens_amd/kernel_util.py -- third party, parity unpinned).  ``supersampled_kernel`` overrides it with a PSF
        already sampled on the supersampled grid."""
        super().__init__(phys_model, sim_config, bs)
        self.device = _native.device()
        self.supersample = int(sim_config.supersample)
        T = (np.eye(2) * sim_config.delta_pix if sim_config.transform_pix2angle is None
             else np.asarray(sim_config.transform_pix2angle, dtype=np.float64))
        # tf/simulator.py:27-29: det of the un-supersampled transform, in float32
        self.conversion_factor = float(np.float32(np.linalg.det(T.astype(np.float32))))
        ss = self.supersample
        Hs, Ws = self.wcs.n_x * ss, self.wcs.n_y * ss
        if sim_config.pix_region is None:  # tf/simulator.py:34-42
            region = np.ones((Hs, Ws), dtype=bool)
            img_region = np.ones((self.wcs.n_x, self.wcs.n_y))
        else:
            img_region = np.asarray(sim_config.pix_region)
            region = np.repeat(np.repeat(img_region, ss, axis=0), ss, axis=1).astype(bool)
        self._region_np = np.argwhere(region)  # == tf.where: row-major [row, col]
        img_X, img_Y = self.wcs.pix2angle(self._region_np[:, 1], self._region_np[:, 0])
        full = self._region_np.shape[0] == Hs * Ws
        pix_index = None if full else (self._region_np[:, 0] * Ws + self._region_np[:, 1]).astype(np.int32)
        self.region = torch.from_numpy(self._region_np).to(self.device)
        self.img_region = torch.from_numpy(img_region.astype(np.float32)).to(self.device)
        self.img_X = torch.from_numpy(img_X).to(self.device)
        self.img_Y = torch.from_numpy(img_Y).to(self.device)
        self.numPix = sim_config.num_pix
        self.depth = len(phys_model.lens_light) + len(phys_model.source_light)
        psf = None
        if supersampled_kernel is not None:
            psf = np.asarray(supersampled_kernel, dtype=np.float32)
        elif sim_config.kernel is not None:
            from gigalens_amd.kernel_util import subgrid_kernel
            psf = np.asarray(subgrid_kernel(np.asarray(sim_config.kernel), ss, odd=True), dtype=np.float32)
        self.kernel = psf
        bodies = []  # user-written profile bodies (profile.py `hip_body`): compiled into this model's kernels
        comps = ([_native.component_of(p, bodies) for p in phys_model.lenses]
                 + [_native.component_of(p, bodies) for p in phys_model.lens_light]
                 + [_native.component_of(p, bodies) for p in phys_model.source_light])
        self._model = _native.Model(comps, len(phys_model.lenses), len(phys_model.lens_light),
                                    len(phys_model.source_light), Hs, Ws, ss, img_X, img_Y, pix_index,
                                    self.conversion_factor, psf, bodies=bodies)
        for i, lens in enumerate(phys_model.lenses):
            if getattr(lens, "_kind", 0) == 10:  # series expansion: the field must live on THIS pixel list
                if lens.x is not self.img_X or lens._coefs is None:
                    lens.set_grid(self.img_X, self.img_Y)
                    lens.set_deriv()
                self._model.set_series(i, lens.series_var_0, lens._coefs)
            elif getattr(lens, "_kind", 0) == 9:  # galaxy catalogues of ScalingRelation lenses (fused dPIE-family kernels)
                self._model.set_catalogue(i, *lens._catalogue())
        self._layout = phys_model._packing()
        assert self._layout.P == self._model.P

    # -- packing between the reference's nested parameter dicts and the native [B, P] rows ---------
    def pack(self, params: Dict[str, List[Dict]]):
        return self._layout.pack(params, self.bs, self.device)

    def beta(self, x, y, lens_params: List[Dict]):
        """tf/simulator.py:72-78 on arbitrary points (plugin-level kernels, one per lens)."""
        beta_x, beta_y = x, y
        for lens, p, c in zip(self.phys_model.lenses, lens_params, self.phys_model.lenses_constants):
            f_xi, f_yi = lens.deriv(x, y, **p, **c)
            beta_x, beta_y = beta_x - f_xi, beta_y - f_yi
        return beta_x, beta_y

    def _lens_maps(self, x, y, lens_params):
        packed = self._pack_partial({"lens_mass": lens_params}) if not torch.is_tensor(lens_params) else lens_params
        series = [(i, l) for i, l in enumerate(self.phys_model.lenses) if getattr(l, "_kind", 0) == 10]
        if not series:
            return self._model.lens_maps(packed, x, y)
        # a series lens answers on its own grid whatever (x, y) it is handed (series_profile.py:76,83 TODO); the
        # reference then silently mixes grids -- here anything but the simulator's grid is refused
        xt = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        yt = torch.as_tensor(y, dtype=torch.float32, device=self.device)
        n = self.img_X.numel()
        if xt.numel() % n or yt.numel() % n or xt.shape[0] != n or yt.shape[0] != n or \
                not (torch.equal(xt.reshape(n, -1)[:, 0], self.img_X) and torch.equal(yt.reshape(n, -1)[:, 0], self.img_Y)):
            raise ValueError("a model with series-expansion lenses maps only the simulator's own grid (img_X, img_Y)")
        for i, lens in series:
            if lens._hcoefs is None:
                lens.set_hessian()
            if getattr(lens, "_hessian_bound", None) is not lens._hcoefs:
                self._model.set_series_hessian(i, lens._hcoefs)
                lens._hessian_bound = lens._hcoefs
        return self._model.lens_maps(packed, None, None)

    def magnification(self, x, y, lens_params: List[Dict]):
        """tf/simulator.py:80-91: ``1 / det(1 - Hessian)`` at ``(x, y)`` (trailing axis = batch)."""
        _, _, fxx, fxy, fyx, fyy = self._lens_maps(x, y, lens_params)
        return 1.0 / ((1 - fxx) * (1 - fyy) - fxy * fyx)

    def convergence(self, x, y, lens_params: List[Dict]):
        """tf/simulator.py:93-98 (sum of the lenses' ``(f_xx + f_yy) / 2``, tf/profile.py:30-34)."""
        _, _, fxx, _, _, fyy = self._lens_maps(x, y, lens_params)
        return 0.5 * (fxx + fyy)

    def shear(self, x, y, lens_params: List[Dict]):
        """tf/simulator.py:100-107: ``(gamma1, gamma2) = ((f_xx - f_yy)/2, f_xy)`` (tf/profile.py:36-42)."""
        _, _, fxx, fxy, _, fyy = self._lens_maps(x, y, lens_params)
        return 0.5 * (fxx - fyy), fxy

    def simulate(self, params, no_deflection=False):
        """tf/simulator.py:109-156.  Returns ``(bs, H, W)`` squeezed like ``tf.squeeze``."""
        packed = params if torch.is_tensor(params) else self.pack(params)
        if no_deflection:  # tf/simulator.py:125-126: sources are rendered on the un-deflected grid
            return self._parts(packed, 2 | 4)
        img = _SimulateFn.apply(packed, self._model)
        return torch.squeeze(img)

    def _parts(self, packed, parts):
        if packed.requires_grad:
            raise NotImplementedError("partial renders are forward-only helpers (no gradient)")
        return torch.squeeze(self._model.simulate_parts(packed, parts))

    def _pack_partial(self, params):
        """Partial renders only receive the groups they use; the unused columns are filled with a valid dummy (1)."""
        if torch.is_tensor(params):
            return params
        cols = []
        for g, i, name, const in self._layout.slots:
            grp = params.get(g)
            v = grp[i].get(name) if grp is not None and i < len(grp) else None
            v = const if v is None else v
            v = torch.as_tensor(1.0 if v is None else v, dtype=torch.float32, device=self.device)
            cols.append(v.reshape(-1).expand(self.bs) if v.numel() != self.bs else v.reshape(self.bs))
        return torch.stack(cols, dim=1)

    def simulate_source(self, params):
        """tf/simulator.py:242-269: the sources on the image grid, without lensing."""
        return self._parts(self._pack_partial(params), 4)

    def simulate_lens_light(self, params):
        """tf/simulator.py:271-297."""
        return self._parts(self._pack_partial(params), 2)

    def simulate_images(self, params):
        """tf/simulator.py:299-328: the lensed sources only."""
        return self._parts(self._pack_partial(params), 1 | 4)

    def lstsq_simulate(self, params, observed_image, err_map, return_stacked=False, return_coeffs=False,
                       no_deflection=False):
        """tf/simulator.py:158-240: render every ``use_lstsq`` light component as unit-amplitude basis images, solve
        ``coeffs = pinv(X^T X, rcond=1e-6) X^T Y`` per sample and return the best-fit image (default), the stack
        ``(bs, H, W, depth)`` or the coefficients ``(bs, depth)``.  All light profiles of the model must have been
        built with ``use_lstsq=True`` (the reference stacks every component, :183-201)."""
        if len(self._layout.linear) != self._model.num_linear():
            raise ValueError("lstsq_simulate needs every light profile built with use_lstsq=True")
        packed = params if torch.is_tensor(params) else self.pack(params)
        parts = (0 if no_deflection else 1) | 2 | 4
        if return_stacked:
            (stack,) = self._model.lstsq(packed, None, None, parts, want="stacked")
            return stack.permute(0, 2, 3, 1)
        obs = torch.as_tensor(observed_image, dtype=torch.float32, device=self.device).contiguous()
        err = torch.as_tensor(err_map, dtype=torch.float32, device=self.device).contiguous()
        if return_coeffs:
            return self._model.lstsq(packed, obs, err, parts, want="coeffs")[0]
        return torch.squeeze(self._model.lstsq(packed, obs, err, parts, want="image")[0])
