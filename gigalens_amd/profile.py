"""Plugin ABCs with the reference's names and attribute semantics (src/gigalens/profile.py:5-83).

A profile here is a *descriptor* -- a kind id plus the reference's parameter-name list -- that the
native library interprets; ``deriv`` / ``light`` keep the reference's call signature and evaluate on
the GPU through ``gl_profile_eval`` (include/gigalens_hip.h).
"""
from abc import ABC
from typing import List

from gigalens_amd import _native


class Parameterized(ABC):
    """src/gigalens/profile.py:5-21."""

    _name: str
    _params: List[str]
    _kind: int = 0  # gl_kind
    # A profile the library has no kind for may carry its body instead (the reference's extension point, profile.py:58-82, is a
    # subclass with a TensorFlow body): ONE HIP C++ function template over a number type R, compiled at run time,
    #     template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy);   (MassProfile;  p = params in order)
    #     template <class R> __device__ R    light(R x, R y, const R* p);                 (LightProfile; amplitude last)
    # It serves the plugin-level calls -- deriv / light on points, differentiable through torch.autograd (forward-mode duals
    # inside; include/gigalens_hip.h gl_user_profile_create) -- and, inside a PhysicalModel, LensSimulator's pixel kernels
    # (gl_model_create_user: the likelihood path's interpreter kernel is compiled with the body in it; <= 16 parameters).
    hip_body: str = ""

    def __init__(self, *args, **kwargs):
        self.name = self._name
        self.params = list(self._params)

    def __str__(self):
        return self.name

    # descriptor consumed by gl_model_create / gl_profile_eval
    def _component(self):
        return (self._kind, 0, 0)

    def _native_params(self):
        """Column names of this component's packed native row (== ``params`` unless amplitudes are solved linearly)."""
        return list(self.params)


class LightProfile(Parameterized, ABC):
    """src/gigalens/profile.py:24-60.  ``use_lstsq`` removes the amplitude from ``params``
    (profile.py:40-41) and turns ``light`` into the stack of unit-amplitude basis images."""

    _amp = ""

    def __init__(self, use_lstsq=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._use_lstsq = use_lstsq
        self.depth = 1
        if not self.use_lstsq:
            self.params.append(self._amp)

    @property
    def use_lstsq(self):
        return self._use_lstsq

    def _native_params(self):
        """With ``use_lstsq`` the amplitude leaves ``params`` (profile.py:40-41) but keeps its native column: the basis
        images are rendered with amplitude 1 and the solved coefficients are written there (LensSimulator.lstsq_simulate)."""
        return list(self.params) + ([self._amp] if self.use_lstsq else [])

    def light(self, x, y, **kwargs):
        """Surface brightness at ``(x, y)``; parameters broadcast on the last axis like the reference."""
        if self.use_lstsq:  # the unit-amplitude basis images, leading axis = depth (sersic.py:30-34, shapelets.py:61-62)
            return _native.profile_basis(self, x, y, kwargs)
        return _native.profile_eval(self, x, y, kwargs)[0]


class MassProfile(Parameterized, ABC):
    """src/gigalens/profile.py:63-82."""

    def deriv(self, x, y, **kwargs):
        """Deflection ``(alpha_x, alpha_y)`` at ``(x, y)``."""
        return _native.profile_eval(self, x, y, kwargs)

    def hessian(self, x, y, **kwargs):
        """``(f_xx, f_xy, f_yx, f_yy)`` as the reference resolves ``hessian``: derivative of ``deriv``
        (src/gigalens/tf/profile.py:9-27; the NFW / Shear / SIS / dPIE overrides equal it), the dPIS override as
        written (piemd.py:62-83).  Evaluated natively with forward-mode duals -- exact, no finite differences."""
        return _native.profile_hessian(self, x, y, kwargs)

    def convergence(self, x, y, **kwargs):
        """tf/profile.py:29-34."""
        f_xx, _, _, f_yy = self.hessian(x, y, **kwargs)
        return (f_xx + f_yy) / 2

    def shear(self, x, y, **kwargs):
        """tf/profile.py:36-42."""
        f_xx, f_xy, _, f_yy = self.hessian(x, y, **kwargs)
        return (f_xx - f_yy) / 2, f_xy
