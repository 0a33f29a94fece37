"""ctypes binding of ``lib/libgigalens_hip.so`` (C ABI: ``include/gigalens_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream; every number on the hot path is
produced by the HIP kernels behind this ABI.  There is deliberately no fallback -- a missing library
or a failed call raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import numpy as np
import torch

_LIB_PATH = os.environ.get("GIGALENS_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                           "libgigalens_hip.so")
_lib = None

GL_FLAG_SHAPELETS_INTERPOLATE = 1


class NativeLibraryError(RuntimeError):
    pass


class gl_component(ctypes.Structure):
    _fields_ = [("kind", c_int32), ("iparam", c_int32), ("flags", c_uint32), ("reserved", c_int32)]


class gl_zcolumn(ctypes.Structure):
    _fields_ = [("param_col", c_int32), ("bijector", c_int32), ("prior", c_int32), ("a", c_float), ("b", c_float),
                ("lo", c_float), ("hi", c_float), ("log_norm", c_float)]


class gl_grid(ctypes.Structure):
    _fields_ = [
        ("height", c_int32), ("width", c_int32), ("supersample", c_int32), ("n_region", c_int32),
        ("grid_x", POINTER(c_float)), ("grid_y", POINTER(c_float)), ("pix_index", POINTER(c_int32)),
        ("conversion_factor", c_float), ("psf", POINTER(c_float)), ("psf_h", c_int32), ("psf_w", c_int32),
    ]


# every symbol include/gigalens_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "gl_model_create": (c_int, [POINTER(gl_component), c_int, c_int, c_int, POINTER(gl_grid), POINTER(c_void_p)]),
    "gl_model_create_user": (c_int, [POINTER(gl_component), c_int, c_int, c_int, POINTER(gl_grid), POINTER(ctypes.c_char_p), c_int,
                                     POINTER(c_void_p)]),
    "gl_model_destroy": (None, [c_void_p]),
    "gl_model_num_params": (c_int, [c_void_p]),
    "gl_model_param_offset": (c_int, [c_void_p, c_int]),
    "gl_model_num_pixels": (c_int64, [c_void_p]),
    "gl_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "gl_simulate_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gl_simulate_parts_fwd": (c_int, [c_void_p, c_void_p, c_int, c_uint32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gl_simulate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gl_loglike_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gl_model_set_prior": (c_int, [c_void_p, POINTER(gl_zcolumn), c_int, POINTER(c_float)]),
    "gl_logprob_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint32, c_void_p, c_size_t,
                                   c_void_p]),
    "gl_model_set_positions": (c_int, [c_void_p, c_int, POINTER(c_int32), POINTER(c_float), POINTER(c_float),
                                       POINTER(c_float), POINTER(c_float)]),
    "gl_positions_fwd_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "gl_series_precompute": (c_int, [c_int, c_int, POINTER(c_int32), c_void_p, POINTER(c_float), c_int, c_int, c_void_p,
                                     c_void_p, c_int64, c_void_p, c_void_p]),
    "gl_model_set_series": (c_int, [c_void_p, c_int, c_float, c_void_p]),
    "gl_series_eval": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                               c_void_p]),
    "gl_adam_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                               c_float, c_int64, c_void_p, c_void_p]),
    "gl_svi_sample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "gl_svi_grad": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "gl_hmc_kick_drift": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "gl_hmc_accept": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                              c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gl_profile_basis": (c_int, [POINTER(gl_component), c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p]),
    "gl_series_precompute_hessian": (c_int, [c_int, c_int, POINTER(c_int32), c_void_p, POINTER(c_float), c_int, c_int,
                                             c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "gl_series_hessian_eval": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "gl_model_set_series_hessian": (c_int, [c_void_p, c_int, c_void_p]),
    "gl_lens_maps": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "gl_model_num_linear": (c_int, [c_void_p]),
    "gl_model_linear_column": (c_int, [c_void_p, c_int]),
    "gl_lstsq_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "gl_lstsq_solve_flags": (c_int, [c_void_p, c_int, ctypes.POINTER(c_size_t)]),
    "gl_lstsq_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_uint32, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p]),
    "gl_model_set_catalogue": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int32), POINTER(c_float)]),
    "gl_scaled_eval": (c_int, [c_int, c_int, POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                               c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gl_scaled_hessian": (c_int, [c_int, c_int, POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                  c_void_p, c_int, c_void_p, c_void_p]),
    "gl_profile_eval": (c_int, [POINTER(gl_component), c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "gl_user_model_compile_count": (ctypes.c_longlong, []),
    "gl_user_profile_check": (c_int, [ctypes.c_char_p, c_int, c_int]),
    "gl_user_points_check": (c_int, [ctypes.c_char_p, c_int]),
    "gl_user_profile_create": (c_int, [ctypes.c_char_p, c_int, c_int, POINTER(c_void_p)]),
    "gl_user_profile_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "gl_user_profile_destroy": (None, [c_void_p]),
    "gl_profile_hessian": (c_int, [POINTER(gl_component), c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p]),
    "gl_kind_num_params": (c_int, [POINTER(gl_component)]),
    "gl_model_set_timing": (c_int, [c_void_p, c_int]),
    "gl_model_last_main_ms": (c_int, [c_void_p, POINTER(c_float)]),
    "gl_model_set_timing_stride": (c_int, [c_void_p, c_int]),
    "gl_model_timing_drain": (c_int, [c_void_p, POINTER(c_float), c_int, POINTER(c_int)]),
    "gl_model_last_main_kernel": (c_int, [c_void_p, ctypes.c_char_p, c_size_t]),
    "gl_model_launch_shape": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    "gl_last_error": (c_char_p, []),
    "gl_version": (c_char_p, []),
}


def lib_path():
    return _LIB_PATH


def lib():
    """Load the HIP library or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NativeLibraryError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). gigalens_amd has no CPU fallback.")
        try:
            h = ctypes.CDLL(_LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NativeLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def _check(rc):
    if rc != 0:
        raise NativeLibraryError(f"gigalens_hip error {rc}: {lib().gl_last_error().decode()}")


def _require_cuda(t, what):
    if not t.is_cuda:
        raise NativeLibraryError(f"{what} must live on the GPU (got device {t.device}); gigalens_amd has no CPU path")


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def component_of(profile, bodies=None):
    """gl_component of a profile.  ``bodies``: the list a model under construction collects user-written bodies in (`hip_body`
    profiles become GL_USER_MASS / GL_USER_LIGHT components pointing into it); None outside a model."""
    kind, iparam, flags = profile._component()
    if not kind and bodies is not None and getattr(profile, "hip_body", ""):
        from gigalens_amd.profile import LightProfile
        n = len(profile._native_params())
        if n > 16:
            raise NativeLibraryError(f"profile {profile.name!r}: a user-written profile inside a model takes at most 16 parameters, got {n}")
        if profile.hip_body not in bodies:
            bodies.append(profile.hip_body)
        # reserved = 1: the LAST parameter of a user-written light is its linear amplitude (LightProfile._amp, profile.py:24-60) --
        # the column the linear-amplitude solve sets to 1 for the basis image and writes the solved coefficient into
        is_light = isinstance(profile, LightProfile)
        return gl_component(20 if is_light else 13, n, bodies.index(profile.hip_body), 1 if (is_light and getattr(profile, "_amp", "")) else 0)
    if not kind:
        raise NativeLibraryError(
            f"profile {getattr(profile, 'name', type(profile).__name__)!r} has no gl_kind: user-defined deriv / light bodies "
            "written in Python cannot run here.  Give the class a `hip_body` -- one HIP C++ function template over a number type "
            "that the library compiles at run time (include/gigalens_hip.h gl_user_profile_create / gl_model_create_user; see "
            "INTEGRATION.md) -- and it serves deriv / light on points as well as LensSimulator's pixel kernels")
    return gl_component(kind, iparam, flags, 0)


# --------------------------------------------------------------------------------------------------
# user-written profile bodies (profile.py: `hip_body`), compiled once per (body, kind, parameter count) by hiprtc
# --------------------------------------------------------------------------------------------------
class _UserProfile:
    def __init__(self, body, is_light, n_params):
        h = c_void_p()
        _check(lib().gl_user_profile_create(body.encode(), int(is_light), int(n_params), ctypes.byref(h)))
        self._h, self.is_light, self.n_params = h, bool(is_light), int(n_params)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().gl_user_profile_destroy(self._h)
        except Exception:
            pass


_USER_PROFILES = {}


def user_profile_of(profile):
    from gigalens_amd.profile import LightProfile
    body = getattr(profile, "hip_body", None)
    if not body:
        return None
    device()
    is_light = isinstance(profile, LightProfile)
    key = (body, is_light, len(profile.params))
    if key not in _USER_PROFILES:
        _USER_PROFILES[key] = _UserProfile(body, is_light, len(profile.params))
    return _USER_PROFILES[key]


def user_points_check(body, n_params):
    """Compile a mass body into the point kernels (image-position likelihood, lens maps: nested duals) without a GPU."""
    _check(lib().gl_user_points_check(body.encode(), int(n_params)))


def user_profile_check(body, is_light, n_params):
    """Compile a body without a GPU (gl_user_profile_check); raises NativeLibraryError with the compiler's message."""
    _check(lib().gl_user_profile_check(body.encode(), int(bool(is_light)), int(n_params)))


class _UserEval(torch.autograd.Function):
    """Values of a compiled user body on points [n_pts, B] with parameters [B, n]; when a gradient is wanted the same launch
    returns the Jacobian with respect to (x, y, parameters) from forward-mode duals, and backward contracts it."""

    @staticmethod
    def forward(ctx, up, xb, yb, P):
        n_pts, B = xb.shape
        n_out = 1 if up.is_light else 2
        out0 = torch.empty_like(xb)
        out1 = None if up.is_light else torch.empty_like(xb)
        need = any(ctx.needs_input_grad[1:])
        jac = torch.empty((n_out, up.n_params + 2, n_pts, B), dtype=torch.float32, device=xb.device) if need else None
        _check(lib().gl_user_profile_eval(up._h, _ptr(xb), _ptr(yb), n_pts, B, 1, _ptr(P), _ptr(out0), _ptr(out1), _ptr(jac),
                                          _stream()))
        ctx.jac = jac
        return (out0,) if up.is_light else (out0, out1)

    @staticmethod
    def backward(ctx, *gs):
        J = ctx.jac
        g = torch.stack([torch.zeros_like(J[0, 0]) if gi is None else gi for gi in gs], dim=0)  # [n_out, n_pts, B]
        full = (g[:, None] * J).sum(dim=0)                                                           # [n + 2, n_pts, B]
        gP = full[2:].sum(dim=1).transpose(0, 1).contiguous() if J.shape[1] > 2 else None            # [B, n]
        return None, full[0], full[1], gP


def _user_eval(up, profile, x, y, kwargs):
    dev = device()
    names = list(profile.params)
    missing = [n for n in names if n not in kwargs]
    if missing:
        raise TypeError(f"{profile.name}: missing parameters {missing}")
    x = torch.as_tensor(x, dtype=torch.float32, device=dev)
    y = torch.as_tensor(y, dtype=torch.float32, device=dev)
    vals = [torch.as_tensor(kwargs[n], dtype=torch.float32, device=dev) for n in names]
    out_shape = torch.broadcast_shapes(x.shape, y.shape, *[v.shape for v in vals])
    B = out_shape[-1] if len(out_shape) else 1
    for n, v in zip(names, vals):
        if v.dim() > 1 and any(s != 1 for s in v.shape[:-1]):
            raise NativeLibraryError(f"{profile.name}.{n}: parameters may only vary along the last (batch) axis")
    if vals:
        P = torch.stack([v.reshape(-1)[-B:].expand(B) if v.numel() > 1 else v.reshape(()).expand(B) for v in vals], dim=1).contiguous()
    else:
        P = torch.zeros((B, 0), dtype=torch.float32, device=dev)
    xb = x.expand(out_shape).reshape(-1, B).contiguous()
    yb = y.expand(out_shape).reshape(-1, B).contiguous()
    return tuple(o.reshape(out_shape) for o in _UserEval.apply(up, xb, yb, P))


def device():
    if not torch.cuda.is_available():
        raise NativeLibraryError("no GPU visible: gigalens_amd's hot path only runs as HIP kernels on gfx950")
    return torch.device("cuda", torch.cuda.current_device())


# --------------------------------------------------------------------------------------------------
# plugin-level point evaluation (MassProfile.deriv / LightProfile.light)
# --------------------------------------------------------------------------------------------------
def _broadcast_points(profile, x, y, kwargs, names, dev):
    missing = [n for n in names if n not in kwargs]
    if missing:
        raise TypeError(f"{profile.name}: missing parameters {missing}")
    x = torch.as_tensor(x, dtype=torch.float32, device=dev)
    y = torch.as_tensor(y, dtype=torch.float32, device=dev)
    vals = [torch.as_tensor(kwargs[n], dtype=torch.float32, device=dev) for n in names]
    out_shape = torch.broadcast_shapes(x.shape, y.shape, *[v.shape for v in vals])
    B = out_shape[-1] if len(out_shape) else 1
    for n, v in zip(names, vals):
        if v.dim() > 1 and any(s != 1 for s in v.shape[:-1]):
            raise NativeLibraryError(f"{profile.name}.{n}: parameters may only vary along the last (batch) axis")
    P = torch.stack([v.reshape(-1)[-B:].expand(B) if v.numel() > 1 else v.reshape(()).expand(B) for v in vals],
                    dim=1).contiguous()
    xb = x.expand(out_shape).reshape(-1, B).contiguous()
    yb = y.expand(out_shape).reshape(-1, B).contiguous()
    return xb, yb, P, B, out_shape


def scaled_eval(profile, x, y, scales):
    """ScalingRelation.deriv (scaling_relation.py:61-70) through gl_scaled_eval."""
    dev = device()
    xb, yb, P, B, out_shape = _broadcast_points(profile, x, y, scales, list(profile.params), dev)
    base_kind, cols, table = profile._catalogue()
    if profile._dev_table is None or profile._dev_table.device != dev:
        profile._dev_table = torch.from_numpy(table).to(dev)
    col_arr = (c_int32 * 3)(*cols)
    out0, out1 = torch.empty_like(xb), torch.empty_like(xb)
    _check(lib().gl_scaled_eval(base_kind, table.shape[0], col_arr, _ptr(profile._dev_table), _ptr(xb), _ptr(yb),
                                xb.shape[0], B, 1, _ptr(P), P.shape[1], _ptr(out0), _ptr(out1), _stream()))
    return out0.reshape(out_shape), out1.reshape(out_shape)


def series_precompute(series, hessian=False):
    """MassSeries.set_deriv / set_hessian (series_profile.py:61-65) through gl_series_precompute[_hessian]: device
    ``[2, order+1, n_points]`` (alpha_x, alpha_y) or ``[3, order+1, n_points]`` (f_xx, f_xy, f_yy)."""
    dev = device()
    base_kind, cols, table, scales = series._series_inputs()
    x = torch.as_tensor(series.x, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    y = torch.as_tensor(series.y, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    tab = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32)).to(dev)
    sc = (c_float * len(scales))(*[float(v) for v in scales])
    out = torch.empty((3 if hessian else 2, series.order + 1, x.numel()), dtype=torch.float32, device=dev)
    fn = lib().gl_series_precompute_hessian if hessian else lib().gl_series_precompute
    _check(fn(int(base_kind), tab.shape[0], (c_int32 * 3)(*cols), _ptr(tab), sc, len(scales), series.order, _ptr(x),
              _ptr(y), x.numel(), _ptr(out), _stream()))
    return out


def series_hessian_eval(series, amplitude, var):
    """MassSeries.hessian (series_profile.py:83-89): ``(f_xx, f_xy, f_xy, f_yy)``, field shape + trailing batch axis."""
    dev = device()
    a = torch.as_tensor(amplitude, dtype=torch.float32, device=dev).reshape(-1)
    v = torch.as_tensor(var, dtype=torch.float32, device=dev).reshape(-1)
    B = max(a.numel(), v.numel())
    a, v = a.expand(B).contiguous(), v.expand(B).contiguous()
    n = series._hcoefs.shape[-1]
    out = torch.empty((3, n, B), dtype=torch.float32, device=dev)
    _check(lib().gl_series_hessian_eval(_ptr(series._hcoefs), series.order, n, B, _ptr(a), _ptr(v),
                                        float(series.series_var_0), _ptr(out), _stream()))
    shape = tuple(torch.as_tensor(series.x).shape)
    if shape and shape[-1] == B and len(shape) > 1:
        shape = shape[:-1]
        out = out.reshape(3, -1, B, B).diagonal(dim1=2, dim2=3)
    out = out.reshape((3,) + shape + (B,))
    return out[0], out[1], out[1], out[2]


def series_eval(series, amplitude, var):
    """MassSeries.deriv (series_profile.py:76-81): field shape + trailing batch axis."""
    dev = device()
    a = torch.as_tensor(amplitude, dtype=torch.float32, device=dev).reshape(-1)
    v = torch.as_tensor(var, dtype=torch.float32, device=dev).reshape(-1)
    B = max(a.numel(), v.numel())
    a, v = a.expand(B).contiguous(), v.expand(B).contiguous()
    n = series._coefs.shape[-1]
    o0 = torch.empty((n, B), dtype=torch.float32, device=dev)
    o1 = torch.empty_like(o0)
    _check(lib().gl_series_eval(_ptr(series._coefs), series.order, n, B, _ptr(a), _ptr(v), float(series.series_var_0),
                                _ptr(o0), _ptr(o1), _stream()))
    shape = tuple(torch.as_tensor(series.x).shape)
    if shape and shape[-1] == B and len(shape) > 1:  # grid given per batch element, as the reference's (N, bs) grids
        shape = shape[:-1]
        o0, o1 = o0.reshape(-1, B, B).diagonal(dim1=1, dim2=2), o1.reshape(-1, B, B).diagonal(dim1=1, dim2=2)
    return o0.reshape(shape + (B,)), o1.reshape(shape + (B,))


def scaled_hessian(profile, x, y, scales):
    """ScalingRelation.hessian (scaling_relation.py:72-83) through gl_scaled_hessian."""
    dev = device()
    xb, yb, P, B, out_shape = _broadcast_points(profile, x, y, scales, list(profile.params), dev)
    base_kind, cols, table = profile._catalogue()
    if profile._dev_table is None or profile._dev_table.device != dev:
        profile._dev_table = torch.from_numpy(table).to(dev)
    out = torch.empty((4,) + tuple(xb.shape), dtype=torch.float32, device=dev)
    _check(lib().gl_scaled_hessian(base_kind, table.shape[0], (c_int32 * 3)(*cols), _ptr(profile._dev_table), _ptr(xb),
                                   _ptr(yb), xb.shape[0], B, 1, _ptr(P), P.shape[1], _ptr(out), _stream()))
    return tuple(out[k].reshape(out_shape) for k in range(4))


def profile_hessian(profile, x, y, kwargs):
    """MassProfile.hessian (tf/profile.py:9-27): ``(f_xx, f_xy, f_yx, f_yy)``."""
    dev = device()
    up = user_profile_of(profile) if not profile._component()[0] else None
    if up is not None:  # a user-written body: the derivative of its deflection from the same duals (d fx / d(x, y), d fy / d(x, y))
        xb, yb, P, B, out_shape = _broadcast_points(profile, x, y, kwargs, list(profile.params), dev)
        n_pts = xb.shape[0]
        out0, out1 = torch.empty_like(xb), torch.empty_like(xb)
        jac = torch.empty((2, up.n_params + 2, n_pts, B), dtype=torch.float32, device=dev)
        _check(lib().gl_user_profile_eval(up._h, _ptr(xb), _ptr(yb), n_pts, B, 1, _ptr(P), _ptr(out0), _ptr(out1), _ptr(jac), _stream()))
        return tuple(jac[i, j].reshape(out_shape) for i, j in ((0, 0), (0, 1), (1, 0), (1, 1)))
    comp = component_of(profile)
    xb, yb, P, B, out_shape = _broadcast_points(profile, x, y, kwargs, list(profile.params), dev)
    out = torch.empty((4,) + tuple(xb.shape), dtype=torch.float32, device=dev)
    _check(lib().gl_profile_hessian(ctypes.byref(comp), _ptr(xb), _ptr(yb), xb.shape[0], B, 1, _ptr(P), _ptr(out),
                                    _stream()))
    return tuple(out[k].reshape(out_shape) for k in range(4))


def profile_eval(profile, x, y, kwargs):
    up = user_profile_of(profile) if not profile._component()[0] else None
    if up is not None:
        return _user_eval(up, profile, x, y, kwargs)
    dev = device()
    comp = component_of(profile)
    names = list(profile.params)
    missing = [n for n in names if n not in kwargs]
    if missing:
        raise TypeError(f"{profile.name}: missing parameters {missing}")
    x = torch.as_tensor(x, dtype=torch.float32, device=dev)
    y = torch.as_tensor(y, dtype=torch.float32, device=dev)
    vals = [torch.as_tensor(kwargs[n], dtype=torch.float32, device=dev) for n in names]
    out_shape = torch.broadcast_shapes(x.shape, y.shape, *[v.shape for v in vals])
    B = out_shape[-1] if len(out_shape) else 1
    for n, v in zip(names, vals):
        if v.dim() > 1 and any(s != 1 for s in v.shape[:-1]):
            raise NativeLibraryError(f"{profile.name}.{n}: parameters may only vary along the last (batch) axis")
    P = torch.stack([v.reshape(-1)[-B:].expand(B) if v.numel() > 1 else v.reshape(()).expand(B) for v in vals],
                    dim=1).contiguous()
    xb = x.expand(out_shape).reshape(-1, B).contiguous()
    yb = y.expand(out_shape).reshape(-1, B).contiguous()
    n_pts = xb.shape[0]
    out0 = torch.empty_like(xb)
    is_mass = comp.kind <= 12
    out1 = torch.empty_like(xb) if is_mass else None
    _check(lib().gl_profile_eval(ctypes.byref(comp), _ptr(xb), _ptr(yb), n_pts, B, 1, _ptr(P), _ptr(out0),
                                 _ptr(out1), _stream()))
    if is_mass:
        return out0.reshape(out_shape), out1.reshape(out_shape)
    return (out0.reshape(out_shape),)


def adam_update(x, grad, m, v, grad_scale, lr, b1, b2, eps, t, t_dev=None):
    """gl_adam_update: one fused optimiser step in place on ``x``, ``m``, ``v`` (contiguous float32 CUDA tensors)."""
    for name, a in (("x", x), ("grad", grad), ("m", m), ("v", v)):
        _require_cuda(a, name)
        if a.dtype != torch.float32 or not a.is_contiguous():
            raise NativeLibraryError(f"adam_update: {name} must be a contiguous float32 tensor")
    if not (x.numel() == grad.numel() == m.numel() == v.numel()):
        raise NativeLibraryError("adam_update: size mismatch")
    _check(lib().gl_adam_update(_ptr(x), _ptr(grad), _ptr(m), _ptr(v), x.numel(), float(grad_scale), float(lr),
                                float(b1), float(b2), float(eps), int(t), _ptr(t_dev), _stream()))


def svi_sample(mu, l_packed, eps, full_rank, diag_shift=1e-6):
    """gl_svi_sample: ``z = mu + L eps`` for the packed surrogate (float32 CUDA tensors)."""
    n, d = eps.shape
    z = torch.empty_like(eps)
    _check(lib().gl_svi_sample(_ptr(mu), _ptr(l_packed), d, int(bool(full_rank)), _ptr(eps), n, float(diag_shift), _ptr(z),
                               _stream()))
    return z


def svi_grad(l_packed, eps, logp, grad_z, full_rank, diag_shift=1e-6):
    """gl_svi_grad: the fused ``[ELBO, dELBO/dmu, dELBO/dl_packed]`` buffer."""
    n, d = eps.shape
    buf = torch.empty(1 + d + l_packed.numel(), dtype=torch.float32, device=eps.device)
    _check(lib().gl_svi_grad(_ptr(l_packed), d, int(bool(full_rank)), _ptr(eps), _ptr(logp), _ptr(grad_z), n,
                             float(diag_shift), _ptr(buf), _stream()))
    return buf


def hmc_kick_drift(p_in, grad, kick, z_in, sigma, eps, p_out, z_out):
    """gl_hmc_kick_drift on contiguous float32 CUDA tensors ``[n, d]`` (``sigma`` ``[d, d]``)."""
    n, d = p_in.shape
    _check(lib().gl_hmc_kick_drift(_ptr(p_in), _ptr(grad), float(kick), _ptr(z_in), _ptr(sigma), float(eps), n, d,
                                   _ptr(p_out), _ptr(z_out), _stream()))


def hmc_accept(z, g, lp, zn, gn, lpn, p0, pn, kick, scale_tril, uniforms, accept_prob):
    """gl_hmc_accept: Metropolis step of one transition, state updated in place."""
    n, d = z.shape
    _check(lib().gl_hmc_accept(_ptr(z), _ptr(g), _ptr(lp), _ptr(zn), _ptr(gn), _ptr(lpn), _ptr(p0), _ptr(pn), float(kick),
                               _ptr(scale_tril), _ptr(uniforms), n, d, _ptr(accept_prob), _stream()))


def profile_basis(profile, x, y, kwargs):
    """``light`` of a ``use_lstsq`` profile (gl_profile_basis): ``(depth,) + broadcast shape`` unit-amplitude images."""
    dev = device()
    comp = component_of(profile)
    names = list(profile.params)  # without the amplitudes (profile.py:40-41)
    missing = [n for n in names if n not in kwargs]
    if missing:
        raise TypeError(f"{profile.name}: missing parameters {missing}")
    x = torch.as_tensor(x, dtype=torch.float32, device=dev)
    y = torch.as_tensor(y, dtype=torch.float32, device=dev)
    vals = [torch.as_tensor(kwargs[n], dtype=torch.float32, device=dev) for n in names]
    out_shape = torch.broadcast_shapes(x.shape, y.shape, *[v.shape for v in vals])
    B = out_shape[-1] if len(out_shape) else 1
    for n, v in zip(names, vals):
        if v.dim() > 1 and any(s != 1 for s in v.shape[:-1]):
            raise NativeLibraryError(f"{profile.name}.{n}: parameters may only vary along the last (batch) axis")
    cols = {n: (v.reshape(-1)[-B:].expand(B) if v.numel() > 1 else v.reshape(()).expand(B)) for n, v in zip(names, vals)}
    one = torch.ones(B, dtype=torch.float32, device=dev)
    P = torch.stack([cols.get(n, one) for n in profile._native_params()], dim=1).contiguous()
    xb = x.expand(out_shape).reshape(-1, B).contiguous()
    yb = y.expand(out_shape).reshape(-1, B).contiguous()
    out = torch.empty((int(profile.depth),) + tuple(xb.shape), dtype=torch.float32, device=dev)
    _check(lib().gl_profile_basis(ctypes.byref(comp), _ptr(xb), _ptr(yb), xb.shape[0], B, 1, _ptr(P), _ptr(out),
                                  _stream()))
    return out.reshape((int(profile.depth),) + tuple(out_shape))


# --------------------------------------------------------------------------------------------------
# model handle
# --------------------------------------------------------------------------------------------------
class Model:
    """Owns one ``gl_model`` (immutable descriptor + grid on the current device)."""

    def __init__(self, components, n_lens, n_lens_light, n_src, height, width, supersample, grid_x, grid_y,
                 pix_index, conversion_factor, psf=None, bodies=None):
        self.device = device()
        L = lib()
        n = len(components)
        arr = (gl_component * max(n, 1))(*components)
        gx = np.ascontiguousarray(grid_x, dtype=np.float32)
        gy = np.ascontiguousarray(grid_y, dtype=np.float32)
        g = gl_grid()
        g.height, g.width, g.supersample, g.n_region = int(height), int(width), int(supersample), int(gx.size)
        g.grid_x = gx.ctypes.data_as(POINTER(c_float))
        g.grid_y = gy.ctypes.data_as(POINTER(c_float))
        if pix_index is not None:
            pi = np.ascontiguousarray(pix_index, dtype=np.int32)
            g.pix_index = pi.ctypes.data_as(POINTER(c_int32))
        g.conversion_factor = float(conversion_factor)
        if psf is not None:
            pk = np.ascontiguousarray(psf, dtype=np.float32)
            g.psf = pk.ctypes.data_as(POINTER(c_float))
            g.psf_h, g.psf_w = pk.shape
        h = c_void_p()
        with torch.cuda.device(self.device):
            if bodies:  # user-written profiles: the interpreter kernel is compiled with them now (seconds, once)
                barr = (ctypes.c_char_p * len(bodies))(*[b.encode() for b in bodies])
                _check(L.gl_model_create_user(arr, n_lens, n_lens_light, n_src, ctypes.byref(g), barr, len(bodies), ctypes.byref(h)))
            else:
                _check(L.gl_model_create(arr, n_lens, n_lens_light, n_src, ctypes.byref(g), ctypes.byref(h)))
        self._h = h
        self.P = L.gl_model_num_params(h)
        self.N = L.gl_model_num_pixels(h)
        self.offsets = [L.gl_model_param_offset(h, i) for i in range(n)]
        self.out_h, self.out_w = height // supersample, width // supersample
        self._ws = {}

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gl_model_destroy(h)

    def set_catalogue(self, component, base_kind, cols, table):
        """Attach the galaxy catalogue of a GL_SCALED lens (gl_model_set_catalogue)."""
        t = np.ascontiguousarray(table, dtype=np.float32)
        col_arr = (c_int32 * 3)(*[int(c) for c in cols])
        with torch.cuda.device(self.device):
            _check(lib().gl_model_set_catalogue(self._h, int(component), int(base_kind), int(t.shape[0]), col_arr,
                                                t.ctypes.data_as(POINTER(c_float))))
        self._ws = {}  # the workspace grows with the catalogue

    def lens_maps(self, params, x, y):
        """gl_lens_maps: ``x, y`` broadcastable to ``(..., B)``; returns ``(6, ...)`` = beta_x, beta_y, f_xx, f_xy, f_yx, f_yy."""
        params = self._params(params)
        B = params.shape[0]
        if x is None and y is None:  # the model's own grid (the only form series-expansion lenses accept)
            out = torch.empty((6, self.N, B), dtype=torch.float32, device=self.device)
            _check(lib().gl_lens_maps(self._h, _ptr(params), B, None, None, self.N, 0, _ptr(out), _stream()))
            return out
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        y = torch.as_tensor(y, dtype=torch.float32, device=self.device)
        shape = torch.broadcast_shapes(x.shape, y.shape, (B,))
        xb = x.expand(shape).reshape(-1, B).contiguous()
        yb = y.expand(shape).reshape(-1, B).contiguous()
        out = torch.empty((6,) + tuple(xb.shape), dtype=torch.float32, device=self.device)
        _check(lib().gl_lens_maps(self._h, _ptr(params), B, _ptr(xb), _ptr(yb), xb.shape[0], 1, _ptr(out), _stream()))
        return out.reshape((6,) + tuple(shape))

    def num_linear(self):
        return lib().gl_model_num_linear(self._h)

    def lstsq(self, params, obs, err, parts, want):
        """gl_lstsq_fwd: ``want`` in {"coeffs", "stacked", "image"} -> 1-tuple with that tensor."""
        params = self._params(params)
        B, D = params.shape[0], self.num_linear()
        nbytes = lib().gl_lstsq_workspace_bytes(self._h, B)
        ws = self._lstsq_ws if getattr(self, "_lstsq_ws", None) is not None and self._lstsq_ws.numel() >= nbytes else None
        if ws is None:
            ws = self._lstsq_ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        dev = params.device
        coeffs = torch.empty((B, D), dtype=torch.float32, device=dev) if want == "coeffs" else None
        stacked = torch.empty((B, D, self.out_h, self.out_w), dtype=torch.float32, device=dev) if want == "stacked" else None
        image = torch.empty((B, self.out_h, self.out_w), dtype=torch.float32, device=dev) if want == "image" else None
        _check(lib().gl_lstsq_fwd(self._h, _ptr(params), _ptr(obs), _ptr(err), B, int(parts), _ptr(coeffs),
                                  _ptr(stacked), _ptr(image), _ptr(ws), ws.numel(), _stream()))
        return ({"coeffs": coeffs, "stacked": stacked, "image": image}[want],)

    def lstsq_solve_flags(self, B):
        """Per-sample flags of the most recent linear solve on ``B`` samples: 0 = solved by the Cholesky attempt, 1 = by the
        eigenvalue solve (a view into the workspace; measurement aid, ``gl_lstsq_solve_flags``)."""
        off = c_size_t()
        _check(lib().gl_lstsq_solve_flags(self._h, B, ctypes.byref(off)))
        return self._lstsq_ws[off.value:off.value + 4 * B].view(torch.int32)

    def set_series(self, component, r0, coeffs):
        """Attach the coefficient field of a GL_SERIES lens (gl_model_set_series)."""
        _require_cuda(coeffs, "series coefficients")
        if coeffs.shape[-1] != self.N:
            raise NativeLibraryError(f"series field has {coeffs.shape[-1]} points, the model grid {self.N}")
        with torch.cuda.device(self.device):
            _check(lib().gl_model_set_series(self._h, int(component), float(r0), _ptr(coeffs.contiguous())))

    def set_series_hessian(self, component, coeffs):
        """Attach the Hessian field of a GL_SERIES lens (gl_model_set_series_hessian)."""
        _require_cuda(coeffs, "series Hessian coefficients")
        if coeffs.shape[-1] != self.N or coeffs.shape[0] != 3:
            raise NativeLibraryError(f"series Hessian field is {tuple(coeffs.shape)}, expected (3, order+1, {self.N})")
        with torch.cuda.device(self.device):
            _check(lib().gl_model_set_series_hessian(self._h, int(component), _ptr(coeffs.contiguous())))

    def set_prior(self, columns, const_row):
        """columns: list of (param_col, bijector, prior, a, b, lo, hi, log_norm); const_row: [P] floats."""
        arr = (gl_zcolumn * max(len(columns), 1))(*[gl_zcolumn(*c) for c in columns])
        cr = np.ascontiguousarray(const_row, dtype=np.float32)
        with torch.cuda.device(self.device):
            _check(lib().gl_model_set_prior(self._h, arr, len(columns), cr.ctypes.data_as(POINTER(c_float))))
        self.d_z = len(columns)

    def set_positions(self, xs, ys, exs, eys):
        """xs, ys, exs, eys: lists (one entry per image family) of 1-D arrays of equal length."""
        sizes = np.asarray([len(np.atleast_1d(x)) for x in xs], dtype=np.int32)
        cat = lambda L: np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(v, dtype=np.float32)) for v in L]))
        x, y, ex, ey = cat(xs), cat(ys), cat(exs), cat(eys)
        if not (x.size == y.size == ex.size == ey.size == int(sizes.sum())):
            raise NativeLibraryError("centroids / errors of a family must have the same length")
        fp = lambda a: a.ctypes.data_as(POINTER(c_float))
        with torch.cuda.device(self.device):
            _check(lib().gl_model_set_positions(self._h, len(sizes), sizes.ctypes.data_as(POINTER(c_int32)), fp(x), fp(y),
                                                fp(ex), fp(ey)))
        self.n_images = int(sizes.sum())
        self._ws = {}  # workspace layout changed

    def positions(self, params, want_grad):
        params = self._params(params)
        B = params.shape[0]
        ws = self._workspace(B)
        ll = torch.empty(B, dtype=torch.float32, device=params.device)
        chi2 = torch.empty_like(ll)
        grad = torch.empty_like(params) if want_grad else None
        _check(lib().gl_positions_fwd_bwd(self._h, _ptr(params), B, _ptr(ll), _ptr(chi2), _ptr(grad), _ptr(ws),
                                          ws.numel(), _stream()))
        return ll, chi2, grad

    def logprob(self, z, obs, err, mask, bg_rms, exp_time, want_grad, chi2_divisor=1.0, terms=1):
        _require_cuda(z, "z")
        if z.dtype != torch.float32 or z.dim() != 2 or z.shape[1] != self.d_z:
            raise NativeLibraryError(f"z must be float32 [B,{self.d_z}], got {z.dtype} {tuple(z.shape)}")
        z = z.contiguous()
        B = z.shape[0]
        ws = self._workspace(B)
        lp = torch.empty(B, dtype=torch.float32, device=z.device)
        ll = torch.empty_like(lp)
        chi2 = torch.empty_like(lp)
        grad = torch.empty_like(z) if want_grad else None
        _check(lib().gl_logprob_fwd_bwd(self._h, _ptr(z), _ptr(obs), _ptr(err), _ptr(mask), float(bg_rms),
                                        float(exp_time), B, _ptr(lp), _ptr(ll), _ptr(chi2), _ptr(grad),
                                        float(chi2_divisor), int(terms), _ptr(ws), ws.numel(), _stream()))
        return lp, ll, chi2, grad

    def set_timing(self, slots=1, stride=1):
        """Ring of ``slots`` HIP-event pairs around every ``stride``-th main-kernel launch (0 / False: off)."""
        _check(lib().gl_model_set_timing(self._h, int(slots)))
        _check(lib().gl_model_set_timing_stride(self._h, max(int(stride), 1)))
        self._timing_slots = int(slots)

    def timing_drain(self):
        """Durations [ms] of the main launches recorded since the last drain (oldest first)."""
        cap = max(int(getattr(self, "_timing_slots", 0)), 1)
        buf = (c_float * cap)()
        n = c_int()
        _check(lib().gl_model_timing_drain(self._h, buf, cap, ctypes.byref(n)))
        return list(buf[:n.value])

    def last_main_kernel(self):
        """Mangled symbol of the kernel the most recent main launch dispatched."""
        buf = ctypes.create_string_buffer(1024)
        _check(lib().gl_model_last_main_kernel(self._h, buf, len(buf)))
        return buf.value.decode()

    def partial_rows(self, B):
        """The per-(sample, chunk) partial rows ``[B, n_chunks, A]`` the most recent gradient call on ``B`` samples left in the
        workspace (a view; measurement aid, see ``gl_model_launch_shape``)."""
        chunk, nc, row, off = c_int(), c_int(), c_int(), c_size_t()
        _check(lib().gl_model_launch_shape(self._h, B, ctypes.byref(chunk), ctypes.byref(nc), ctypes.byref(row), ctypes.byref(off)))
        ws = self._workspace(B)
        n = B * nc.value * row.value
        return ws[off.value:off.value + 4 * n].view(torch.float32).view(B, nc.value, row.value)

    def last_main_ms(self):
        ms = c_float()
        _check(lib().gl_model_last_main_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def _workspace(self, B):
        ws = self._ws.get(B)
        if ws is None:
            nbytes = lib().gl_workspace_bytes(self._h, B)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            self._ws = {B: ws}  # keep only the latest batch size
        return ws

    def _params(self, params):
        _require_cuda(params, "params")
        if params.dtype != torch.float32 or params.dim() != 2 or params.shape[1] != self.P:
            raise NativeLibraryError(f"params must be float32 [B,{self.P}], got {params.dtype} {tuple(params.shape)}")
        return params.contiguous()

    def simulate_fwd(self, params):
        params = self._params(params)
        B = params.shape[0]
        ws = self._workspace(B)
        img = torch.empty((B, self.out_h, self.out_w), dtype=torch.float32, device=params.device)
        _check(lib().gl_simulate_fwd(self._h, _ptr(params), B, _ptr(img), _ptr(ws), ws.numel(), _stream()))
        return img

    def simulate_parts(self, params, parts):
        params = self._params(params)
        B = params.shape[0]
        ws = self._workspace(B)
        img = torch.empty((B, self.out_h, self.out_w), dtype=torch.float32, device=params.device)
        _check(lib().gl_simulate_parts_fwd(self._h, _ptr(params), B, int(parts), _ptr(img), _ptr(ws), ws.numel(),
                                           _stream()))
        return img

    def simulate_bwd(self, params, grad_img):
        params = self._params(params)
        B = params.shape[0]
        ws = self._workspace(B)
        grad_img = grad_img.to(torch.float32).expand(B, self.out_h, self.out_w).contiguous()
        grad = torch.empty_like(params)
        _check(lib().gl_simulate_bwd(self._h, _ptr(params), _ptr(grad_img), B, _ptr(grad), _ptr(ws), ws.numel(),
                                     _stream()))
        return grad

    def loglike(self, params, obs, err, mask, bg_rms, exp_time, want_grad):
        params = self._params(params)
        B = params.shape[0]
        ws = self._workspace(B)
        ll = torch.empty(B, dtype=torch.float32, device=params.device)
        chi2 = torch.empty(B, dtype=torch.float32, device=params.device)
        grad = torch.empty_like(params) if want_grad else None
        _check(lib().gl_loglike_fwd_bwd(self._h, _ptr(params), _ptr(obs), _ptr(err), _ptr(mask), float(bg_rms),
                                        float(exp_time), B, _ptr(ll), _ptr(chi2), _ptr(grad), _ptr(ws), ws.numel(),
                                        _stream()))
        return ll, chi2, grad
