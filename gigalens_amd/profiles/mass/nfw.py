from gigalens_amd.profile import MassProfile


class NFW(MassProfile):
    """Spherical NFW halo (reference: src/gigalens/tf/profiles/mass/nfw.py:5-52)."""

    _name = "NFW"
    _params = ["Rs", "alpha_Rs", "center_x", "center_y"]
    _kind = 3


class NFW_ELLIPSE(MassProfile):
    """NFW with an elliptical potential: the spherical profile on coordinates stretched by ``sqrt(1 -+ e)``
    (reference: src/gigalens/tf/profiles/mass/nfw.py:97-134)."""

    _name = "NFW_ELLIPSE"
    _params = ["Rs", "alpha_Rs", "e1", "e2", "center_x", "center_y"]
    _kind = 11
