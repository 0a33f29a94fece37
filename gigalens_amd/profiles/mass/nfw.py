from gigalens_amd.profile import MassProfile


class NFW(MassProfile):
    """Spherical NFW halo (reference: src/gigalens/tf/profiles/mass/nfw.py:5-52)."""

    _name = "NFW"
    _params = ["Rs", "alpha_Rs", "center_x", "center_y"]
    _kind = 3
