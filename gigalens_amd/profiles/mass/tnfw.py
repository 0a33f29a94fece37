from gigalens_amd.profile import MassProfile


class TNFW(MassProfile):
    """Truncated NFW (reference: src/gigalens/tf/profiles/mass/tnfw.py:10-62)."""

    _name = "TNFW"
    _params = ["Rs", "alpha_Rs", "r_trunc", "center_x", "center_y"]
    _kind = 12
