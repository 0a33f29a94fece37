from gigalens_amd.profile import MassProfile


class SIE(MassProfile):
    """Singular isothermal ellipsoid (reference: src/gigalens/tf/profiles/mass/sie.py:5-42; core s == 0)."""

    _name = "SIE"
    _params = ["theta_E", "e1", "e2", "center_x", "center_y"]
    _kind = 2
