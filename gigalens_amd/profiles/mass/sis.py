from gigalens_amd.profile import MassProfile


class SIS(MassProfile):
    """Singular isothermal sphere (reference: src/gigalens/tf/profiles/mass/sis.py:5-17)."""

    _name = "SIS"
    _params = ["theta_E", "center_x", "center_y"]
    _kind = 5
