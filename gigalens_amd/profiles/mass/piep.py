from gigalens_amd.profile import MassProfile


class DPIEP(MassProfile):
    """Dual pseudo-isothermal elliptical *potential*: the dPIS evaluated on coordinates stretched by
    ``sqrt(1 -+ e)`` (reference: src/gigalens/tf/profiles/mass/piep.py:18-55; it shares the name "dPIE")."""

    _name = "dPIE"
    _params = ["theta_E", "Ra", "Rs", "center_x", "center_y", "e1", "e2"]
    _kind = 8
