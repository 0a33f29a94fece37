from gigalens_amd.profile import MassProfile


class Shear(MassProfile):
    """External shear (reference: src/gigalens/tf/profiles/mass/shear.py:5-16)."""

    _name = "SHEAR"
    _params = ["gamma1", "gamma2"]
    _kind = 4
