from gigalens_amd.profile import LightProfile


class Sersic(LightProfile):
    """Spherical Sersic profile (reference: src/gigalens/tf/profiles/light/sersic.py:9-63)."""

    _name = "SERSIC"
    _params = ["R_sersic", "n_sersic", "center_x", "center_y"]
    _amp = "Ie"
    _kind = 16

    def __init__(self, use_lstsq=False):
        super().__init__(use_lstsq=use_lstsq)


class SersicEllipse(Sersic):
    """Elliptical Sersic profile (reference: sersic.py:66-80)."""

    _name = "SERSIC_ELLIPSE"
    _params = ["R_sersic", "n_sersic", "e1", "e2", "center_x", "center_y"]
    _amp = "Ie"
    _kind = 17


class CoreSersic(Sersic):
    """Core-Sersic profile, with the reference's expression as written (sersic.py:83-131)."""

    _name = "CORE_SERSIC"
    _params = ["R_sersic", "n_sersic", "Rb", "alpha", "gamma", "e1", "e2", "center_x", "center_y"]
    _amp = "Ie"
    _kind = 19
