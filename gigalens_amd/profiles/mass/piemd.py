from gigalens_amd.profile import MassProfile


class DPIS(MassProfile):
    """Dual pseudo-isothermal sphere (reference: src/gigalens/tf/profiles/mass/piemd.py:21-94).

    ``deriv`` follows piemd.py:33-60 including the ``_sort_ra_rs`` ordering/clamping of the two radii and the
    0/0 = NaN on the centre.  Kernel maths: gigalens_amd/csrc/gl_dpie.h.
    """

    _name = "dPIS"
    _params = ["theta_E", "r_core", "r_cut", "center_x", "center_y"]
    _kind = 6
    _r_min = 0.0001


class DPIE(MassProfile):
    """Dual pseudo-isothermal elliptical mass distribution, Kassiola & Kovner (1993) as implemented in Lenstool
    (reference: src/gigalens/tf/profiles/mass/piemd.py:97-255)."""

    _name = "dPIE"
    _params = ["theta_E", "r_core", "r_cut", "center_x", "center_y", "e1", "e2"]
    _kind = 7
    _r_min = 0.0001
