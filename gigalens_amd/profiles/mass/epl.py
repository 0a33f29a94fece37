from gigalens_amd.profile import MassProfile


class EPL(MassProfile):
    """Elliptical power law (reference: src/gigalens/tf/profiles/mass/epl.py:5-57).

    ``niter`` caps the Tessore-Metcalf angular series exactly as in the reference (epl.py:15,51).
    """

    _name = "EPL"
    _params = ["theta_E", "gamma", "e1", "e2", "center_x", "center_y"]
    _kind = 1

    def __init__(self, niter=50):
        super().__init__()
        self.niter = int(niter)

    def _component(self):
        return (self._kind, self.niter, 0)
