from gigalens_amd.profile import LightProfile


class Shapelets(LightProfile):
    """2-D Gauss-Hermite (shapelet) basis (reference: src/gigalens/tf/profiles/light/shapelets.py:11-85).

    ``interpolate=True`` (the reference default) evaluates the basis by linear interpolation of a
    6000-node table on [-5, 5]; ``False`` uses the Hermite recurrence.  Amplitudes are named
    ``amp{i:0w}`` with ``w = len(str(n_layers))`` in the reference's (n1, n2) order (shapelets.py:26-46).
    """

    _name = "SHAPELETS"
    _params = ["beta", "center_x", "center_y"]
    _kind = 18

    def __init__(self, n_max, use_lstsq=False, interpolate=True):
        super().__init__(use_lstsq=use_lstsq)
        self.params = list(self._params)  # the reference re-adds numbered amplitudes instead of `_amp`
        self.n_max = int(n_max)
        self.n_layers = int((n_max + 1) * (n_max + 2) / 2)
        self.interpolate = bool(interpolate)
        self.N1, self.N2 = [], []
        n1 = n2 = 0
        width = len(str(self.n_layers))
        self._amp_names = []
        for i in range(self.n_layers):
            name = f"amp{str(i).zfill(width)}"
            self._amp_names.append(name)
            self.N1.append(n1)
            self.N2.append(n2)
            if n1 == 0:
                n1, n2 = n2 + 1, 0
            else:
                n1, n2 = n1 - 1, n2 + 1
        if not use_lstsq:
            self.params += self._amp_names
        self.depth = self.n_layers

    def _component(self):
        return (self._kind, self.n_max, 1 if self.interpolate else 0)

    def _native_params(self):
        return list(self._params) + list(self._amp_names)
