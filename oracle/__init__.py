"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  The product
package (``gigalens_amd``) never imports it and fails loudly when its HIP
library is missing.

Parity status (see DESIGN.md "Oracle"): the reference is Python on TensorFlow /
TFP / lenstronomy, none of which is installed in the build container, so the
reference itself cannot be executed.  The restatement in ``ref_torch.py``
follows the reference's TF substrate line by line (file:line cited on every
function) and is pinned by (i) the reference's one executable known-answer test
(``tests/test_profiles.py:17-26``), (ii) the reference's own test recipes
re-targeted at independently restated published formulas (``published.py``),
(iii) closed-form identities between independent code paths, (iv) the one piece of
reference-GENERATED data available -- the tf-demo image ``src/gigalens/assets/demo.npy``,
which the restatement explains at reduced chi^2 = 0.9989 from the notebook's truth
parameters (tests/test_reference_demo.py).  Third-party
pieces that could not be pinned (TFP bijector layout, lenstronomy
``subgrid_kernel`` for supersample>1) are marked "parity unpinned" where used.
"""
