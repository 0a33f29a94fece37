"""The wave-per-system eigen / pseudo-inverse solver of the linear-amplitude step (gigalens_amd/csrc/gl_eigh.h), run
through its serial context on the host against numpy's float64 ``pinv`` with the reference's cutoff
(tf/simulator.py:233: ``tf.linalg.pinv(XtX, rcond=1e-6)``)."""
from ctypes import POINTER, c_float, c_int

import numpy as np
import pytest


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def _solve(hostmath, A, rhs, rcond=1e-6, shortcut=False, want_path=False):
    """shortcut=False: the general path (eigenvectors, implicit QL); True: let the Sturm test pick the LDL^T short cut."""
    n = A.shape[0]
    A32 = np.ascontiguousarray(A, dtype=np.float32)
    r32 = np.ascontiguousarray(rhs, dtype=np.float32)
    co, ev = np.zeros(n, np.float32), np.zeros(n, np.float32)
    took = hostmath.hm_eigh_pinv_f32(_fp(A32), c_int(n), _fp(r32), c_float(rcond), c_int(int(shortcut)), _fp(co), _fp(ev))
    if want_path:
        return co, took, A32, r32
    return co, ev, A32, r32


def _gram(n, rows, cond, seed):
    r = np.random.default_rng(seed)
    U, _ = np.linalg.qr(r.normal(size=(rows, n)))
    W, _ = np.linalg.qr(r.normal(size=(n, n)))
    s = np.geomspace(1.0, 1.0 / np.sqrt(cond), n)
    X = (U * s) @ W.T * 37.0
    return X.T @ X, X


@pytest.mark.parametrize("n", [1, 2, 3, 8, 21, 66, 67, 79])
@pytest.mark.parametrize("cond", [1e1, 1e4])
def test_eigenvalues_and_solution_of_well_posed_systems(hostmath, n, cond):
    A, X = _gram(n, 4 * n + 3, cond, seed=n)
    y = np.random.default_rng(1).normal(size=X.shape[0])
    co, ev, A32, r32 = _solve(hostmath, A, X.T @ y)
    A64 = A32.astype(np.float64)
    w = np.linalg.eigvalsh(A64)
    scale = np.abs(np.diag(A64)).max()
    assert np.allclose(np.sort(ev) * scale, w, rtol=0, atol=2e-5 * w.max())
    want = np.linalg.pinv(A64, rcond=1e-6, hermitian=True) @ r32.astype(np.float64)
    # forward error of a float32 solve ~ cond * eps
    assert np.abs(co - want).max() <= 30 * cond * 6e-8 * np.abs(want).max() + 1e-6 * np.abs(want).max()
    # and the residual of the normal equations is at rounding level whatever the conditioning
    res = A64 @ co.astype(np.float64) - r32
    assert np.abs(res).max() <= 2e-5 * (np.abs(A64).sum(1).max() * np.abs(co).max())


def test_rank_deficient_system_uses_the_pseudo_inverse(hostmath):
    """Duplicate and all-zero basis images (a source that misses the image, two identical components): eigenvalues below
    rcond * max are cut, the minimum-norm solution is returned."""
    r = np.random.default_rng(3)
    X = r.normal(size=(200, 12))
    X[:, 5] = X[:, 2]          # duplicate column -> an exact zero eigenvalue
    X[:, 9] = 0.0              # empty basis image
    y = r.normal(size=200)
    A, rhs = X.T @ X, X.T @ y
    co, ev, A32, r32 = _solve(hostmath, A, rhs)
    want = np.linalg.pinv(A32.astype(np.float64), rcond=1e-6, hermitian=True) @ r32.astype(np.float64)
    assert np.abs(co - want).max() <= 2e-4 * np.abs(want).max()
    assert abs(co[9]) <= 1e-6 * np.abs(co).max() and abs(co[5] - co[2]) <= 1e-4 * abs(co[2])
    assert (np.abs(ev) <= 1e-6 * np.abs(ev).max()).sum() == 2


def test_degenerate_inputs(hostmath):
    co, ev, _, _ = _solve(hostmath, np.zeros((5, 5)), np.ones(5))
    assert np.all(co == 0.0)
    D = np.diag([4.0, 1.0, 9.0, 0.25])
    co, ev, _, _ = _solve(hostmath, D, np.array([4.0, 2.0, 18.0, 1.0]))
    assert np.allclose(co, [1.0, 2.0, 2.0, 4.0], rtol=1e-6)
    # an already tridiagonal matrix and a matrix with a huge dynamic range of entries
    T = np.diag([2.0] * 6) + np.diag([-1.0] * 5, 1) + np.diag([-1.0] * 5, -1)
    co, ev, A32, r32 = _solve(hostmath, T * 1e12, np.arange(6.0) * 1e12)
    assert np.allclose(co, np.linalg.solve(T, np.arange(6.0)), rtol=2e-5)


@pytest.mark.parametrize("n", [1, 2, 5, 21, 66, 79])
@pytest.mark.parametrize("cond", [1e1, 1e4])
def test_short_cut_equals_the_general_path_when_nothing_is_cut(hostmath, n, cond):
    """No eigenvalue under the cutoff: pinv = inverse, and the tridiagonal LDL^T between two reflector sweeps must give
    what the eigenvector path gives (both within the float32 forward error of the float64 pinv)."""
    A, X = _gram(n, 4 * n + 3, cond, seed=100 + n)
    y = np.random.default_rng(2).normal(size=X.shape[0])
    fast, took, A32, r32 = _solve(hostmath, A, X.T @ y, shortcut=True, want_path=True)
    slow, took0, _, _ = _solve(hostmath, A, X.T @ y, shortcut=False, want_path=True)
    assert took == 1 and took0 == 0
    want = np.linalg.pinv(A32.astype(np.float64), rcond=1e-6, hermitian=True) @ r32.astype(np.float64)
    tol = 30 * cond * 6e-8 * np.abs(want).max() + 1e-6 * np.abs(want).max()
    assert np.abs(fast - want).max() <= tol and np.abs(slow - want).max() <= tol


@pytest.mark.parametrize("case", ["duplicate", "zero_column", "cond_1e8", "negative"])
def test_short_cut_is_refused_when_the_cutoff_bites(hostmath, case):
    r = np.random.default_rng(7)
    X = r.normal(size=(300, 14))
    if case == "duplicate":
        X[:, 3] = X[:, 11]
    elif case == "zero_column":
        X[:, 6] = 0.0
    elif case == "cond_1e8":
        A, X = _gram(14, 80, 1e8, seed=9)
    A = X.T @ X
    if case == "negative":  # not a Gram matrix: a large negative eigenvalue (kept by pinv through |lambda|)
        A = A - 1.5 * np.outer(X[0], X[0]) * 300
    y = r.normal(size=X.shape[0])
    rhs = X.T @ y
    co, took, A32, r32 = _solve(hostmath, A, rhs, shortcut=True, want_path=True)
    assert took == 0
    want = np.linalg.pinv(A32.astype(np.float64), rcond=1e-6, hermitian=True) @ r32.astype(np.float64)
    # a float32 eigenvalue just above the cutoff (1e-6 lambda_max) is known to eps / rcond ~ 6 %: that is the accuracy of
    # the component of the solution along its direction, here and in any float32 pinv
    assert np.abs(co - want).max() <= (3e-2 if case == "cond_1e8" else 5e-3) * np.abs(want).max()


def test_spectrum_near_the_cutoff_is_left_to_the_eigenvalue_path(hostmath):
    """lambda_min = 2.5e-6 lambda_max: nothing is cut, but the spectrum comes within the 4x safety margin of the
    cutoff, so the cut decision is taken on converged eigenvalues, not on a Sturm count."""
    n = 12
    r = np.random.default_rng(11)
    Q, _ = np.linalg.qr(r.normal(size=(n, n)))
    lam = np.geomspace(1.0, 2.5e-6, n)
    A = (Q * lam) @ Q.T
    rhs = A @ r.normal(size=n)
    co, took, A32, r32 = _solve(hostmath, A, rhs, shortcut=True, want_path=True)
    assert took == 0 and np.isfinite(co).all()
    res = A32.astype(np.float64) @ co.astype(np.float64) - r32
    assert np.abs(res).max() <= 1e-4 * np.abs(r32).max()
