"""Table-mode shapelets without the table: csrc/gl_shp.hip.h (ShpNodeGen) generates the basis values at the two nodes that
bracket a coordinate instead of loading them -- phi_0 by one exponential, the higher orders by the normalised three-term
recurrence, and the DIFFERENCE to the next node by its own recurrence (no cancellation in the slope of the interpolant).

This file restates that float32 arithmetic in numpy, operation by operation, and compares it over ALL 6000 nodes of the
reference's grid (tf/profiles/light/shapelets.py:39-40, phi_n(linspace(-5, 5, 6000))) with the float64 table the round-2 kernels
load (csrc/gl_host_tables.h build_shapelet_table).  The GPU parity tests (tests/test_gpu_parity.py, table-mode cases) compare
the kernel itself with the oracle; this test pins WHY that works and how close the generated nodes are."""
import numpy as np

F = np.float32
N_NODES, N_MAX = 6000, 10
C0 = 0.75112554446494248286  # pi^(-1/4)


def table_f64():
    x = -5.0 + 10.0 * np.arange(N_NODES) / (N_NODES - 1)
    tab = np.zeros((N_NODES, N_MAX + 1))
    hm2, hm1 = np.zeros(N_NODES), C0 * np.exp(-0.5 * x * x)
    tab[:, 0] = hm1
    for n in range(1, N_MAX + 1):
        h = np.sqrt(2.0 / n) * x * hm1 - (np.sqrt((n - 1.0) / n) * hm2 if n >= 2 else 0.0)
        tab[:, n] = h
        hm2, hm1 = hm1, h
    return x, tab


def fma(a, b, c):
    """float32 fused multiply-add (exact product and sum in float64, one rounding)."""
    return (a.astype(np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def generated_nodes():
    """ShpNodeGen::init / advance for every node index as the node below (t = 0)."""
    top = F(N_NODES - 1)
    h = F(10.0) / top
    fb = np.arange(N_NODES - 1).astype(F)                 # node below: 0 .. 5998
    u0 = ((fb - F(0.5) * top).astype(F) * h).astype(F)
    e0 = np.exp2((((u0 * u0).astype(F)) * F(-0.5 * 1.4426950408889634)).astype(F)).astype(F)
    z = fma(u0, -h, -(F(0.5) * h * h))
    em1 = fma(z, F(1.0 / 6.0), F(0.5))
    em1 = (fma(z, em1, F(1.0)) * z).astype(F)
    # monic scaling phi_n = K_n P_n, K_n = sqrt(2^n / n!):  P_{n+1} = u0 P_n - (n/2) P_{n-1},  Q_{n+1} = (u0 + h) Q_n - (n/2) Q_{n-1} + h P_n
    u1 = (u0 + h).astype(F)
    P = np.zeros((N_NODES - 1, N_MAX + 1), F)
    Q = np.zeros_like(P)
    P[:, 0] = (e0 * F(C0)).astype(F)
    Q[:, 0] = (P[:, 0] * em1).astype(F)
    for n in range(N_MAX):
        bn = F(0.5 * n)
        if n == 0:
            P[:, 1] = (u0 * P[:, 0]).astype(F)
            dn = (u1 * Q[:, 0]).astype(F)
        else:
            P[:, n + 1] = fma(u0, P[:, n], -(P[:, n - 1] * bn).astype(F))
            dn = fma(u1, Q[:, n], -(Q[:, n - 1] * bn).astype(F))
        Q[:, n + 1] = fma(np.full_like(dn, h), P[:, n], dn)
    import math
    K = np.array([math.sqrt(2.0 ** n / math.factorial(n)) for n in range(N_MAX + 1)])
    # (the kernel never multiplies by K_n: the amplitude matrix carries K_n1 K_n2; float64 here, to compare like with like)
    return P.astype(np.float64) * K, Q.astype(np.float64) * K


def test_generated_node_values_and_differences_match_the_float64_table():
    x, tab = table_f64()
    V, D = generated_nodes()
    amp = np.abs(tab).max(axis=0)                         # 0.75 .. 0.53
    err_v = np.abs(V.astype(np.float64) - tab[:-1]).max(axis=0) / amp
    assert err_v.max() <= 2e-6, err_v                     # the reference's own float32 node positions move phi_n by ~1e-6 of its amplitude
    dtab = tab[1:] - tab[:-1]                             # exact differences between neighbouring nodes
    err_d = np.abs(D.astype(np.float64) - dtab).max(axis=0) / np.abs(dtab).max(axis=0)
    assert err_d.max() <= 2e-5, err_d                     # slope of the interpolant: no cancellation (V(i+1) - V(i) in float32: ~3e-4)
    naive = np.abs((V[1:].astype(F) - V[:-1].astype(F)).astype(np.float64) - dtab[:-1]).max(axis=0) / np.abs(dtab).max(axis=0)
    assert naive.max() > 5 * err_d.max()                  # what the difference recurrence buys
    # continuity: node i reached from interval i - 1 (V + D) agrees with node i of interval i
    jump = np.abs((V[:-1] + D[:-1]).astype(np.float64) - V[1:]).max(axis=0) / amp
    assert jump.max() <= 3e-6, jump


def test_float32_table_differences_for_scale():
    """The table path of round 2 subtracts neighbouring float32 table entries: its slope noise, for comparison."""
    _, tab = table_f64()
    t32 = tab.astype(F)
    dtab = tab[1:] - tab[:-1]
    err = np.abs((t32[1:] - t32[:-1]).astype(np.float64) - dtab).max(axis=0) / np.abs(dtab).max(axis=0)
    assert 2e-5 < err.max() <= 1e-4, err  # 7e-5: the generated differences (<= 2e-5 above) are the more accurate ones
