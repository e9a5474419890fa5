"""Many random models with strong low-velocity zones (velocities drawn unsorted from the whole
prior range) against the oracle.  The answer of the sequential search is a lattice point; which
one depends on the SIGN of the period equation at abscissae that come within 1e-7 relative of the
root, and for ill-conditioned models the double-precision value itself is only good to ~1e-9
relative in the position of its zero (an extended-precision evaluation of the oracle's own
formulas moves it by that much, scratch/secular_ld.py).  Two correct double-precision
implementations -- this library, the oracle, disba on another compiler -- therefore disagree by one
lattice step (<= 2e-6 relative) on a curve in some ten thousand.  What must hold: the two device
searches agree with each other bit for bit, status and evaluation counts equal the oracle's,
disagreements in value are that rare and that small -- and every one of them sits where the
period equation is ill-conditioned: where the zero of the oracle's double-precision evaluation and
the zero of the same formulas in extended precision (tests/helpers/secular_ld.py) lie more than
2e-10 relative apart (1e-13 for a well-conditioned model, tests/test_oracle_conditioning.py), so
that a real defect cannot hide in the allowance."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))

pytestmark = pytest.mark.gpu


def _zero_shift(oracle, m, nl, vpvs, T, c_a, c_b):
    """|zero of the double-precision period equation - zero in extended precision| / zero, near
    the two neighbouring answers c_a, c_b of the search for period T of model m."""
    import secular_ld as SL
    lt = oracle.layer_table(m, nl, vpvs)
    lo, hi = min(c_a, c_b) * (1 - 3e-6), max(c_a, c_b) * (1 + 3e-6)
    zd = SL.zero_crossing(lambda c: oracle.secular(*lt, T, c), lo, hi)
    zl = SL.zero_crossing(lambda c: float(SL.secular_ld(*lt, T, c)), lo, hi)
    return abs(zd - zl) / zl


def test_random_models_with_low_velocity_zones(oracle):
    import torch
    from mcmc_amd.engine import DeviceProblem
    B = 2048
    curves = differing = 0
    worst = 0.0
    shifts = []
    for r, (nl, P, lo, hi) in enumerate([(6, 100, 0.15, 1.59), (7, 100, 0.43, 0.98), (9, 100, 0.0, 0.9),
                                         (5, 20, 0.29, 0.86), (11, 64, 0.2, 1.2), (3, 40, 0.3, 1.3)]):
        rng = np.random.default_rng(500 + r)
        periods = np.flip(1 / np.logspace(lo, hi, P)).copy()
        vpvs = float(rng.uniform(1.5, 2.2))
        d = np.sort(rng.uniform(0.001, 0.15, (B, nl)), axis=1)
        v = rng.uniform(0.1, 2.0, (B, nl + 1))
        params = np.concatenate([d, v], axis=1)
        ost, ovel, olit = oracle.forward_batch(params, nl, vpvs, periods, nthreads=8)
        res = []
        for search in (2, 1):
            prob = DeviceProblem(nl, vpvs, periods, search=search)
            ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
            vel, st = prob.forward(torch.tensor(params, device=prob.device), n_evals=ne)
            res.append((vel.cpu().numpy(), st.cpu().numpy(), int(ne[0])))
        (v2, s2, l2), (v1, s1, l1) = res
        assert np.array_equal(v2, v1) and np.array_equal(s2, s1) and l2 == l1
        assert np.array_equal(s2, ost) and l2 == olit
        ok = ost == 0
        assert ok.sum() > B // 2
        diff = (v2[ok] != ovel[ok]).any(axis=1)
        curves += int(ok.sum())
        differing += int(diff.sum())
        if diff.any():
            worst = max(worst, float(np.abs(v2[ok][diff] / ovel[ok][diff] - 1).max()))
        for i in np.where(ok)[0][diff]:
            # the first period at which the curves part: the two answers are neighbouring lattice
            # points, and the equation that decides between them is ill-conditioned there
            k = int(np.argmax(v2[i] != ovel[i]))
            assert abs(v2[i, k] / ovel[i, k] - 1) <= 3.0e-6
            shift = _zero_shift(oracle, params[i], nl, vpvs, periods[k], v2[i, k], ovel[i, k])
            shifts.append(shift)
            assert shift > 2e-10, (r, int(i), k, shift)
    print("curves %d, differing %d, worst %.2e, zero shifts %s" %
          (curves, differing, worst, ["%.1e" % x for x in shifts]))
    assert differing <= max(1, curves // 5000), (differing, curves)
    assert worst <= 3.0e-6, worst
