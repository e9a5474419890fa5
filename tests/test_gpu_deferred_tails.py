"""The deferring period search infers the last halvings of a period from an interpolated root and
evaluates them one round later.  In the product build a wrong inference is rare; the test build
(libmcmc_amd_sabotage.so, -DMHD_DEFER_SABOTAGE) makes every third one wrong, which must change
nothing but the time taken."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wrong_inferences_are_caught_and_the_periods_searched_again():
    lib = os.path.join(ROOT, "mcmc_amd", "lib", "libmcmc_amd_sabotage.so")
    assert os.path.exists(lib), "build it: python -m mcmc_amd.build"
    env = dict(os.environ, MCMC_AMD_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sabotage_check.py")],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sabotage check ok" in r.stdout
