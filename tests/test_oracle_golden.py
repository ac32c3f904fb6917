"""The oracle's pin: it must reproduce the outputs the UNMODIFIED reference produced for the committed
golden inputs (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from helpers import GOLDEN_CASES, load_golden, oracle_run


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_oracle_reproduces_reference_output(name):
    g = load_golden(name)
    out = oracle_run(g)
    exp = g['expected']
    assert out.shape == exp.shape and out.dtype == np.float64
    # float64 end to end; the only differences are libm ulps (numpy SIMD log/exp vs glibc)
    tol = 1e-11 * np.abs(exp).max()
    assert np.abs(out - exp).max() <= tol


@pytest.mark.parametrize('name', ['c1_baryonify', 'lowz_baryonify', 'rdelta_baryonify'])
def test_golden_mass_conservation(name):
    g = load_golden(name)
    assert np.isclose(g['expected'].sum(), g['map_in'].sum())     # HealpixRunner.py:344-346
