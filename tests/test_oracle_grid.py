"""CPU: the grid oracle (oracle/grid.py + bfg_oracle.c) against the reference-generated grid fixtures (the oracle's
pin for the regular-grid path), and the numpy P(k) restatement against independent properties."""
import numpy as np
import pytest

import helpers as H
from oracle import grid as G


@pytest.mark.parametrize('name', H.GRID_RUNNER_CASES)
def test_grid_oracle_matches_reference(name):
    g = H.load_grid_golden(name)
    out = H.grid_oracle_run(g)
    exp = g['expected']
    assert out.shape == exp.shape
    assert np.abs(out - exp).max() <= 1e-13 * np.abs(exp).max()        # measured: 0 (baryonify), < 1e-16 (paint)
    if g['kind'] == 'baryonify':
        assert np.isclose(out.sum(), g['map_in'].sum())


@pytest.mark.parametrize('name', ['grid2d_regrid', 'grid3d_regrid'])
def test_regrid_pixels_oracle_is_exact(name):
    g = H.load_grid_golden(name)
    grid = G.regrid_pixels(np.zeros((g['npix'],) * g['ndim']), g['pos'], g['val'])
    assert np.array_equal(grid, g['expected'])
    assert np.isclose(grid.sum(), g['val'].sum(), rtol=1e-13)


@pytest.mark.parametrize('name', ['grid2d_make_map', 'grid3d_make_map'])
def test_make_map_oracle_is_exact(name):
    g = H.load_grid_golden(name)
    out = G.make_map([g['xyz'][:, d] for d in range(g['ndim'])], g['mass'], float(g['L']), int(g['N_grid']))
    assert np.array_equal(out, g['expected'])
    # against numpy itself, on the same edges
    edges = np.linspace(0, float(g['L']), int(g['N_grid']) + 1)
    ref = np.histogramdd(g['xyz'][:, :g['ndim']], bins=(edges,) * g['ndim'], weights=g['mass'])[0]
    assert np.array_equal(out, ref)


def test_power_spectrum_restatement_properties():
    rng = np.random.default_rng(0)
    N, L, Nk = 16, 40.0, 7
    Map = rng.normal(size=(N, N, N))
    k_cen, Pk, k_c = G.power_spectrum(Map, L, Nk)
    # every mode between the fundamental and the Nyquist frequency is counted exactly once
    klin = 2 * np.pi / L * np.fft.fftfreq(N, 1.0 / N)
    k = np.sqrt(klin[:, None, None] ** 2 + klin[None, :, None] ** 2 + klin[None, None, :] ** 2)
    kf, kn = 2 * np.pi / L, 2 * np.pi / L * N / 2
    assert k_c.sum() == int(((k >= kf) & (k < kn)).sum())
    assert np.all((k_cen >= kf) & (k_cen <= kn))
    # white noise of unit variance: <|F|^2> = N^3
    assert abs(np.average(Pk, weights=k_c) / N ** 3 - 1) < 0.1
    # a single plane wave lands in one bin
    x = np.arange(N)
    wave = np.cos(2 * np.pi * 3 * x / N)[:, None, None] * np.ones((N, N, N))
    _, Pw, cw = G.power_spectrum(wave, L, Nk)
    b = int(np.floor((3 * kf - kf) / ((kn - kf) / Nk)))
    assert np.argmax(Pw) == b and np.isclose(Pw[b] * cw[b], 2 * (N ** 3 / 2) ** 2)


@pytest.mark.parametrize('name', H.SNAPSHOT_CASES)
def test_snapshot_oracle_matches_reference(name):
    """BaryonifySnapshot (reference run with scipy's own periodic KDTree) vs the brute-force restatement"""
    g = H.load_snapshot_golden(name)
    out = H.snapshot_oracle_run(g)
    assert np.array_equal(np.isnan(out), np.isnan(g['expected']))
    assert np.nanmax(np.abs(out - g['expected'])) <= 1e-13 * g['L']          # measured: 0
    assert g['moved_idx'].size > 100
    assert np.all((out[~np.isnan(out)] >= 0) & (out[~np.isnan(out)] <= g['L']))
