"""
Synthetic inputs for tests and bench.py, exactly as SURVEY.md section 8(d) specifies them
(numpy.random.default_rng / PCG64, fixed seeds and draw order).  No reference code involved:
the reference's example data are remote downloads that are unavailable offline.
"""
import numpy as np

from .utils.cosmology import Cosmology, MassDef

COSMO = {'Omega_m': 0.3175, 'Omega_b': 0.049, 'h': 0.6711, 'sigma8': 0.834, 'n_s': 0.9649, 'w0': -1.0}
SEED_CATALOG = 20250103
SEED_MAP = 20250104


def make_catalog(N, seed=SEED_CATALOG, z_lo=0.2, z_hi=0.3, logM_lo=12.0, logM_hi=15.0):
    """dn/dlog M ~ M^-0.9 on [1e12, 1e15], z ~ U(0.2, 0.3), uniform on the sphere."""
    rng = np.random.default_rng(seed)
    k = 0.9 * np.log(10.0)
    span = logM_hi - logM_lo
    u = rng.random(N)
    log10M = logM_lo - np.log(1.0 - u * (1.0 - np.exp(-span * k))) / k
    z = rng.uniform(z_lo, z_hi, N)
    ra = rng.uniform(0, 360, N)
    dec = np.degrees(np.arcsin(rng.uniform(-1, 1, N)))
    return {'M': 10.0 ** log10M, 'z': z, 'ra': ra, 'dec': dec}


def make_map(nside, seed=SEED_MAP, lam=8.0):
    """Poisson(8) mass map (about 0.03 % zero pixels exercise the `orig_map > 0` mask)."""
    rng = np.random.default_rng(seed)
    return rng.poisson(lam, 12 * nside * nside).astype(np.float64)


def table_grid(cat, Nz=10, NM=10, NR=500, R_min=1e-3, R_max=3e2, pad=0.0):
    """z/M edges = catalog min/max (README.md:78-80); `pad` > 0 widens them by that relative amount."""
    z_lo, z_hi = cat['z'].min(), cat['z'].max()
    M_lo, M_hi = cat['M'].min(), cat['M'].max()
    if pad > 0:
        z_lo, z_hi = z_lo * (1 - pad), z_hi * (1 + pad)
        M_lo, M_hi = M_lo * (1 - pad), M_hi * (1 + pad)
    z = np.geomspace(z_lo, z_hi, Nz)
    M = np.geomspace(M_lo, M_hi, NM)
    r = np.geomspace(R_min, R_max, NR)
    return z, M, r


def _Rc(z, M, cosmo=COSMO):
    """comoving R200c [Mpc] on the (z, M) grid nodes"""
    c = Cosmology.from_dict(cosmo)
    md = MassDef(200, 'critical')
    a = 1.0 / (1.0 + z)
    return np.stack([md.get_radius(c, M, ai) / ai for ai in a], axis=0)      # [Nz, NM]


def displacement_table(z, M, r, cosmo=COSMO):
    """closed-form plumbing table: d = -0.05 R_c x exp(-x) / (1 + x^2), x = r / R_c  [comoving Mpc]"""
    Rc = _Rc(z, M, cosmo)[:, :, None]
    x = r[None, None, :] / Rc
    return -0.05 * Rc * x * np.exp(-x) / (1.0 + x * x)


def paint_table(z, M, r, cosmo=COSMO):
    """closed-form plumbing profile: ln P = -x - 2 ln(1 + x); returns P (the table holds P, the
    interpolator its log, Tabulate.py:237-238)"""
    Rc = _Rc(z, M, cosmo)[:, :, None]
    x = r[None, None, :] / Rc
    return np.exp(-x - 2.0 * np.log1p(x))


# examples/default_config.npy of the reference (decoded in SURVEY.md section 5), cdelta = 7, proj_cutoff = 50
S19_PARAMS = dict(epsilon=4.0, theta_ej=4.0, theta_co=0.1, M_c=1e14, mu_beta=0.1, gamma=2.5, delta=7.0, eta=0.3, eta_delta=0.1,
                  tau=-1.5, tau_delta=0.0, A=0.055, M1=3e11, epsilon_h=0.015, a=0.3, n=2.0, p=0.3, q=0.707, cdelta=7.0,
                  alpha_nt=0.2, nu_nt=0.5, gamma_nt=0.5, cutoff=1000.0, proj_cutoff=50.0, mu_theta_ej=0.1, mu_theta_co=0.0,
                  M_theta_ej=5e13, M_theta_co=5e13)


def s19_displacement_table(z, M, r, cosmo=COSMO, params=None):
    """Benchmark table (ii): d(z, M, r) from the Schneider19 one-halo DMO/DMB profiles through the GPU table builders
    (Baryonification2D.setup_interpolator's inner loop: get_masses x2 + displacement_rows per redshift)."""
    import warnings
    from . import tables
    from .Profiles import Baryonification2D, DarkMatterBaryon, DarkMatterOnly
    par = dict(S19_PARAMS if params is None else params)
    c = Cosmology.from_dict(cosmo)
    model = Baryonification2D(DarkMatterOnly(**par), DarkMatterBaryon(**par), c, epsilon_max=20)
    out = np.zeros((z.size, M.size, r.size))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for j, zj in enumerate(z):
            a = 1.0 / (1.0 + zj)
            out[j], _ = tables.displacement_rows(r, model.get_masses(model.DMO, r, M, a), model.get_masses(model.DMB, r, M, a))
    return out
