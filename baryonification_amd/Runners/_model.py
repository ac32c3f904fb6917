"""
Turns a model object (ours or the reference's own Baryonification2D / TabulatedProfile /
ParamTabulatedProfile, duck-typed through the raw_input_* attributes they keep --
BaryonCorrection.py:309-316, Tabulate.py:231-238, :553-561) into the bfgx_model the C ABI takes.
"""
import numpy as np

from .. import _lib
from ..utils.cosmology import cosmo_to_dict, massdef_to_tuple


def _axes(model, p_keys):
    try:
        axes = [model.raw_input_z_range, model.raw_input_M_range, model.raw_input_r_range]
    except AttributeError:
        raise NameError("No Table created. Run setup_interpolator() method first")
    axes += [getattr(model, 'raw_input_%s_range' % k) for k in p_keys]
    return [np.asarray(a, dtype=np.float64) for a in axes]


# sampling of the table a callable (non-tabulated) model is given on first use: the README's radial grid (README.md:78-80) on the
# (z, M) support of the runner's catalog
BRIDGE_N_Z, BRIDGE_N_M, BRIDGE_N_R, BRIDGE_R_MIN, BRIDGE_R_MAX = 10, 20, 500, 1e-3, 3e2


def _support(runner):
    """(z_min, z_max, M_min, M_max) of the runner's catalog, opened by 1e-6 relative so that no halo sits on a table edge"""
    shell = hasattr(runner, 'HaloLightConeCatalog')
    cat = runner.HaloLightConeCatalog.cat if shell else runner.HaloNDCatalog.cat
    M = np.asarray(cat['M'], dtype=np.float64)
    if shell:                                             # (an ND catalog's 'z' column is a position)
        z = np.asarray(cat['z'], dtype=np.float64)
        z0, z1 = float(z.min()), float(z.max())
    else:                                                 # a box at one redshift (GridRunner / SnapshotRunner): two slices around it
        z0 = z1 = float(runner.HaloNDCatalog.redshift)
    z0, z1 = max(z0 * (1 - 1e-6) - 1e-6, 0.0), z1 * (1 + 1e-6) + 1e-6
    return z0, z1, float(M.min()) * (1 - 1e-6), float(M.max()) * (1 + 1e-6)


def _fingerprint(model):
    """the model's parameters as far as they can be read off the object (scalars, strings, arrays, nested dicts / lists of those): a
    table tabulated from a callable is reused only while this has not changed"""
    import hashlib

    def enc(v, depth=0):
        if isinstance(v, (bool, int, float, complex, str, bytes, type(None))):
            return repr(v)
        if isinstance(v, np.ndarray):
            return 'nd%r%s' % (v.shape, hashlib.blake2b(np.ascontiguousarray(v).tobytes(), digest_size=8).hexdigest())
        if isinstance(v, np.generic):
            return repr(v.item())
        if depth < 3 and isinstance(v, (list, tuple)):
            return '[' + ','.join(enc(x, depth + 1) for x in v) + ']'
        if depth < 3 and isinstance(v, dict):
            return '{' + ','.join('%r:%s' % (k, enc(v[k], depth + 1)) for k in sorted(v, key=repr)) + '}'
        if depth < 2 and hasattr(v, '__dict__') and not callable(v):
            return type(v).__name__ + enc(vars(v), depth + 1)
        return type(v).__name__ + '@%x' % id(v)
    try:
        items = {k: v for k, v in vars(model).items() if not k.startswith('_bfgx')}
    except TypeError:
        return None
    return hashlib.blake2b(enc(items).encode(), digest_size=8).hexdigest()


_WARNED = set()


def tabulate_callable(runner, kind):
    """The reference's runners call `model.displacement(r, M, a)` / `model.projected(cosmo, r, M, a)` / `model.real(cosmo, r, M, a)`
    once per halo on ANY object (HealpixRunner.py:321, :441; Map2DRunner.py:534, :752, :781; SnapshotRunner.py:228).  A model that carries no table (no raw_input_*) and cannot build one itself (no
    setup_interpolator) is tabulated here, once, on the (z, M) support of the runner's catalog -- BRIDGE_N_Z x BRIDGE_N_M x
    BRIDGE_N_R samples, z linear, M and r logarithmic -- and the table holder is cached on the model (`_bfgx_tabulated`), keyed by
    the support.  Returns the holder (a Baryonification2D / TabulatedProfile of this package)."""
    import warnings
    model = runner.model
    z0, z1, M0, M1 = _support(runner)
    shell = hasattr(runner, 'HaloLightConeCatalog')
    # sampling: model.bfgx_table_grid = (Nz, NM, NR) [, (R_min, R_max)] overrides the default BRIDGE_* grid
    grid = getattr(model, 'bfgx_table_grid', None)
    nz, nm, nrad, rmin, rmax = (BRIDGE_N_Z if shell else 2), BRIDGE_N_M, BRIDGE_N_R, BRIDGE_R_MIN, BRIDGE_R_MAX
    if grid is not None:
        g = tuple(grid)
        if len(g) == 2 and np.ndim(g[0]) == 1:
            (nz, nm, nrad), (rmin, rmax) = (int(v) for v in g[0]), (float(v) for v in g[1])
        else:
            nz, nm, nrad = (int(v) for v in g)
        if nz < 2 or nm < 2 or nrad < 2 or not (0 < rmin < rmax):
            raise ValueError("model.bfgx_table_grid = (Nz, NM, NR)[, (R_min, R_max)] needs at least 2 samples per axis and 0 < R_min < R_max")
    key = (kind, z0, z1, M0, M1, nz, nm, nrad, rmin, rmax, _fingerprint(model))
    cached = getattr(model, '_bfgx_tabulated', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    wkey = (type(model).__name__, kind, nz, nm, nrad)
    if wkey not in _WARNED:
        _WARNED.add(wkey)
        warnings.warn("%s has no table: its %s() is tabulated once on %d x %d x %d samples (z linear in [%.4g, %.4g], M logarithmic in "
                      "[%.3g, %.3g], r logarithmic in [%.3g, %.3g] Mpc) and the GPU reads that table out, where the reference calls the "
                      "method per halo (HealpixRunner.py:321, :441): results differ by the interpolation error of the table (a few 1e-3 of the "
                      "effect at the default sampling).  Set model.bfgx_table_grid = (Nz, NM, NR) for a finer one."
                      % (type(model).__name__, kind, nz, nm, nrad, z0, z1, M0, M1, rmin, rmax), RuntimeWarning, stacklevel=3)
    from ..Profiles.BaryonCorrection import Baryonification2D
    from ..utils.Tabulate import TabulatedProfile
    from ..utils.cosmology import Cosmology, cosmo_to_dict
    cosmo = getattr(model, 'cosmo', None) or runner.cosmo
    cosmo_obj = cosmo if isinstance(cosmo, Cosmology) else Cosmology.from_dict(cosmo_to_dict(cosmo))
    z = np.linspace(z0, z1, nz)
    Mg = np.geomspace(M0, M1, nm)
    r = np.geomspace(rmin, rmax, nrad)
    if kind == 'displacement':
        d = np.zeros((z.size, Mg.size, r.size))
        for i, zi in enumerate(z):
            for j, Mj in enumerate(Mg):
                d[i, j] = np.asarray(model.displacement(r, Mj, 1.0 / (1.0 + zi)), dtype=np.float64).reshape(r.size)
        # the model applies its own cut inside displacement() (BaryonCorrection.py:381-382); the holder must not cut again
        holder = Baryonification2D(None, None, cosmo_obj, epsilon_max=float(getattr(model, 'epsilon_max', np.inf)),
                                   mass_def=getattr(model, 'mass_def', None))
        holder.set_table(z, Mg, r, d)
    else:
        # the runner paints model.projected(cosmo, r_sep / a, M, a) as it comes (HealpixRunner.py:441): no factor of `a` here (the
        # reference's TabulatedProfile multiplies its OWN table by a, Tabulate.py:226 -- a raw profile is painted without it)
        t2 = np.zeros((z.size, Mg.size, r.size))
        profile = model.projected if kind == 'projected' else model.real
        for i, zi in enumerate(z):
            a = 1.0 / (1.0 + zi)
            try:
                row = np.asarray(profile(cosmo_obj, r, Mg, a), dtype=np.float64)
                assert row.shape == (Mg.size, r.size)
            except Exception:        # noqa: BLE001  a profile that takes scalar masses only
                row = np.stack([np.asarray(profile(cosmo_obj, r, Mj, a), dtype=np.float64).reshape(r.size) for Mj in Mg])
            t2[i] = row
        holder = TabulatedProfile(None, cosmo_obj, mass_def=getattr(model, 'mass_def', None))
        holder.set_table(z, Mg, r, t2)
    try:
        model._bfgx_tabulated = (key, holder)
    except AttributeError:
        pass
    return holder


class _Proxy(object):
    """a runner whose model is the tabulated holder (everything else falls through)"""

    def __init__(self, runner, model):
        self.__dict__['_r'], self.__dict__['model'] = runner, model

    def __getattr__(self, name):
        return getattr(self._r, name)


EXACT_MAX_HALOS = 50_000        # catalogs up to this size: a callable model is called per halo, as the reference does (model.bfgx_exact overrides)


def plain_callable(model, kind):
    """is `model` an object the reference would simply CALL -- a displacement(r, M, a) / projected(cosmo, r, M, a) method, no table of its
    own (raw_input_*) and no way to build one (setup_interpolator)?"""
    attr = 'raw_input_d' if kind == 'displacement' else ('raw_input_2D' if kind == 'projected' else 'raw_input_3D')
    return (model is not None and not hasattr(model, attr) and not hasattr(model, 'setup_interpolator')
            and callable(getattr(model, kind, None)))


def wants_exact(runner, kind):
    """the per-halo route for a plain callable: model.bfgx_exact = True / False, default by the size of the catalog (the route makes one
    Python call per halo, as the reference's loop does: HealpixRunner.py:321, :441)"""
    if not plain_callable(runner.model, kind):
        return False
    flag = getattr(runner.model, 'bfgx_exact', None)
    return runner.HaloLightConeCatalog.cat.size <= EXACT_MAX_HALOS if flag is None else bool(flag)


def process_callable_exact(runner, kind, orig_map=None):
    """BaryonifyShell / PaintProfilesShell .process() for a model that is a plain Python callable, evaluated exactly as the reference does:
    once per halo, on the separations r_sep / a_j of that halo's own pixels (HealpixRunner.py:314-331, :436-445).  The device finds the discs
    and the separations (bfgx_shell_pairs_begin / _radii), this loop calls the model, the device turns the values into pixel offsets and
    regrids / adds the painted values (bfgx_shell_pairs_apply).  No table, hence no interpolation error: the result equals the reference's to
    the rounding of the geometry (1e-10)."""
    import ctypes as C
    from ..utils.cosmology import Cosmology
    model = runner.model
    cat = runner.HaloLightConeCatalog.cat
    paint = kind != 'displacement'
    # the runner's side of the geometry only (cosmology, mass definition, epsilon_max); the table is a placeholder nobody reads
    table, keep = _lib.make_table([np.array([0.0, 1.0])] * 3, np.zeros((2, 2, 2)), False, False, 0.0)
    m = _lib.bfgx_model()
    m.table = table
    m.cosmo_runner = m.cosmo_model = _lib.make_cosmo(cosmo_to_dict(runner.cosmo))
    D, rho = massdef_to_tuple(runner.mass_def)
    m.massdef_runner = m.massdef_model = _lib.make_massdef(D, rho)
    m.eps_runner = float(runner.epsilon_max)
    cols = [_lib.f8(cat[k]) for k in ('M', 'z', 'ra', 'dec')]
    c, ckeep = _lib.make_catalog_host(cols[0], cols[1], cols[2], cols[3], [])
    nside = int(runner.LightconeShell.NSIDE)
    lib = _lib.load()
    h = C.c_void_p()
    counts = np.zeros(max(cat.size, 1), dtype=np.int64)
    _lib.check(lib.bfgx_shell_pairs_begin(C.byref(c), C.byref(m), nside, int(paint), int(runner.device), C.byref(h), counts.ctypes.data))
    try:
        off = np.concatenate([[0], np.cumsum(counts[:cat.size])]).astype(np.int64)
        r = np.empty(max(int(off[-1]), 1))
        _lib.check(lib.bfgx_shell_pairs_radii(h, r.ctypes.data))
        vals = np.zeros_like(r)
        cosmo = getattr(model, 'cosmo', None) or runner.cosmo
        cosmo_obj = cosmo if isinstance(cosmo, Cosmology) else Cosmology.from_dict(cosmo_to_dict(cosmo))
        M, z = cols[0], cols[1]
        with np.errstate(all='ignore'):
            for j in range(cat.size):
                sl = slice(int(off[j]), int(off[j + 1]))
                a_j = 1.0 / (1.0 + z[j])                                                      # HealpixRunner.py:295
                if paint:
                    vals[sl] = np.asarray(model.projected(cosmo_obj, r[sl], M[j], a_j), dtype=np.float64).reshape(-1)     # :441
                else:
                    vals[sl] = np.asarray(model.displacement(r[sl], M[j], a_j), dtype=np.float64).reshape(-1)              # :321
        npix = 12 * nside * nside
        new_map = _lib.pinned_empty(npix)
        stats = _lib.bfgx_stats()
        src = None if paint else _lib.f8(orig_map)
        _lib.check(lib.bfgx_shell_pairs_apply(h, vals.ctypes.data, None if paint else src.ctypes.data, new_map.ctypes.data, 1, C.byref(stats)))
    finally:
        lib.bfgx_shell_pairs_end(h)
    runner.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
    del keep, ckeep
    return new_map


def build_model(runner, kind, runner_cosmo=None):
    """kind = 'displacement' (BaryonifyShell / BaryonifyGrid), 'projected' (PaintProfilesShell, PaintProfilesGrid on
    2D maps) or 'real' (PaintProfilesGrid on 3D maps).  `runner_cosmo` overrides the runner-side cosmology dict (the
    grid runners drop w0, Map2DRunner.py:456-459).  Returns (bfgx_model, p_keys, keepalive)."""
    model = runner.model
    p_keys = list(vars(model).get('p_keys', []))                      # HealpixRunner.py:282
    if kind == 'displacement':
        if not hasattr(model, 'raw_input_d'):
            if hasattr(model, 'setup_interpolator'):          # a Baryonification2D/3D that was never set up: as the reference's displacement()
                raise NameError("No Table created. Run setup_interpolator() method first")
            if callable(getattr(model, 'displacement', None)):
                return build_model(_Proxy(runner, tabulate_callable(runner, kind)), kind, runner_cosmo)
            raise TypeError("BaryonifyShell needs a model with a displacement(r, M, a) method or a displacement table "
                            "(Baryonification2D/3D)")
        values = np.asarray(model.raw_input_d, dtype=np.float64)
        rdelta = bool(getattr(model, 'Rdelta_sampling', False))
        logv = False
        eps_model = float(model.epsilon_max)
    else:
        attr = 'raw_input_2D' if kind == 'projected' else 'raw_input_3D'
        if not hasattr(model, attr):
            if hasattr(model, 'setup_interpolator'):
                raise NameError("No Table created. Run setup_interpolator() method first")
            if callable(getattr(model, kind, None)):
                return build_model(_Proxy(runner, tabulate_callable(runner, kind)), kind, runner_cosmo)
            raise TypeError("painting needs a profile with a projected / real (cosmo, r, M, a) method or a tabulated profile "
                            "(TabulatedProfile / ParamTabulatedProfile)")
        with np.errstate(divide='ignore', invalid='ignore'):
            values = np.log(np.asarray(getattr(model, attr), dtype=np.float64))   # Tabulate.py:237-238, :560-561
        rdelta, logv, eps_model = False, True, 0.0
    table, keep = _lib.make_table(_axes(model, p_keys), values, rdelta, logv, eps_model)

    m = _lib.bfgx_model()
    m.table = table
    m.cosmo_runner = _lib.make_cosmo(cosmo_to_dict(runner.cosmo if runner_cosmo is None else runner_cosmo))
    D, rho = massdef_to_tuple(runner.mass_def)
    m.massdef_runner = _lib.make_massdef(D, rho)
    mc = getattr(model, 'cosmo', None)
    m.cosmo_model = _lib.make_cosmo(cosmo_to_dict(mc if mc is not None else runner.cosmo))
    D, rho = massdef_to_tuple(getattr(model, 'mass_def', None))
    m.massdef_model = _lib.make_massdef(D, rho)
    m.eps_runner = float(runner.epsilon_max)
    return m, p_keys, keep
