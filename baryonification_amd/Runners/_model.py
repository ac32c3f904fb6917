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


def build_model(runner, kind, runner_cosmo=None):
    """kind = 'displacement' (BaryonifyShell / BaryonifyGrid), 'projected' (PaintProfilesShell, PaintProfilesGrid on
    2D maps) or 'real' (PaintProfilesGrid on 3D maps).  `runner_cosmo` overrides the runner-side cosmology dict (the
    grid runners drop w0, Map2DRunner.py:456-459).  Returns (bfgx_model, p_keys, keepalive)."""
    model = runner.model
    p_keys = list(vars(model).get('p_keys', []))                      # HealpixRunner.py:282
    if kind == 'displacement':
        if not hasattr(model, 'raw_input_d'):
            if hasattr(model, 'displacement'):
                raise NameError("No Table created. Run setup_interpolator() method first")
            raise TypeError("BaryonifyShell needs a tabulated displacement model (Baryonification2D/3D)")
        values = np.asarray(model.raw_input_d, dtype=np.float64)
        rdelta = bool(getattr(model, 'Rdelta_sampling', False))
        logv = False
        eps_model = float(model.epsilon_max)
    else:
        attr = 'raw_input_2D' if kind == 'projected' else 'raw_input_3D'
        if not hasattr(model, attr):
            raise TypeError("painting on the GPU needs a tabulated profile (TabulatedProfile / "
                            "ParamTabulatedProfile); wrap the profile and call setup_interpolator()/set_table()")
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
