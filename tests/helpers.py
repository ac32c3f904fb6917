"""Shared helpers: golden fixtures -> inputs for the oracle and for the product's runners."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
GOLDEN_CASES = ['c1_baryonify', 'lowz_baryonify', 'rdelta_baryonify', 'lowz_paint', 'c1_paint', 'param1_paint', 'param2_paint']
COSMO_KEYS = ('Omega_m', 'Omega_b', 'h', 'sigma8', 'n_s', 'w0')


def load_golden(name):
    f = np.load(os.path.join(GOLDEN, name + '.npz'))
    g = {k: f[k] for k in f.files}
    g['kind'] = str(g['kind'])
    g['nside'] = int(g['nside'])
    g['rdelta'] = bool(g['rdelta'])
    g['eps_runner'] = float(g['eps_runner'])
    g['eps_model'] = float(g['eps_model'])
    g['cosmo_runner'] = dict(zip(COSMO_KEYS, g['cosmo_runner'].tolist()))
    g['cosmo_model'] = dict(zip(COSMO_KEYS, g['cosmo_model'].tolist()))
    g['cat'] = {'M': g['cat_M'], 'z': g['cat_z'], 'ra': g['cat_ra'], 'dec': g['cat_dec']}
    g['p_keys'] = [str(k) for k in g['p_keys']] if 'p_keys' in g else []
    g['p_axes'] = {k: g['p_axis_' + k] for k in g['p_keys']}          # per-halo property columns (model.p_keys)
    for k in g['p_keys']:
        g['cat'][k] = g['cat_' + k]
    g['map_in'] = g['map_in'].astype(np.float64)
    return g


def oracle_run(g):
    from oracle import oracle as O
    axes = [np.log(1 + g['tab_z']), np.log(g['tab_M']), np.log(g['tab_r'])] + [g['p_axes'][k] for k in g['p_keys']]
    bg_r, bg_m = O.Background.from_dict(g['cosmo_runner']), O.Background.from_dict(g['cosmo_model'])
    if g['kind'] == 'baryonify':
        tab = O.Table(axes, g['tab_values'], g['rdelta'], g['eps_model'], p_keys=g['p_keys'])
        return O.baryonify_shell(g['nside'], g['map_in'], g['cat'], tab, g['eps_runner'], bg_r, bg_m)
    with np.errstate(divide='ignore'):
        tab = O.Table(axes, np.log(g['tab_values']), p_keys=g['p_keys'])
    return O.paint_shell(g['nside'], g['cat'], tab, g['eps_runner'], bg_r)


def product_runner(g, acc_f64=None):
    """Build the product's drop-in objects exactly as a BaryonForge user would."""
    import baryonification_amd as bfg
    cat = g['cat']
    Catalog = bfg.utils.HaloLightConeCatalog(ra=cat['ra'], dec=cat['dec'], M=cat['M'], z=cat['z'], cosmo=g['cosmo_runner'],
                                             **{k: cat[k] for k in g['p_keys']})
    cosmo_model = bfg.utils.Cosmology.from_dict(g['cosmo_model'])
    if g['kind'] == 'baryonify':
        Shell = bfg.utils.LightconeShell(map=g['map_in'], cosmo=g['cosmo_runner'])
        model = bfg.Profiles.Baryonification2D(None, None, cosmo_model, epsilon_max=g['eps_model'])
        model.set_table(g['tab_z'], g['tab_M'], g['tab_r'], g['tab_values'], Rdelta_sampling=g['rdelta'])
        runner = bfg.Runners.BaryonifyShell(Catalog, Shell, g['eps_runner'], model, verbose=False)
    else:
        Shell = bfg.utils.LightconeShell(map=np.zeros(12 * g['nside'] ** 2), cosmo=g['cosmo_runner'])
        if g['p_keys']:
            model = bfg.utils.ParamTabulatedProfile(None, cosmo_model)
            model.set_table(g['tab_z'], g['tab_M'], g['tab_r'], g['tab_values'], other_params=g['p_axes'])
        else:
            model = bfg.utils.TabulatedProfile(None, cosmo_model)
            model.set_table(g['tab_z'], g['tab_M'], g['tab_r'], g['tab_values'])
        runner = bfg.Runners.PaintProfilesShell(Catalog, Shell, g['eps_runner'], model, verbose=False)
    runner.acc_f64 = acc_f64
    return runner
