"""
Tabulated profile holders, interface-compatible with BaryonForge/utils/Tabulate.py
(`TabulatedProfile` :67-358, `ParamTabulatedProfile` :362-701) for what the HEALPix runners use:
the raw tables `raw_input_2D/3D`, axes `raw_input_{z,M,r}_range` (+ `raw_input_<key>_range`, `p_keys`)
and the host read-outs `real()` / `projected()` (exp of a linear interpolation of the log table,
NaN outside the grid).

The per-halo hot path (PaintProfilesShell.process) never calls these host read-outs: it uploads the
raw table to the GPU (see Runners/HealpixRunner.py).  They exist so user scripts that evaluate a
profile at a handful of radii keep working.
"""
import numpy as np
from scipy import interpolate

from .cosmology import MassDef

__all__ = ['TabulatedProfile', 'ParamTabulatedProfile']


class _TabulatedBase(object):

    def __init__(self, model=None, cosmo=None, mass_def=None):
        self.model = model
        self.cosmo = cosmo
        self.mass_def = mass_def if mass_def is not None else MassDef(200, 'critical')
        self.p_keys = []

    def set_table(self, z_range, M_range, r_range, table_2D, table_3D=None, other_params=None):
        """Load an externally built table: z_range, M_range, r_range are the (linear) sample points,
        table_2D = projected profile * a on that grid (Tabulate.py:226, :547)."""
        other_params = dict(other_params or {})
        self.p_keys = list(other_params.keys())
        self.raw_input_2D = np.asarray(table_2D, dtype=np.float64)
        self.raw_input_3D = np.asarray(table_3D if table_3D is not None else table_2D, dtype=np.float64)
        self.raw_input_z_range = np.log(1 + np.asarray(z_range, dtype=np.float64))
        self.raw_input_M_range = np.log(np.asarray(M_range, dtype=np.float64))
        self.raw_input_r_range = np.log(np.asarray(r_range, dtype=np.float64))
        for k, v in other_params.items():
            setattr(self, 'raw_input_%s_range' % k, np.asarray(v, dtype=np.float64))
        grid = tuple([self.raw_input_z_range, self.raw_input_M_range, self.raw_input_r_range] +
                     [np.asarray(other_params[k], dtype=np.float64) for k in self.p_keys])
        with np.errstate(divide='ignore', invalid='ignore'):
            self.interp3D = interpolate.RegularGridInterpolator(grid, np.log(self.raw_input_3D), bounds_error=False)
            self.interp2D = interpolate.RegularGridInterpolator(grid, np.log(self.raw_input_2D), bounds_error=False)
        return self

    def setup_interpolator(self, z_min=1e-2, z_max=5, N_samples_z=30, z_linear_sampling=False,
                           M_min=1e12, M_max=1e16, N_samples_Mass=30, R_min=1e-3, R_max=1e2, N_samples_R=100,
                           other_params={}, verbose=True, N_samples_M=None):
        """Tabulates model.real and model.projected * a on (z, M, r[, params]) (Tabulate.py:160-243, :468-566).
        Profiles from baryonification_amd.Profiles project on the GPU (line-of-sight kernel)."""
        from itertools import product
        if N_samples_M is not None:
            N_samples_Mass = N_samples_M
        M_range = np.geomspace(M_min, M_max, N_samples_Mass)
        r = np.geomspace(R_min, R_max, N_samples_R)
        z_range = np.linspace(z_min, z_max, N_samples_z) if z_linear_sampling else np.geomspace(z_min, z_max, N_samples_z)
        other_params = {k: np.asarray(v, dtype=np.float64) for k, v in other_params.items()}
        p_keys = list(other_params.keys())
        shape = [z_range.size, M_range.size, r.size] + [other_params[k].size for k in p_keys]
        t3, t2 = np.full(shape, np.nan), np.full(shape, np.nan)
        for j in range(z_range.size):
            a_j = 1 / (1 + z_range[j])
            for c in product(*[np.arange(other_params[k].size) for k in p_keys]):
                for k_i, key in enumerate(p_keys):
                    self.model.set_parameter(key, other_params[key][c[k_i]])
                index = tuple([j, slice(None), slice(None)] + list(c))
                t3[index] = self.model.real(self.cosmo, r, M_range, a_j)
                t2[index] = self.model.projected(self.cosmo, r, M_range, a_j) * a_j
        return _TabulatedBase.set_table(self, z_range, M_range, r, t2, t3, other_params)

    def _readout(self, r, M, a, table, **kwargs):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        prof = np.zeros([M_use.size, r_use.size])
        empty = np.ones_like(r_use, dtype=np.float64)
        z_in, r_in = np.log(1 / a) * empty, np.log(r_use)
        k_in = [kwargs[k] * empty for k in self.p_keys]
        for i in range(M_use.size):
            prof[i] = np.exp(table(tuple([z_in, np.log(M_use[i]) * empty, r_in] + k_in)))
        if np.ndim(r) == 0:
            prof = np.squeeze(prof, axis=-1)
        if np.ndim(M) == 0:
            prof = np.squeeze(prof, axis=0)
        return prof

    def _check(self, kwargs, what):
        if not (hasattr(self, 'interp3D') and hasattr(self, 'interp2D')):
            raise NameError("No Table created. Run setup_interpolator() method first")
        for k in self.p_keys:
            assert k in kwargs.keys(), "Need to provide %s as input into `%s'. Table was built with this." % (k, what)

    def real(self, cosmo, r, M, a, **kwargs):
        self._check(kwargs, 'real')
        return self._readout(r, M, a, self.interp3D, **kwargs)

    def projected(self, cosmo, r, M, a, **kwargs):
        self._check(kwargs, 'projected')
        return self._readout(r, M, a, self.interp2D, **kwargs)


class TabulatedProfile(_TabulatedBase):
    """(z, M, r) table; the reference class forbids extra parameters."""

    def set_table(self, z_range, M_range, r_range, table_2D, table_3D=None, other_params=None):
        assert not other_params, "TabulatedProfile takes no extra parameters; use ParamTabulatedProfile"
        return super().set_table(z_range, M_range, r_range, table_2D, table_3D)


class ParamTabulatedProfile(_TabulatedBase):
    """(z, M, r, *params) table (Tabulate.py:524-561)."""

    def __init__(self, model=None, cosmo=None, mass_def=None):
        assert not isinstance(model, TabulatedProfile), "Input model cannot be 'TabulatedProfile' object."
        super().__init__(model, cosmo, mass_def)
