"""
Displacement-model holder, interface-compatible with BaryonForge/Profiles/BaryonCorrection.py:89-431
(`BaryonificationClass`, `Baryonification2D`): constructor `(DMO, DMB, cosmo, epsilon_max=20, mass_def=...)`,
attributes `raw_input_d`, `raw_input_{z,M,r}_range`, `Rdelta_sampling`, `p_keys`, `epsilon_max`,
`mass_def`, `cosmo`, and the host read-out `displacement(r, M, a, **kw)`.

BaryonifyShell.process() never calls the host read-out: it ships `raw_input_d` to the GPU, where the
(z, M) blend and the ln r interpolation of BaryonCorrection.py:356-382 run inside the per-halo kernel.
"""
import numpy as np
from scipy import interpolate

from ..utils.cosmology import MassDef

__all__ = ['BaryonificationClass', 'Baryonification2D']


class BaryonificationClass(object):

    def __init__(self, DMO=None, DMB=None, cosmo=None, epsilon_max=20, mass_def=None):
        self.DMO, self.DMB = DMO, DMB
        for prof in (DMO, DMB):                        # BaryonCorrection.py:100-101
            if prof is not None and hasattr(prof, 'set_parameter'):
                prof.set_parameter('cutoff', 1000)
        self.cosmo = cosmo
        self.epsilon_max = epsilon_max
        self.mass_def = mass_def if mass_def is not None else MassDef(200, 'critical')

    def set_table(self, z_range, M_range, r_range, d_interp, Rdelta_sampling=False, other_params=None):
        """Load a displacement table d[z, M, r(, params...)] in comoving Mpc.  `r_range` holds r, or
        r/R_Delta when Rdelta_sampling (BaryonCorrection.py:306-316)."""
        other_params = dict(other_params or {})
        self.p_keys = list(other_params.keys())
        self.raw_input_d = np.asarray(d_interp, dtype=np.float64)
        self.raw_input_z_range = np.log(1 + np.asarray(z_range, dtype=np.float64))
        self.raw_input_M_range = np.log(np.asarray(M_range, dtype=np.float64))
        self.raw_input_r_range = np.log(np.asarray(r_range, dtype=np.float64))
        for k, v in other_params.items():
            setattr(self, 'raw_input_%s_range' % k, np.asarray(v, dtype=np.float64))
        grid = tuple([self.raw_input_z_range, self.raw_input_M_range, self.raw_input_r_range] +
                     [np.asarray(other_params[k], dtype=np.float64) for k in self.p_keys])
        self.interp_d = interpolate.RegularGridInterpolator(grid, self.raw_input_d, bounds_error=False, fill_value=np.nan)
        self.Rdelta_sampling = bool(Rdelta_sampling)
        return self

    def get_masses(self, model, r, M, a):
        raise NotImplementedError("Implement a get_masses() method first")

    def setup_interpolator(self, *args, **kwargs):
        raise NotImplementedError("displacement-table construction is not part of this build yet; "
                                  "load a table with set_table()")

    def _readout(self, r, M, a, **kwargs):
        r_use, M_use = np.atleast_1d(r), np.atleast_1d(M)
        displ = np.zeros([M_use.size, r_use.size])
        empty = np.ones_like(r_use, dtype=np.float64)
        z_in, r_in = np.log(1 / a) * empty, np.log(r_use)
        k_in = [kwargs[k] * empty for k in self.p_keys]
        for i in range(M_use.size):
            M_in = np.log(M_use[i]) * empty
            R = self.mass_def.get_radius(self.cosmo, M_use[i], a) / a        # comoving Mpc
            R = float(np.atleast_1d(R)[0])
            r_axis = r_in - np.log(R) if self.Rdelta_sampling else r_in
            d = self.interp_d(tuple([z_in, M_in, r_axis] + k_in))
            displ[i] = np.where(r_use < self.epsilon_max * R, d, 0)
        if np.ndim(r) == 0:
            displ = np.squeeze(displ, axis=-1)
        if np.ndim(M) == 0:
            displ = np.squeeze(displ, axis=0)
        return displ

    def displacement(self, r, M, a, **kwargs):
        if not hasattr(self, 'interp_d'):
            raise NameError("No Table created. Run setup_interpolator() method first")
        for k in self.p_keys:
            assert k in kwargs.keys(), "Need to provide %s as input into `displacement'. Table was built with this." % k
        return self._readout(r, M, a, **kwargs)


class Baryonification2D(BaryonificationClass):
    """Projected (2D) displacement model: d(r_p) = M_DMB,p^-1(M_DMO,p(r_p)) - r_p."""
    pass
