"""
Displacement-model holder, interface-compatible with BaryonForge/Profiles/BaryonCorrection.py:89-431
(`BaryonificationClass`, `Baryonification2D`): constructor `(DMO, DMB, cosmo, epsilon_max=20, mass_def=...)`,
attributes `raw_input_d`, `raw_input_{z,M,r}_range`, `Rdelta_sampling`, `p_keys`, `epsilon_max`,
`mass_def`, `cosmo`, and the host read-out `displacement(r, M, a, **kw)`.

BaryonifyShell.process() never calls the host read-out: it ships `raw_input_d` to the GPU, where the
(z, M) blend and the ln r interpolation of BaryonCorrection.py:356-382 run inside the per-halo kernel.
"""
import warnings

import numpy as np
from scipy import interpolate

from .. import tables

from ..utils.cosmology import MassDef

__all__ = ['BaryonificationClass', 'Baryonification2D', 'Baryonification3D']


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

    def setup_interpolator(self, z_min=1e-2, z_max=5, N_samples_z=30, z_linear_sampling=False,
                           M_min=1e12, M_max=1e16, N_samples_Mass=30, R_min=1e-3, R_max=1e2, N_samples_R=100,
                           Rdelta_min=1e-3, Rdelta_max=10, Rdelta_sampling=False, other_params={}, verbose=True,
                           N_samples_M=None):
        """Builds the displacement table d[z, M, r(, params)] (BaryonCorrection.py:136-321).  The enclosed-mass
        profiles and the inversion d = M_DMB^-1(M_DMO(r)) - r run on the GPU (tables.enclosed_mass_2d,
        tables.displacement_rows); only the 3-D densities are evaluated on the host.  `N_samples_M` is accepted as
        an alias of `N_samples_Mass` (the reference README passes it, README.md:79)."""
        from itertools import product
        if N_samples_M is not None:
            N_samples_Mass = N_samples_M
        if z_min <= 0:
            assert z_linear_sampling, f"Geometric series not possible for {z_min} < z < {z_max}. Set z_linear_sampling = True, or z_min > 0"
        M_range = np.geomspace(M_min, M_max, N_samples_Mass)
        r = np.geomspace(R_min, R_max, N_samples_R)
        z_range = np.linspace(z_min, z_max, N_samples_z) if z_linear_sampling else np.geomspace(z_min, z_max, N_samples_z)
        a_range = 1 / (1 + z_range)
        other_params = {k: np.asarray(v, dtype=np.float64) for k, v in other_params.items()}
        p_keys = list(other_params.keys())
        d_interp = np.zeros([z_range.size, M_range.size, r.size] + [other_params[k].size for k in p_keys])
        rdelta_range = np.geomspace(Rdelta_min, Rdelta_max, N_samples_R) if Rdelta_sampling else None
        for j in range(z_range.size):
            for c in product(*[np.arange(other_params[k].size) for k in p_keys]):
                for k_i, key in enumerate(p_keys):
                    for prof in (self.DMO, self.DMB):
                        prof.set_parameter(key, other_params[key][c[k_i]])
                M_DMO = self.get_masses(self.DMO, r, M_range, a_range[j])
                M_DMB = self.get_masses(self.DMB, r, M_range, a_range[j])
                offset, status = tables.displacement_rows(r, M_DMO, M_DMB)
                for i in np.nonzero(status)[0]:                      # the reference's warnings (:252-265, :292-297)
                    warnings.warn(f"Mass profile of log10(M) = {np.log10(M_range[i])} is nearly constant over radius, "
                                  "or fewer than 5 datapoints are usable. Defaulting to d = 0.", UserWarning)
                if Rdelta_sampling:                                  # :286-288
                    Rdelta = np.atleast_1d(self.mass_def.get_radius(self.cosmo, M_range, a_range[j])) / a_range[j]
                    offset = np.stack([np.interp(rdelta_range, r / Rdelta[i], offset[i]) for i in range(M_range.size)])
                d_interp[tuple([j, slice(None), slice(None)] + list(c))] = offset
        return self.set_table(z_range, M_range, rdelta_range if Rdelta_sampling else r, d_interp,
                              Rdelta_sampling=Rdelta_sampling, other_params=other_params)

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

    def get_masses(self, model, r, M, a):
        """Projected enclosed mass M_enc(r) = cumsum(2 pi r^2 Sigma dln r) on a 50 000-point grid, read out with a
        log-log PCHIP (BaryonCorrection.py:585-665).  For profiles that project in real space the whole chain
        (line-of-sight integral, prefix sum, PCHIP) runs on the GPU from the 3-D density on ~110 nodes."""
        r = np.asarray(r, dtype=np.float64)
        scalar = isinstance(M, (float, int))
        M_use = np.atleast_1d(np.asarray(M, dtype=np.float64))
        r_int = tables.r_int_2d(r)
        realspace = getattr(getattr(model, '_projected', None), '__name__', '') == '_projected_realspace'
        if realspace:
            l = tables.los_grid(r_int, model.padding_lo_proj, model.padding_hi_proj, model.n_per_decade_proj, model.proj_cutoff)
            rho = np.atleast_2d(model.real(self.cosmo, l, M_use, a))
            M_f = tables.enclosed_mass_2d(l, rho, a, r)
        else:                                            # e.g. pixel-convolved profiles: host projection, GPU integral
            Sigma = np.atleast_2d(model.projected(self.cosmo, r_int, M_use, a)) * a
            M_f = tables.enclosed_mass_from_sigma(r_int, Sigma, r)
        return M_f[0] if scalar else M_f


class Baryonification3D(BaryonificationClass):
    """3-D displacement model: d(r) = M_DMB^-1(M_DMO(r)) - r from the enclosed masses of the 3-D density profiles."""

    def get_masses(self, model, r, M, a):
        """M_enc(r) = cumsum(4 pi r^3 rho dln r) on a 50 000-point grid, read out with a log-log PCHIP over the points
        with rho > 0 (BaryonCorrection.py:470-548).  The density is sampled on the host; prefix sum and PCHIP run on
        the GPU."""
        r = np.asarray(r, dtype=np.float64)
        scalar = isinstance(M, (float, int))
        M_use = np.atleast_1d(np.asarray(M, dtype=np.float64))
        r_int = tables.r_int_3d(r)
        rho = np.atleast_2d(model.real(self.cosmo, r_int, M_use, a))
        M_f = tables.enclosed_mass_3d(r_int, rho, r)
        return M_f[0] if scalar else M_f
