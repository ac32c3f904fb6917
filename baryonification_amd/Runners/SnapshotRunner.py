"""
Particle-snapshot runner: drop-in for BaryonForge/Runners/SnapshotRunner.py (`DefaultRunnerSnapshot` :9-92,
`BaryonifySnapshot` :95-262).  Same constructor and attributes; `process()` returns a copy of the snapshot's
structured array with displaced, periodically re-wrapped x, y(, z).  No KD-tree is built: the HIP path bins the halos
into a periodic cell grid and gathers per particle (csrc/bfgx_snapshot.hpp); `KDTree_kwargs` is accepted and ignored,
`tree` is None.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..utils.cosmology import MassDef
from ._model import build_model

__all__ = ['DefaultRunnerSnapshot', 'BaryonifySnapshot']


class DefaultRunnerSnapshot(object):

    def __init__(self, HaloNDCatalog, ParticleSnapshot, epsilon_max, model, mass_def=MassDef(200, 'critical'), verbose=True,
                 KDTree_kwargs={}):
        self.HaloNDCatalog = HaloNDCatalog
        self.ParticleSnapshot = ParticleSnapshot
        self.epsilon_max = epsilon_max
        self.cosmo = HaloNDCatalog.cosmology
        self.model = model
        self.mass_def = mass_def
        self.verbose = verbose
        self.tree = None               # the reference keeps a scipy KDTree here; the GPU path needs none
        self.device = 0
        self.use_records = True        # engine knob: hand the structured array to the library as it is (False: gather the columns on the host)
        self.last_stats = None

    def enforce_periodicity(self, dx):
        L = self.ParticleSnapshot.L
        dx = np.where(dx > L / 2, dx - L, dx)
        dx = np.where(dx < -L / 2, dx + L, dx)
        return dx

    def compute_distance(self, *args):
        d = 0
        for dx in args:
            d = d + self.enforce_periodicity(dx) ** 2
        return np.sqrt(d)


class BaryonifySnapshot(DefaultRunnerSnapshot):
    """Moves every particle within epsilon_max * R200c of a halo radially by the model's displacement."""

    def _setup(self):
        snap = self.ParticleSnapshot
        hcat = self.HaloNDCatalog.cat
        is2D = snap.is2D
        cosmo = dict(self.cosmo)
        cosmo['w0'] = -1.0                                            # SnapshotRunner.py:204-207 does not pass w0
        model, p_keys, keep = build_model(self, 'displacement', cosmo)
        if p_keys:
            raise NotImplementedError("BaryonifySnapshot passes no halo properties to the model (SnapshotRunner.py:240)")
        with np.errstate(invalid='ignore', divide='ignore'):
            lnM = np.log(np.asarray(hcat['M'], dtype=np.float32)).astype(np.float64)     # float32 log, as BaryonifyGrid
        cat, cols = _lib.make_grid_catalog_host(hcat['M'], hcat['x'], hcat['y'], None if is2D else hcat['z'], lnM)
        return snap, is2D, model, keep, cat, cols

    def process_make_map(self, N_grid):
        """`ParticleSnapshot(cat=self.process(), ...).make_map(N_grid)` (SnapshotRunner.py:173-262 followed by io.py:622-670) as ONE call for
        the common case that only the map of the baryonified particles is wanted (the reference's notebook 10): the records go to the device
        once, the displaced coordinates exist only as the deposit's sort keys, the map comes back.  Not in the reference's API; the result
        equals the two calls (cell for cell with unit masses, to the order of the sums inside a cell otherwise).  Snapshots whose `cat` is
        not one C-contiguous structured array of float64 fields take the two calls."""
        snap, is2D, model, keep, cat, cols = self._setup()
        rec = snap.cat
        fields = rec.dtype.fields or {}
        need = ('x', 'y', 'M') if is2D else ('x', 'y', 'z', 'M')
        ndim = 2 if is2D else 3
        ok = (isinstance(rec, np.ndarray) and rec.ndim == 1 and rec.flags.c_contiguous and rec.dtype.itemsize % 8 == 0 and rec.dtype.itemsize >= 16 and
              all(k in fields and fields[k][0] == np.float64 and fields[k][1] % 8 == 0 for k in need))
        if ok:
            edges = np.linspace(0, snap.L, N_grid + 1)                # io.py:640
            out = _lib.pinned_empty(N_grid ** ndim).reshape((N_grid,) * ndim)
            opts = _lib.bfgx_opts(int(self.device), 1, 1, 0, 1, 0)
            stats = _lib.bfgx_stats()
            rc = _lib.load().bfgx_baryonify_snapshot_records_map(
                C.byref(cat), C.byref(model), ndim, float(snap.L), float(self.HaloNDCatalog.redshift), rec.size,
                rec.ctypes.data if rec.size else None, rec.dtype.itemsize, fields['x'][1], fields['y'][1], 0 if is2D else fields['z'][1],
                fields['M'][1], int(N_grid), edges.ctypes.data, out.ctypes.data, C.byref(opts), C.byref(stats))
            if rc == _lib.ERR_UNSUPPORTED:
                ok = False                                            # (a grid the tile-owned deposit does not take)
            else:
                _lib.check(rc)
                self.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
                del keep, cols
                return out
        from ..utils.io import ParticleSnapshot
        new = ParticleSnapshot.__new__(ParticleSnapshot)
        new.__dict__.update(snap.__dict__)
        new.cat = self.process()
        return new.make_map(N_grid)

    def process(self):
        snap, is2D, model, keep, cat, cols = self._setup()
        opts = _lib.bfgx_opts(int(self.device), 1, 1, 0, 1, 0)
        stats = _lib.bfgx_stats()
        rec = snap.cat
        fields = rec.dtype.fields or {}
        ok = (getattr(self, 'use_records', True) and isinstance(rec, np.ndarray) and rec.ndim == 1 and rec.flags.c_contiguous and rec.dtype.itemsize % 8 == 0 and rec.dtype.itemsize >= 16 and
              all(k in fields and fields[k][0] == np.float64 and fields[k][1] % 8 == 0 for k in (('x', 'y') if is2D else ('x', 'y', 'z'))))
        if ok:
            # the records as they are: uploaded in chunks, displaced in place on the device, downloaded into the new catalog -- the host
            # neither gathers the strided columns nor scatters them back (`new_cat = cat.copy(); new_cat['x'] = ...`, SnapshotRunner.py:254-262)
            new_cat = _lib.pinned_empty(rec.size * rec.dtype.itemsize, np.uint8).view(rec.dtype) if rec.size else rec.copy()
            rc = _lib.load().bfgx_baryonify_snapshot_records(
                C.byref(cat), C.byref(model), 2 if is2D else 3, float(snap.L), float(self.HaloNDCatalog.redshift), rec.size,
                rec.ctypes.data if rec.size else None, new_cat.ctypes.data if rec.size else None, rec.dtype.itemsize, fields['x'][1], fields['y'][1],
                0 if is2D else fields['z'][1], C.byref(opts), C.byref(stats))
            _lib.check(rc)
            self.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
            del keep, cols
            return new_cat
        x, y = _lib.f8(snap.cat['x']), _lib.f8(snap.cat['y'])
        z = None if is2D else _lib.f8(snap.cat['z'])
        s = _lib.bfgx_snapshot(2 if is2D else 3, 0, x.size, x.ctypes.data, y.ctypes.data, None if is2D else z.ctypes.data,
                               float(snap.L), float(self.HaloNDCatalog.redshift))
        ox, oy = np.empty_like(x), np.empty_like(y)
        oz = None if is2D else np.empty_like(z)
        rc = _lib.load().bfgx_baryonify_snapshot(C.byref(cat), C.byref(model), C.byref(s), ox.ctypes.data, oy.ctypes.data,
                                                 None if is2D else oz.ctypes.data, C.byref(opts), C.byref(stats))
        _lib.check(rc)
        self.last_stats = {k: getattr(stats, k) for k, _ in stats._fields_}
        new_cat = snap.cat.copy()
        new_cat['x'], new_cat['y'] = ox, oy
        if not is2D:
            new_cat['z'] = oz
        del keep, cols
        return new_cat
