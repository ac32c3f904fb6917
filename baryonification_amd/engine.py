"""
Resident (inputs-already-in-HBM) front-end of the C ABI: a thin object around `bfgx_plan`.
Pointers are plain integers (e.g. `tensor.data_ptr()`), the stream a raw hipStream_t handle
(e.g. `torch.cuda.current_stream().cuda_stream`); this module does not import torch.

Used by bench.py and by the multi-GPU driver (parallel.py); the drop-in runners use the one-shot
host entry points instead.
"""
import ctypes as C

import numpy as np

from . import _lib


class ShellPlan(object):

    def __init__(self, model, keepalive, nside, max_halos, device=0, stream=0):
        self._keep = keepalive
        self.nside = int(nside)
        self.npix = 12 * self.nside * self.nside
        self.max_halos = int(max_halos)
        h = C.c_void_p()
        _lib.check(_lib.load().bfgx_plan_create(int(device), C.c_void_p(int(stream) or None), self.nside,
                                               self.max_halos, C.byref(model), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            try:
                _lib.load().bfgx_plan_destroy(self._h)
            except Exception:          # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    __del__ = close

    def offsets(self, cat_dev, offsets_ptr, acc_f64=False):
        """K0 + K1 (HealpixRunner.py:291-331): offsets[npix][3] += ..."""
        _lib.check(_lib.load().bfgx_offsets_device(self._h, C.byref(cat_dev), C.c_void_p(int(offsets_ptr)), int(acc_f64)))

    def regrid(self, map_in_ptr, offsets_ptr, map_out_ptr, sums_ptr=0, acc_f64=False):
        """K2 (HealpixRunner.py:333-346): map_out[npix] (zeroed) += regrid; sums[2] optional"""
        _lib.check(_lib.load().bfgx_regrid_device(self._h, C.c_void_p(int(map_in_ptr)), C.c_void_p(int(offsets_ptr)),
                                                 int(acc_f64), C.c_void_p(int(map_out_ptr)),
                                                 C.c_void_p(int(sums_ptr) or None)))

    def baryonify(self, cat_dev, map_in_ptr, offsets_work_ptr, map_out_ptr, sums_ptr=0, acc_f64=False):
        """K0 + K1 + K2 in one enqueue-only call (HealpixRunner.py:291-346 on device buffers); offsets_work: [npix][3] scratch"""
        _lib.check(_lib.load().bfgx_baryonify_device(self._h, C.byref(cat_dev), C.c_void_p(int(map_in_ptr)), C.c_void_p(int(offsets_work_ptr)),
                                                    int(acc_f64), C.c_void_p(int(map_out_ptr)), C.c_void_p(int(sums_ptr) or None)))

    def route_count(self, n, rings_ptr, ring_bounds, counts_ptr):
        """routing pass 1 (enqueue-only): counts[world] (int32, device) = halos whose ring range touches each rank's rings"""
        rb = np.ascontiguousarray(ring_bounds, dtype=np.int32)
        _lib.check(_lib.load().bfgx_route_count_device(self._h, int(n), C.c_void_p(int(rings_ptr) or None), int(rb.size - 1), rb.ctypes.data,
                                                      C.c_void_p(int(counts_ptr))))

    def route_fill(self, n, rings_ptr, ring_bounds, start, col_ptrs, cursor_ptr, rows_ptr):
        """routing pass 2 (enqueue-only): rows[start[j] ...] = packed rows (len(col_ptrs) doubles per halo) bound for rank j"""
        rb = np.ascontiguousarray(ring_bounds, dtype=np.int32)
        st = np.ascontiguousarray(start, dtype=np.int64)
        cols = (C.c_void_p * len(col_ptrs))(*[int(c) for c in col_ptrs])
        _lib.check(_lib.load().bfgx_route_fill_device(self._h, int(n), C.c_void_p(int(rings_ptr) or None), int(rb.size - 1), rb.ctypes.data,
                                                     st.ctypes.data, len(col_ptrs), cols, C.c_void_p(int(cursor_ptr)),
                                                     C.c_void_p(int(rows_ptr) or None)))

    def route_pack(self, n, rings_ptr, ring_bounds, blockcap, col_ptrs, cursor_ptr, blocks_ptr, overflow_ptr):
        """routing in one pass with nothing read back (enqueue-only): blocks[world][len(col_ptrs)][blockcap], rows a destination does not
        receive keep M = NaN (column 0); *overflow (int32, device, zeroed by the caller) is set when a block is too small"""
        rb = np.ascontiguousarray(ring_bounds, dtype=np.int32)
        cols = (C.c_void_p * len(col_ptrs))(*[int(c) for c in col_ptrs])
        _lib.check(_lib.load().bfgx_route_pack_device(self._h, int(n), C.c_void_p(int(rings_ptr) or None), int(rb.size - 1), rb.ctypes.data,
                                                     int(blockcap), len(col_ptrs), cols, C.c_void_p(int(cursor_ptr)), C.c_void_p(int(blocks_ptr)),
                                                     C.c_void_p(int(overflow_ptr))))

    def route_step(self, cat_dev, rank, ring_bounds, blockcap, col_ptrs, cursor_ptr, send_ptr, recv_ptr, overflow_ptr):
        """the routing of one resident multi-GPU step in one call (ring ranges + packing; two launches): send = (world - 1) blocks
        [len(col_ptrs)][blockcap] in rank order without this rank, recv = world blocks with this rank's own rows written into the LAST one"""
        rb = np.ascontiguousarray(ring_bounds, dtype=np.int32)
        cols = (C.c_void_p * len(col_ptrs))(*[int(c) for c in col_ptrs])
        _lib.check(_lib.load().bfgx_route_step_device(self._h, C.byref(cat_dev), int(rb.size - 1), int(rank), rb.ctypes.data, int(blockcap), len(col_ptrs), cols,
                                                     C.c_void_p(int(cursor_ptr)), C.c_void_p(int(send_ptr) or None), C.c_void_p(int(recv_ptr)),
                                                     C.c_void_p(int(overflow_ptr))))

    def set_catalog_blocks(self, rows, stride):
        """the catalogs of the following K0 launches are blocks [block][column][rows] (stride = columns x rows); rows = 0: plain columns"""
        _lib.check(_lib.load().bfgx_plan_set_catalog_blocks(self._h, int(rows), int(stride)))

    def offsets_regrid_bands(self, cat_dev, B0, B1, offsets_ptr, band0, band1, map_in_ptr, out_slice_ptr, sums_ptr=0, foreign_ptr=0, acc_f64=False):
        """K0 + K1 for the bands [B0, B1), the banded regrid of [band0, band1) into the slice, far deposits applied, sums: one enqueue-only call"""
        _lib.check(_lib.load().bfgx_offsets_regrid_bands_device(self._h, C.byref(cat_dev), int(B0), int(B1), C.c_void_p(int(offsets_ptr)), int(acc_f64),
                                                               int(band0), int(band1), C.c_void_p(int(map_in_ptr)), C.c_void_p(int(out_slice_ptr)),
                                                               C.c_void_p(int(sums_ptr) or None), C.c_void_p(int(foreign_ptr) or None)))

    def bands_max_offset2(self, band0, band1, out_ptr):
        """largest |offset|^2 (float32, device) of the slice offsets_bands() has just written for the bands [band0, band1): reduced
        from the per-tile maxima K1 leaves, no pass over the slice"""
        _lib.check(_lib.load().bfgx_bands_max_offset2_device(self._h, int(band0), int(band1), C.c_void_p(int(out_ptr))))

    def max_offset2(self, offsets_ptr, npixels, out_ptr, acc_f64=False):
        """enqueue-only: *out (float32, device) = largest |offset|^2 over npixels pixels of pix_offsets"""
        _lib.check(_lib.load().bfgx_max_offset2_device(self._h, C.c_void_p(int(offsets_ptr) or None), int(npixels), int(acc_f64),
                                                      C.c_void_p(int(out_ptr))))

    def tile_shape(self):
        """(rings per band, columns per tile): band b holds the rings [1 + b R, 1 + (b + 1) R)"""
        br, w = C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.load().bfgx_plan_tile_shape(self._h, C.byref(br), C.byref(w)))
        return int(br.value), int(w.value)

    def disc_rings(self, cat_dev, rings_ptr):
        """per halo the ring range [first, last] (1-based, inclusive, 2 rings of margin) its disc can touch -> int32 [n][2]"""
        _lib.check(_lib.load().bfgx_disc_rings_device(self._h, C.byref(cat_dev), C.c_void_p(int(rings_ptr) or None)))

    def set_route_margin(self, rings):
        """disc_rings widens every halo's ring range by `rings` from now on (a rank that computes the bands next to its own as well)"""
        _lib.check(_lib.load().bfgx_plan_set_route_margin(self._h, int(rings)))

    def offsets_bands(self, cat_dev, band0, band1, offsets_slice_ptr, acc_f64=False):
        """K0 + K1 for the tiles of bands [band0, band1) only; the slice starts at the first pixel of band0 ([p1 - p0][3])"""
        _lib.check(_lib.load().bfgx_offsets_bands_device(self._h, C.byref(cat_dev), int(band0), int(band1), C.c_void_p(int(offsets_slice_ptr)),
                                                        int(acc_f64)))

    def paint_bands(self, cat_dev, band0, band1, map_slice_ptr, acc_f64=True):
        """K0 + K3 for the tiles of bands [band0, band1) only; the slice starts at the first pixel of band0"""
        _lib.check(_lib.load().bfgx_paint_bands_device(self._h, C.byref(cat_dev), int(band0), int(band1), C.c_void_p(int(map_slice_ptr)),
                                                      int(acc_f64)))

    def bands(self):
        """first RING pixel of every band of rings the tiling uses (+ npix): the unit of multi-GPU pixel ownership"""
        nb = C.c_int32(0)
        _lib.check(_lib.load().bfgx_plan_bands(self._h, C.byref(nb), None))
        first = np.zeros(nb.value + 1, dtype=np.int64)
        _lib.check(_lib.load().bfgx_plan_bands(self._h, C.byref(nb), first.ctypes.data))
        return first

    def reach_rings(self, max_offset):
        """rings of apron the banded regrid needs for summed pix_offsets whose largest |offset| is `max_offset` [rad]"""
        r = C.c_int32(0)
        _lib.check(_lib.load().bfgx_plan_reach_rings(self._h, float(max_offset), C.byref(r)))
        return int(r.value)

    def set_band_reach(self, rings):
        """every rank of a banded regrid must use the same apron (default 1 ring): see reach_rings"""
        _lib.check(_lib.load().bfgx_plan_set_band_reach(self._h, int(rings)))

    def band_apron(self, band0, band1):
        """pixel range [olo, ohi) of summed pix_offsets the owner of bands [band0, band1) needs: its pixels + the band
        reach (set_band_reach) in rings either side"""
        lo, hi = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().bfgx_plan_band_apron(self._h, int(band0), int(band1), C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def regrid_bands(self, band0, band1, map_in_ptr, offsets_ptr, olo, ohi, out_slice_ptr, sums_ptr=0, acc_f64=False):
        """K2 for the OUTPUT pixels of bands [band0, band1): offsets_ptr -> summed pix_offsets of pixels [olo, ohi) (band_apron),
        out_slice_ptr -> the bands' own pixels (stored once, no zero-fill).  Far deposits: far_fetch()."""
        _lib.check(_lib.load().bfgx_regrid_bands_device(self._h, int(band0), int(band1), C.c_void_p(int(map_in_ptr)),
                                                       C.c_void_p(int(offsets_ptr)), int(olo), int(ohi), int(acc_f64),
                                                       C.c_void_p(int(out_slice_ptr)), C.c_void_p(int(sums_ptr) or None)))

    def far_apply(self, out_slice_ptr, p0, p1, foreign_ptr=0):
        """enqueue-only: add the listed deposits that fall into pixels [p0, p1) to the slice; count the others into *foreign_ptr"""
        _lib.check(_lib.load().bfgx_plan_far_apply_device(self._h, C.c_void_p(int(out_slice_ptr)), int(p0), int(p1),
                                                         C.c_void_p(int(foreign_ptr) or None)))

    def far_fetch(self):
        """(pixels, values) of the deposits the last regrid listed instead of applying (banded regrid only); blocking"""
        n = C.c_int64(0)
        _lib.check(_lib.load().bfgx_plan_far_fetch(self._h, 0, None, None, C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0)
        pix, val = np.zeros(n.value, dtype=np.int64), np.zeros(n.value)
        _lib.check(_lib.load().bfgx_plan_far_fetch(self._h, n.value, pix.ctypes.data, val.ctypes.data, C.byref(n)))
        return pix, val

    def paint(self, cat_dev, map_out_ptr, acc_f64=True):
        """K0 + K3 (HealpixRunner.py:418-445).  acc_f64: True / 1 = fp64 throughout (double map), False / 0 = fp32 (float map),
        2 = fp32 pair math accumulated in fp64 into a double map"""
        _lib.check(_lib.load().bfgx_paint_device(self._h, C.byref(cat_dev), C.c_void_p(int(map_out_ptr)), int(acc_f64)))

    def count_pairs(self, cat_dev, fallback4=True, counts_ptr=0):
        tot = C.c_int64(0)
        _lib.check(_lib.load().bfgx_count_pairs_device(self._h, C.byref(cat_dev), int(fallback4),
                                                      C.c_void_p(int(counts_ptr) or None), C.byref(tot)))
        return int(tot.value)

    def precision(self, acc_f64=-1):
        """(BFGX_ACC_* a request resolves to on this plan, largest displacement of the plan's table in pixel sides of its NSIDE)"""
        r, d = C.c_int32(0), C.c_double(0.0)
        _lib.check(_lib.load().bfgx_plan_precision(self._h, int(acc_f64), C.byref(r), C.byref(d)))
        return int(r.value), float(d.value)

    def set_algo(self, algo):
        """1 = tile-owned LDS accumulators (default), 0 = one wave per halo + global float atomics"""
        _lib.check(_lib.load().bfgx_plan_set_algo(self._h, int(algo)))

    def status(self):
        _lib.check(_lib.load().bfgx_plan_status(self._h))

    def regrid_stats(self):
        """what the last full-map regrid did (blocking): far deposits listed, list overflowed, tiles run by the walking kernel, largest reach"""
        far, ovf, walked, reach = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.load().bfgx_plan_regrid_stats(self._h, C.byref(far), C.byref(ovf), C.byref(walked), C.byref(reach)))
        return {"far_listed": int(far.value), "far_overflowed": bool(ovf.value), "tiles_walked": int(walked.value), "max_reach_rings": int(reach.value)}

    def timing_enable(self, on=True):
        _lib.check(_lib.load().bfgx_plan_timing_enable(self._h, int(on)))

    def timing_read(self):
        """{kernel kind: (summed ms, launches)} since the last read; synchronises the stream"""
        ms = np.zeros(len(_lib.KERNEL_KINDS))
        n = np.zeros(len(_lib.KERNEL_KINDS), dtype=np.int64)
        _lib.check(_lib.load().bfgx_plan_timing_read(self._h, ms.ctypes.data, n.ctypes.data))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_lib.KERNEL_KINDS)}


def model_from_tables(axes, values, cosmo, eps_runner, eps_model=None, rdelta_sampling=False, log_values=False,
                      cosmo_model=None, massdef=(200.0, 'critical'), massdef_model=None):
    """bfgx_model from raw arrays: axes = [ln(1+z), ln M, ln r], values C-order."""
    table, keep = _lib.make_table(axes, values, rdelta_sampling, log_values,
                                  eps_model if eps_model is not None else eps_runner)
    m = _lib.bfgx_model()
    m.table = table
    m.cosmo_runner = _lib.make_cosmo(cosmo)
    m.cosmo_model = _lib.make_cosmo(cosmo_model if cosmo_model is not None else cosmo)
    m.massdef_runner = _lib.make_massdef(*massdef)
    m.massdef_model = _lib.make_massdef(*(massdef_model if massdef_model is not None else massdef))
    m.eps_runner = float(eps_runner)
    return m, keep


class GridPlan(object):
    """Resident front-end of the regular-grid path (`bfgx_grid_plan`): periodic 2D / 3D maps at one redshift."""

    def __init__(self, model, keepalive, bins, ndim, redshift, max_halos, device=0, stream=0):
        self._keep = keepalive
        self.ndim = int(ndim)
        self.npix = int(len(bins))
        self.ntot = self.npix ** self.ndim
        self.max_halos = int(max_halos)
        self.device = int(device)
        self.stream = int(stream)
        grid, gkeep = _lib.make_grid(bins, ndim, redshift)
        h = C.c_void_p()
        _lib.check(_lib.load().bfgx_grid_plan_create(self.device, C.c_void_p(self.stream or None), C.byref(grid),
                                                    self.max_halos, C.byref(model), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            try:
                _lib.load().bfgx_grid_plan_destroy(self._h)
            except Exception:
                pass
            self._h = None

    __del__ = close

    def offsets(self, cat_dev, offsets_ptr):
        """halo loop of BaryonifyGrid (Map2DRunner.py:476-575): offsets[npix^d][d] f64, zeroed then filled.
        Returns the number of contributing (halo, pixel) pairs."""
        n = C.c_int64(0)
        _lib.check(_lib.load().bfgx_grid_offsets_device(self._h, C.byref(cat_dev), C.c_void_p(int(offsets_ptr)), C.byref(n)))
        return int(n.value)

    def paint(self, cat_dev, map_out_ptr):
        """halo loop of PaintProfilesGrid (Map2DRunner.py:708-812)"""
        n = C.c_int64(0)
        _lib.check(_lib.load().bfgx_grid_paint_device(self._h, C.byref(cat_dev), C.c_void_p(int(map_out_ptr)), C.byref(n)))
        return int(n.value)

    def regrid(self, map_in_ptr, offsets_ptr, map_out_ptr, sums_ptr=0):
        """post-loop regrid (Map2DRunner.py:577-605): map_out zeroed then filled; sums[2] optional (zeroed by caller)"""
        _lib.check(_lib.load().bfgx_grid_regrid_device(self._h, C.c_void_p(int(map_in_ptr)), C.c_void_p(int(offsets_ptr)),
                                                      C.c_void_p(int(map_out_ptr)), C.c_void_p(int(sums_ptr) or None)))

    def baryonify(self, cat_dev, map_in_ptr, map_out_ptr, sums_ptr=0):
        """BaryonifyGrid.process() on device arrays as one cell-owned pass (no pix_offsets array): halos listed per block of cells,
        every cell sums its offsets in registers and is regridded at once.  Same map_out as offsets() + regrid() up to the order
        of fp64 sums.  Returns the number of contributing (halo, pixel) pairs."""
        n = C.c_int64(0)
        _lib.check(_lib.load().bfgx_grid_baryonify_device(self._h, C.byref(cat_dev), C.c_void_p(int(map_in_ptr)), C.c_void_p(int(map_out_ptr)),
                                                         C.c_void_p(int(sums_ptr) or None), C.byref(n)))
        return int(n.value)

    def deposit_baryonify(self, cat_dev, n_particles, x_ptr, y_ptr, z_ptr, mass_ptr, edges_ptr, map_in_ptr, map_out_ptr, sums_ptr=0):
        """ParticleSnapshot.make_map + BaryonifyGrid.process() in one call (device arrays): the histogram of the particles goes to
        map_in AND, as the start value of the cell-owned pass, to map_out while it is being stored -- the map is not read again to copy
        and sum it.  Returns the number of contributing (halo, pixel) pairs."""
        n = C.c_int64(0)
        _lib.check(_lib.load().bfgx_grid_deposit_baryonify_device(
            self._h, C.byref(cat_dev), int(n_particles), C.c_void_p(int(x_ptr)), C.c_void_p(int(y_ptr)), C.c_void_p(int(z_ptr) or None),
            C.c_void_p(int(mass_ptr) or None), C.c_void_p(int(edges_ptr)), C.c_void_p(int(map_in_ptr)), C.c_void_p(int(map_out_ptr)),
            C.c_void_p(int(sums_ptr) or None), C.byref(n)))
        return int(n.value)

    def set_slab(self, plane_lo, plane_n):
        """slab decomposition over GPUs: this plan owns the planes [plane_lo, plane_lo + plane_n) of the first array axis;
        offsets() / paint() then fill plane_n x npix (x npix) cells (pass the whole catalog), regrid_slab() regrids them"""
        _lib.check(_lib.load().bfgx_grid_plan_set_slab(self._h, int(plane_lo), int(plane_n)))

    def regrid_slab(self, map_in_ptr, offsets_ptr, apron, map_out_ptr, sums_ptr=0, missed_ptr=0):
        """regrid of the slab's source cells into map_out = slab + `apron` planes either side (zeroed by the call, periodic);
        *missed (int32, zeroed by the caller) is set if a deposit fell outside that buffer"""
        _lib.check(_lib.load().bfgx_grid_regrid_slab_device(self._h, C.c_void_p(int(map_in_ptr)), C.c_void_p(int(offsets_ptr)), int(apron),
                                                           C.c_void_p(int(map_out_ptr)), C.c_void_p(int(sums_ptr) or None),
                                                           C.c_void_p(int(missed_ptr) or None)))

    def timing_enable(self, on=True):
        _lib.check(_lib.load().bfgx_grid_plan_timing_enable(self._h, int(on)))

    def timing_read(self):
        ms = np.zeros(len(_lib.KERNEL_KINDS))
        n = np.zeros(len(_lib.KERNEL_KINDS), dtype=np.int64)
        _lib.check(_lib.load().bfgx_grid_plan_timing_read(self._h, ms.ctypes.data, n.ctypes.data))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_lib.KERNEL_KINDS)}


def deposit_particles_slab_device(x_ptr, y_ptr, z_ptr, mass_ptr, n, n_grid, edges_ptr, plane_lo, plane_n, map_out_ptr, ndim=3, device=0, stream=0):
    """ParticleSnapshot.make_map for the planes [plane_lo, plane_lo + plane_n) of the first axis (particles elsewhere are dropped)"""
    _lib.check(_lib.load().bfgx_deposit_particles_slab_device(int(device), C.c_void_p(int(stream) or None), int(ndim), int(n),
                                                             C.c_void_p(int(x_ptr)), C.c_void_p(int(y_ptr)), C.c_void_p(int(z_ptr) or None),
                                                             C.c_void_p(int(mass_ptr) or None), int(n_grid), C.c_void_p(int(edges_ptr)),
                                                             int(plane_lo), int(plane_n), C.c_void_p(int(map_out_ptr))))


def fft_slab_planes_device(map_ptr, n_grid, planes, work_ptr, device=0, stream=0):
    """slab P(k), step 1: transforms along the last two axes of `planes` planes; work = complex128 [planes][n][n/2+1]"""
    _lib.check(_lib.load().bfgx_fft_slab_planes_device(int(device), C.c_void_p(int(stream) or None), int(n_grid), int(planes),
                                                      C.c_void_p(int(map_ptr)), C.c_void_p(int(work_ptr))))


def fft_slab_axis0_pk_device(work_ptr, n_grid, ncols, col0, L, nk, pk_sum_ptr, k_sum_ptr, counts_ptr, device=0, stream=0):
    """slab P(k), step 3 (after the transpose): transform along the first axis of work [n][ncols][n/2+1] + partial bin sums"""
    _lib.check(_lib.load().bfgx_fft_slab_axis0_pk_device(int(device), C.c_void_p(int(stream) or None), int(n_grid), int(ncols), int(col0),
                                                        C.c_void_p(int(work_ptr)), float(L), int(nk), C.c_void_p(int(pk_sum_ptr)),
                                                        C.c_void_p(int(k_sum_ptr)), C.c_void_p(int(counts_ptr))))


def deposit_particles_device(x_ptr, y_ptr, z_ptr, mass_ptr, n, n_grid, edges_ptr, map_out_ptr, ndim=3, device=0, stream=0):
    """ParticleSnapshot.make_map on device-resident columns (io.py:622-670)"""
    _lib.check(_lib.load().bfgx_deposit_particles_device(int(device), C.c_void_p(int(stream) or None), int(ndim), int(n),
                                                        C.c_void_p(int(x_ptr)), C.c_void_p(int(y_ptr)),
                                                        C.c_void_p(int(z_ptr) or None), C.c_void_p(int(mass_ptr) or None),
                                                        int(n_grid), C.c_void_p(int(edges_ptr)), C.c_void_p(int(map_out_ptr))))


def route_particles_count_device(x_ptr, n, n_grid, edges_ptr, world, owner_ptr, counts_ptr, device=0, stream=0):
    """pass 1 of the particle routing of a slab-decomposed grid (enqueue-only): owner[n] uint8, counts[world] int32"""
    _lib.check(_lib.load().bfgx_route_particles_count_device(int(device), C.c_void_p(int(stream) or None), int(n), C.c_void_p(int(x_ptr) or None), int(n_grid),
                                                            C.c_void_p(int(edges_ptr)), int(world), C.c_void_p(int(owner_ptr) or None), C.c_void_p(int(counts_ptr))))


def route_particles_fill_device(col_ptrs, n, owner_ptr, world, start, total, cursor_ptr, cols_out_ptr, device=0, stream=0):
    """pass 2: cols_out[c][start[d] + ...] = column c of the particles bound for rank d (len(col_ptrs) <= 4 columns)"""
    st = np.ascontiguousarray(start, dtype=np.int64)
    cp = (C.c_void_p * len(col_ptrs))(*[int(q) for q in col_ptrs])
    _lib.check(_lib.load().bfgx_route_particles_fill_device(int(device), C.c_void_p(int(stream) or None), int(n), len(col_ptrs), cp, C.c_void_p(int(owner_ptr) or None),
                                                           int(world), st.ctypes.data, int(total), C.c_void_p(int(cursor_ptr)), C.c_void_p(int(cols_out_ptr) or None)))


def fft_pitch(n_grid):
    """complex values per row of the half-spectrum work arrays: n/2 + 1 rounded up to whole 128-byte lines"""
    return int(_lib.load().bfgx_fft_pitch(int(n_grid)))


def power_spectrum_work_doubles(n_grid):
    """doubles of scratch power_spectrum_device needs (half spectrum, rows padded to whole 128-byte lines)"""
    return int(_lib.load().bfgx_power_spectrum_work_doubles(int(n_grid)))


def power_spectrum_device(map_ptr, n_grid, L, nk, work_ptr, pk_sum_ptr, k_sum_ptr, counts_ptr, device=0, stream=0):
    """FFT + |F|^2 + linear k-bins on a device-resident map; work = power_spectrum_work_doubles(n_grid) doubles of scratch"""
    _lib.check(_lib.load().bfgx_power_spectrum_device(int(device), C.c_void_p(int(stream) or None), int(n_grid),
                                                     C.c_void_p(int(map_ptr)), float(L), int(nk), C.c_void_p(int(work_ptr)),
                                                     C.c_void_p(int(pk_sum_ptr)), C.c_void_p(int(k_sum_ptr)),
                                                     C.c_void_p(int(counts_ptr))))


def power_spectrum(Map, Lbox, Nk=180, device=0):
    """P(k) summary of a cubic map as in examples/10_Reproduce_Schneider_deltaPk.ipynb (cells 12, 15): |FFT|^2 averaged in
    Nk linear bins between the fundamental and the Nyquist frequency.  Returns (k_cen, Pk, k_count)."""
    Map = _lib.f8(Map)
    if Map.ndim != 3 or len(set(Map.shape)) != 1:
        raise ValueError("power_spectrum needs a cubic 3-D map")
    pk, kc = np.empty(Nk), np.empty(Nk)
    cnt = np.empty(Nk, dtype=np.int64)
    _lib.check(_lib.load().bfgx_power_spectrum(int(device), Map.shape[0], Map.ctypes.data, float(Lbox), int(Nk),
                                              pk.ctypes.data, kc.ctypes.data, cnt.ctypes.data))
    return kc, pk, cnt


def baryonify_snapshot_device(model, halos_dev, part_ptrs, n_part, L, redshift, out_ptrs, device=0, stream=0):
    """BaryonifySnapshot on device-resident columns: part_ptrs / out_ptrs = (x, y[, z]) device pointers.
    Returns the number of displaced (halo, particle) pairs."""
    ndim = len(part_ptrs)
    s = _lib.bfgx_snapshot(ndim, 0, int(n_part), int(part_ptrs[0]), int(part_ptrs[1]), int(part_ptrs[2]) if ndim == 3 else None,
                           float(L), float(redshift))
    n = C.c_int64(0)
    _lib.check(_lib.load().bfgx_baryonify_snapshot_device(int(device), C.c_void_p(int(stream) or None), C.byref(halos_dev), C.byref(model),
                                                         C.byref(s), C.c_void_p(int(out_ptrs[0])), C.c_void_p(int(out_ptrs[1])),
                                                         C.c_void_p(int(out_ptrs[2])) if ndim == 3 else None, C.byref(n)))
    return int(n.value)


class SnapshotPlan(object):
    """Resident front-end of the particle-snapshot path (`bfgx_snapshot_plan`): the model and the halo-cell workspace stay
    on the device between calls."""

    def __init__(self, model, keepalive, ndim, L, redshift, max_halos, device=0, stream=0):
        self._keep = keepalive
        self.ndim = int(ndim)
        h = C.c_void_p()
        _lib.check(_lib.load().bfgx_snapshot_plan_create(int(device), C.c_void_p(int(stream) or None), C.byref(model), self.ndim,
                                                        float(L), float(redshift), int(max_halos), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            try:
                _lib.load().bfgx_snapshot_plan_destroy(self._h)
            except Exception:
                pass
            self._h = None

    __del__ = close

    def displace(self, halos_dev, n_part, part_ptrs, out_ptrs):
        """BaryonifySnapshot.process (SnapshotRunner.py:199-262) on device columns; returns the number of displaced pairs"""
        n = C.c_int64(0)
        z_in = C.c_void_p(int(part_ptrs[2])) if self.ndim == 3 else None
        z_out = C.c_void_p(int(out_ptrs[2])) if self.ndim == 3 else None
        _lib.check(_lib.load().bfgx_snapshot_displace_device(self._h, C.byref(halos_dev), int(n_part), C.c_void_p(int(part_ptrs[0])),
                                                            C.c_void_p(int(part_ptrs[1])), z_in, C.c_void_p(int(out_ptrs[0])),
                                                            C.c_void_p(int(out_ptrs[1])), z_out, C.byref(n)))
        return int(n.value)

    def displace_deposit(self, halos_dev, n_part, part_ptrs, mass_ptr, n_grid, edges_ptr, map_out_ptr):
        """BaryonifySnapshot.process() followed by ParticleSnapshot.make_map(n_grid) of the result (SnapshotRunner.py:173-262, io.py:622-670)
        when only the map is wanted: map_out [n_grid^ndim] float64 on the device; the displaced coordinates are never stored.  mass_ptr 0 =
        unit masses.  Returns the number of displaced pairs."""
        n = C.c_int64(0)
        z_in = C.c_void_p(int(part_ptrs[2])) if self.ndim == 3 else None
        _lib.check(_lib.load().bfgx_snapshot_displace_deposit_device(self._h, C.byref(halos_dev), int(n_part), C.c_void_p(int(part_ptrs[0])),
                                                                    C.c_void_p(int(part_ptrs[1])), z_in, C.c_void_p(int(mass_ptr) or None),
                                                                    int(n_grid), C.c_void_p(int(edges_ptr)), C.c_void_p(int(map_out_ptr)), C.byref(n)))
        return int(n.value)
