"""
Stand-in for the `healpy` functions BaryonForge's HEALPix runner calls
(HealpixRunner.py:303,306,310,312,333,334,337; io.py:344,350), written in vectorised numpy
from the published HEALPix RING-scheme definitions (Gorski et al. 2005, ApJ 622, 759).

TEST INFRASTRUCTURE ONLY: it lets the *unmodified* reference runner be imported and run in the
build container (where healpy cannot be installed) in order to generate tests/golden/*.npz.
It is a second, independent implementation next to oracle/bfg_oracle.c; the two are
cross-checked in tests/test_oracle_healpix.py.  Never imported by the product.
"""
import numpy as np

__all__ = ['nside2npix', 'npix2nside', 'nside2resol', 'ang2vec', 'vec2ang', 'pix2vec', 'pix2ang',
           'query_disc', 'get_interp_weights', 'read_map']


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def npix2nside(npix):
    nside = int(round(np.sqrt(npix / 12.0)))
    if 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    return nside


def nside2resol(nside, arcmin=False):
    resol = np.sqrt(4 * np.pi / nside2npix(nside))
    return np.rad2deg(resol) * 60 if arcmin else resol


def read_map(*args, **kwargs):
    raise NotImplementedError("refshim.healpy has no FITS reader")


def _lonlat2thetaphi(lon, lat):
    return np.pi / 2.0 - np.radians(lat), np.radians(lon)


def _thetaphi2lonlat(theta, phi):
    return np.degrees(phi), 90.0 - np.degrees(theta)


def ang2vec(theta, phi, lonlat=False):
    if lonlat:
        theta, phi = _lonlat2thetaphi(theta, phi)
    theta, phi = np.asarray(theta, dtype=np.float64), np.asarray(phi, dtype=np.float64)
    if np.any(theta < 0) or np.any(theta > np.pi):
        raise ValueError("THETA is out of range [0,pi]")
    st = np.sin(theta)
    return np.array([st * np.cos(phi), st * np.sin(phi), np.cos(theta)]).T


def vec2ang(vectors, lonlat=False):
    vectors = np.asarray(vectors, dtype=np.float64).reshape(-1, 3)
    dnorm = np.sqrt(np.sum(np.square(vectors), axis=1))
    theta = np.arccos(vectors[:, 2] / dnorm)
    phi = np.arctan2(vectors[:, 1], vectors[:, 0])
    phi[phi < 0] += 2 * np.pi
    if lonlat:
        return _thetaphi2lonlat(theta, phi)
    return theta, phi


# --- ring bookkeeping ---------------------------------------------------------------------

def _ring_layout(nside, ring):
    """(start pixel, pixels in ring, shifted?, z of ring) for ring numbers 1..4nside-1 (arrays)."""
    ring = np.asarray(ring, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    north = ring < nside
    south = ring > 3 * nside
    rs = 4 * nside - ring                       # ring index counted from the south pole
    npr = np.where(north, 4 * ring, np.where(south, 4 * rs, 4 * nside))
    start = np.where(north, 2 * ring * (ring - 1),
                     np.where(south, npix - 2 * rs * (rs + 1), ncap + (ring - nside) * 4 * nside))
    shifted = np.where(north | south, True, ((ring - nside) % 2) == 0)
    z = np.where(north, 1.0 - ring.astype(np.float64) ** 2 * (4.0 / npix),
                 np.where(south, rs.astype(np.float64) ** 2 * (4.0 / npix) - 1.0,
                          (2 * nside - ring) * (2.0 / (3.0 * nside))))
    return start, npr, shifted, z


def _pix2ring(nside, ipix):
    ipix = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    ring = np.empty(ipix.shape, dtype=np.int64)
    n = ipix < ncap
    s = ipix >= npix - ncap
    e = ~(n | s)
    # north cap: ring i holds pixels [2i(i-1), 2i(i+1))
    rn = ((1 + np.floor(np.sqrt(1 + 2 * ipix[n].astype(np.float64) + 0.5))).astype(np.int64)) // 2
    rn = np.where(2 * rn * (rn - 1) > ipix[n], rn - 1, rn)
    rn = np.where(2 * rn * (rn + 1) <= ipix[n], rn + 1, rn)
    ring[n] = rn
    ring[e] = (ipix[e] - ncap) // (4 * nside) + nside
    q = npix - 1 - ipix[s]                       # mirror
    rsn = ((1 + np.floor(np.sqrt(1 + 2 * q.astype(np.float64) + 0.5))).astype(np.int64)) // 2
    rsn = np.where(2 * rsn * (rsn - 1) > q, rsn - 1, rsn)
    rsn = np.where(2 * rsn * (rsn + 1) <= q, rsn + 1, rsn)
    ring[s] = 4 * nside - rsn
    return ring


def pix2ang(nside, ipix, nest=False, lonlat=False):
    assert not nest
    ipix = np.asarray(ipix, dtype=np.int64)
    ring = _pix2ring(nside, ipix)
    start, npr, shifted, z = _ring_layout(nside, ring)
    k = ipix - start
    phi = (k + np.where(shifted, 0.5, 0.0)) * (2 * np.pi / npr)
    theta = np.arccos(z)
    if lonlat:
        return _thetaphi2lonlat(theta, phi)
    return theta, phi


def pix2vec(nside, ipix, nest=False):
    assert not nest
    ipix = np.asarray(ipix, dtype=np.int64)
    ring = _pix2ring(nside, ipix)
    start, npr, shifted, z = _ring_layout(nside, ring)
    k = ipix - start
    phi = (k + np.where(shifted, 0.5, 0.0)) * (2 * np.pi / npr)
    # sin(theta): near the poles use the cancellation-free form
    npix = 12 * nside * nside
    rr = np.minimum(ring, 4 * nside - ring).astype(np.float64)
    tmp = rr * rr * (4.0 / npix)
    sth = np.where(np.abs(z) > 0.99, np.sqrt(tmp * (2.0 - tmp)), np.sqrt((1.0 - z) * (1.0 + z)))
    return sth * np.cos(phi), sth * np.sin(phi), z


def _ring_above(nside, z):
    z = np.asarray(z, dtype=np.float64)
    az = np.abs(z)
    eq = (nside * (2.0 - 1.5 * z)).astype(np.int64)
    cap = (nside * np.sqrt(3.0 * (1.0 - az))).astype(np.int64)
    return np.where(az <= 2.0 / 3.0, eq, np.where(z > 0, cap, 4 * nside - cap - 1))


def query_disc(nside, vec, radius, inclusive=False, fact=4, nest=False, buff=None):
    """Pixels whose CENTRE lies within `radius` [rad] of `vec`; ascending RING order."""
    assert (not inclusive) and (not nest)
    vec = np.asarray(vec, dtype=np.float64)
    theta0 = np.arctan2(np.sqrt(vec[0] ** 2 + vec[1] ** 2), vec[2])
    phi0 = np.arctan2(vec[1], vec[0])
    if phi0 < 0:
        phi0 += 2 * np.pi
    npix = 12 * nside * nside
    if radius >= np.pi:
        return np.arange(npix, dtype=np.int64)
    z0 = np.cos(theta0)
    s0 = np.sqrt((1 - z0) * (1 + z0))
    cosr = np.cos(radius)
    pieces = []
    rlat1, rlat2 = theta0 - radius, theta0 + radius
    irmin = int(_ring_above(nside, np.cos(rlat1))) + 1
    irmax = int(_ring_above(nside, np.cos(rlat2)))
    if rlat1 <= 0 and irmin > 1:                           # north pole inside the disc
        st, npr, _, _ = _ring_layout(nside, irmin - 1)
        pieces.append(np.arange(0, int(st + npr), dtype=np.int64))
    if irmax >= irmin:
        rings = np.arange(irmin, irmax + 1, dtype=np.int64)
        start, npr, shifted, z = _ring_layout(nside, rings)
        x = (cosr - z * z0) / s0
        ysq = 1 - z * z - x * x
        ok = ysq > 0
        dphi = np.where(ok, np.arctan2(np.sqrt(np.where(ok, ysq, 0.0)), x), 0.0)
        shift = np.where(shifted, 0.5, 0.0)
        ip_lo = np.floor(npr / (2 * np.pi) * (phi0 - dphi) - shift).astype(np.int64) + 1
        ip_hi = np.floor(npr / (2 * np.pi) * (phi0 + dphi) - shift).astype(np.int64)
        for r in range(rings.size):
            if not (dphi[r] > 0):
                continue
            lo, hi, n, st = int(ip_lo[r]), int(ip_hi[r]), int(npr[r]), int(start[r])
            if hi >= n:
                lo -= n
                hi -= n
            if lo < 0:
                pieces.append(np.arange(st, st + hi + 1, dtype=np.int64))
                pieces.append(np.arange(st + lo + n, st + n, dtype=np.int64))
            else:
                pieces.append(np.arange(st + lo, st + hi + 1, dtype=np.int64))
    if rlat2 >= np.pi and irmax + 1 < 4 * nside:          # south pole inside the disc
        st, _, _, _ = _ring_layout(nside, irmax + 1)
        pieces.append(np.arange(int(st), npix, dtype=np.int64))
    if not pieces:
        return np.zeros(0, dtype=np.int64)
    return np.concatenate(pieces)


def _ring_theta(nside, ring):
    """colatitude of ring centres (rings 1..4nside-1), the way get_interp_weights needs them"""
    ring = np.asarray(ring, dtype=np.int64)
    npix = 12 * nside * nside
    nr = np.where(ring > 2 * nside, 4 * nside - ring, ring)
    tmp = nr.astype(np.float64) ** 2 * (4.0 / npix)
    th_cap = np.arctan2(np.sqrt(tmp * (2 - tmp)), 1 - tmp)
    th_eq = np.arccos(np.clip((2 * nside - nr) * (2.0 / (3.0 * nside)), -1, 1))
    th = np.where(nr < nside, th_cap, th_eq)
    return np.where(ring > 2 * nside, np.pi - th, th)


def get_interp_weights(nside, theta, phi=None, nest=False, lonlat=False):
    """Bilinear interpolation neighbours and weights, shapes (4, N) like healpy."""
    assert not nest
    if lonlat:
        theta, phi = _lonlat2thetaphi(theta, phi)
    scalar = np.ndim(theta) == 0
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    phi = np.atleast_1d(np.asarray(phi, dtype=np.float64))
    npix = 12 * nside * nside
    N = theta.size
    z = np.cos(theta)
    ir1 = _ring_above(nside, z)
    ir2 = ir1 + 1
    pix = np.zeros((4, N), dtype=np.int64)
    wgt = np.zeros((4, N))
    th = [np.zeros(N), np.zeros(N)]
    for half, ir in enumerate((ir1, ir2)):
        valid = (ir >= 1) & (ir <= 4 * nside - 1)
        irc = np.clip(ir, 1, 4 * nside - 1)
        start, npr, shifted, _ = _ring_layout(nside, irc)
        th[half] = _ring_theta(nside, irc)
        dphi = 2 * np.pi / npr
        sh = np.where(shifted, 0.5, 0.0)
        tmp = phi / dphi - sh
        i1 = np.where(tmp < 0, tmp.astype(np.int64) - 1, tmp.astype(np.int64))
        w1 = (phi - (i1 + sh) * dphi) / dphi
        i2 = i1 + 1
        i1 = np.where(i1 < 0, i1 + npr, i1)
        i2 = np.where(i2 >= npr, i2 - npr, i2)
        pix[2 * half] = np.where(valid, start + i1, 0)
        pix[2 * half + 1] = np.where(valid, start + i2, 0)
        wgt[2 * half] = np.where(valid, 1 - w1, 0.0)
        wgt[2 * half + 1] = np.where(valid, w1, 0.0)
    theta1, theta2 = th
    npole = ir1 == 0
    spole = ir2 == 4 * nside
    mid = ~(npole | spole)
    # regular case
    wt = np.where(mid, (theta - theta1) / np.where(mid, theta2 - theta1, 1.0), 0.0)
    for k in (0, 1):
        wgt[k] = np.where(mid, wgt[k] * (1 - wt), wgt[k])
        wgt[k + 2] = np.where(mid, wgt[k + 2] * wt, wgt[k + 2])
    # north pole: missing upper ring -> the 4 polar pixels at 1/4 each
    wtn = theta / theta2
    facn = (1 - wtn) * 0.25
    w2n, w3n = wgt[2] * wtn + facn, wgt[3] * wtn + facn
    p0n, p1n = (pix[2] + 2) & 3, (pix[3] + 2) & 3
    # south pole
    wts = (theta - theta1) / (np.pi - theta1)
    facs = wts * 0.25
    w0s, w1s = wgt[0] * (1 - wts) + facs, wgt[1] * (1 - wts) + facs
    p2s, p3s = ((pix[0] + 2) & 3) + npix - 4, ((pix[1] + 2) & 3) + npix - 4
    wgt[0] = np.where(npole, facn, np.where(spole, w0s, wgt[0]))
    wgt[1] = np.where(npole, facn, np.where(spole, w1s, wgt[1]))
    wgt[2] = np.where(npole, w2n, np.where(spole, facs, wgt[2]))
    wgt[3] = np.where(npole, w3n, np.where(spole, facs, wgt[3]))
    pix[0] = np.where(npole, p0n, pix[0])
    pix[1] = np.where(npole, p1n, pix[1])
    pix[2] = np.where(spole, p2s, pix[2])
    pix[3] = np.where(spole, p3s, pix[3])
    if scalar:
        return pix[:, 0], wgt[:, 0]
    return pix, wgt
