"""The Rayleigh period equation in extended precision (numpy longdouble, 64-bit mantissa on x86):
an arbiter between two double-precision evaluations near a root.  Test infrastructure: the same
formulas as the oracle's dltar4 restatement (oracle/surf96_oracle.c), term by term."""
import numpy as np

LD = np.longdouble


def compound_ld(d, a, b, rho, period, c):
    """All five components of the compound vector at the surface."""
    d, a, b, rho = (np.asarray(x, dtype=LD) for x in (d, a, b, rho))
    omega = LD(2) * LD(np.pi) / LD(period)
    omega = 2 * np.arctan2(LD(0), LD(-1)) / LD(period)          # pi in extended precision
    k = omega / LD(c)
    k2 = k * k
    L = len(d)
    xka, xkb = omega / a[-1], omega / b[-1]
    ra = np.sqrt((k + xka) * abs(k - xka)); rb = np.sqrt((k + xkb) * abs(k - xkb))
    t = b[-1] / omega; gammk = 2 * t * t; gam = gammk * k2; gamm1 = gam - 1
    r1 = rho[-1]
    e = np.array([r1 * r1 * (gamm1 * gamm1 - gam * gammk * ra * rb), -r1 * ra,
                  r1 * (gamm1 - gammk * ra * rb), r1 * rb, k2 - ra * rb], dtype=LD)
    for m in range(L - 2, -1, -1):
        xka, xkb = omega / a[m], omega / b[m]
        t = b[m] / omega; gammk = 2 * t * t; gam = gammk * k2
        ra = np.sqrt((k + xka) * abs(k - xka)); rb = np.sqrt((k + xkb) * abs(k - xkb))
        dp = d[m]; r1 = rho[m]; p = ra * dp; q = rb * dp
        pex = sex = LD(0)
        if k < xka:
            sinp = np.sin(p); w = sinp / ra; x = -ra * sinp; cosp = np.cos(p)
        elif k == xka:
            cosp = LD(1); w = dp; x = LD(0)
        else:
            pex = p; fac = np.exp(-2 * p) if p < 16 else LD(0)
            cosp = (1 + fac) / 2; sinp = (1 - fac) / 2; w = sinp / ra; x = ra * sinp
        if k < xkb:
            sinq = np.sin(q); y = sinq / rb; z = -rb * sinq; cosq = np.cos(q)
        elif k == xkb:
            cosq = LD(1); y = dp; z = LD(0)
        else:
            sex = q; fac = np.exp(-2 * q) if q < 16 else LD(0)
            cosq = (1 + fac) / 2; sinq = (1 - fac) / 2; y = sinq / rb; z = rb * sinq
        exa = pex + sex
        a0 = np.exp(-exa) if exa < 60 else LD(0)
        cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz = cosp * cosq, cosp * y, cosp * z, cosq * w, cosq * x, x * y, x * z, w * y, w * z
        gamm1 = gam - 1; twgm1 = gam + gamm1; gmgmk = gam * gammk; gmgm1 = gam * gamm1
        gm1sq = gamm1 * gamm1; rho2 = r1 * r1; a0pq = a0 - cpcq; tt = -2 * k2
        ca = np.zeros((5, 5), dtype=LD)
        ca[0, 0] = cpcq - 2 * gmgm1 * a0pq - gmgmk * xz - k2 * gm1sq * wy
        ca[0, 1] = (k2 * cpy - cqx) / r1
        ca[0, 2] = -(twgm1 * a0pq + gammk * xz + k2 * gamm1 * wy) / r1
        ca[0, 3] = (cpz - k2 * cqw) / r1
        ca[0, 4] = -(2 * k2 * a0pq + xz + k2 * k2 * wy) / rho2
        ca[1, 0] = (gmgmk * cpz - gm1sq * cqw) * r1; ca[1, 1] = cpcq
        ca[1, 2] = gammk * cpz - gamm1 * cqw; ca[1, 3] = -wz; ca[1, 4] = ca[0, 3]
        ca[3, 0] = (gm1sq * cpy - gmgmk * cqx) * r1; ca[3, 1] = -xy
        ca[3, 2] = gamm1 * cpy - gammk * cqx; ca[3, 3] = ca[1, 1]; ca[3, 4] = ca[0, 1]
        ca[4, 0] = -(2 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2
        ca[4, 1] = ca[3, 0]
        ca[4, 2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * r1
        ca[4, 3] = ca[1, 0]; ca[4, 4] = ca[0, 0]
        ca[2, 0] = tt * ca[4, 2]; ca[2, 1] = tt * ca[3, 2]; ca[2, 2] = a0 + 2 * (cpcq - ca[0, 0])
        ca[2, 3] = tt * ca[1, 2]; ca[2, 4] = tt * ca[0, 2]
        ee = e @ ca
        e = ee / np.abs(ee).max()
    return e


def secular_ld(d, a, b, rho, period, c):
    return compound_ld(d, a, b, rho, period, c)[0]




def zero_crossing(f, lo, hi, iters=70):
    """The sign change of f inside [lo, hi] by bisection (f(lo) and f(hi) differ in sign)."""
    flo = f(lo) > 0
    assert (f(hi) > 0) != flo
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        if (f(mid) > 0) == flo:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)
