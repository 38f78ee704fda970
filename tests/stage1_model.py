"""NumPy restatement of the stage-1 filter of k_select (csrc/hem.hip: is_regular, spd_det64, make_filter, white_smd) --
test infrastructure: lets the CPU suite attack the filter's bound against the reference arithmetic (the oracle's KLD)
without a GPU, and the GPU suite compare the device's records with it (tests/test_hem_gpu.py)."""
import numpy as np

DET_TOL = 0.04
U = 2.0 ** -24


def det6_f32(c):
    """vec.hpp:863-866 in float32, operand order of gsr_math.h::det6."""
    c = c.astype(np.float32)
    e00, e01, e02, e11, e12, e22 = (c[:, k] for k in range(6))
    f = np.float32
    return (-e02 * e02 * e11 + f(2.0) * e01 * e02 * e12 - e00 * e12 * e12 - e01 * e01 * e22 + e00 * e11 * e22).astype(np.float32)


def inverse6_f32(c, det):
    c = c.astype(np.float32)
    e00, e01, e02, e11, e12, e22 = (c[:, k] for k in range(6))
    r = np.stack([e11 * e22 - e12 * e12, e02 * e12 - e01 * e22, e01 * e12 - e02 * e11,
                  e00 * e22 - e02 * e02, e02 * e01 - e00 * e12, e00 * e11 - e01 * e01], 1).astype(np.float32)
    inv = (np.float32(1.0) / det.astype(np.float32)).astype(np.float32)
    return (r * inv[:, None]).astype(np.float32)


def spd_det64(m):
    a00, a01, a02, a11, a12, a22 = (m[:, k].astype(np.float64) for k in range(6))
    m2 = a00 * a11 - a01 * a01
    det = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02)
    mag = (np.abs(a00) * (np.abs(a11 * a22) + a12 * a12) + np.abs(a01) * (np.abs(a01 * a22) + np.abs(a12 * a02))
           + np.abs(a02) * (np.abs(a01 * a12) + np.abs(a11 * a02)))
    ok = (a00 > 0) & (a11 > 0) & (a22 > 0) & (m2 > 1e-12 * a00 * a11) & (det > 1e-9 * mag) & (mag < 1e300)
    return ok, det


def is_regular(cov6, xyz):
    with np.errstate(all="ignore"):
        c = cov6.astype(np.float32)
        det = det6_f32(c)
        big = 1e12
        ok = (np.abs(xyz.astype(np.float64)) < big).all(1) & (np.abs(c.astype(np.float64)) < big).all(1)
        ok &= (det >= np.float32(1e-18)) & (det <= np.float32(1e18))
        spd, d64 = spd_det64(c)
        ok &= spd & (np.abs(det.astype(np.float64) - d64) <= DET_TOL * d64)
    return ok, det


def make_filter(pcov6, pxyz, thr):
    """-> dict(white, T1, U (n,6 float32: u00 u01 u02 u11 u12 u22), pinv (float32), det_p, regular)"""
    with np.errstate(all="ignore"):
        reg, det_p = is_regular(pcov6, pxyz)
        M = inverse6_f32(pcov6, det_p)
        m = M.astype(np.float64)
        m00, m01, m02, m11, m12, m22 = (m[:, k] for k in range(6))
        finite = (np.abs(m) < 1e30).all(1)
        spd, detM = spd_det64(M)
        trM = m00 + m11 + m22
        e2 = (m00 * m11 - m01 * m01) + (m00 * m22 - m02 * m02) + (m11 * m22 - m12 * m12)
        K = trM * e2 / detM * 1.000001
        theta = 6.1 * U * K + 2.0 * U
        G = -(np.log(detM) + np.log(det_p.astype(np.float64))) + 2e-6
        thr2 = 2.0 * float(np.float32(thr))
        smin = (thr2 * (1.0 + 2.0 * U) + G + DET_TOL + 3.2 * theta / (1.0 - theta) + 4e-5) / (1.0 - theta)
        T1 = smin * (1.0 + theta) ** 2 * (1.0 + 1e-5)
        u00 = np.sqrt(m00); u01 = m01 / u00; u02 = m02 / u00
        t11 = m11 - u01 * u01; u11 = np.sqrt(t11); u12 = (m12 - u01 * u02) / u11
        t22 = m22 - u02 * u02 - u12 * u12; u22 = np.sqrt(t22)
        white = reg & (thr2 >= 0) & (thr2 < 1e30) & finite & spd & (theta <= 0.25) & (np.abs(G) <= 1.0) & (T1 > 0) & (T1 < 1e30) & (t11 > 0) & (t22 > 0)
        Uf = np.stack([u00, u01, u02, u11, u12, u22], 1).astype(np.float32)
        T1f = (T1 * (1.0 + 2.0 * U)).astype(np.float32)
    return {"white": white, "T1": np.where(white, T1f, np.float32(np.inf)), "U": Uf, "pinv": M, "det_p": det_p, "regular": reg, "theta": theta, "G": G}


def white_smd(Uf, pm, cm):
    """|U d|^2 in float32 (no fused multiply-add on the host: the bound covers either rounding)."""
    f = np.float32
    d = (cm.astype(f) - pm.astype(f)).astype(f)
    y2 = Uf[:, 5] * d[:, 2]
    y1 = Uf[:, 3] * d[:, 1] + Uf[:, 4] * d[:, 2]
    y0 = Uf[:, 0] * d[:, 0] + (Uf[:, 1] * d[:, 1] + Uf[:, 2] * d[:, 2])
    return (y0 * y0 + (y1 * y1 + y2 * y2)).astype(f)
