"""
Shape conversions between reduced shear g, distortion e and eta
(reference: ngmix/shape.py).  Host scalar helpers used by the API shell.
"""
import numpy as np

from .gexceptions import GMixRangeError

ONE_MINUS_EPS = 0.9999999999999999


def shear_reduced(g1, g2, s1, s2):
    """apply shear (s1,s2) to the reduced-shear shape (g1,g2)"""
    A = 1 + g1 * s1 + g2 * s2
    B = g2 * s1 - g1 * s2
    denom_inv = 1.0 / (A * A + B * B)
    g1o = (A * (g1 + s1) + B * (g2 + s2)) * denom_inv
    g2o = (A * (g2 + s2) - B * (g1 + s1)) * denom_inv
    return g1o, g2o


def _convert(a1, a2, eta_factor, what):
    """shared body of g<->e: |out| = tanh(eta_factor * atanh(|in|))"""
    mag = np.sqrt(a1 * a1 + a2 * a2)
    if isinstance(a1, np.ndarray):
        if np.any(mag >= 1.0):
            raise GMixRangeError("some %s were out of bounds" % what)
        out = np.tanh(eta_factor * np.arctanh(mag))
        np.clip(out, 0.0, ONE_MINUS_EPS, out)
        o1 = np.zeros(mag.size)
        o2 = np.zeros(mag.size)
        w, = np.where(mag != 0.0)
        if w.size > 0:
            fac = out[w] / mag[w]
            o1[w] = fac * a1[w]
            o2[w] = fac * a2[w]
        return o1, o2
    if mag >= 1.0:
        raise GMixRangeError("%s out of bounds: %s" % (what, mag))
    if mag == 0.0:
        return 0.0, 0.0
    out = np.tanh(eta_factor * np.arctanh(mag))
    if out >= 1.0:
        out = ONE_MINUS_EPS
    fac = out / mag
    return fac * a1, fac * a2


def g1g2_to_e1e2(g1, g2):
    """reduced shear -> distortion (ixx-iyy)/(ixx+iyy)"""
    return _convert(g1, g2, 2.0, "g")


def e1e2_to_g1g2(e1, e2):
    """distortion -> reduced shear"""
    return _convert(e1, e2, 0.5, "e")


def g1g2_to_eta1eta2(g1, g2):
    g = np.sqrt(g1 * g1 + g2 * g2)
    if isinstance(g1, np.ndarray):
        if np.any(g >= 1.0):
            raise GMixRangeError("some g were out of bounds")
        eta1 = np.zeros(g.size)
        eta2 = np.zeros(g.size)
        w, = np.where(g != 0.0)
        if w.size > 0:
            fac = 2 * np.arctanh(g[w]) / g[w]
            eta1[w] = fac * g1[w]
            eta2[w] = fac * g2[w]
        return eta1, eta2
    if g >= 1.0:
        raise GMixRangeError("g out of bounds: %s converting to eta" % g)
    if g == 0.0:
        return 0.0, 0.0
    fac = 2 * np.arctanh(g) / g
    return fac * g1, fac * g2


def eta1eta2_to_g1g2(eta1, eta2):
    eta = np.sqrt(eta1 * eta1 + eta2 * eta2)
    if isinstance(eta1, np.ndarray):
        g = np.tanh(0.5 * eta)
        if np.any(g >= 1.0):
            raise GMixRangeError("some g were out of bounds")
        g1 = np.zeros(g.size)
        g2 = np.zeros(g.size)
        w, = np.where(eta != 0.0)
        if w.size > 0:
            fac = g[w] / eta[w]
            g1[w] = fac * eta1[w]
            g2[w] = fac * eta2[w]
        return g1, g2
    g = np.tanh(0.5 * eta)
    if g >= 1.0:
        raise GMixRangeError("g out of bounds: %s converting from eta" % g)
    if g == 0.0:
        return 0.0, 0.0
    fac = g / eta
    return fac * eta1, fac * eta2


def e1e2_to_eta1eta2(e1, e2):
    """distortion (e1, e2) -> eta-space shape: |eta| = atanh(|e|), the same
    position angle (reference: ngmix/shape.py:350-393); |e| >= 1 raises
    GMixRangeError; scalars in, scalars out"""
    scalar = not isinstance(e1, np.ndarray)
    a1 = np.atleast_1d(np.asarray(e1, dtype="f8"))
    a2 = np.atleast_1d(np.asarray(e2, dtype="f8"))
    mag = np.sqrt(a1 * a1 + a2 * a2)
    if np.any(mag >= 1.0):
        raise GMixRangeError("some e were out of bounds")
    eta1 = np.zeros(mag.size)
    eta2 = np.zeros(mag.size)
    w, = np.where(mag > 0.0)
    if w.size > 0:
        fac = np.arctanh(mag[w]) / mag[w]
        eta1[w] = fac * a1[w]
        eta2[w] = fac * a2[w]
    if scalar:
        return eta1[0], eta2[0]
    return eta1, eta2


def dgs_by_dgo_jacob(g1, g2, s1, s2):
    """|d g_sheared / d g_observed| at fixed shear (s1, s2) for reduced-shear
    shapes: (1 - |s|^2)^2 / (1 + 2 g.s + |g|^2 |s|^2)^2 (reference:
    ngmix/shape.py:443-468)"""
    ssq = s1 * s1 + s2 * s2
    # (the reference's association: 1 + 2 g1 s1 + 2 g2 s2 + g1^2 ssq + g2^2 ssq)
    root = 1 + 2 * g1 * s1 + 2 * g2 * s2 + g1 ** 2 * ssq + g2 ** 2 * ssq
    return (ssq - 1) ** 2 / root ** 2


def get_round_factor(g1, g2):
    """T_round = T * factor under the shear that rounds the shape"""
    gsq = g1 ** 2 + g2 ** 2
    return (1 - gsq) / (1 + gsq)


def rotate_shape(g1, g2, theta):
    """rotate the shape by theta radians"""
    twotheta = 2.0 * theta
    cos2 = np.cos(twotheta)
    sin2 = np.sin(twotheta)
    return g1 * cos2 + g2 * sin2, -g1 * sin2 + g2 * cos2


class Shape(object):
    """a reduced-shear shape (g1, g2)"""

    def __init__(self, g1, g2):
        self.set_g1g2(g1, g2)

    def set_g1g2(self, g1, g2):
        """(g1, g2) and their magnitude .g; |g| >= 1 is a GMixRangeError (the
        components are stored first, as in the reference)"""
        self.g1 = g1
        self.g2 = g2
        g = np.sqrt(g1 * g1 + g2 * g2)
        if g >= 1.0:
            raise GMixRangeError("g out of range: %.16g" % g)
        self.g = g

    def get_sheared(self, s1, s2=None):
        if isinstance(s1, Shape):
            s1, s2 = s1.g1, s1.g2
        elif s2 is None:
            raise ValueError("send s1,s2 or a Shape")
        g1, g2 = shear_reduced(self.g1, self.g2, s1, s2)
        return Shape(g1, g2)

    def __neg__(self):
        return Shape(-self.g1, -self.g2)

    def get_rotated(self, theta_radians):
        g1, g2 = rotate_shape(self.g1, self.g2, theta_radians)
        return Shape(g1, g2)

    def rotate(self, theta_radians):
        g1, g2 = rotate_shape(self.g1, self.g2, theta_radians)
        self.set_g1g2(g1, g2)

    def copy(self):
        return Shape(self.g1, self.g2)

    def __repr__(self):
        return "(%.16g, %.16g)" % (self.g1, self.g2)
