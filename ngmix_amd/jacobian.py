"""
Jacobian of the (row, col) -> (v, u) transformation about a centre
(reference: ngmix/jacobian/jacobian.py).  The 64-byte record these classes own
is passed by value to every kernel (include/ngmix_hip.h ngmix_jacobian); the
scalar transforms go through the same C ABI as the reference's njit helpers.
"""
import ctypes

import numpy as np

from . import _lib

__all__ = ["Jacobian", "DiagonalJacobian", "UnitJacobian"]

_jacobian_dtype = _lib.JACOBIAN_DTYPE
_ROWCOL_REQ = ("row", "col", "dvdrow", "dvdcol", "dudrow", "dudcol")
_XY_REQ = ("x", "y", "dudx", "dudy", "dvdx", "dvdy")


class Jacobian(object):
    """
    Send either
        row=, col=, dvdrow=, dvdcol=, dudrow=, dudcol=     or
        x=, y=, dudx=, dudy=, dvdx=, dvdy=
    or replace the derivatives with wcs= (an object with .dudx .dudy .dvdx
    .dvdy).  x is the column, y the row.
    """

    def __init__(self, **kw):
        self._data = np.zeros(1, dtype=_jacobian_dtype)
        if "x" in kw:
            cen, req = ("y", "x"), _XY_REQ
            names = ("dvdy", "dvdx", "dudy", "dudx")
        elif "row" in kw:
            cen, req = ("row", "col"), _ROWCOL_REQ
            names = ("dvdrow", "dvdcol", "dudrow", "dudcol")
        else:
            raise ValueError("send by row,col or x,y")
        if "wcs" in kw:
            wcs = kw["wcs"]
            derivs = (wcs.dvdy, wcs.dvdx, wcs.dudy, wcs.dudx)
        else:
            # a missing keyword -- the other half of the centre included -- is
            # a ValueError naming it (jacobian.py:262-292)
            for k in req:
                if k not in kw:
                    raise ValueError("missing keyword: '%s'" % k)
            derivs = tuple(kw[n] for n in names)
        self._fill(kw[cen[0]], kw[cen[1]], *derivs)

    def _fill(self, row0, col0, dvdrow, dvdcol, dudrow, dudcol):
        d = self._data
        d["row0"] = row0
        d["col0"] = col0
        d["dvdrow"] = dvdrow
        d["dvdcol"] = dvdcol
        d["dudrow"] = dudrow
        d["dudcol"] = dudcol
        d["det"] = dvdrow * dudcol - dvdcol * dudrow
        d["scale"] = np.sqrt(np.abs(d["det"]))

    def get_data(self):
        """the underlying 1-element record array (a reference)"""
        return self._data

    def get_cen(self):
        return self._data["row0"][0], self._data["col0"][0]

    def get_row0(self):
        return self._data["row0"][0]

    def get_col0(self):
        return self._data["col0"][0]

    def get_dvdrow(self):
        return self._data["dvdrow"][0]

    def get_dvdcol(self):
        return self._data["dvdcol"][0]

    def get_dudrow(self):
        return self._data["dudrow"][0]

    def get_dudcol(self):
        return self._data["dudcol"][0]

    def get_det(self):
        return self._data["det"][0]

    def get_scale(self):
        return self._data["scale"][0]

    def get_area(self):
        return self.scale ** 2

    cen = property(fget=get_cen)
    row0 = property(fget=get_row0)
    col0 = property(fget=get_col0)
    dvdrow = property(fget=get_dvdrow)
    dvdcol = property(fget=get_dvdcol)
    dudrow = property(fget=get_dudrow)
    dudcol = property(fget=get_dudcol)
    det = property(fget=get_det)
    scale = property(fget=get_scale)
    area = property(fget=get_area)

    def _record(self):
        return np.ascontiguousarray(self._data)

    def get_vu(self, row, col):
        """(v, u) of image position(s) (row, col): jacobian_nb.py:4-16"""
        if np.ndim(row) > 0 or np.ndim(col) > 0:
            d = self._data
            rowdiff = np.asarray(row, dtype="f8") - d["row0"][0]
            coldiff = np.asarray(col, dtype="f8") - d["col0"][0]
            v = d["dvdrow"][0] * rowdiff + d["dvdcol"][0] * coldiff
            u = d["dudrow"][0] * rowdiff + d["dudcol"][0] * coldiff
            return v, u
        v, u = ctypes.c_double(), ctypes.c_double()
        rec = self._record()
        _lib.lib().ngmix_jacobian_get_vu(_lib.ptr(rec), float(row), float(col),
                                         ctypes.byref(v), ctypes.byref(u))
        return v.value, u.value

    def get_rowcol(self, v, u):
        """(row, col) of tangent-plane position(s) (v, u): jacobian_nb.py:19-30"""
        if np.ndim(v) > 0 or np.ndim(u) > 0:
            d = self._data
            v = np.asarray(v, dtype="f8")
            u = np.asarray(u, dtype="f8")
            rowdiff = d["dudcol"][0] * v - d["dvdcol"][0] * u
            coldiff = -d["dudrow"][0] * v + d["dvdrow"][0] * u
            return (d["row0"][0] + rowdiff / d["det"][0],
                    d["col0"][0] + coldiff / d["det"][0])
        row, col = ctypes.c_double(), ctypes.c_double()
        rec = self._record()
        st = _lib.lib().ngmix_jacobian_get_rowcol(
            _lib.ptr(rec), float(v), float(u), ctypes.byref(row), ctypes.byref(col))
        _lib.check(st, "ngmix_jacobian_get_rowcol")
        return row.value, col.value

    def __call__(self, row, col):
        return self.get_vu(row, col)

    def set_cen(self, **kw):
        """reset the centre: row=,col= or x=,y="""
        if "row" in kw:
            self._data["row0"] = kw["row"]
            self._data["col0"] = kw["col"]
        elif "x" in kw:
            self._data["row0"] = kw["y"]
            self._data["col0"] = kw["x"]
        else:
            raise ValueError("expected row=,col= or x=,y=")

    def copy(self):
        return Jacobian(row=self.row0, col=self.col0, dudrow=self.dudrow,
                        dudcol=self.dudcol, dvdrow=self.dvdrow,
                        dvdcol=self.dvdcol)

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        result = self.copy()
        memo[id(self)] = result
        return result

    def __eq__(self, jacobian):
        if not isinstance(jacobian, Jacobian):
            raise ValueError(f"expected Jacobian, got {type(jacobian)}")
        return np.all(self.get_data() == jacobian.get_data())

    def __repr__(self):
        return ("ngmix.Jacobian(row=%r, col=%r, dvdrow=%r, dvdcol=%r, "
                "dudrow=%r, dudcol=%r)" % (self.row0, self.col0, self.dvdrow,
                                           self.dvdcol, self.dudrow, self.dudcol))


class DiagonalJacobian(Jacobian):
    """u varies with column only and v with row only, both by `scale`"""

    def __init__(self, scale=1.0, **kw):
        if "x" in kw:
            assert "y" in kw, "send both x= and y="
            super().__init__(x=kw["x"], y=kw["y"], dudx=scale, dudy=0.0,
                             dvdx=0.0, dvdy=scale)
        elif "row" in kw:
            assert "col" in kw, "send both row= and col="
            super().__init__(row=kw["row"], col=kw["col"], dvdrow=scale,
                             dvdcol=0.0, dudrow=0.0, dudcol=scale)
        else:
            raise ValueError("expected row=,col= or x=,y=")


class UnitJacobian(DiagonalJacobian):
    """DiagonalJacobian with scale 1"""

    def __init__(self, **kw):
        super().__init__(scale=1.0, **kw)
