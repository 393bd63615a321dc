"""
Exception types of the ngmix API (names and str() behaviour as in the
reference, ngmix/gexceptions.py:1-67): device status codes are mapped onto
these by ngmix_amd._lib.check.
"""


class NGmixBaseException(Exception):
    """root of the ngmix exception tree; str() is repr(value)"""

    def __init__(self, value):
        super().__init__(value)
        self.value = value

    def __str__(self):
        return repr(self.value)


class GMixRangeError(NGmixBaseException):
    """a number was out of range (det/T too low, g >= 1, ...)"""


class GMixFatalError(NGmixBaseException):
    """unrecoverable problem with the inputs (e.g. no positive weights)"""


class GMixMaxIterEM(NGmixBaseException):
    """EM reached its iteration limit"""


class PSFFluxFailure(NGmixBaseException):
    """psf flux fit failed"""


class BootPSFFailure(NGmixBaseException):
    """psf bootstrap failed"""


class BootGalFailure(NGmixBaseException):
    """galaxy bootstrap failed"""


class FFTRangeError(NGmixBaseException):
    """inconsistent FFT size"""
