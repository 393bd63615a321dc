"""
Result flag bits and their descriptions.  The numeric values are part of the
result parity contract (reference: ngmix/flags.py:3-57); kernels write the
admom/em bits directly into result records (include/ngmix_hip.h NGMIX_FLAG_*).
"""
import numpy as np

NO_ATTEMPT = 1 << 0
CEN_SHIFT = 1 << 1
NONPOS_FLUX = 1 << 2
NONPOS_SIZE = 1 << 3
LOW_DET = 1 << 4
MAXITER = 1 << 5
NONPOS_VAR = 1 << 6
GMIX_RANGE_ERROR = 1 << 7
NONPOS_SHAPE_VAR = 1 << 8
LM_SINGULAR_MATRIX = 1 << 9
LM_NEG_COV_EIG = 1 << 10
LM_NEG_COV_DIAG = 1 << 11
LM_FUNC_NOTFINITE = 1 << 12
EIG_NOTFINITE = 1 << 13
DIV_ZERO = 1 << 14
ZERO_DOF = 1 << 15

EM_RANGE_ERROR = GMIX_RANGE_ERROR
EM_MAXITER = MAXITER
BAD_VAR = NONPOS_VAR

NAME_MAP = {
    NO_ATTEMPT: 'no attempt',
    CEN_SHIFT: 'center shifted too far',
    NONPOS_FLUX: 'flux <= 0',
    NONPOS_SIZE: 'T <= 0',
    LOW_DET: 'determinant near zero',
    MAXITER: 'max iterations reached',
    NONPOS_VAR: 'non-positive (definite) variance',
    NONPOS_SHAPE_VAR: 'non-positive shape variance',
    GMIX_RANGE_ERROR: 'GMixRangeError raised',
    LM_SINGULAR_MATRIX: 'singular matrix in LM',
    LM_NEG_COV_EIG: 'negative covariance eigenvalue in LM',
    LM_NEG_COV_DIAG: 'negative covariance diagional value in LM',
    LM_FUNC_NOTFINITE: 'function not finite in LM',
    EIG_NOTFINITE: 'eigenvalues of covariance cannot be found in LM',
    DIV_ZERO: 'divide by zero',
    ZERO_DOF: 'degrees of freedom for it is zero (no chi^2/dof possible)',
}


def get_flags_str(val, name_map=None):
    """'|'-joined descriptions of the bits set in val (bits 0..30)"""
    if name_map is None:
        name_map = NAME_MAP
    if val < 0:
        raise ValueError(f"Flag value {val} must be non-negative.")
    val = int(val) & 0xFFFFFFFF
    val = int(np.uint32(val))
    names = []
    for bit in range(31):
        fval = 1 << bit
        if val & fval:
            names.append(name_map.get(fval, "bit 2**%d" % bit))
    return "|".join(names)
