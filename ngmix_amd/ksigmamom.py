"""kept for the reference's import path (ngmix/ksigmamom.py)"""
from .prepsfmom import KSigmaMom  # noqa: F401
