"""Kept for older imports: the reference's op sequence now lives in ``oracle/reference_ops.py`` (test infrastructure)."""
from oracle.reference_ops import eager_forward, regions_token_major  # noqa: F401
