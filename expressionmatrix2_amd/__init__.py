"""MI355X-native implementation of ExpressionMatrix2's LSH similar-pairs path.

Only the hot path named in BASELINE.json lives here: signature projection, the findSimilarPairs4 /
findSimilarPairs5 Hamming scans and the SimilarPairs / Lsh file formats around them, behind the reference's
ExpressionMatrix method names.  The compute is in hand-written HIP (csrc/), reached through the C ABI of
include/em2_lsh.h.
"""
from . import capi, files  # noqa: F401
from .expression_matrix import ExpressionMatrix  # noqa: F401

__all__ = ["capi", "files", "ExpressionMatrix"]
