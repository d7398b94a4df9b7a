"""bonsai_amd -- MI355X-native classify hot path of dnbaker/bonsai behind a C ABI.

Python here is a thin host mirror over include/bonsai_amd.h (ctypes); all compute runs in the
hand-written HIP kernels of bonsai_amd/csrc.  There is no CPU fallback.
"""
from ._lib import (BonsaiAmdError, LAYOUT_BUCKET, LAYOUT_KHASH, LAYOUT_MINBUCKET, SCORE_ENTROPY_PATH,  # noqa: F401
                   SCORE_ENTROPY_STRING, SCORE_LEX, TAX_ABSENT, load)
from .context import Context, concat_reads, pack_reads  # noqa: F401

__all__ = ["Context", "concat_reads", "pack_reads", "BonsaiAmdError", "LAYOUT_BUCKET", "LAYOUT_KHASH", "LAYOUT_MINBUCKET", "TAX_ABSENT", "load",
           "SCORE_LEX", "SCORE_ENTROPY_PATH", "SCORE_ENTROPY_STRING"]
