"""sdr_amd: MI355X (gfx950) engine for batched SDR set algebra.

The product is ``libsdr_amd.so`` (HIP kernels behind the C ABI of ``include/sdr_amd.h``);
this package is the Python host side mirroring the reference's operation names.
"""
from ._lib import (DATA_ARITH, DATA_EMPTY, DATA_UNIT, FLAG_FORCE_BITSLICE, FLAG_FORCE_MERGE, FLAG_FORCE_MFMA, FLAG_FORCE_POPCOUNT, FLAG_SWAP,
                   OP_AND, OP_OR, OP_RM, OP_XOR, SdrbError, load)
from . import csrfile
from .batch import POS_NONE, ChunkedElemResult, ElemResult, SDRBatch, VisitResult, rows_to_csr, synth_csr

__all__ = [
    "SDRBatch", "ElemResult", "ChunkedElemResult", "VisitResult", "POS_NONE", "csrfile", "SdrbError", "rows_to_csr", "synth_csr", "load",
    "OP_AND", "OP_OR", "OP_XOR", "OP_RM", "DATA_EMPTY", "DATA_UNIT", "DATA_ARITH",
    "FLAG_SWAP", "FLAG_FORCE_MERGE", "FLAG_FORCE_BITSLICE", "FLAG_FORCE_POPCOUNT", "FLAG_FORCE_MFMA",
]
