"""CSR database files: a flat, mmap-able container for a packed batch (``sdrb_file_*`` in
``include/sdr_amd.h``), with a streaming writer.  Host side only, except
:func:`upload_rows`, which sends a row range of a file straight into a device batch.

The reference has no serialisation (``SDR.hpp:1670-1680`` prints text only); the container
is this engine's own and holds exactly what a batch holds.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from ._lib import DATA_ARITH, DATA_EMPTY, DATA_UNIT, check, load


def _path(p) -> bytes:
    return os.fsencode(p)


class Writer:
    """Appends rows in chunks of any size, then finishes the file::

        with Writer("db.sdrb", width=2048) as w:
            for off, ids, data in chunks:          # offsets relative to the chunk, first 0
                w.append(off, ids, data)
    """

    def __init__(self, path, *, width: int = 0, payloads: bool = False, kind: Optional[str] = None):
        """``kind``: "empty", "unit" (payloads in [0,1]) or "arith" (any finite float); by default
        "unit" when ``payloads`` is set, else "empty"."""
        self._lib = load()
        self._h = C.c_void_p()
        code = {None: DATA_UNIT if payloads else DATA_EMPTY, "empty": DATA_EMPTY, "unit": DATA_UNIT,
                "arith": DATA_ARITH}[kind]
        self.payloads = code != DATA_EMPTY
        check(self._lib.sdrb_file_writer_open(_path(path), width, code, C.byref(self._h)))

    def append(self, offsets, ids, data=None) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if off.ndim != 1 or off.size < 1:
            raise ValueError("offsets must be a 1-D array of n_rows+1 entries")
        if self.payloads and data is None:
            raise ValueError("this file holds payloads: data is required")
        d = None if data is None or not self.payloads else np.ascontiguousarray(data, dtype=np.float32)
        if int(off[-1]) != ids.size or (d is not None and d.size != ids.size):
            raise ValueError("offsets[-1] must equal len(ids) (and len(data))")
        check(self._lib.sdrb_file_writer_append(self._h, off.size - 1, off.ctypes.data, ids.ctypes.data,
                                                None if d is None else d.ctypes.data))

    def close(self) -> None:
        if self._h:
            h, self._h = self._h, C.c_void_p()
            check(self._lib.sdrb_file_writer_close(h))

    def abort(self) -> None:
        if self._h:
            h, self._h = self._h, C.c_void_p()
            self._lib.sdrb_file_writer_abort(h)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:
            self.abort()
        return False


def write(path, offsets, ids, data=None, *, width: int = 0, kind: Optional[str] = None) -> None:
    """Writes a whole host CSR as one file."""
    with Writer(path, width=width, payloads=data is not None, kind=kind) as w:
        w.append(offsets, ids, data)


def info(path) -> dict:
    lib = load()
    n_rows, nnz, width, kind = C.c_uint64(), C.c_uint64(), C.c_int32(), C.c_int()
    check(lib.sdrb_file_info(_path(path), C.byref(n_rows), C.byref(nnz), C.byref(width), C.byref(kind)))
    return dict(n_rows=int(n_rows.value), nnz=int(nnz.value), width=int(width.value),
                payloads=kind.value != DATA_EMPTY,
                kind={DATA_EMPTY: "empty", DATA_UNIT: "unit", DATA_ARITH: "arith"}[kind.value])


def read_rows(path, row_lo: int = 0, row_hi: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """Rows ``[row_lo, row_hi)`` as a host CSR with offsets rebased to zero."""
    lib = load()
    meta = info(path)
    if row_hi is None:
        row_hi = meta["n_rows"]
    span = C.c_uint64()
    check(lib.sdrb_file_row_span(_path(path), row_lo, row_hi, C.byref(span)))
    off = np.empty(row_hi - row_lo + 1, dtype=np.uint64)
    ids = np.empty(int(span.value), dtype=np.int32)
    data = np.empty(int(span.value), dtype=np.float32) if meta["payloads"] else None
    check(lib.sdrb_file_read_rows(_path(path), row_lo, row_hi, off.ctypes.data, ids.ctypes.data,
                                  None if data is None else data.ctypes.data))
    return off, ids, data


def upload_rows(path, row_lo: int = 0, row_hi: Optional[int] = None, *, device: int = 0, payloads: Optional[bool] = None,
                stream=None):
    """Packs rows ``[row_lo, row_hi)`` of a file into a new device batch (e.g. one rank's shard of a
    database file).  ``payloads=False`` drops the payloads of the file."""
    from .batch import SDRBatch, _stream_handle
    meta = info(path)
    if row_hi is None:
        row_hi = meta["n_rows"]
    use_payloads = meta["payloads"] if payloads is None else payloads
    code = DATA_EMPTY if not use_payloads else DATA_ARITH if meta["kind"] == "arith" else DATA_UNIT
    b = SDRBatch(device, meta["width"], code)
    check(b._lib.sdrb_batch_upload_file(b._h, _path(path), row_lo, row_hi, _stream_handle(device, stream)))
    b._refresh_info()
    return b
