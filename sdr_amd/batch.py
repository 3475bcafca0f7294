"""Host-side mirror of the reference's set-algebra interface, batched.

The reference evaluates ``a.ands(b)``, ``a.ande(b)`` ... for one pair of SDRs
(``SDR.hpp:151-318``).  :class:`SDRBatch` packs many SDRs of one element type into a
device-resident CSR batch and offers the same operation names over every pair
``(self[i], other[j])``; ``self[i]`` plays ``this`` and ``other[j]`` plays ``arg`` unless
``swap=True``.  All computation happens in ``libsdr_amd.so`` through the C ABI of
``include/sdr_amd.h``; torch is used only to allocate result buffers and to name streams.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (DATA_ARITH, DATA_EMPTY, DATA_UNIT, FLAG_FORCE_BITMAP, FLAG_FORCE_LANEMERGE, FLAG_FORCE_LANERANK, FLAG_FORCE_BITSLICE, FLAG_FORCE_MERGE, FLAG_FORCE_MFMA, FLAG_FORCE_POPCOUNT, FLAG_FORCE_RANK, FLAG_FORCE_ROWTABLE, FLAG_ROWWISE, FLAG_SWAP,
                   OP_AND, OP_OR, OP_RM, OP_XOR, check)

_KINDS = {"empty": DATA_EMPTY, "unit": DATA_UNIT, "arith": DATA_ARITH, DATA_EMPTY: DATA_EMPTY, DATA_UNIT: DATA_UNIT,
          DATA_ARITH: DATA_ARITH}


def _torch():
    import torch  # plumbing only: device buffers and streams
    return torch


def _stream_handle(device: int, stream) -> int:
    if stream is None:
        torch = _torch()
        return int(torch.cuda.current_stream(device).cuda_stream)
    if hasattr(stream, "cuda_stream"):
        return int(stream.cuda_stream)
    return int(stream)


def rows_to_csr(rows: Iterable, with_data: bool):
    """``rows``: iterable of id sequences, or of ``(ids, data)`` pairs when ``with_data``."""
    offsets = [0]
    ids, data = [], []
    for row in rows:
        if with_data:
            r_ids, r_data = row
            if len(r_ids) != len(r_data):
                raise ValueError("ids and data of a row differ in length")
            data.extend(float(x) for x in r_data)
        else:
            r_ids = row
        ids.extend(int(x) for x in r_ids)
        offsets.append(len(ids))
    return (np.asarray(offsets, dtype=np.uint64), np.asarray(ids, dtype=np.int32),
            np.asarray(data, dtype=np.float32) if with_data else None)


class ElemResult:
    """CSR result of an element op over all pairs: pair (i, j) owns
    ``ids[offsets[i*nb+j] : offsets[i*nb+j+1]]`` (and ``data`` likewise, results with payloads only).

    A result made by the single-call path has buffers sized by an upper bound and its element count still on the
    device: ``total``, ``ids`` and ``data`` read that count (one stream synchronisation) the first time they are
    asked for; ``offsets`` and the untrimmed ``ids_buffer`` / ``data_buffer`` never wait.  A result whose buffers
    were sized by an ESTIMATE (ande of a few rows) is computed a second time, with room, if the estimate turns
    out too small when the count is first read: the two batches must still hold the rows they held at the call
    (an upload or in-place op in between raises here instead of yielding the result of other data)."""

    def __init__(self, na, nb, offsets, ids, data, total=None, total_dev=None, refill=None):
        self.na = na
        self.nb = nb
        self.offsets = offsets            # torch.int32 [na*nb + 1] (values are uint32)
        self.ids_buffer = ids             # torch.int32 / int64 [>= total]
        self.data_buffer = data           # torch.float32 [>= total] or None
        self._total = total
        self._total_dev = total_dev       # torch.int64 [1] on the device, or None
        self._refill = refill             # buffers sized by an estimate: called with the exact total if they were too small

    @property
    def total(self) -> int:
        if self._total is None:
            self._total = int(self._total_dev.item())
            if self._refill is not None and self._total > self.ids_buffer.numel():
                self.offsets, self.ids_buffer, self.data_buffer = self._refill(self._total)
            self._refill = None
        return self._total

    @property
    def ids(self):
        return self.ids_buffer[: self.total]

    @property
    def data(self):
        return None if self.data_buffer is None else self.data_buffer[: self.total]

    def ids_as(self, dtype):
        """The ids in another integer type -- the reference's ``ande<SDRElem<long>, ...>`` and friends, which
        retarget the id type of the result (SDR.hpp:151-152; /root/reference/src/test.cpp:204-229)."""
        return self.ids.to(dtype)

    def to_host(self):
        off = self.offsets.cpu().numpy().view(np.uint32).astype(np.uint64)
        ids = self.ids.cpu().numpy()
        data = self.data.cpu().numpy() if self.data is not None else None
        return off, ids, data


class ChunkedElemResult:
    """An element-op result too long for one CSR with 32-bit offsets: rows ``starts[c] .. starts[c+1]`` of A
    are held by ``chunks[c]``, an :class:`ElemResult` over those rows of A and all rows of B."""

    def __init__(self, na, nb, starts, chunks):
        self.na, self.nb, self.starts, self.chunks = na, nb, list(starts), list(chunks)

    @property
    def total(self) -> int:
        return sum(c.total for c in self.chunks)

    def chunk_of(self, i: int):
        """``(chunk, row of A inside it)`` for row i of A."""
        for c in range(len(self.chunks)):
            if self.starts[c] <= i < self.starts[c + 1]:
                return self.chunks[c], i - self.starts[c]
        raise IndexError(i)

    def pair(self, i: int, j: int):
        """Device tensors ``(ids, data)`` of result (i, j)."""
        chunk, r = self.chunk_of(i)
        chunk.total      # a chunk whose buffers were sized by an estimate is completed first
        lo, hi = (int(x) & 0xFFFFFFFF for x in chunk.offsets[r * self.nb + j: r * self.nb + j + 2].tolist())
        return chunk.ids_buffer[lo:hi], None if chunk.data_buffer is None else chunk.data_buffer[lo:hi]

    def to_host(self):
        """One CSR with 64-bit offsets on the host (only sensible for results that fit host memory)."""
        offs, ids, data, base = [np.zeros(1, np.uint64)], [], [], 0
        for c in self.chunks:
            o, i, d = c.to_host()
            offs.append(o[1:] + np.uint64(base))
            ids.append(i)
            data.append(d)
            base += int(o[-1])
        return (np.concatenate(offs), np.concatenate(ids),
                None if data[0] is None else np.concatenate(data))


@dataclass
class VisitResult:
    """What the reference's visitors are called with, for all pairs: pair (i, j) owns records
    ``offsets[i*nb+j] : offsets[i*nb+j+1]``, each a (position in ``this``, position in ``arg``); the absent
    side of an ``orv`` single is ``POS_NONE``."""
    na: int
    nb: int
    offsets: "object"            # torch.int32 [na*nb + 1] (values are uint32)
    this_pos: "object"           # torch.int32 [total] (values are uint32)
    arg_pos: "object"            # torch.int32 [total]
    total: int

    def to_host(self):
        return (self.offsets.cpu().numpy().view(np.uint32).astype(np.uint64),
                self.this_pos.cpu().numpy().view(np.uint32), self.arg_pos.cpu().numpy().view(np.uint32))

    def window(self, this_lo: int = 0, this_hi: int = 0xFFFFFFFF, arg_lo: int = 0, arg_hi: int = 0xFFFFFFFF) -> "VisitResult":
        """The calls of the reference's sub-range ``andv(arg, visitor, this_pos, this_end, arg_pos, arg_end)``
        (SDR.hpp:196-202) when every pair is given the same windows: the records whose position in ``this`` lies in
        ``[this_lo, this_hi)`` and whose position in ``arg`` lies in ``[arg_lo, arg_hi)`` -- two sorted sequences meet
        in a common id inside both windows exactly where the whole rows do.  Positions stay those of the whole rows."""
        import torch
        tp = self.this_pos.to(torch.int64) & 0xFFFFFFFF
        ap = self.arg_pos.to(torch.int64) & 0xFFFFFFFF
        keep = (tp >= this_lo) & (tp < this_hi) & (ap >= arg_lo) & (ap < arg_hi)
        kept_upto = torch.cat([torch.zeros(1, dtype=torch.int64, device=keep.device), torch.cumsum(keep, 0)])
        off = kept_upto[self.offsets.to(torch.int64) & 0xFFFFFFFF]
        off32 = torch.where(off >= 2 ** 31, off - 2 ** 32, off).to(torch.int32)
        return VisitResult(self.na, self.nb, off32, self.this_pos[keep], self.arg_pos[keep], int(kept_upto[-1]))


    def orv_tails(self, this_lens, arg_lens):
        """The reference's five-callback ``orv`` (SDR.hpp:284-285, 1099-1150) for every pair of an ``orv`` result: it
        calls the single-side visitors only until one side is exhausted and then hands the rest of the other side over
        in ONE call -- ``arg_only_position(arg position)`` when ``this`` runs out (``this`` is asked first, so also
        when both end together), else ``this_only_position(this position)``.  ``this_lens`` [na] and ``arg_lens`` [nb]
        are the row lengths.  Returns host arrays over the pairs: ``head_end`` (records ``offsets[p] : head_end[p]`` are
        the visitor calls made before the position call), ``arg_only`` (bool: which position call) and ``position``
        (where the handed-over tail starts in its row; may be the row's end)."""
        off, tp, ap = self.to_host()
        this_len = np.repeat(np.asarray(this_lens, dtype=np.int64), self.nb)
        arg_len = np.tile(np.asarray(arg_lens, dtype=np.int64), self.na)
        n_pairs = self.na * self.nb
        counts = np.diff(off).astype(np.int64)
        pid = np.repeat(np.arange(n_pairs), counts)
        has_t, has_a = tp != POS_NONE, ap != POS_NONE
        cum_t = np.concatenate([[0], np.cumsum(has_t)])
        cum_a = np.concatenate([[0], np.cumsum(has_a)])
        start = off[:-1].astype(np.int64)
        seen_t = cum_t[1:] - cum_t[start][pid]
        seen_a = cum_a[1:] - cum_a[start][pid]
        end_t = has_t & (seen_t == this_len[pid])
        end_a = has_a & (seen_a == arg_len[pid]) & ~end_t
        k = np.arange(tp.size, dtype=np.int64)
        first = np.full(n_pairs, np.iinfo(np.int64).max)
        np.minimum.at(first, pid, np.where(end_t | end_a, k, np.iinfo(np.int64).max))
        head_end = start.copy()
        arg_only = this_len == 0
        position = np.zeros(n_pairs, dtype=np.int64)
        live = (this_len > 0) & (arg_len > 0)
        f = first[live]
        head_end[live] = f + 1
        arg_only[live] = end_t[f]
        position[live] = np.where(end_t[f], seen_a[f], seen_t[f])
        return head_end, arg_only, position


POS_NONE = 0xFFFFFFFF


class SDRBatch:
    """A device-resident CSR pack of SDRs (ids int32, optional UnitData float payloads)."""

    def __init__(self, device: int = 0, width: int = 0, data_kind="empty", id_bits: int = 32):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.device = int(device)
        self.width = int(width)
        self.data_kind = _KINDS[data_kind]
        self.id_bits = int(id_bits)
        if self.id_bits == 64:      # the reference's SDRElem<long, ...>: no declared width, merge kernels
            if self.width:
                raise ValueError("batches with 64-bit ids have no declared width")
            check(self._lib.sdrb_batch_create64(self.device, self.data_kind, C.byref(self._h)))
        elif self.id_bits == 32:
            check(self._lib.sdrb_batch_create(self.device, self.width, self.data_kind, C.byref(self._h)))
        else:
            raise ValueError("id_bits must be 32 or 64")
        self.n_rows = 0
        self.nnz = 0
        self.max_row_len = 0
        self._generation = 0        # counts the uploads: a result computed lazily checks that its inputs are unchanged

    # -- construction -------------------------------------------------------------
    @classmethod
    def from_csr(cls, offsets, ids, data=None, *, width: int = 0, device: int = 0, stream=None, kind=None):
        """``kind``: "empty" / "unit" / "arith"; by default "unit" when payloads are given (UnitData,
        values in [0,1]); "arith" takes any finite float (ArithData<float>, for :meth:`inner`)."""
        id_bits = 64 if np.asarray(ids).dtype == np.int64 else 32     # int64 arrays make a 64-bit batch
        b = cls(device, width, kind if kind is not None else ("unit" if data is not None else "empty"), id_bits)
        b.upload(offsets, ids, data, stream=stream)
        return b

    @classmethod
    def from_rows(cls, rows: Sequence, *, with_data: bool = False, width: int = 0, device: int = 0):
        off, ids, data = rows_to_csr(rows, with_data)
        return cls.from_csr(off, ids, data, width=width, device=device)

    def upload(self, offsets, ids, data=None, *, stream=None) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.int64 if self.id_bits == 64 else np.int32)
        if off.ndim != 1 or off.size < 1:
            raise ValueError("offsets must be a 1-d array with n_rows+1 entries")
        if int(off[-1]) != ids.size:
            raise ValueError("offsets[-1] must equal len(ids)")
        dptr = None
        if self.data_kind != DATA_EMPTY:
            if data is None:
                raise ValueError("a batch with payloads needs them")
            data = np.ascontiguousarray(data, dtype=np.float32)
            if data.size != ids.size:
                raise ValueError("data and ids differ in length")
            dptr = data.ctypes.data_as(C.c_void_p)
        elif data is not None:
            raise ValueError("EmptyData batch takes no payloads")
        n_rows = off.size - 1
        self._generation += 1
        upload = self._lib.sdrb_batch_upload_csr64 if self.id_bits == 64 else self._lib.sdrb_batch_upload_csr
        check(upload(self._h, n_rows, off.ctypes.data_as(C.c_void_p),
                                              ids.ctypes.data_as(C.c_void_p), dptr,
                                              _stream_handle(self.device, stream)))
        self.n_rows = n_rows
        self.nnz = int(ids.size)
        mx = C.c_uint32()
        check(self._lib.sdrb_batch_info(self._h, None, None, None, None, None, C.byref(mx)))
        self.max_row_len = int(mx.value)

    def _refresh_info(self) -> None:
        """Re-reads rows / elements / longest row from the handle (after an upload done in C)."""
        n, nnz, mx = C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(self._lib.sdrb_batch_info(self._h, None, C.byref(n), C.byref(nnz), None, None, C.byref(mx)))
        self.n_rows, self.nnz, self.max_row_len = int(n.value), int(nnz.value), int(mx.value)

    def upload_device(self, offsets, ids, data=None, *, stream=None) -> None:
        """Packs a CSR that already lives on this device: ``offsets`` int32 tensor [n_rows+1]
        (uint32 values), ``ids`` int32 tensor, ``data`` float32 tensor for UnitData batches."""
        torch = _torch()
        if offsets.dtype != torch.int32 or ids.dtype != torch.int32 or offsets.dim() != 1 or offsets.numel() < 1:
            raise ValueError("offsets and ids must be 1-d int32 tensors")
        if not (offsets.is_contiguous() and ids.is_contiguous()):
            raise ValueError("offsets and ids must be contiguous")
        dptr = None
        if self.data_kind != DATA_EMPTY:
            if data is None or data.dtype != torch.float32 or not data.is_contiguous() or data.numel() != ids.numel():
                raise ValueError("UnitData batch needs a contiguous float32 payload tensor as long as ids")
            dptr = data.data_ptr() if data.numel() else None
        elif data is not None:
            raise ValueError("EmptyData batch takes no payloads")
        n_rows = offsets.numel() - 1
        self._generation += 1
        check(self._lib.sdrb_batch_upload_device_csr(self._h, n_rows, offsets.data_ptr(),
                                                     ids.data_ptr() if ids.numel() else None, dptr,
                                                     _stream_handle(self.device, stream)))
        self.n_rows = n_rows
        nnz, mx = C.c_uint64(), C.c_uint32()
        check(self._lib.sdrb_batch_info(self._h, None, None, C.byref(nnz), None, None, C.byref(mx)))
        self.nnz = int(nnz.value)
        self.max_row_len = int(mx.value)

    def adopt_device(self, offsets, ids, data=None, *, max_row_len: int, min_id: int, max_id: int) -> None:
        """Makes this batch a zero-copy VIEW of CSR tensors on its device that are known to hold
        an already validated batch (``max_row_len``/``min_id``/``max_id`` are that batch's
        statistics, see :meth:`stats`).  Nothing is checked, copied or synchronised; the tensors
        are kept referenced here so they outlive the view."""
        torch = _torch()
        if offsets.dtype != torch.int32 or ids.dtype != torch.int32 or not offsets.is_contiguous() \
                or not ids.is_contiguous():
            raise ValueError("offsets and ids must be contiguous int32 tensors")
        if (self.data_kind != DATA_EMPTY) != (data is not None):
            raise ValueError("payload tensor must be given exactly for UnitData batches")
        n_rows, nnz = offsets.numel() - 1, ids.numel()
        self._generation += 1
        check(self._lib.sdrb_batch_adopt_device_csr(
            self._h, n_rows, nnz, offsets.data_ptr(), ids.data_ptr() if nnz else None,
            data.data_ptr() if (data is not None and nnz) else None, max_row_len, min_id, max_id))
        self._views = (offsets, ids, data)
        self.n_rows, self.nnz, self.max_row_len = n_rows, nnz, int(max_row_len)

    def to_wire(self, stream=None):
        """The batch in wire form: one int32 device tensor ``[offsets (n_rows+1) | ids (nnz) |
        payload bits (nnz, UnitData only)]`` (a device-to-device copy), the sending side of
        :meth:`adopt_device`."""
        torch = _torch()
        unit = self.data_kind != DATA_EMPTY
        words = self.n_rows + 1 + self.nnz * (2 if unit else 1)
        wire = torch.empty(words, dtype=torch.int32, device=f"cuda:{self.device}")
        ids_t = wire[self.n_rows + 1: self.n_rows + 1 + self.nnz]
        data_t = wire[self.n_rows + 1 + self.nnz:]
        check(self._lib.sdrb_batch_export_device_csr(
            self._h, wire.data_ptr(), ids_t.data_ptr() if self.nnz else None,
            data_t.data_ptr() if (unit and self.nnz) else None, _stream_handle(self.device, stream)))
        return wire

    def table_width(self) -> int:
        """The width the table kernels index this batch by: the declared one, or largest id + 1 for rows uploaded
        without a declared width and without a negative id; 0 when there is none."""
        w = C.c_int32()
        check(self._lib.sdrb_batch_table_width(self._h, C.byref(w)))
        return int(w.value)

    def stats(self) -> dict:
        """Shape and id statistics of the uploaded batch (what a zero-copy view needs)."""
        lo, hi = C.c_int32(), C.c_int32()
        check(self._lib.sdrb_batch_id_span(self._h, C.byref(lo), C.byref(hi)))
        return dict(n_rows=self.n_rows, nnz=self.nnz, max_row_len=self.max_row_len,
                    min_id=int(lo.value), max_id=int(hi.value), has_data=self.data_kind != DATA_EMPTY)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.sdrb_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return self.n_rows

    # -- size ops (SDR.hpp:175-176, 229-230, 257-258, 317-318) -------------------------
    def _flags(self, swap: bool, algo: Optional[str], rowwise: bool = False) -> int:
        f = (FLAG_SWAP if swap else 0) | (FLAG_ROWWISE if rowwise else 0)
        if algo == "merge":
            f |= FLAG_FORCE_MERGE
        elif algo == "bitslice":
            f |= FLAG_FORCE_BITSLICE
        elif algo == "popcount":
            f |= FLAG_FORCE_POPCOUNT
        elif algo == "rowtable":
            f |= FLAG_FORCE_ROWTABLE
        elif algo == "rank":        # element fill by the wave-per-pair rank-table kernel (counts: default choice)
            f |= FLAG_FORCE_RANK
        elif algo == "bitmap":      # element fill by the dense-bitset kernel
            f |= FLAG_FORCE_BITMAP
        elif algo == "lanemerge":   # element fill by the lane-per-pair merge kernel
            f |= FLAG_FORCE_LANEMERGE
        elif algo == "lanerank":    # element fill by the lane-per-pair rank kernel
            f |= FLAG_FORCE_LANERANK
        elif algo == "lanerows":    # size ops with payloads by the lane-per-row match kernel
            f |= _lib.FLAG_FORCE_SPARSE
        elif algo == "masktable":   # size ops by the 64-row mask-table kernel
            f |= _lib.FLAG_FORCE_MASKTABLE
        elif algo == "bitslice_long":   # id-only size ops by the bit-sliced kernel for long rows / wide tables
            f |= _lib.FLAG_FORCE_BITSLICE_LONG
        elif algo == "search":      # element fill / visitors by the wave-per-pair binary-search kernel (ids as they are)
            f |= _lib.FLAG_FORCE_SEARCH
        elif algo is not None:
            raise ValueError("algo must be None, 'merge', 'bitslice', 'popcount', 'rowtable', 'rank', 'bitmap', "
                             "'lanemerge', 'lanerank', 'lanerows', 'masktable', 'bitslice_long' or 'search'")
        return f

    def _size_op_narrow(self, op: int, other: "SDRBatch", swap: bool, out, stream, dtype):
        """Counts as uint8 / int16 (uint16 values): the bit-sliced kernel only; refused when a count might not fit."""
        torch = _torch()
        nbytes = {torch.uint8: 1, torch.int16: 2}.get(dtype)
        if nbytes is None:
            raise ValueError("dtype must be torch.int32 (default), torch.uint8 or torch.int16 (uint16 values)")
        ld = (other.n_rows + 3) // 4 * 4
        if out is None:
            out = torch.empty((max(self.n_rows, 1), max(ld, 4)), dtype=dtype, device=f"cuda:{other.device}")
        if out.dtype != dtype or out.dim() != 2 or out.shape[0] < self.n_rows or out.stride(1) != 1 or out.stride(0) < other.n_rows:
            raise ValueError("out must be a [>=na, >=nb] tensor of the requested dtype with unit inner stride")
        check(self._lib.sdrb_size_op_narrow(self._h, other._h, op, FLAG_SWAP if swap else 0, out.data_ptr(), out.stride(0),
                                            nbytes, _stream_handle(other.device, stream)))
        return out[: self.n_rows, : other.n_rows]

    def _size_op(self, op: int, other: "SDRBatch", swap: bool, out, stream, algo, dtype=None):
        torch = _torch()
        if dtype is not None and dtype != torch.int32:
            if algo not in (None, "bitslice"):
                raise ValueError("narrow counts come from the bit-sliced kernel only")
            return self._size_op_narrow(op, other, swap, out, stream, dtype)
        if out is None:
            out = torch.empty((self.n_rows, other.n_rows), dtype=torch.int32,
                              device=f"cuda:{other.device}")
        if out.dtype != torch.int32 or out.dim() != 2 or out.shape[0] < self.n_rows \
                or out.shape[1] < other.n_rows or out.stride(1) != 1:
            raise ValueError("out must be an int32 [>=na, >=nb] tensor with unit inner stride")
        check(self._lib.sdrb_size_op(self._h, other._h, op, self._flags(swap, algo),
                                     out.data_ptr(), out.stride(0),
                                     _stream_handle(other.device, stream)))
        return out

    def transpose(self, *, stream=None) -> "SDRBatch":
        """The batch regrouped by id (CSR -> CSC): row c of the result lists the rows of ``self`` that
        contain id c, ascending, with their payloads -- ``SDR::transpose`` (SDR.hpp:1807-1826) for a
        matrix stored as the rows of a batch.  Needs a declared width."""
        t = SDRBatch(self.device, 0, self.data_kind)
        check(self._lib.sdrb_batch_transpose(self._h, t._h, _stream_handle(self.device, stream)))
        t._refresh_info()
        t.width = self.n_rows
        return t

    # -- range and point queries -----------------------------------------------------------------
    def range_size(self, lo, hi, *, stream=None):
        """``out[r, j] = self[j].ands(lo[r], hi[r])``: ids of row j in ``[lo, hi)`` (SDR.hpp:931-956)."""
        torch = _torch()
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        out = torch.empty((max(lo.size, 1), max(self.n_rows, 1)), dtype=torch.int32, device=f"cuda:{self.device}")
        check(self._lib.sdrb_range_size(self._h, lo.size, lo.ctypes.data, hi.ctypes.data, out.data_ptr(),
                                        out.stride(0), _stream_handle(self.device, stream)))
        return out[: lo.size, : self.n_rows]

    def range_elems(self, lo, hi, *, stream=None) -> ElemResult:
        """CSR of ``self[j].ande(lo[r], hi[r])`` over the pairs (r, j) (SDR.hpp:881-929)."""
        torch = _torch()
        dev = f"cuda:{self.device}"
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        s = _stream_handle(self.device, stream)
        pairs = lo.size * self.n_rows
        offsets = torch.empty(pairs + 1, dtype=torch.int32, device=dev)
        total = C.c_uint64()
        check(self._lib.sdrb_range_elems_count(self._h, lo.size, lo.ctypes.data, hi.ctypes.data, offsets.data_ptr(),
                                               C.byref(total), s))
        n = int(total.value)
        ids_buf = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        data_buf = torch.empty(max(n, 1), dtype=torch.float32, device=dev) if self.data_kind != DATA_EMPTY else None
        check(self._lib.sdrb_range_elems_fill(self._h, lo.size, lo.ctypes.data, hi.ctypes.data, offsets.data_ptr(),
                                              ids_buf.data_ptr(),
                                              data_buf.data_ptr() if data_buf is not None else None, s))
        return ElemResult(lo.size, self.n_rows, offsets, ids_buf, data_buf, n)

    def point_query(self, ids, *, stream=None):
        """``out[q, j]`` = position of ``ids[q]`` in row j or -1: the batched ``ande(id)``
        (SDR.hpp:860-879); the payload of a hit is ``data[row_offsets[j] + position]``."""
        torch = _torch()
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = torch.empty((max(ids.size, 1), max(self.n_rows, 1)), dtype=torch.int32, device=f"cuda:{self.device}")
        check(self._lib.sdrb_point_query(self._h, ids.size, ids.ctypes.data, out.data_ptr(), out.stride(0),
                                         _stream_handle(self.device, stream)))
        return out[: ids.size, : self.n_rows]

    def inner(self, other: "SDRBatch", *, out=None, stream=None, algo=None):
        """``out[i, j] = self[i].inner(other[j])`` (float32): the sparse dot product over the id
        matches in ascending id order, product and sum rounded separately, bit-exact against the
        reference (``SDR.hpp:1682-1692``).  ``self`` must be an "arith" batch, ``other`` "arith" or
        "empty".  With ``self`` = the rows of a row-major matrix and ``other`` = the columns of a
        column-major one this is ``diff_mul``; with ``self`` = one vector and ``other`` = matrix
        rows, ``row_major_mul_vec``."""
        torch = _torch()
        if out is None:
            out = torch.empty((self.n_rows, other.n_rows), dtype=torch.float32, device=f"cuda:{other.device}")
        if out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] < self.n_rows \
                or out.shape[1] < other.n_rows or out.stride(1) != 1:
            raise ValueError("out must be a float32 [>=na, >=nb] tensor with unit inner stride")
        flags = {None: 0, "sparse": _lib.FLAG_FORCE_SPARSE, "table": FLAG_FORCE_ROWTABLE, "merge": FLAG_FORCE_MERGE}[algo]
        check(self._lib.sdrb_inner(self._h, other._h, flags, out.data_ptr(), out.stride(0),
                                   _stream_handle(other.device, stream)))
        return out

    def allpairs_topk(self, other: "SDRBatch", k: int, *, exclude_offset: int = -1, stream=None, algo=None):
        """For every row i of ``self``: the ``k`` rows j of ``other`` with the largest id overlap
        (``self[i].ands(other[j])`` on EmptyData views), overlap descending then j ascending.
        ``exclude_offset >= 0`` skips the pair (i, i + exclude_offset).  Returns
        ``(counts, rows)``: int32 ``[len(self), k]`` tensors on the device; unused slots are
        (0, -1).  The full count matrix is never materialised.  ``algo``: None (bit-sliced kernel
        when it applies, else matrix cores), ``"bitslice"`` or ``"mfma"``."""
        flags = {None: 0, "bitslice": FLAG_FORCE_BITSLICE, "mfma": FLAG_FORCE_MFMA}[algo]
        torch = _torch()
        dev = f"cuda:{other.device}"
        counts = torch.empty((max(self.n_rows, 1), k), dtype=torch.int32, device=dev)[: self.n_rows]
        rows = torch.empty((max(self.n_rows, 1), k), dtype=torch.int32, device=dev)[: self.n_rows]
        check(self._lib.sdrb_allpairs_topk(self._h, other._h, k, exclude_offset, flags, counts.data_ptr(),
                                           rows.data_ptr(), _stream_handle(other.device, stream)))
        return counts, rows

    def ands(self, other, *, swap=False, out=None, stream=None, algo=None, dtype=None):
        """``out[i, j] = self[i].ands(other[j])`` (uint32 counts in an int32 tensor).  ``dtype=torch.uint8`` /
        ``torch.int16`` asks for narrow counts (a view of a row-padded tensor): every count is bounded by the
        longest rows, and the call is refused rather than truncating when the bound does not fit."""
        return self._size_op(OP_AND, other, swap, out, stream, algo, dtype)

    def ors(self, other, *, swap=False, out=None, stream=None, algo=None, dtype=None):
        return self._size_op(OP_OR, other, swap, out, stream, algo, dtype)

    def xors(self, other, *, swap=False, out=None, stream=None, algo=None, dtype=None):
        return self._size_op(OP_XOR, other, swap, out, stream, algo, dtype)

    def rms(self, other, *, swap=False, out=None, stream=None, algo=None, dtype=None):
        return self._size_op(OP_RM, other, swap, out, stream, algo, dtype)

    # -- element ops (SDR.hpp:151-152, 209-212, 237-240, 292-293) ---------------------
    SINGLE_CALL_BYTES = 24 << 30    # largest bound-sized result buffer the single-call path allocates

    def _elem_op(self, op: int, other: "SDRBatch", swap: bool, stream, algo, rowwise: bool = False) -> ElemResult:
        torch = _torch()
        dev = f"cuda:{other.device}"
        if rowwise and self.n_rows != other.n_rows:
            raise ValueError("row-wise evaluation needs batches of equal length")
        pairs = other.n_rows if rowwise else self.n_rows * other.n_rows
        s = _stream_handle(other.device, stream)
        offsets = torch.empty(pairs + 1, dtype=torch.int32, device=dev)
        total = C.c_uint64()
        flags = self._flags(swap, algo, rowwise)
        this_kind = other.data_kind if swap else self.data_kind
        if not rowwise and self.id_bits == 32 and other.id_bits == 32 and pairs:
            # more elements than 32-bit offsets address (certainly for unions, possibly for intersections):
            # the rows of A are cut into chunks whose results fit, one CSR each
            chunked = self._elem_op_chunked(op, other, swap, stream, algo, flags)
            if chunked is not None:
                return chunked
        table_width = other.table_width() if (op == OP_AND and not rowwise and algo is None and pairs and self.n_rows <= 32) else 0
        if (op == OP_AND and not rowwise and algo is None and self.id_bits == 32 and other.id_bits == 32 and pairs
                and self.n_rows <= 32 and 0 < table_width <= 4096 and other.max_row_len <= 64
                and self.data_kind != DATA_ARITH and other.data_kind != DATA_ARITH):
            # One query, or a handful, against a database: offsets and elements in one enqueue, the element buffers
            # sized by the expected number of matches (ids spread evenly over the width) with room to spare.
            expected = self.nnz * other.nnz / table_width
            cap = int(min(min(self.nnz * other.n_rows, other.nnz * self.n_rows), expected * 1.3 + 8.0 * expected ** 0.5 + 4096))
            has_data = this_kind != DATA_EMPTY

            def run(capacity):
                off_t = torch.empty(pairs + 1, dtype=torch.int32, device=dev)
                ids_t = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
                data_t = torch.empty(max(capacity, 1), dtype=torch.float32, device=dev) if has_data else None
                tot_t = torch.empty(1, dtype=torch.int64, device=dev)
                check(self._lib.sdrb_ande_few(self._h, other._h, flags & FLAG_SWAP, off_t.data_ptr(), ids_t.data_ptr(),
                                              data_t.data_ptr() if data_t is not None else None, capacity,
                                              tot_t.data_ptr(), s))
                return off_t, ids_t, data_t, tot_t

            made_from = (self._generation, other._generation)

            def refill(total):
                if (self._generation, other._generation) != made_from:
                    raise RuntimeError("this ande result outgrew its estimate-sized buffers and has to be computed again, "
                                       "but one of its batches has been uploaded into since the call: read .total (or "
                                       ".ids) of such a result before changing its inputs")
                return run(total)[:3]

            try:
                off_t, ids_t, data_t, tot_t = run(cap)
                return ElemResult(self.n_rows, other.n_rows, off_t, ids_t, data_t, total_dev=tot_t, refill=refill)
            except _lib.SdrbError as e:
                if e.code != -5:        # not served (tables too large for LDS): the count / fill protocol below
                    raise
        if op != OP_AND and self.id_bits == 32 and other.id_bits == 32 and pairs:
            # Unions and differences: the shapes bound the result tightly (|A_i| + |B_j| per pair), so the buffers
            # are sized by the bound and count, scan and fill are enqueued in one call, with no wait in between.
            bound = C.c_uint64()
            check(self._lib.sdrb_elem_op_bound(self._h, other._h, op, flags, C.byref(bound)))
            cap = int(bound.value)
            if 0 < cap <= 0xFFFFFFFF and cap * (8 if this_kind != DATA_EMPTY else 4) <= self.SINGLE_CALL_BYTES:
                ids_buf = torch.empty(cap, dtype=torch.int32, device=dev)
                data_buf = torch.empty(cap, dtype=torch.float32, device=dev) if this_kind != DATA_EMPTY else None
                total_dev = torch.empty(1, dtype=torch.int64, device=dev)
                check(self._lib.sdrb_elem_op(self._h, other._h, op, flags, offsets.data_ptr(), ids_buf.data_ptr(),
                                             data_buf.data_ptr() if data_buf is not None else None, cap,
                                             total_dev.data_ptr(), s))
                return ElemResult(1 if rowwise else self.n_rows, other.n_rows, offsets, ids_buf, data_buf,
                                  total_dev=total_dev)
        check(self._lib.sdrb_elem_op_count(self._h, other._h, op, flags, offsets.data_ptr(),
                                           C.byref(total), s))
        n = int(total.value)
        # buffers keep at least one element so an empty result still has a valid pointer
        this_bits = other.id_bits if swap else self.id_bits      # a result holds ids of `this`'s width
        ids_buf = torch.empty(max(n, 1), dtype=torch.int64 if this_bits == 64 else torch.int32, device=dev)
        data_buf = torch.empty(max(n, 1), dtype=torch.float32, device=dev) if this_kind != DATA_EMPTY else None
        fill = self._lib.sdrb_elem_op_fill64 if this_bits == 64 else self._lib.sdrb_elem_op_fill
        check(fill(self._h, other._h, op, flags, offsets.data_ptr(),
                                          ids_buf.data_ptr(),
                                          data_buf.data_ptr() if data_buf is not None else None, s))
        return ElemResult(1 if rowwise else self.n_rows, other.n_rows, offsets, ids_buf, data_buf, n)

    def _elem_op_chunked(self, op, other, swap, stream, algo, flags):
        """None when one call can hold the result; else a :class:`ChunkedElemResult`."""
        bound = C.c_uint64()
        check(self._lib.sdrb_elem_op_bound(self._h, other._h, op, flags, C.byref(bound)))
        limit = 0xFFFFFFFF
        if int(bound.value) <= limit:
            return None
        torch = _torch()
        if op == OP_AND:
            # the bound of an intersection is loose: the exact total decides
            sizes = self._size_op(op, other, swap, None, stream, algo if algo in ("merge", "rowtable") else None)
            per_row = sizes.to(torch.int64).sum(dim=1).cpu().numpy()
            del sizes
        else:
            a_off = self.to_wire(stream=stream)[: self.n_rows + 1].to(torch.int64).cpu().numpy() & 0xFFFFFFFF
            a_len = np.diff(a_off)
            if op == OP_RM:
                per_row = a_len * other.n_rows if not swap else np.full(self.n_rows, other.nnz, dtype=np.int64)
            else:
                per_row = a_len * other.n_rows + other.nnz
        if int(per_row.sum()) <= limit:
            return None
        if int(per_row.max(initial=0)) > limit:
            raise _lib.SdrbError(-6, "one row of A against all rows of B already yields more than 2^32-1 elements: "
                                     "split B")
        # greedy chunks of consecutive rows of A
        starts, acc = [0], 0
        for i, n in enumerate(per_row.tolist()):
            if acc + n > limit:
                starts.append(i)
                acc = 0
            acc += n
        starts.append(self.n_rows)
        wire = self.to_wire(stream=stream)
        n, nnz = self.n_rows, self.nnz
        off_t, ids_t = wire[: n + 1], wire[n + 1: n + 1 + nnz]
        data_t = wire[n + 1 + nnz:].view(torch.float32) if self.data_kind != DATA_EMPTY else None
        stats = self.stats()
        off_host = off_t.to(torch.int64).cpu().numpy() & 0xFFFFFFFF
        chunks = []
        for c in range(len(starts) - 1):
            i0, i1 = starts[c], starts[c + 1]
            lo, hi = int(off_host[i0]), int(off_host[i1])
            part = SDRBatch(self.device, self.width, self.data_kind)
            part.adopt_device((off_t[i0: i1 + 1] - off_t[i0]).contiguous(), ids_t[lo:hi],
                              data_t[lo:hi] if data_t is not None else None, max_row_len=stats["max_row_len"],
                              min_id=stats["min_id"], max_id=stats["max_id"])
            chunks.append(part._elem_op(op, other, swap, stream, algo))
        return ChunkedElemResult(self.n_rows, other.n_rows, starts, chunks)

    # -- row-wise evaluation and the in-place ops ------------------------------------------------
    def rowwise_size(self, name: str, other: "SDRBatch", *, swap=False, stream=None):
        """``out[j] = self[j].<name>(other[j])`` for ``name`` in ands/ors/xors/rms and two batches
        of equal length (int32 tensor [n])."""
        torch = _torch()
        op = {"ands": OP_AND, "ors": OP_OR, "xors": OP_XOR, "rms": OP_RM}[name]
        out = torch.empty(max(other.n_rows, 1), dtype=torch.int32, device=f"cuda:{other.device}")
        check(self._lib.sdrb_size_op(self._h, other._h, op, self._flags(swap, None, True), out.data_ptr(),
                                     other.n_rows, _stream_handle(other.device, stream)))
        return out[: other.n_rows]

    def rowwise_elem(self, name: str, other: "SDRBatch", *, swap=False, stream=None) -> ElemResult:
        """CSR of ``self[j].<name>(other[j])`` for ``name`` in ande/ore/xore/rme (one result per row)."""
        op = {"ande": OP_AND, "ore": OP_OR, "xore": OP_XOR, "rme": OP_RM}[name]
        return self._elem_op(op, other, swap, stream, None, rowwise=True)

    def _inplace(self, name: str, other: "SDRBatch", stream) -> "SDRBatch":
        res = self.rowwise_elem(name, other, stream=stream)
        self.upload_device(res.offsets, res.ids, res.data, stream=stream)
        return self

    def shift(self, amount: int, *, stream=None) -> "SDRBatch":
        """Adds ``amount`` to every id of every row, in place (``SDR::shift``, SDR.hpp:1579-1590).  Refused when an
        id would leave 32 bits (the reference asserts the same) or the declared width; 64-bit ids stay within 64 bits."""
        check(self._lib.sdrb_batch_shift(self._h, int(amount), _stream_handle(self.device, stream)))
        self._generation += 1
        return self

    def copy(self, *, stream=None) -> "SDRBatch":
        """A batch of its own with the same rows (device-to-device; batches of 32-bit ids)."""
        if self.id_bits != 32:
            raise ValueError("copy serves batches with 32-bit ids")
        torch = _torch()
        wire = self.to_wire(stream=stream)
        n, nnz = self.n_rows, self.nnz
        b = SDRBatch(self.device, self.width, self.data_kind)
        data = wire[n + 1 + nnz:].view(torch.float32) if self.data_kind != DATA_EMPTY else None
        b.upload_device(wire[: n + 1], wire[n + 1: n + 1 + nnz], data, stream=stream)
        return b

    # the reference's shift operators (SDR.hpp:461-468): << and >> give a shifted copy, <<= and >>= shift in place
    def __lshift__(self, amount: int) -> "SDRBatch":
        return self.copy().shift(amount)

    def __rshift__(self, amount: int) -> "SDRBatch":
        return self.copy().shift(-int(amount))

    def __ilshift__(self, amount: int) -> "SDRBatch":
        return self.shift(amount)

    def __irshift__(self, amount: int) -> "SDRBatch":
        return self.shift(-int(amount))

    def andi(self, other, *, stream=None) -> "SDRBatch":
        """Every row becomes ``row.andi(other[j])`` (SDR.hpp:1027-1081); the results equal ande's."""
        return self._inplace("ande", other, stream)

    def ori(self, other, *, stream=None) -> "SDRBatch":
        """Every row becomes ``row.ori(other[j])`` (SDR.hpp:1239-1292)."""
        return self._inplace("ore", other, stream)

    def xori(self, other, *, stream=None) -> "SDRBatch":
        """Every row becomes ``row.xori(other[j])`` (SDR.hpp:1366-1422)."""
        return self._inplace("xore", other, stream)

    def rmi(self, other, *, combine=None, stream=None) -> "SDRBatch":
        """Every row becomes ``row.rmi(other[j])`` (SDR.hpp:1436-1513).

        ``combine`` is the reference's ``rmi_combiner`` (SDR.hpp:304-310, 1447-1513) in batched form: it is called
        ONCE with the payloads of all common ids -- ``combine(this_values, arg_values)``, float32 device tensors
        in row order, ascending ids within a row (``arg_values`` is None for an EmptyData argument) -- and returns
        the new payloads of those elements of ``this``; an element whose new payload is not ``rm_relevant`` (below
        0.1 for UnitData, UnitData.hpp:22-28; never for ArithData, ArithData.hpp:22-28) is removed, every other
        element stays.  The default combiner is the payload type's own ``rmi``."""
        if combine is None:
            return self._inplace("rme", other, stream)
        if self.data_kind == DATA_EMPTY:
            raise ValueError("a combiner needs payloads on this batch (EmptyData has nothing to combine)")
        if self.id_bits != 32 or other.id_bits != 32:
            raise ValueError("rmi with a combiner serves batches with 32-bit ids")
        torch = _torch()
        hits = self.andv(other, rowwise=True, stream=stream)
        wire = self.to_wire(stream=stream)
        n, nnz = self.n_rows, self.nnz
        off = wire[: n + 1].to(torch.int64) & 0xFFFFFFFF
        ids, data = wire[n + 1: n + 1 + nnz], wire[n + 1 + nnz:].view(torch.float32).clone()
        # positions are per row: make them positions in the element arrays
        lens = hits.offsets.to(torch.int64) & 0xFFFFFFFF
        row_of_hit = torch.repeat_interleave(torch.arange(n, device=wire.device), lens[1:] - lens[:-1])
        t_at = (hits.this_pos.to(torch.int64) & 0xFFFFFFFF) + off[row_of_hit]
        arg_vals = None
        if other.data_kind != DATA_EMPTY:
            o_wire = other.to_wire(stream=stream)
            o_off = o_wire[: other.n_rows + 1].to(torch.int64) & 0xFFFFFFFF
            o_data = o_wire[other.n_rows + 1 + other.nnz:].view(torch.float32)
            arg_vals = o_data[(hits.arg_pos.to(torch.int64) & 0xFFFFFFFF) + o_off[row_of_hit]]
        new_vals = combine(data[t_at], arg_vals)
        if new_vals.dtype != torch.float32 or new_vals.shape != t_at.shape:
            raise ValueError("the combiner must return one float32 payload per common id")
        data[t_at] = new_vals
        keep = torch.ones(nnz, dtype=torch.bool, device=wire.device)
        if self.data_kind == _KINDS["unit"]:
            keep[t_at] = new_vals >= 0.1        # binary32 compare: same verdict as the reference's double compare
        # new row starts: how many kept elements precede each old row start
        kept_upto = torch.cat([torch.zeros(1, dtype=torch.int64, device=wire.device), torch.cumsum(keep, 0)])
        new_off = kept_upto[off]
        new_off32 = torch.where(new_off >= 2 ** 31, new_off - 2 ** 32, new_off).to(torch.int32)
        self.upload_device(new_off32, ids[keep].contiguous(), data[keep].contiguous(), stream=stream)
        self._refresh_info()
        return self

    def equals(self, other: "SDRBatch", *, rowwise=False, stream=None):
        """``out[i, j] = (self[i] == other[j])`` as 0/1 (int32), the reference's ``operator==`` (SDR.hpp:470-483):
        same ids in the same positions and, where both batches carry payloads, equal payloads.  ``rowwise``:
        ``out[j] = (self[j] == other[j])``."""
        torch = _torch()
        dev = f"cuda:{other.device}"
        if rowwise:
            if self.n_rows != other.n_rows:
                raise ValueError("row-wise evaluation needs batches of equal length")
            out = torch.empty(max(other.n_rows, 1), dtype=torch.int32, device=dev)
        else:
            out = torch.empty((max(self.n_rows, 1), max(other.n_rows, 1)), dtype=torch.int32, device=dev)
        check(self._lib.sdrb_equal(self._h, other._h, FLAG_ROWWISE if rowwise else 0, out.data_ptr(),
                                   other.n_rows, _stream_handle(other.device, stream)))
        return out[: other.n_rows] if rowwise else out[: self.n_rows, : other.n_rows]

    # -- batched visitors ------------------------------------------------------------------------
    def _visit(self, which: int, other: "SDRBatch", swap, stream, algo, rowwise) -> VisitResult:
        torch = _torch()
        dev = f"cuda:{other.device}"
        if rowwise and self.n_rows != other.n_rows:
            raise ValueError("row-wise evaluation needs batches of equal length")
        pairs = other.n_rows if rowwise else self.n_rows * other.n_rows
        s = _stream_handle(other.device, stream)
        flags = self._flags(swap, algo, rowwise)
        offsets = torch.empty(pairs + 1, dtype=torch.int32, device=dev)
        total = C.c_uint64()
        check(self._lib.sdrb_visit_count(self._h, other._h, which, flags, offsets.data_ptr(), C.byref(total), s))
        n = int(total.value)
        tp = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ap = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        check(self._lib.sdrb_visit_fill(self._h, other._h, which, flags, offsets.data_ptr(), tp.data_ptr(),
                                        ap.data_ptr(), s))
        return VisitResult(1 if rowwise else self.n_rows, other.n_rows, offsets, tp[:n], ap[:n], n)

    def andv(self, other: "SDRBatch", *, swap=False, stream=None, algo=None, rowwise=False) -> VisitResult:
        """For every pair, the calls ``self[i].andv(other[j], visitor)`` makes (SDR.hpp:186-194): one record
        per common id, ascending -- where it sits in ``self[i]`` and in ``other[j]``."""
        return self._visit(0, other, swap, stream, algo, rowwise)

    def orv(self, other: "SDRBatch", *, swap=False, stream=None, algo=None, rowwise=False) -> VisitResult:
        """For every pair, the calls of ``self[i].orv(other[j], this_visitor, arg_visitor, both_visitor)``
        (SDR.hpp:268-269): one record per id of the union, ascending; ``POS_NONE`` marks the absent side."""
        return self._visit(1, other, swap, stream, algo, rowwise)

    # The reference's operator sugar (SDR.hpp:431-460), batch against batch.  Python has no overloadable
    # && and ||, so the size ops keep their names.
    def __and__(self, other):
        return self.ande(other)

    def __mul__(self, other):
        return self.ande(other)

    def __or__(self, other):
        return self.ore(other)

    def __add__(self, other):
        return self.ore(other)

    def __xor__(self, other):
        return self.xore(other)

    def __sub__(self, other):
        return self.rme(other)

    def __iand__(self, other):
        return self.andi(other)

    def __imul__(self, other):
        return self.andi(other)

    def __ior__(self, other):
        return self.ori(other)

    def __iadd__(self, other):
        return self.ori(other)

    def __ixor__(self, other):
        return self.xori(other)

    def __isub__(self, other):
        return self.rmi(other)

    # -- views of a batch under another payload type ------------------------------------------------
    def _view(self, kind, data_t, stream):
        torch = _torch()
        if self.id_bits != 32:
            raise ValueError("views exist for batches with 32-bit ids")
        wire = self.to_wire(stream=stream)
        n, nnz = self.n_rows, self.nnz
        st = self.stats()
        v = SDRBatch(self.device, self.width, kind)
        if data_t is None and kind != "empty":
            data_t = torch.ones(nnz, dtype=torch.float32, device=wire.device)
        v.adopt_device(wire[: n + 1], wire[n + 1: n + 1 + nnz], data_t if kind != "empty" else None,
                       max_row_len=st["max_row_len"], min_id=st["min_id"], max_id=st["max_id"])
        return v

    def ids_only(self, *, stream=None) -> "SDRBatch":
        """The same rows as an EmptyData batch (a device copy of offsets and ids; payloads dropped)."""
        return self if self.data_kind == DATA_EMPTY else self._view("empty", None, stream)

    def as_unit(self, *, stream=None) -> "SDRBatch":
        """An EmptyData batch as UnitData: every element gets ``UnitData(EmptyData)`` = 1 (UnitData.hpp:13-14)."""
        if self.data_kind != DATA_EMPTY:
            raise ValueError("as_unit converts EmptyData batches")
        return self._view("unit", None, stream)

    def _elem_op_ret(self, op, other, ret, swap, stream, algo):
        """Element op with the reference's result-type argument (``ande<ret_t, c_ret_t>``, SDR.hpp:151-152,
        209-212, 237-240, 292-293).  ``ret``: None keeps ``this``'s payload type; "empty": the data of a match is
        converted to EmptyData before its relevance is asked, so the payloads play no part (EmptyData.hpp:27-37:
        intersections keep every common id, xore and rme drop every common id); "unit" on an EmptyData ``this``:
        a match gets ``EmptyData::op<UnitData>()`` = ``UnitData()`` = 1 and is kept (EmptyData.hpp:44-45,53-54,
        64-65; 1 is relevant), elements of ``this`` alone become ``UnitData(EmptyData)`` = 1, elements of the
        argument alone keep their payload -- for ore AND xore that is the union with ones on ``this``'s side;
        rme has no such form (``EmptyData::rme`` returns EmptyData whatever is asked for, EmptyData.hpp:73-74)."""
        if ret is None:
            return self._elem_op(op, other, swap, stream, algo)
        if swap:
            raise ValueError("a result type is chosen for this.op(arg): evaluate the swapped call on the other batch")
        if ret == "empty":
            return self.ids_only(stream=stream)._elem_op(op, other.ids_only(stream=stream), False, stream, algo)
        if ret != "unit":
            raise ValueError('ret must be None, "empty" or "unit"')
        if self.data_kind != DATA_EMPTY:
            if self.data_kind != _KINDS["unit"]:
                raise ValueError('ret="unit" needs an EmptyData or UnitData batch')
            return self._elem_op(op, other, False, stream, algo)
        if op == OP_RM:
            raise ValueError("rme of an EmptyData batch yields EmptyData whatever result type is asked for")
        if op == OP_AND:
            res = self._elem_op(op, other.ids_only(stream=stream), False, stream, algo)
            for c in (res.chunks if isinstance(res, ChunkedElemResult) else [res]):
                c.total     # settles the buffers first (an estimate-sized result may be computed again, with room)
                c.data_buffer = _torch().ones_like(c.ids_buffer, dtype=_torch().float32)
            return res
        return self.as_unit(stream=stream)._elem_op(OP_OR, other, False, stream, algo)

    def ande(self, other, *, swap=False, stream=None, algo=None, ret=None) -> ElemResult:
        return self._elem_op_ret(OP_AND, other, ret, swap, stream, algo)

    def ore(self, other, *, swap=False, stream=None, algo=None, ret=None) -> ElemResult:
        return self._elem_op_ret(OP_OR, other, ret, swap, stream, algo)

    def xore(self, other, *, swap=False, stream=None, algo=None, ret=None) -> ElemResult:
        return self._elem_op_ret(OP_XOR, other, ret, swap, stream, algo)

    def rme(self, other, *, swap=False, stream=None, algo=None, ret=None) -> ElemResult:
        return self._elem_op_ret(OP_RM, other, ret, swap, stream, algo)


def synth_csr(seed: int, first_row: int, n_rows: int, width: int, active: int, with_data: bool = False):
    """Deterministic synthetic rows (see ``sdrb_synth_csr`` in ``include/sdr_amd.h``)."""
    lib = _lib.load()
    off = np.empty(n_rows + 1, dtype=np.uint64)
    ids = np.empty(n_rows * active, dtype=np.int32)
    data = np.empty(n_rows * active, dtype=np.float32) if with_data else None
    check(lib.sdrb_synth_csr(seed, first_row, n_rows, width, active, off.ctypes.data_as(C.c_void_p),
                             ids.ctypes.data_as(C.c_void_p),
                             data.ctypes.data_as(C.c_void_p) if with_data else None))
    return off, ids, data
