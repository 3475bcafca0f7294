"""TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED ENGINE.

ctypes access to the CPU oracle (``libsdr_oracle.so``, the plain-C restatement in
``sdr_oracle.c``) and, when it has been built, to the real reference behind
``ref_driver.cpp`` (``_ref/libsdr_ref.so``).  Only tests, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libsdr_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libsdr_ref.so")
REF_DBG_SO = os.path.join(_HERE, "_ref", "libsdr_ref_dbg.so")
REF_FUZZ = os.path.join(_HERE, "_ref", "fuzz_sdr")

EMPTY, UNIT, ARITH = 0, 1, 2
AND, OR, XOR, RM = 0, 1, 2, 3
OP_NAMES = {AND: "and", OR: "or", XOR: "xor", RM: "rm"}

_vp = C.c_void_p
_BATCH_ARGS = [_vp, _vp, _vp, C.c_int, C.c_uint64] * 2  # (off, ids, data, kind, n) x2


def build(ref: bool = True) -> None:
    """Compiles the oracle (and the reference driver where /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "-s"] + ([] if ref else ["libsdr_oracle.so"]), check=True)


class Batch:
    """Host CSR pack: offsets uint64[n+1], ids int32, data float32 or None.  Payloads are UnitData unless
    ``arith`` says they are ArithData<float>."""

    def __init__(self, offsets, ids, data=None, arith: bool = False):
        self.off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.data = None if data is None else np.ascontiguousarray(data, dtype=np.float32)
        self.kind = EMPTY if data is None else (ARITH if arith else UNIT)
        self.n = self.off.size - 1
        assert int(self.off[-1]) == self.ids.size
        assert self.data is None or self.data.size == self.ids.size

    @classmethod
    def from_rows(cls, rows, with_data=False):
        off, ids, data = [0], [], []
        for row in rows:
            if with_data:
                r_ids, r_data = row
                data.extend(r_data)
            else:
                r_ids = row
            ids.extend(r_ids)
            off.append(len(ids))
        return cls(off, ids, data if with_data else None)

    def csr(self):
        return self.off, self.ids, self.data

    def args(self):
        return [self.off.ctypes.data_as(_vp), self.ids.ctypes.data_as(_vp),
                self.data.ctypes.data_as(_vp) if self.data is not None else None, self.kind, self.n]

    def row(self, r):
        lo, hi = int(self.off[r]), int(self.off[r + 1])
        return self.ids[lo:hi], (self.data[lo:hi] if self.data is not None else None)

    def nbytes(self):
        return self.off.nbytes + self.ids.nbytes + (self.data.nbytes if self.data is not None else 0)


class _Impl:
    """One implementation (restatement or real reference) of the batch entry points."""

    def __init__(self, path: str, prefix: str, has_inplace_flag: bool):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.is_reference = has_inplace_flag
        f = getattr(self.lib, prefix + "batch_size_op")
        f.restype, f.argtypes = None, [C.c_int] + _BATCH_ARGS + [_vp]
        f = getattr(self.lib, prefix + "batch_elem_op")
        f.restype = C.c_uint64
        f.argtypes = [C.c_int] + ([C.c_int] if has_inplace_flag else []) + _BATCH_ARGS + [_vp, _vp, _vp]
        f = getattr(self.lib, prefix + "time_size_op")
        f.restype = C.c_double
        f.argtypes = [C.c_int] + _BATCH_ARGS + [C.c_int] + ([C.c_int] if has_inplace_flag else []) + [_vp]

    def size_op(self, op: int, t: Batch, a: Batch) -> np.ndarray:
        out = np.zeros((t.n, a.n), dtype=np.uint32)
        getattr(self.lib, self.prefix + "batch_size_op")(op, *t.args(), *a.args(), out.ctypes.data_as(_vp))
        return out

    def range_size(self, b: Batch, lo, hi):
        """(counts uint32 [n_ranges, b.n], firsts uint32 [n_ranges, b.n]) -- restatement only."""
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        fn = self.lib.osdr_batch_range_size
        fn.restype, fn.argtypes = None, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp]
        counts = np.zeros((lo.size, b.n), dtype=np.uint32)
        firsts = np.zeros((lo.size, b.n), dtype=np.uint32)
        fn(b.off.ctypes.data_as(_vp), b.ids.ctypes.data_as(_vp), b.n, lo.size, lo.ctypes.data_as(_vp),
           hi.ctypes.data_as(_vp), counts.ctypes.data_as(_vp), firsts.ctypes.data_as(_vp))
        return counts, firsts

    def point(self, b: Batch, qid):
        """int32 [n_ids, b.n]: position of the id in the row or -1 -- restatement only."""
        qid = np.ascontiguousarray(qid, dtype=np.int32)
        fn = self.lib.osdr_batch_point
        fn.restype, fn.argtypes = None, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp]
        pos = np.zeros((qid.size, b.n), dtype=np.int32)
        fn(b.off.ctypes.data_as(_vp), b.ids.ctypes.data_as(_vp), b.n, qid.size, qid.ctypes.data_as(_vp),
           pos.ctypes.data_as(_vp))
        return pos

    def ref_range(self, b: Batch, lo, hi):
        """(counts, digests) from the reference's ands(lo, hi) / ande(lo, hi) -- reference build only."""
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        fn = self.lib.rsdr_batch_range
        fn.restype = None
        fn.argtypes = [_vp, _vp, _vp, C.c_int, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp]
        counts = np.zeros((lo.size, b.n), dtype=np.uint32)
        digests = np.zeros((lo.size, b.n), dtype=np.int64)
        fn(*b.args(), lo.size, lo.ctypes.data_as(_vp), hi.ctypes.data_as(_vp), counts.ctypes.data_as(_vp),
           digests.ctypes.data_as(_vp))
        return counts, digests

    def ref_point(self, b: Batch, qid):
        """(found uint8, value float32) from the reference's ande(id) -- reference build only."""
        qid = np.ascontiguousarray(qid, dtype=np.int32)
        fn = self.lib.rsdr_batch_point
        fn.restype = None
        fn.argtypes = [_vp, _vp, _vp, C.c_int, C.c_uint64, C.c_uint64, _vp, _vp, _vp]
        found = np.zeros((qid.size, b.n), dtype=np.uint8)
        value = np.zeros((qid.size, b.n), dtype=np.float32)
        fn(*b.args(), qid.size, qid.ctypes.data_as(_vp), found.ctypes.data_as(_vp), value.ctypes.data_as(_vp))
        return found, value

    def ref_transpose(self, b: Batch, width: int):
        """(col_off uint64 [width+1], rows int32, vals float32) from the reference's transpose(), in the
        order the reference emitted the elements -- reference build only."""
        fn = self.lib.rsdr_transpose
        fn.restype, fn.argtypes = None, [_vp, _vp, _vp, C.c_uint64, C.c_int32, _vp, _vp, _vp]
        col_off = np.zeros(width + 1, dtype=np.uint64)
        rows = np.zeros(b.ids.size, dtype=np.int32)
        vals = np.zeros(b.ids.size, dtype=np.float32)
        fn(b.off.ctypes.data_as(_vp), b.ids.ctypes.data_as(_vp), b.data.ctypes.data_as(_vp), b.n, width,
           col_off.ctypes.data_as(_vp), rows.ctypes.data_as(_vp), vals.ctypes.data_as(_vp))
        return col_off, rows, vals

    def ref_op64(self, op: int, elems: bool, t_csr, a_csr):
        """The reference on rows with `long` ids: (off uint64, ids int64, data-or-None) per side.  Returns
        sizes uint32 [nt, na], or (offsets, ids int64, data) of the element op -- reference build only."""
        def args(csr):
            off, ids, data = csr
            off = np.ascontiguousarray(off, dtype=np.uint64)
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            data = None if data is None else np.ascontiguousarray(data, dtype=np.float32)
            keep.extend([off, ids, data])
            return [off.ctypes.data_as(_vp), ids.ctypes.data_as(_vp),
                    data.ctypes.data_as(_vp) if data is not None else None, UNIT if data is not None else EMPTY,
                    off.size - 1]
        keep = []
        fn = self.lib.rsdr_batch_op64
        fn.restype = C.c_uint64
        fn.argtypes = [C.c_int, C.c_int] + [_vp, _vp, _vp, C.c_int, C.c_uint64] * 2 + [_vp, _vp, _vp, _vp]
        ta, aa = args(t_csr), args(a_csr)
        nt, na = ta[4], aa[4]
        if not elems:
            sizes = np.zeros((nt, na), dtype=np.uint32)
            fn(op, 0, *ta, *aa, sizes.ctypes.data_as(_vp), None, None, None)
            return sizes
        off = np.zeros(nt * na + 1, dtype=np.uint64)
        total = fn(op, 1, *ta, *aa, None, off.ctypes.data_as(_vp), None, None)
        ids = np.zeros(total, dtype=np.int64)
        data = np.zeros(total, dtype=np.float32) if t_csr[2] is not None else None
        fn(op, 1, *ta, *aa, None, off.ctypes.data_as(_vp), ids.ctypes.data_as(_vp),
           data.ctypes.data_as(_vp) if data is not None else None)
        return off, ids, data

    def inner(self, t: Batch, a: Batch) -> np.ndarray:
        """float32 [t.n, a.n]: this_i.inner(arg_j) with ArithData<float> payloads on ``t`` and
        ArithData<float> (``a.data`` given) or EmptyData (``a.data`` None) on ``a``."""
        fn = getattr(self.lib, self.prefix + "batch_inner")
        fn.restype, fn.argtypes = None, [_vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, _vp]
        if t.data is None:
            raise ValueError("inner needs payloads on the this side")
        out = np.zeros((t.n, a.n), dtype=np.float32)
        fn(t.off.ctypes.data_as(_vp), t.ids.ctypes.data_as(_vp), t.data.ctypes.data_as(_vp), t.n,
           a.off.ctypes.data_as(_vp), a.ids.ctypes.data_as(_vp),
           a.data.ctypes.data_as(_vp) if a.data is not None else None, a.n, out.ctypes.data_as(_vp))
        return out

    def equal(self, t: Batch, a: Batch) -> np.ndarray:
        """uint32 [t.n, a.n]: 1 where this_i == arg_j (SDR::operator==, SDR.hpp:470-483)."""
        fn = getattr(self.lib, self.prefix + "batch_equal")
        fn.restype, fn.argtypes = None, _BATCH_ARGS + [_vp]
        out = np.zeros((t.n, a.n), dtype=np.uint32)
        fn(*t.args(), *a.args(), out.ctypes.data_as(_vp))
        return out

    def visit(self, which: int, t: Batch, a: Batch):
        """The callbacks of andv (``which`` 0) / orv (``which`` 1) of every pair, in call order:
        (offsets uint64 [t.n*a.n+1], this_pos uint32, arg_pos uint32); 0xFFFFFFFF marks the absent side of
        an orv single.  Payloads play no part."""
        fn = getattr(self.lib, self.prefix + "batch_visit")
        fn.restype = C.c_uint64
        fn.argtypes = [C.c_int, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, _vp, _vp, _vp]
        off = np.zeros(t.n * a.n + 1, dtype=np.uint64)
        ids = [t.off.ctypes.data_as(_vp), t.ids.ctypes.data_as(_vp), t.n, a.off.ctypes.data_as(_vp),
               a.ids.ctypes.data_as(_vp), a.n]
        total = fn(which, *ids, off.ctypes.data_as(_vp), None, None)
        tp = np.zeros(total, dtype=np.uint32)
        ap = np.zeros(total, dtype=np.uint32)
        fn(which, *ids, off.ctypes.data_as(_vp), tp.ctypes.data_as(_vp), ap.ctypes.data_as(_vp))
        return off, tp, ap

    def elem_op(self, op: int, t: Batch, a: Batch, inplace: bool = False):
        """Returns (offsets uint64[t.n*a.n+1], ids, data-or-None) of this_i.op(arg_j)."""
        fn = getattr(self.lib, self.prefix + "batch_elem_op")
        head = [op] + ([1 if inplace else 0] if self.is_reference else [])
        if inplace and not self.is_reference:
            raise ValueError("the restatement exposes in-place ops through inplace_op()")
        off = np.zeros(t.n * a.n + 1, dtype=np.uint64)
        total = fn(*head, *t.args(), *a.args(), off.ctypes.data_as(_vp), None, None)
        ids = np.zeros(total, dtype=np.int32)
        data = np.zeros(total, dtype=np.float32) if t.kind != EMPTY else None
        fn(*head, *t.args(), *a.args(), off.ctypes.data_as(_vp), ids.ctypes.data_as(_vp),
           data.ctypes.data_as(_vp) if data is not None else None)
        return off, ids, data

    def time_size_op(self, op: int, t: Batch, a: Batch, repeats: int = 3, threads: int = 1):
        """Best-of-``repeats`` seconds for one pass over all pairs, and the result checksum."""
        chk = C.c_uint64()
        fn = getattr(self.lib, self.prefix + "time_size_op")
        tail = [repeats] + ([threads] if self.is_reference else []) + [C.byref(chk)]
        if threads != 1 and not self.is_reference:
            raise ValueError("the C restatement is timed on one thread")
        sec = fn(op, *t.args(), *a.args(), *tail)
        return sec, int(chk.value)


_port = None
_refs = {}


def port() -> _Impl:
    """The plain-C restatement (always available once built)."""
    global _port
    if _port is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        _port = _Impl(ORACLE_SO, "osdr_", False)
        _port.lib.osdr_inplace_op.restype = C.c_size_t
        _port.lib.osdr_inplace_op.argtypes = [C.c_int, _vp, _vp, C.c_size_t, C.c_int, _vp]
    return _port


def have_reference() -> bool:
    return os.path.exists(REF_SO)


def reference(debug: bool = False) -> _Impl:
    """The real reference (oracle/_ref); ``debug`` keeps its asserts on."""
    path = REF_DBG_SO if debug else REF_SO
    if path not in _refs:
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built (needs /root/reference; run make -C oracle ref)")
        impl = _Impl(path, "rsdr_", True)
        impl.lib.rsdr_inplace_self.restype = C.c_uint64
        impl.lib.rsdr_inplace_self.argtypes = [C.c_int, _vp, _vp, C.c_uint64, C.c_int]
        _refs[path] = impl
    return _refs[path]


class _View(C.Structure):
    _fields_ = [("ids", _vp), ("data", _vp), ("n", C.c_size_t), ("kind", C.c_int)]


def port_inplace(op: int, t_ids, t_data, a_ids, a_data, alias: bool = False):
    """this.opi(arg) on the restatement; ``alias`` passes this's own arrays as the argument."""
    p = port()
    t_kind = EMPTY if t_data is None else UNIT
    n_t = len(t_ids)
    n_a = n_t if alias else len(a_ids)
    ids = np.zeros(n_t + n_a + 1, dtype=np.int32)
    ids[:n_t] = t_ids
    data = np.zeros(n_t + n_a + 1, dtype=np.float32)
    if t_data is not None:
        data[:n_t] = t_data
    if alias:
        view = _View(ids.ctypes.data, data.ctypes.data if t_data is not None else None, n_t, t_kind)
    else:
        a_ids = np.ascontiguousarray(a_ids, dtype=np.int32)
        a_data_arr = None if a_data is None else np.ascontiguousarray(a_data, dtype=np.float32)
        view = _View(a_ids.ctypes.data, a_data_arr.ctypes.data if a_data_arr is not None else None,
                     n_a, EMPTY if a_data is None else UNIT)
    n = p.lib.osdr_inplace_op(op, ids.ctypes.data_as(_vp),
                              data.ctypes.data_as(_vp) if t_data is not None else None, n_t, t_kind,
                              C.byref(view))
    return ids[:n].copy(), (data[:n].copy() if t_data is not None else None)
