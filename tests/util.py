"""Shared helpers of the test-suite: deterministic inputs and result digests."""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_MASK = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per uint64 state value (vectorised)."""
    with np.errstate(over="ignore"):
        z = (x + GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_rows_numpy(seed: int, first_row: int, n_rows: int, width: int, active: int, with_data=False):
    """Independent numpy statement of the synthetic-row generator: row r is the sorted first
    ``active`` entries of a partial Fisher-Yates shuffle of 0..width-1 driven by splitmix64
    seeded with seed ^ (r * GOLDEN); payloads (x >> 40) * 2^-24 follow, in id order."""
    rows = np.arange(first_row, first_row + n_rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        state = np.uint64(seed) ^ (rows * GOLDEN)
    perm = np.tile(np.arange(width, dtype=np.int32), (n_rows, 1))
    ar = np.arange(n_rows)
    for i in range(active):
        x = splitmix64(state)
        with np.errstate(over="ignore"):
            state = state + GOLDEN
        j = i + (x % np.uint64(width - i)).astype(np.int64)
        pi = perm[ar, i].copy()
        perm[ar, i] = perm[ar, j]
        perm[ar, j] = pi
    ids = np.sort(perm[:, :active], axis=1).astype(np.int32)
    data = None
    if with_data:
        data = np.empty((n_rows, active), dtype=np.float32)
        for k in range(active):
            x = splitmix64(state)
            with np.errstate(over="ignore"):
                state = state + GOLDEN
            data[:, k] = (x >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
        data = data.reshape(-1)
    off = (np.arange(n_rows + 1, dtype=np.uint64) * np.uint64(active))
    return off, ids.reshape(-1), data


# payload palette that sits on and around the relevance threshold 0.1
_F = np.float32
PALETTE = np.array([0.0, 1.0, 0.1, np.nextafter(_F(0.1), _F(0)), np.nextafter(_F(0.1), _F(1)),
                    0.5, 0.25, 0.2, 0.31622776, 0.3162278, 0.9, 0.05, 0.11111111, 0.9999999,
                    0.099999994, 0.7], dtype=np.float32)


def fuzz_payload(seed: int, sdr_number: np.ndarray, bit: np.ndarray) -> np.ndarray:
    """Payload of element `bit` of fuzz SDR `sdr_number`: half palette, half uniform 24-bit."""
    with np.errstate(over="ignore"):
        x = splitmix64(np.uint64(seed) ^ (sdr_number.astype(np.uint64) * GOLDEN)
                       ^ (bit.astype(np.uint64) << np.uint64(48)))
    pal = PALETTE[((x >> np.uint64(8)) % np.uint64(len(PALETTE))).astype(np.int64)]
    uni = (x >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return np.where((x & np.uint64(1)) == 0, pal, uni).astype(np.float32)


def fuzz_universe(n: int, seed: int | None):
    """The reference fuzzer's inputs (src/fuzz.cpp:185-219): SDR v holds the set-bit positions
    of v, for v in [0, n).  Payloads (seed given) are NOT filtered for relevance (fuzz.cpp:209)."""
    off, ids, sdr_no = [0], [], []
    for v in range(n):
        bits = [i for i in range(32) if (v >> i) & 1]
        ids.extend(bits)
        sdr_no.extend([v] * len(bits))
        off.append(len(ids))
    ids = np.asarray(ids, dtype=np.int32)
    data = None
    if seed is not None:
        data = fuzz_payload(seed, np.asarray(sdr_no, dtype=np.int64), ids.astype(np.int64))
    return np.asarray(off, dtype=np.uint64), ids, data


_P = 0x100000001B3  # odd multiplier; powers taken mod 2^64


def _powers(n: int) -> np.ndarray:
    p = np.empty(n + 1, dtype=np.uint64)
    acc = 1
    for k in range(n + 1):
        p[k] = acc
        acc = (acc * _P) & _MASK
    return p


def pair_digests(off: np.ndarray, ids: np.ndarray, data: np.ndarray | None) -> np.ndarray:
    """Order- and content-sensitive 32-bit digest of every CSR segment:
    sum over elements k of (id+1)*P^(2k+1) + (payload_bits+1)*P^(2k+2)  (mod 2^64), folded."""
    off = np.asarray(off, dtype=np.uint64)
    nseg = off.size - 1
    out = np.zeros(nseg, dtype=np.uint64)
    total = int(off[-1])
    if total:
        lens = np.diff(off).astype(np.int64)
        pos = np.arange(total, dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), lens)
        pw = _powers(2 * int(lens.max()) + 2)
        with np.errstate(over="ignore"):
            term = (ids.astype(np.int64).astype(np.uint64) + np.uint64(1)) * pw[2 * pos + 1]
            if data is not None:
                bits = np.asarray(data, dtype=np.float32).view(np.uint32).astype(np.uint64)
                term = term + (bits + np.uint64(1)) * pw[2 * pos + 2]
        nz = lens > 0
        starts = off[:-1].astype(np.int64)[nz]
        out[nz] = np.add.reduceat(term, starts)
    return ((out >> np.uint64(32)) ^ (out & np.uint64(0xFFFFFFFF))).astype(np.uint32)


def visit_digests(off: np.ndarray, this_pos: np.ndarray, arg_pos: np.ndarray) -> np.ndarray:
    """One order-sensitive 64-bit digest per pair of its visitor records (this position, arg position)."""
    off = np.asarray(off, dtype=np.uint64)
    if off.size <= 1:
        return np.zeros(0, np.uint64)
    lens = np.diff(off).astype(np.int64)
    k = np.arange(this_pos.size, dtype=np.uint64) - np.repeat(off[:-1], lens)
    with np.errstate(over="ignore"):
        term = (k + np.uint64(1)) * (this_pos.astype(np.uint64) * np.uint64(0x9E3779B1)
                                     + arg_pos.astype(np.uint64) * np.uint64(0x85EBCA77) + np.uint64(1))
        csum = np.concatenate([np.zeros(1, np.uint64), np.cumsum(term, dtype=np.uint64)])
        return csum[off[1:].astype(np.int64)] - csum[off[:-1].astype(np.int64)]


def ragged_rows(rng: np.random.Generator, n_rows: int, width: int, max_len: int, with_data=False,
                lo: int = 0):
    """Random ragged batch: row lengths uniform in [0, max_len], ids in [lo, lo+width)."""
    off, ids, data = [0], [], []
    for _ in range(n_rows):
        ln = int(rng.integers(0, max_len + 1))
        ln = min(ln, width)
        row = np.sort(rng.choice(width, size=ln, replace=False)).astype(np.int64) + lo
        ids.extend(row.tolist())
        if with_data:
            x = rng.integers(0, 1 << 62, size=ln, dtype=np.uint64)
            pal = PALETTE[(x >> np.uint64(8)) % np.uint64(len(PALETTE))]
            uni = rng.random(ln, dtype=np.float32)
            data.extend(np.where((x & np.uint64(1)) == 0, pal, uni).tolist())
        off.append(len(ids))
    return (np.asarray(off, dtype=np.uint64), np.asarray(ids, dtype=np.int32),
            np.asarray(data, dtype=np.float32) if with_data else None)


def arith_rows(rng: np.random.Generator, n_rows: int, width: int, max_len: int):
    """Ragged rows with ArithData<float>-style payloads: signed values over six decades, so that
    a sum's bits depend on the order of its terms."""
    off, ids, _ = ragged_rows(rng, n_rows, width, max_len, False)
    mag = 10.0 ** rng.uniform(-3.0, 3.0, ids.size)
    data = (rng.standard_normal(ids.size) * mag).astype(np.float32)
    return off, ids, data
