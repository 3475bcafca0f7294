"""CSR database files on the host: streaming writer, row-range reads, rejection of invalid rows
and of damaged files.  The reference has no serialisation, so there is nothing of the
reference to compare with: the property is the round trip (parity unpinned)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import util  # noqa: E402
from sdr_amd import SdrbError, csrfile  # noqa: E402


@pytest.mark.parametrize("payloads", [False, True])
def test_streamed_chunks_round_trip(tmp_path, payloads):
    rng = np.random.default_rng(1)
    off, ids, data = util.ragged_rows(rng, 1000, 2048, 40, payloads)
    path = tmp_path / "db.sdrb"
    cuts = [0, 1, 250, 250, 777, 1000]        # chunks of 1, 249, 0, 527 and 223 rows
    with csrfile.Writer(path, width=2048, payloads=payloads) as w:
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            e0, e1 = int(off[lo]), int(off[hi])
            w.append(off[lo:hi + 1] - off[lo], ids[e0:e1], None if data is None else data[e0:e1])
    assert not os.path.exists(str(path) + ".payload.tmp")
    meta = csrfile.info(path)
    assert meta == dict(n_rows=1000, nnz=int(off[-1]), width=2048, payloads=payloads,
                        kind="unit" if payloads else "empty")
    r_off, r_ids, r_data = csrfile.read_rows(path)
    assert np.array_equal(r_off, off) and np.array_equal(r_ids, ids)
    assert (r_data is None) == (data is None) and (data is None or np.array_equal(r_data.view(np.uint32), data.view(np.uint32)))
    # a row range comes back rebased
    s_off, s_ids, s_data = csrfile.read_rows(path, 300, 420)
    assert np.array_equal(s_off, off[300:421] - off[300])
    assert np.array_equal(s_ids, ids[int(off[300]):int(off[420])])
    # sections are 64-byte aligned and the file is no larger than header + padded sections
    size = os.path.getsize(path)
    assert size <= 64 + (ids.nbytes + 63) // 64 * 64 + ((data.nbytes + 63) // 64 * 64 if payloads else 0) + off.nbytes


def test_empty_database_and_empty_rows(tmp_path):
    path = tmp_path / "empty.sdrb"
    csrfile.write(path, np.zeros(1, np.uint64), np.zeros(0, np.int32))
    assert csrfile.info(path)["n_rows"] == 0
    off, ids, data = csrfile.read_rows(path)
    assert off.tolist() == [0] and ids.size == 0 and data is None
    csrfile.write(path, np.array([0, 0, 2, 2], np.uint64), np.array([5, 9], np.int32))
    off, ids, _ = csrfile.read_rows(path, 1, 3)
    assert off.tolist() == [0, 2, 2] and ids.tolist() == [5, 9]


def test_arith_payloads(tmp_path):
    """ArithData<float> rows: any finite value is stored, bit for bit; non-finite ones are refused."""
    rng = np.random.default_rng(7)
    off, ids, _ = util.ragged_rows(rng, 200, 512, 30, False)
    data = (rng.standard_normal(ids.size) * 1e3).astype(np.float32)
    path = tmp_path / "arith.sdrb"
    csrfile.write(path, off, ids, data, width=512, kind="arith")
    assert csrfile.info(path)["kind"] == "arith" and csrfile.info(path)["payloads"]
    _, r_ids, r_data = csrfile.read_rows(path)
    assert np.array_equal(r_ids, ids) and np.array_equal(r_data.view(np.uint32), data.view(np.uint32))
    with pytest.raises(SdrbError) as e:
        csrfile.write(path, np.array([0, 1], np.uint64), np.array([1], np.int32), np.array([np.inf], np.float32), kind="arith")
    assert e.value.code == -7


def test_writer_rejects_what_an_upload_rejects(tmp_path):
    path = tmp_path / "bad.sdrb"
    with pytest.raises(SdrbError) as e:
        csrfile.write(path, np.array([0, 3], np.uint64), np.array([1, 3, 2], np.int32))
    assert e.value.code == -2 and not path.exists()                      # aborted writers leave no file
    with pytest.raises(SdrbError) as e:
        csrfile.write(path, np.array([0, 2], np.uint64), np.array([1, 2048], np.int32), width=2048)
    assert e.value.code == -3
    with pytest.raises(SdrbError) as e:
        csrfile.write(path, np.array([0, 1], np.uint64), np.array([1], np.int32), np.array([1.5], np.float32))
    assert e.value.code == -7
    with pytest.raises(SdrbError) as e:
        csrfile.read_rows(tmp_path / "missing.sdrb")
    assert e.value.code == -8


def test_damaged_files_are_refused(tmp_path):
    rng = np.random.default_rng(2)
    off, ids, _ = util.ragged_rows(rng, 50, 300, 20, False)
    path = tmp_path / "db.sdrb"
    csrfile.write(path, off, ids, width=300)
    raw = bytearray(path.read_bytes())
    for what, mutate in [("magic", lambda b: b.__setitem__(0, ord("X"))),
                         ("truncated", lambda b: b.__delitem__(slice(len(b) - 40, len(b)))),
                         ("rows", lambda b: b.__setitem__(slice(24, 32), (10 ** 12).to_bytes(8, "little"))),
                         # n_rows + 1 wraps to zero offsets: must not pass as "fits"
                         ("rows_wrap", lambda b: b.__setitem__(slice(24, 32), (2 ** 64 - 1).to_bytes(8, "little"))),
                         # the offsets section moved onto the ids section (64-byte aligned, inside the file)
                         ("overlap", lambda b: b.__setitem__(slice(56, 64), (64).to_bytes(8, "little")))]:
        damaged = bytearray(raw)
        mutate(damaged)
        bad = tmp_path / f"{what}.sdrb"
        bad.write_bytes(bytes(damaged))
        with pytest.raises(SdrbError) as e:
            csrfile.info(bad)
        assert e.value.code == -9, what
    with pytest.raises(SdrbError) as e:
        csrfile.read_rows(path, 10, 51)
    assert e.value.code == -1
