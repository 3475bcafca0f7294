#!/usr/bin/env python3
"""Headline benchmark: batched `ands`, 4096 queries x 1 Mi-row CSR database per GPU,
width 2048 / 40 active ids, EmptyData, uint32 counts[Q][N] (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one query batch: (N > 1 only) RCCL broadcast of the
packed queries from rank 0, then the size-op call on this rank's row shard.  Inputs are
resident in HBM before the timed region.  The database is sharded by rows, every rank holds
1 Mi rows (weak scaling) and counts stay sharded.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WIDTH, ACTIVE = 2048, 40
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def algorithmic_bytes(nq: int, nd: int) -> int:
    """Read every input once, write every output once: CSR rows are 4*A ids + a 4-byte offset,
    counts are uint32."""
    return nd * (4 * ACTIVE + 4) + nq * (4 * ACTIVE + 4) + nq * nd * 4


def cpu_baseline(nq: int, nd: int, threads: int):
    """The reference's own std::vector path (oracle/_ref, built from its headers) on this host's
    cores over a bounded sample of the same workload; falls back to the C restatement."""
    import oracle
    import sdr_amd
    q = oracle.Batch(*sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE))
    d = oracle.Batch(*sdr_amd.synth_csr(1, 0, nd, WIDTH, ACTIVE))
    if oracle.have_reference():
        impl, kind = oracle.reference(), "reference"
    else:
        impl, kind, threads = oracle.port(), "port", 1
    sec1, chk1 = impl.time_size_op(oracle.AND, q, d, repeats=2, threads=1)
    one_thread = nq * nd / sec1
    if threads > 1:
        sec, chk = impl.time_size_op(oracle.AND, q, d, repeats=3, threads=threads)
        assert chk == chk1
    else:
        sec = sec1
    # BASELINE.json configs[0] itself: ONE query against 10 000 rows, one thread
    q1 = oracle.Batch(*sdr_amd.synth_csr(2, 0, 1, WIDTH, ACTIVE))
    d1 = oracle.Batch(*sdr_amd.synth_csr(1, 0, 10_000, WIDTH, ACTIVE))
    sec_c1, _ = impl.time_size_op(oracle.AND, q1, d1, repeats=5, threads=1)
    return {
        "value": nq * nd / sec, "unit": "pair-ops/s", "cores": threads, "kind": kind,
        "sample": f"ands, {nq} queries x {nd} rows, width {WIDTH}/{ACTIVE} active, std::vector rows, "
                  f"g++ -O2, best of 3; single thread: {one_thread:.3e} pair-ops/s; the reference's own case "
                  f"(1 query x 10000 rows, one thread, best of 5): {10_000 / sec_c1:.3e} pair-ops/s -- the rate does "
                  f"not depend on the sample size once the rows fit the caches",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--rows", type=int, default=1 << 20, help="database rows per GPU")
    ap.add_argument("--op", default="ands", choices=["ands", "ors", "xors", "rms"])
    ap.add_argument("--algo", default=None, choices=[None, "merge", "bitslice", "popcount"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other BASELINE.json configs reported beside the metric (element ops, UnitData, all-pairs)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise RCCL and broadcast the queries every step even with one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import sdr_amd
    from sdr_amd.shard import HipEngine, ShardedDatabase

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    trace_on = os.environ.get("SDRB_BENCH_TRACE") == "1"

    def trace(what):   # SDRB_BENCH_TRACE=1: where every rank is, on stderr
        if trace_on:
            print(f"[bench rank {rank}] {what}", file=sys.stderr, flush=True)

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # SDRB_BENCH_REHEARSE_ON_ONE_GPU=1: every rank uses GPU 0 and the collectives run over gloo (RCCL refuses two ranks
    # on one device) -- a rehearsal of the N > 1 control flow on a one-GPU box; its line says so and is no measurement
    rehearsal = os.environ.get("SDRB_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    use_dist = world > 1 or args.force_collectives
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    trace("process group up" if use_dist else "single process")
    nq, nd = args.queries, args.rows
    engine = HipEngine(local_rank)

    # ---- inputs, resident in HBM before the timed region
    q_off, q_ids, _ = sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE)
    d_off, d_ids, _ = sdr_amd.synth_csr(1, rank * nd, nd, WIDTH, ACTIVE)   # this rank's rows
    db = ShardedDatabase(engine.pack_host(d_off, d_ids, None, WIDTH), rank * nd, (rank + 1) * nd,
                         world * nd, WIDTH, engine)
    # the query batch in wire form; its plan (shape + id statistics) is known on every rank
    from sdr_amd.shard import PackedQueries
    packed_q = PackedQueries.from_host_csr(q_off, q_ids, None, width=WIDTH, engine=engine)
    plan = packed_q.plan
    recv = torch.empty(plan.words, dtype=torch.int32, device=dev) if rank != 0 else None
    if rank != 0:
        packed_q = None
    out = torch.empty((nq, nd), dtype=torch.int32, device=dev)
    trace("inputs packed")
    state = {"queries": db.broadcast_packed(packed_q, plan, src=0, force_collective=args.force_collectives, recv=recv)}
    trace("queries exchanged once")

    def step():
        # N > 1: one RCCL broadcast of the packed queries, adopted in place, then the kernel
        if use_dist:
            if rehearsal:
                # two processes that each keep several 3 ms kernels queued on ONE GPU while gloo stages the exchange
                # through the host did not finish on the test boxes (steps 5, warm-up 2: every launch returned, the
                # device never went idle); with the device idle at each exchange they do.  A rank has its own GPU in a
                # real run, and RCCL orders the broadcast on the device.
                torch.cuda.synchronize()
            state["queries"] = db.broadcast_packed(packed_q, plan, src=0, force_collective=args.force_collectives,
                                                   recv=recv, into=state["queries"])
        getattr(state["queries"], args.op)(db.shard, out=out, algo=args.algo)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    trace("warm-up enqueued")
    fence()
    trace("warm-up done")
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    trace("timed steps enqueued")
    fence()
    trace("timed steps done")
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # device time of the K steps on the launch stream
    if use_dist:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    trace("times reduced")
    # ---- parity gate on the timed output (outside the timed region)
    verified = None
    if not args.no_verify and args.op == "ands":
        # checksum of checksums: sum of all overlaps == sum over ids of (#queries with id)*(#rows with id)
        hq = np.bincount(q_ids, minlength=WIDTH).astype(np.int64)
        hd = np.bincount(d_ids, minlength=WIDTH).astype(np.int64)
        got = int(out.sum(dtype=torch.int64))
        verified = got == int((hq * hd).sum())
        if rank == 0:
            import oracle
            rng = np.random.default_rng(0)
            qi = rng.integers(0, nq, 16)
            dj = rng.integers(0, nd, 512)
            want = oracle.port().size_op(oracle.AND, oracle.Batch(*_gather_rows(q_off, q_ids, qi)),
                                         oracle.Batch(*_gather_rows(d_off, d_ids, dj)))
            sample = out[torch.from_numpy(qi).to(dev)][:, torch.from_numpy(dj).to(dev)].cpu().numpy()
            verified = verified and bool(np.array_equal(sample.view(np.uint32), want))
        if not verified:
            raise SystemExit("parity check failed: device counts differ from the oracle")

    if rank == 0:
        pair_ops = world * nq * nd * args.steps
        ms_per_step = elapsed * 1e3 / args.steps
        kernel_ms = dev_ms / args.steps
        ach = algorithmic_bytes(nq, nd) / (kernel_ms * 1e-3) / 1e9
        # HBM traffic cannot be counted inside this run (the counters need rocprofv3 passes of their own): the figure
        # is the one those passes left for this very shape, and `traffic_source` says where it comes from
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("queries") == nq and tj.get("rows") == nd and tj.get("op") == args.op:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = {"file": "profiles/hbm_traffic.json", "tag": tj.get("tag"),
                                      "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command "
                                             "(tools/profile_bench.sh), not measured in this run"}
            except Exception:
                traffic, traffic_source = None, None
        line = {
            "metric": "SDR pair-ops/sec (ands, 2048-bit/40-active)",
            "value": pair_ops / elapsed,
            "unit": "pair-ops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"batched {args.op}: {nq} queries x {nd} rows per GPU CSR database, "
                                   f"width {WIDTH}, {ACTIVE} active, EmptyData, uint32 counts[Q][N]",
                       "queries": nq, "rows_per_gpu": nd, "width": WIDTH, "active": ACTIVE,
                       "sharding": f"rows x{world}, counts stay sharded"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kernel_ms, "algorithmic_bytes": algorithmic_bytes(nq, nd)},
            "parity_checked": verified,
        }
        if rehearsal:
            line["rehearsal"] = "all ranks on GPU 0, collectives over gloo: control flow only, not a measurement"
        if world == 1 and args.op == "ands" and nq > 1:
            n0 = int(q_off[1])
            line["one_query"] = one_query_regime(engine.pack_host(q_off[:2], q_ids[:n0], None, WIDTH), db.shard, nd, dev)
        if world == 1 and not args.no_secondary and args.op == "ands":
            # the other configs of BASELINE.json, one GPU's share each, outside the timed region and never
            # part of `value`; the 17 GB of counts are released first
            del out
            state.clear()
            torch.cuda.empty_cache()
            line["secondary"] = secondary_configs(dev)
            line["secondary"]["ingest"] = ingest_rates(dev)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(64, 100_000, threads)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def one_query_regime(q1, shard, nd: int, dev) -> dict:
    """The other regime of the same op, reported beside the metric (never part of `value`): ONE query
    against the resident rows, where reading the database, not writing counts, is the bound
    (4*ACTIVE + 4 bytes of row, 4 bytes of count per pair-op)."""
    import torch
    out1 = torch.empty((1, nd), dtype=torch.int32, device=dev)
    for _ in range(3):
        q1.ands(shard, out=out1)
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        q1.ands(shard, out=out1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = algorithmic_bytes(1, nd)
    ach = nbytes / (ms * 1e-3) / 1e9
    res = {"workload": f"ands: 1 query x {nd} rows", "value": nd / (ms * 1e-3), "unit": "pair-ops/s",
           "ms": ms, "algorithmic_bytes": nbytes, "achieved": ach, "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS}
    # BASELINE.json configs[0], the reference's own CPU-runnable case, on the GPU: one query against 10 000 rows, as a
    # caller sees it (launch included, counts left in HBM) and with the counts brought back to the host
    import sdr_amd
    from sdr_amd import SDRBatch
    n_small = 10_000
    small = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, n_small, WIDTH, ACTIVE)[:2], None, width=WIDTH)
    out_s = torch.empty((1, n_small), dtype=torch.int32, device=dev)
    host = torch.empty((1, n_small), dtype=torch.int32).pin_memory()
    for _ in range(3):
        q1.ands(small, out=out_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        q1.ands(small, out=out_s)
    torch.cuda.synchronize()
    call_us = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for _ in range(200):
        q1.ands(small, out=out_s)
        host.copy_(out_s, non_blocking=True)
        torch.cuda.synchronize()
    round_trip_us = (time.perf_counter() - t0) / 200 * 1e6
    res["config1_1x10000"] = {"workload": f"ands: 1 query x {n_small} rows (the reference's CPU case)",
                              "us_per_call_pipelined": call_us, "pair_ops_per_s_pipelined": n_small / (call_us * 1e-6),
                              "us_per_call_with_counts_on_host": round_trip_us,
                              "pair_ops_per_s_with_counts_on_host": n_small / (round_trip_us * 1e-6)}
    return res


def ingest_rates(dev) -> dict:
    """Getting a database into HBM (never part of `value`): packing 1 Mi std::vector rows into host CSR through the
    C++ front end and uploading it, timed by the small binary built beside the library (rows with the memory layout
    of the reference's default SDR), and the upload of the same CSR through the C ABI as this process sees it --
    each next to a plain copy of the same bytes from page-locked memory measured in the same run."""
    import subprocess

    import torch

    import sdr_amd
    from sdr_amd import SDRBatch
    out = {}
    exe = os.path.join(ROOT, "sdr_amd", "bench_ingest")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe, str(1 << 20), str(WIDTH), str(ACTIVE), "5"], capture_output=True, text=True, timeout=240)
            out["cpp_front_end"] = json.loads(r.stdout) if r.returncode == 0 else {"error": r.stderr[-300:]}
        except Exception as e:      # noqa: BLE001 -- the bench line still goes out
            out["cpp_front_end"] = {"error": repr(e)}
    else:
        out["cpp_front_end"] = {"error": "sdr_amd/bench_ingest has not been built"}
    nd = 1 << 20
    for name, with_data in (("empty", False), ("unit", True)):
        off, ids, data = sdr_amd.synth_csr(1, 0, nd, WIDTH, ACTIVE, with_data)
        nbytes = ids.nbytes * (2 if with_data else 1) + (nd + 1) * 4
        b = SDRBatch(int(str(dev).split(":")[1]), WIDTH, "unit" if with_data else "empty")
        b.upload(off, ids, data)
        best = 1e30
        for _ in range(5):
            t0 = time.perf_counter()
            b.upload(off, ids, data)        # checks included; synchronises
            best = min(best, time.perf_counter() - t0)
        pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        target = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        best_pin = 1e30
        for _ in range(5):
            t0 = time.perf_counter()
            target.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
            best_pin = min(best_pin, time.perf_counter() - t0)
        # the same upload from arrays the caller already holds page-locked (no staging for ids and payloads)
        ids_pin = torch.from_numpy(ids).pin_memory().numpy()
        data_pin = torch.from_numpy(data).pin_memory().numpy() if with_data else None
        b.upload(off, ids_pin, data_pin)
        best_direct = 1e30
        for _ in range(5):
            t0 = time.perf_counter()
            b.upload(off, ids_pin, data_pin)
            best_direct = min(best_direct, time.perf_counter() - t0)
        del ids_pin, data_pin
        out[f"upload_{name}"] = {"workload": f"sdrb_batch_upload_csr: {nd} rows, width {WIDTH}/{ACTIVE}, "
                                             f"{'UnitData' if with_data else 'EmptyData'} host CSR (pageable numpy arrays)",
                                 "bytes": int(nbytes), "upload_ms": best * 1e3, "upload_GBps": nbytes / best / 1e9,
                                 "pinned_h2d_ms": best_pin * 1e3, "pinned_h2d_GBps": nbytes / best_pin / 1e9,
                                 "upload_vs_pinned": best_pin / best,
                                 "upload_from_page_locked_arrays_ms": best_direct * 1e3,
                                 "upload_from_page_locked_arrays_GBps": nbytes / best_direct / 1e9}
        del b, pinned, target
    return out


I8_PEAK_OPS = 5.03e15   # dense int8 MFMA: 256 CUs x 4 SIMDs x 2048 ops/clk x 2.4 GHz


def _timed(fn, reps=5):
    import torch
    fn()
    torch.cuda.synchronize()
    best, res = 1e30, None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        res = fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, res


def _elem_parity(res, counts, op, q_csr, d_csr, nd, sample_rows, with_data):
    """Result lengths equal the size op's counts for EVERY pair, and the elements (ids, payload bits) of every
    query against a sample of rows equal the oracle's."""
    import oracle
    import torch
    nq = res.na
    off = res.offsets
    lens = (off[1:] - off[:-1]).view(nq, nd)
    ok = bool(torch.equal(lens, counts[:nq, :nd]))
    q = oracle.Batch(*q_csr)
    d_off, d_ids, d_data = d_csr
    s_off, s_ids, _ = _gather_rows(d_off, d_ids, sample_rows)
    s_data = np.concatenate([d_data[int(d_off[r]):int(d_off[r + 1])] for r in sample_rows]) if with_data else None
    w_off, w_ids, w_data = oracle.port().elem_op(op, q, oracle.Batch(s_off, s_ids, s_data))
    pair = torch.as_tensor(np.add.outer(np.arange(nq) * nd, sample_rows).reshape(-1), device=off.device)
    lo = off[pair].cpu().numpy().view(np.uint32).astype(np.int64)
    hi = off[pair + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    ids_host = [res.ids_buffer[a:b].cpu().numpy() for a, b in zip(lo, hi)]
    got_ids = np.concatenate(ids_host) if ids_host else np.zeros(0, np.int32)
    ok = ok and np.array_equal(hi - lo, np.diff(w_off).astype(np.int64)) and np.array_equal(got_ids, w_ids)
    if with_data and ok:
        got_data = np.concatenate([res.data_buffer[a:b].cpu().numpy() for a, b in zip(lo, hi)])
        ok = np.array_equal(got_data.view(np.uint32), w_data.view(np.uint32))
    return bool(ok)


def secondary_configs(dev) -> dict:
    """BASELINE.json configs 3, 4 and 5 on one GPU: time through the API (count, host read of the total,
    allocation and fill for the element ops), algorithmic bytes = every input once + every output once, the
    fraction of the 8 TB/s peak they amount to (int8 operations against the matrix-core peak for the dense
    all-pairs kernel), and a parity bit from the oracle."""
    import torch

    import oracle
    import sdr_amd
    from sdr_amd import SDRBatch
    out = {}
    rng = np.random.default_rng(7)

    def line(ms, nbytes, parity, **extra):
        ach = nbytes / (ms * 1e-3) / 1e9
        return dict(ms=ms, algorithmic_bytes=int(nbytes), achieved_GBps=ach, frac=ach / HBM_PEAK_GBS, parity=parity, **extra)

    # config 3: element ops with variable-length outputs over 1 Mi rows, EmptyData 2048/40
    nd = 1 << 20
    d_csr = sdr_amd.synth_csr(1, 0, nd, WIDTH, ACTIVE)
    d = SDRBatch.from_csr(*d_csr, width=WIDTH)
    sample = np.sort(rng.choice(nd, 48, replace=False))
    for nq in (1, 16):
        q_csr = sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE)
        q = SDRBatch.from_csr(*q_csr, width=WIDTH)
        in_bytes = (nd + nq) * (4 * ACTIVE + 4)
        for name, size_name, op in (("ande", "ands", oracle.AND), ("ore", "ors", oracle.OR)):
            ms, res = _timed(lambda: getattr(q, name)(d))
            counts = getattr(q, size_name)(d)
            nbytes = in_bytes + (nq * nd + 1) * 4 + res.total * 4
            out[f"c3_{name}_q{nq}"] = line(ms, nbytes, _elem_parity(res, counts, op, q_csr, d_csr, nd, sample, False),
                                           workload=f"{name}: {nq} x {nd} rows, width {WIDTH}/{ACTIVE}, EmptyData",
                                           elements=res.total)
            del res, counts
    # ids without a usable width, a 64-query slab of the headline shape.  Ids far apart and partly negative (times 100003,
    # less 10^9: nothing to declare or derive): the default choice -- small labels, then the table kernels -- beside the
    # per-pair merge kernel it replaced; the same rows with a declared width, with none declared but none negative (one is
    # derived), and with every id 1000 lower (close together: moved to zero for the call).  Elements of the far ids: a
    # wave per pair with binary searches, beside the lane-per-pair merge.
    nq = 64
    q_csr = sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE)
    counts = torch.empty((nq, nd), dtype=torch.int32, device=dev)
    q = SDRBatch.from_csr(*q_csr, width=WIDTH)
    ms_declared, _ = _timed(lambda: q.ands(d, out=counts))
    want = counts.clone()
    qu, du = SDRBatch.from_csr(*q_csr), SDRBatch.from_csr(*d_csr)
    ms_undeclared, _ = _timed(lambda: qu.ands(du, out=counts))
    same = bool(torch.equal(counts, want))
    qm = SDRBatch.from_csr(q_csr[0], q_csr[1] - 1000, None)
    dm = SDRBatch.from_csr(d_csr[0], d_csr[1] - 1000, None)
    ms_moved, _ = _timed(lambda: qm.ands(dm, out=counts))
    same = same and bool(torch.equal(counts, want))
    far = lambda x: (x.astype(np.int64) * 100003 - 1000000000).astype(np.int32)   # noqa: E731
    qn = SDRBatch.from_csr(q_csr[0], far(q_csr[1]), None)
    dn = SDRBatch.from_csr(d_csr[0], far(d_csr[1]), None)
    ms_labels, _ = _timed(lambda: qn.ands(dn, out=counts))
    same = same and bool(torch.equal(counts, want))
    ms_merge, _ = _timed(lambda: qn.ands(dn, out=counts, algo="merge"), reps=3)
    same = same and bool(torch.equal(counts, want))
    cols = torch.as_tensor(sample, device=dev)
    same = same and bool(np.array_equal(
        want[:, cols].cpu().numpy().view(np.uint32),
        oracle.port().size_op(oracle.AND, oracle.Batch(*q_csr[:2], None), oracle.Batch(*_gather_rows(d_csr[0], d_csr[1], sample)[:2], None))))
    ms_far_ande, res_far = _timed(lambda: qn.ande(dn), reps=3)
    ms_far_ande_merge, res_merge = _timed(lambda: qn.ande(dn, algo="merge"), reps=2)
    res_declared = q.ande(d)
    same = same and bool(torch.equal(res_far.offsets, res_merge.offsets) and torch.equal(res_far.ids, res_merge.ids) and
                         torch.equal(res_far.offsets, res_declared.offsets) and res_far.total == res_declared.total)
    del res_far, res_merge, res_declared
    entry = line(ms_labels, (nd + nq) * (4 * ACTIVE + 4) + nq * nd * 4, same,
                 workload=f"ands: {nq} x {nd} rows, width {WIDTH}/{ACTIVE}, EmptyData, ids far apart and partly negative, no width declared")
    entry.update(declared_width_ms=ms_declared, undeclared_nonnegative_ms=ms_undeclared, ids_1000_lower_moved_to_zero_ms=ms_moved,
                 merge_kernel_ms=ms_merge, merge_pair_ops_per_s=nq * nd / (ms_merge * 1e-3),
                 ande_far_ids_ms=ms_far_ande, ande_far_ids_lane_merge_ms=ms_far_ande_merge)
    out["widthless_ands_q64"] = entry
    del d, q, qu, du, qn, dn, qm, dm, counts, want
    torch.cuda.empty_cache()

    # config 4: UnitData 16384 / 328 active, one GPU's 128 Ki rows of the 1 Mi-row database, 64 queries
    w4, a4, nq4, nd4 = 16384, 328, 64, 1 << 17
    d_csr = sdr_amd.synth_csr(1, 0, nd4, w4, a4, True)
    q_csr = sdr_amd.synth_csr(2, 0, nq4, w4, a4, True)
    d = SDRBatch.from_csr(*d_csr, width=w4)
    q = SDRBatch.from_csr(*q_csr, width=w4)
    in_bytes = (nd4 + nq4) * (8 * a4 + 4)
    ms_s, counts = _timed(lambda: q.ands(d))
    sample = np.sort(rng.choice(nd4, 16, replace=False))
    want = oracle.port().size_op(oracle.AND, oracle.Batch(*q_csr),
                                 oracle.Batch(*_gather_rows(d_csr[0], d_csr[1], sample)[:2],
                                              np.concatenate([d_csr[2][int(d_csr[0][r]):int(d_csr[0][r + 1])] for r in sample])))
    got = counts[:, torch.as_tensor(sample, device=dev)].cpu().numpy().view(np.uint32)
    out["c4_ands"] = line(ms_s, in_bytes + nq4 * nd4 * 4, bool(np.array_equal(got, want)),
                          workload=f"ands: {nq4} x {nd4} rows, width {w4}/{a4}, UnitData (multiply + relevance filter)")
    ms_e, res = _timed(lambda: q.ande(d))
    out["c4_ande"] = line(ms_e, in_bytes + (nq4 * nd4 + 1) * 4 + res.total * 8,
                          _elem_parity(res, counts, oracle.AND, q_csr, d_csr, nd4, sample, True),
                          workload=f"ande: {nq4} x {nd4} rows, width {w4}/{a4}, UnitData", elements=res.total)
    del res, counts, d, q
    torch.cuda.empty_cache()

    # config 5: all-pairs overlap + top-16, one GPU's 32 Ki rows against all 256 Ki rows, both kernels
    na5, nb5, k5 = 1 << 15, 1 << 18, 16
    off, ids, _ = sdr_amd.synth_csr(1, 0, nb5, WIDTH, ACTIVE)
    b = SDRBatch.from_csr(off, ids, None, width=WIDTH)
    a = SDRBatch.from_csr(off[: na5 + 1].copy(), ids[: int(off[na5])], None, width=WIDTH)
    pick = np.linspace(0, na5 - 1, 4).astype(np.int64)
    want = oracle.port().size_op(oracle.AND, oracle.Batch(*_gather_rows(off, ids, pick)), oracle.Batch(off, ids, None))
    results = {}
    for algo in ("bitslice", "mfma"):
        ms, (cnt, rows) = _timed(lambda: a.allpairs_topk(b, k5, exclude_offset=0, algo=algo), reps=3)
        gc, gr = cnt.cpu().numpy(), rows.cpu().numpy()
        results[algo] = (gc, gr)
        ok = True
        for n, i in enumerate(pick):
            c = want[n].astype(np.int64)
            cand = np.arange(nb5)
            cand = cand[cand != i]
            order = cand[np.lexsort((cand, -c[cand]))][:k5]
            ok = ok and np.array_equal(gr[i], order) and np.array_equal(gc[i], c[order])
        ops = 2.0 * na5 * nb5 * WIDTH
        entry = dict(ms=ms, pair_ops_per_s=na5 * nb5 / (ms * 1e-3), parity=bool(ok),
                     workload=f"all-pairs overlap + top-{k5}: {na5} x {nb5} rows, width {WIDTH}/{ACTIVE}, self excluded")
        if algo == "mfma":
            entry.update(int8_ops_per_s=ops / (ms * 1e-3), peak_int8_ops_per_s=I8_PEAK_OPS, frac=ops / (ms * 1e-3) / I8_PEAK_OPS)
        out[f"c5_allpairs_{algo}"] = entry
    same = np.array_equal(results["bitslice"][0], results["mfma"][0]) and np.array_equal(results["bitslice"][1], results["mfma"][1])
    out["c5_allpairs_bitslice"]["parity"] = out["c5_allpairs_bitslice"]["parity"] and bool(same)
    out["c5_allpairs_mfma"]["parity"] = out["c5_allpairs_mfma"]["parity"] and bool(same)
    del a, b, results
    torch.cuda.empty_cache()

    # the headline shape with payloads: UnitData ands (product + relevance filter per match) and ArithData dot
    # products, 4096 x 1 Mi rows, 2048/40; the oracle checks 8 queries x 256 rows
    nq, nd = 4096, 1 << 20
    qi, dj = np.sort(rng.choice(nq, 8, replace=False)), np.sort(rng.choice(nd, 256, replace=False))

    def rows_with(csr, rows, vals):
        off, ids, _ = _gather_rows(csr[0], csr[1], rows)
        return off, ids, np.concatenate([vals[int(csr[0][r]):int(csr[0][r + 1])] for r in rows])

    out_buf = torch.empty((nq, nd), dtype=torch.int32, device=dev)
    # the headline shape with narrow counts (every count is at most 40 / 80 here): the count matrix is the traffic
    d_csr = sdr_amd.synth_csr(1, 0, nd, WIDTH, ACTIVE)
    q_csr = sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE)
    d, q = SDRBatch.from_csr(*d_csr, width=WIDTH), SDRBatch.from_csr(*q_csr, width=WIDTH)
    sub_q, sub_d = oracle.Batch(*_gather_rows(q_csr[0], q_csr[1], qi)), oracle.Batch(*_gather_rows(d_csr[0], d_csr[1], dj))
    for name, op, dtype, nb_out in (("ands", oracle.AND, torch.uint8, 1), ("xors", oracle.XOR, torch.uint8, 1), ("ors", oracle.OR, torch.int16, 2)):
        narrow_buf = out_buf.view(dtype)[:, : nd]          # rows stay 4 * nd bytes apart: the leading dimension is the caller's
        ms, _ = _timed(lambda: getattr(q, name)(d, out=narrow_buf, dtype=dtype), reps=5)
        want = oracle.port().size_op(op, sub_q, sub_d)
        got = narrow_buf[torch.as_tensor(qi, device=dev)][:, torch.as_tensor(dj, device=dev)].cpu().numpy()
        got = got.astype(np.uint32) if nb_out == 1 else got.view(np.uint16).astype(np.uint32)
        out[f"{name}_u{8 * nb_out}"] = line(ms, (nd + nq) * (4 * ACTIVE + 4) + nq * nd * nb_out, bool(np.array_equal(got, want)),
                                            workload=f"{name}: {nq} x {nd} rows, width {WIDTH}/{ACTIVE}, EmptyData, uint{8 * nb_out} counts[Q][N]")
    del d, q
    d_csr = sdr_amd.synth_csr(1, 0, nd, WIDTH, ACTIVE, True)
    q_csr = sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE, True)
    d, q = SDRBatch.from_csr(*d_csr, width=WIDTH), SDRBatch.from_csr(*q_csr, width=WIDTH)
    nbytes = (nd + nq) * (8 * ACTIVE + 4) + nq * nd * 4
    sub_q, sub_d = oracle.Batch(*rows_with(q_csr, qi, q_csr[2])), oracle.Batch(*rows_with(d_csr, dj, d_csr[2]))
    for name, op in (("ands", oracle.AND), ("xors", oracle.XOR), ("rms", oracle.RM)):
        ms, _ = _timed(lambda: getattr(q, name)(d, out=out_buf), reps=3)
        want = oracle.port().size_op(op, sub_q, sub_d)
        got = out_buf[torch.as_tensor(qi, device=dev)][:, torch.as_tensor(dj, device=dev)].cpu().numpy().view(np.uint32)
        out[f"unit_{name}_4096"] = line(ms, nbytes, bool(np.array_equal(got, want)),
                                        workload=f"{name}: {nq} x {nd} rows, width {WIDTH}/{ACTIVE}, UnitData on both sides")
    del d, q
    q_val = rng.standard_normal(q_csr[1].size).astype(np.float32)
    d_val = rng.standard_normal(d_csr[1].size).astype(np.float32)
    d = SDRBatch.from_csr(d_csr[0], d_csr[1], d_val, width=WIDTH, kind="arith")
    q = SDRBatch.from_csr(q_csr[0], q_csr[1], q_val, width=WIDTH, kind="arith")
    out_f = out_buf.view(torch.float32)
    ms, _ = _timed(lambda: q.inner(d, out=out_f), reps=3)
    want = oracle.port().inner(oracle.Batch(*rows_with(q_csr, qi, q_val), arith=True), oracle.Batch(*rows_with(d_csr, dj, d_val), arith=True))
    got = out_f[torch.as_tensor(qi, device=dev)][:, torch.as_tensor(dj, device=dev)].cpu().numpy().view(np.uint32)
    out["inner_4096"] = line(ms, nbytes, bool(np.array_equal(got, want.view(np.uint32))),
                             workload=f"inner (sparse dot products): {nq} x {nd} rows, width {WIDTH}/{ACTIVE}, ArithData<float>")
    return out


def _gather_rows(off, ids, rows):
    lens = (off[rows + 1] - off[rows]).astype(np.int64)
    new_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    parts = [ids[int(off[r]):int(off[r + 1])] for r in rows]
    return new_off, (np.concatenate(parts) if parts else np.zeros(0, np.int32)), None


if __name__ == "__main__":
    main()
