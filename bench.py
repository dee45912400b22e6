#!/usr/bin/env python3
"""bench.py -- the LSH similar-pairs hot path on MI355X, BASELINE.json's metric and configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] = the configuration the metric is quoted on --
1M synthetic cells x 30k genes (1% nnz), 1024-bit signatures, findSimilarPairs4 (k=100, threshold 0.2) --
which fits one GPU; with N GPUs the SAME problem is row-sharded (signature shards all-gathered over RCCL,
every rank scans its rows against all columns), i.e. strong scaling.

One step = one pass of the hot path over the synthetic matrix resident in HBM:
    signature projection of this rank's cells -> all-gather of signature shards -> all-pairs Hamming scan with
    the reference's per-cell top-k selection for this rank's rows.
value = unordered cell pairs of the whole problem, N(N-1)/2 (the reference's own accounting,
src/ExpressionMatrixLsh.cpp:220,274), per second of wall time, whole job.

`python bench.py --gpus N` with N > 1 and no launcher environment starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a CHILD process -- before this process has
imported torch or touched a GPU -- relays its output (rank 0's one JSON line) and exits with its code.

Also on the JSON line:
    roofline      the dominant kernel.  1 GPU, 129..2048-bit signatures: fsp4ScanMatrixPinnedKernel / fsp4ScanMatrixWideKernel,
                  the triangle part of the scan as FP4 dot products (0 / 1 operands up to 1024 bits, +-1 above) on the matrix cores -- bound "mfma": 2 x 1024 (2048)
                  flop per (row, column) pair the launcher counted / the kernel's duration (HIP events on the launch stream,
                  recorded inside the library) against the dense FP4 MFMA peak (10 PFLOP/s), with the in-kernel clock and the
                  fraction at that clock beside it; `hbm_view` keeps SURVEY.md 8(d)'s byte model (2*8*W bytes per unordered
                  pair against 8 TB/s; operands are cache / register resident, so that one is NOT bounded by 1).  Other
                  forms (row shards, other widths): the byte model with `valu_frac`, the fraction of the v_xor/v_bcnt
                  instruction roofline.  `traffic` = fabric bytes per launch from the committed PMC digest of THIS
                  configuration (PROFILE_DIGESTS; a digest of another configuration is refused, there is no fallback).
    roofline_projection   the signature projection against its 8(d) HBM bytes, with the digest's traffic and the ratio, and
                          (gather_view) against the L2 -> L1 line rate its first tier is really bound by.
    cpu_baseline  the CPU oracle's literal findSimilarPairs4 (oracle/, kind "port": the reference itself cannot
                  be built in this image) on the first cells of the same signatures, one thread, ~15-20 s.
Before timing AND after the last timed step, the GPU result is checked against the oracle bit for bit: the signatures of
64 cells and >= 10^4 SimilarPairs rows x all columns at 1M cells (the whole result for configs[1]), the
oracle's rows spread over the host's threads -- a run whose check fails prints no number.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_FP4_PEAK_TFLOPS = 10000.0   # MI355X_MICROARCH.md: FP4/FP6 block-scaled MFMA, ~10 PFLOP/s dense
# v_xor_b32 / v_bcnt_u32_b32 issue one wave64 instruction per 4 clocks per SIMD on gfx950 (measured:
# profiles/r01_pmc_1Mcells_scan_projection.json, SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles): 16 lanes/clk.
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9     # 256 CUs x 4 SIMD x 16 lanes/clk x 2.4 GHz


# The PMC digests `traffic` may come from (tools/profile_bench.sh -> tools/profile_digest.py), one per workload; each names
# the configuration it was taken on and is used for that configuration only (tests/test_bench_plumbing_cpu.py).
PROFILE_DIGESTS = {
    "fsp4": "r06_pmc_bench_1Mcells_1gpu.json",
    "fsp5": "r05_pmc_bench_fsp5_1Mcells_2048bit.json",
    "chain": "r06_pmc_bench_chain_1Mcells.json",
}
def usable_cpus():
    """The host threads worth starting: os.cpu_count(), or the CPU quota of the control group when that is smaller (the GPU boxes
    of the pool show 256 hardware threads under a quota of 16 CPUs: threads beyond it only get the process throttled)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


TRAFFIC_NOTE = "FETCH_SIZE x 2 for 16-byte loads + WRITE_SIZE, bytes per launch of the profiled run"


def profile_query(workload, cells, lsh_count, k, n_gpus=1, genes=None, slice_length=None, bucket_overflow=None):
    """The (key, value) pairs a digest's `config` must hold to stand for this run."""
    query = {"cells": cells, "lsh_count": lsh_count, "k": k, "n_gpus": n_gpus}
    if workload == "fsp4":
        query["genes"] = genes
    if workload == "fsp5":
        query["slice_length"] = slice_length
        query["bucket_overflow"] = bucket_overflow
    return query


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain command line: the N ranks are started by torch.distributed.run as a child of this
    process, which has not imported torch and never touches a GPU (a GPU-initialised process must not exec, and does not need
    to: it waits).  The child's stdout/stderr are this process's own, its exit code is returned."""
    import socket
    import subprocess
    # Under a profiler this hop would be a launcher between the profiler's preloaded library (which may have initialised the
    # GPU) and the ranks: refuse, the profiled command is `... -- python3 -m torch.distributed.run ... bench.py --gpus N`.
    preload = " ".join(os.environ.get(name, "") for name in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
    if "rocprof" in preload.lower():
        raise SystemExit("bench.py --gpus %d under a profiler: start the ranks with torch.distributed.run yourself "
                         "(rocprofv3 ... -- python3 -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, args.gpus, args.gpus))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    code = 1
    for attempt in range(3):                # (the port is free when it is picked, not necessarily when the ranks bind it)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        done = subprocess.run(command, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(done.stderr)
        code = done.returncode
        if code == 0 or "address already in use" not in done.stderr.lower():
            break
    return code


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--cells", type=int, default=1000000)
    p.add_argument("--genes", type=int, default=30000)
    p.add_argument("--density", type=float, default=0.01)
    p.add_argument("--lsh-count", type=int, default=1024)
    p.add_argument("--k", type=int, default=100)
    p.add_argument("--threshold", type=float, default=0.2)
    p.add_argument("--seed", type=int, default=231)
    p.add_argument("--cpu-baseline-cells", type=int, default=50000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--check-rows", type=int, default=10240,
                   help="SimilarPairs rows (x all columns) compared with the oracle before and after the timed steps "
                        "(SURVEY.md 8(d): >= 10^4 at 1M cells); 0 = every row this rank owns")
    p.add_argument("--check-threads", type=int, default=0, help="host threads of the oracle in the parity gates (0: all, at most 64)")
    p.add_argument("--fsp5-check-cells", type=int, default=2048)
    p.add_argument("--no-extra", action="store_true", help="skip the configs[1] block")
    p.add_argument("--workload", choices=["fsp4", "fsp5", "chain"], default="fsp4",
                   help="fsp4 (default): the headline, BASELINE configs[2]; fsp5: configs[3], bucketed findSimilarPairs5 at 2048 "
                        "bits; chain: configs[4], findSimilarPairs4 -> createCellGraph -> label propagation.  fsp5 and chain "
                        "print secondary lines (1 GPU), never the headline")
    p.add_argument("--slice-length", type=int, default=20)
    p.add_argument("--bucket-overflow", type=int, default=1000)
    p.add_argument("--graph-k", type=int, default=20)
    p.add_argument("--no-check", action="store_true",
                   help="diagnostic runs only (EM2_MATRIX_DIAG makes results wrong on purpose): skip the parity gates; "
                        "the line says so and is not a measurement")
    return p.parse_args()


class Watchdog:
    """Bounded time for every stage of a multi-rank run: a collective that one rank never enters (or a kernel that never
    ends) would otherwise hold the whole job until the driver's own limit.  arm(name, seconds) starts the clock of a stage,
    disarm() stops it; a stage that overruns ends THIS process (os._exit: the GPU and the communicator may be wedged, nothing
    of this process can be trusted to unwind).  Every rank runs the same watchdog with the same limits, so on a hung
    collective all ranks leave within a second of each other.  on_expiry, when set (rank 0 with a finished first leg),
    prints the line of what was measured before and turns the exit code into 0."""

    def __init__(self, rank):
        import threading
        self.rank = rank
        self.deadline = None
        self.stage = ""
        self.on_expiry = None
        self.exit_code = 1
        self.lock = threading.Lock()
        thread = threading.Thread(target=self._run, daemon=True)
        thread.start()

    def arm(self, stage, seconds):
        seconds = float(os.environ.get("EM2_BENCH_STAGE_LIMIT", seconds))          # (tests shorten the limits)
        with self.lock:
            self.stage, self.deadline = stage, time.monotonic() + seconds

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                expired = self.deadline is not None and time.monotonic() > self.deadline
                stage, handler, code = self.stage, self.on_expiry, self.exit_code
            if expired:
                print("[bench] rank %d: stage '%s' exceeded its time limit; leaving" % (self.rank, stage), file=sys.stderr, flush=True)
                if handler is not None:
                    try:
                        handler(stage)
                    except Exception as error:            # noqa: BLE001 -- nothing may stop the exit
                        print("[bench] rank %d: %s" % (self.rank, error), file=sys.stderr, flush=True)
                        code = 1
                sys.stdout.flush()
                os._exit(code)


def device_state(device_index):
    """Clock / power / temperature as rocm-smi reports them for this GPU, or None: recorded next to the measurement so that a
    reader can tell a throttled box from a regression (the scan kernel's own in-kernel clock is in roofline.clock_ghz)."""
    import subprocess
    # (under a profiler whose preloaded library initialises the GPU in every child, rocm-smi -- an `env python3` script -- would
    # be an exec from a GPU-initialised process, which the GPU pool refuses: the profile has its own clock counters)
    if any("rocprof" in os.environ.get(name, "").lower() for name in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) \
            or any(name.startswith("ROCPROF") for name in os.environ):
        return None
    try:
        out = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(out.stdout).values()))
        keep = {}
        for key, value in card.items():
            low = key.lower()
            if any(word in low for word in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor edge)",
                                            "performance level")):
                keep[key] = value
        return keep or None
    except Exception:                                     # noqa: BLE001 -- a diagnostic, never a reason to fail
        return None


def profiled_traffic(file_name, config, kernel_prefixes):
    """Fabric bytes per launch (FETCH_SIZE x 2 for 16-byte loads + WRITE_SIZE, as tools/profile_digest.py corrects them) of the
    kernels whose names start with one of kernel_prefixes, from a committed digest of tools/profile_bench.sh -- when that
    digest is of THIS configuration (its `config` holds every (key, value) of `config`).  (bytes per launch summed over the
    kernels, {kernel: launches per pass}) or (None, None).  PMC passes are separate runs: the figure is per launch of the
    profiled run, not of this one."""
    path = os.path.join(ROOT, "profiles", file_name)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        digest = json.load(f)
    if any(digest.get("config", {}).get(key) != value for key, value in config.items()):
        return None, None
    total, launches = 0.0, {}
    for name, entry in digest.get("kernels", {}).items():
        if any(name.startswith(prefix) for prefix in kernel_prefixes) and "fabric_fetch_bytes_per_launch" in entry:
            total += entry["fabric_fetch_bytes_per_launch"] + entry.get("fabric_write_bytes_per_launch", 0.0)
            launches[name] = entry["launches_per_pass"]
    return (total, launches) if launches else (None, None)


def sample_ranges(owned, rows_wanted, places=16):
    """Row ranges totalling about rows_wanted rows out of the owned [begin, end) ranges: one contiguous owned range is sampled in
    `places` evenly spread spans (the first and the last rows included), a list of 64-row blocks by evenly spread whole blocks.
    rows_wanted <= 0 or >= what is owned: everything."""
    total = sum(e - b for b, e in owned)
    if rows_wanted <= 0 or rows_wanted >= total:
        return list(owned)
    if len(owned) == 1:
        begin, end = owned[0]
        span = max(1, rows_wanted // places)
        starts = sorted(set(begin + (end - begin - span) * i // (places - 1) for i in range(places)))
        out = []
        for s0 in starts:                                    # (merge overlapping spans)
            if out and s0 < out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], s0 + span))
            else:
                out.append((s0, s0 + span))
        return out
    count = max(1, min(len(owned), -(-rows_wanted // max(1, owned[0][1] - owned[0][0]))))
    return [owned[i] for i in sorted(set((len(owned) - 1) * i // max(1, count - 1) for i in range(count)))]


def oracle_rows_parallel(oracle, sig_host, L, k, thr, ranges, threads=0):
    """The oracle's per-cell findSimilarPairs4 contract (rows against all columns) for the row ranges, spread over host threads
    (the ctypes call releases the GIL; rows are independent).  Returns [(begin, end, cell, sim, used)] in pieces."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, min(64, usable_cpus()))
    total = sum(e - b for b, e in ranges)
    per = max(1, -(-total // (4 * threads)))
    pieces = [(s0, min(e, s0 + per)) for b, e in ranges for s0 in range(b, e, per)]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        results = list(pool.map(lambda piece: oracle.find_similar_pairs4_rows(sig_host, L, k, thr, piece[0], piece[1]), pieces))
    return [(b, e) + tuple(r) for (b, e), r in zip(pieces, results)]


def parity_gate(pipe, oracle, synthetic, sig_host, toc, data, vectors_host, check_rows, threads=0):
    """The result of the pipeline's LAST step against the CPU oracle, bit for bit: the signatures of a few of this
    rank's cells and sampled SimilarPairs rows this rank owns against all columns (check_rows of them; 0 = all).
    Exits on a difference.  Returns (signature cells checked, rows checked)."""
    C, G, L, k, thr = pipe.cell_count, pipe.gene_count, pipe.lsh_count, pipe.k, pipe.thr
    if not pipe.rows:
        return 0, 0
    sample = min(64, pipe.rows)
    t_h, g_h, c_h = synthetic.csr_to_host(toc[:sample + 1], data[:int(toc[sample].item())])
    expect = oracle.compute_signatures(t_h, g_h, c_h, G, vectors_host, L)
    got = sig_host[pipe.row_begin:pipe.row_begin + sample]
    if not np.array_equal(expect, got):
        raise SystemExit("PARITY FAILURE: signatures differ from the oracle")
    ranges = sample_ranges(pipe.owned_ranges(), check_rows)
    fetched = {(b, e): pipe.results_for(b, e) for b, e in ranges}
    rows_checked = 0
    for begin, end, cell, sim, oused in oracle_rows_parallel(oracle, sig_host, L, k, thr, ranges, threads):
        r_begin, (pairs, used) = next((b, fetched[(b, e)]) for b, e in ranges if b <= begin and end <= e)
        lo, hi = begin - r_begin, end - r_begin
        ok = (np.array_equal(used[lo:hi], oused) and np.array_equal(pairs["cell"][lo:hi], cell) and
              np.array_equal(pairs["similarity"][lo:hi].view(np.uint32), sim.view(np.uint32)))
        if not ok:
            raise SystemExit("PARITY FAILURE: SimilarPairs rows %d..%d differ from the oracle" % (begin, end))
        rows_checked += end - begin
    return sample, rows_checked


def rows_form_block(args, capi, pipe, oracle, sig_host, torch, threads, shards=((2, 1), (4, 1), (8, 3)), repeats=3, oracle_rows=1024):
    """North_star's partitioning across GPUs, measured on ONE: this GPU plays rank r of P -- the rank's contiguous rows against
    all columns (em2_dev_find_similar_pairs4 with a row range: the rows form on the matrix cores) on the signatures of the
    headline run.  EVERY row of each shard must equal the headline's result (the symmetric form, which the gates held against
    the oracle), and oracle_rows rows per shard are compared with the oracle directly.  Scan only: a rank's projection is 1/P of
    the headline's, its all_gather is not played."""
    C, L, k, thr = pipe.cell_count, pipe.lsh_count, pipe.k, pipe.thr
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    for world, rank in shards:
        per = -(-C // world)
        begin, end = min(C, rank * per), min(C, (rank + 1) * per)
        rows = end - begin
        ws_bytes = capi.dev_find_similar_pairs4_workspace(C, rows, L, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pipe.device)
        pairs = torch.zeros((rows, k, 2), dtype=torch.int32, device=pipe.device)
        used = torch.zeros(rows, dtype=torch.int32, device=pipe.device)
        wall, kernel = [], []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.dev_find_similar_pairs4(pipe.full_sig.data_ptr(), C, begin, end, L, k, thr, pairs.data_ptr(), used.data_ptr(),
                                         ws.data_ptr(), ws_bytes, stream)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
            kernel.append(capi.dev_find_similar_pairs4_last_launch()["matrix_kernel_ms"])
        capi.dev_find_similar_pairs4_status(ws.data_ptr(), rows, k, stream)
        launch = capi.dev_find_similar_pairs4_last_launch()
        same = bool(torch.equal(pairs, pipe.pairs[begin:end]) and torch.equal(used, pipe.used[begin:end]))
        if not same:
            raise SystemExit("PARITY FAILURE: rows form, rank %d of %d: rows differ from the symmetric form's" % (rank, world))
        ranges = sample_ranges([(begin, end)], oracle_rows)
        host_pairs = pairs.cpu().numpy().view(np.uint32)
        host_used = used.cpu().numpy().view(np.uint32)
        checked = 0
        for b, e, cell, sim, oused in oracle_rows_parallel(oracle, sig_host, L, k, thr, ranges, threads):
            lo, hi = b - begin, e - begin
            if not (np.array_equal(host_used[lo:hi], oused) and np.array_equal(host_pairs[lo:hi, :, 0], cell) and
                    np.array_equal(host_pairs[lo:hi, :, 1], sim.view(np.uint32))):
                raise SystemExit("PARITY FAILURE: rows form, rank %d of %d: rows %d..%d differ from the oracle" % (rank, world, b, e))
            checked += e - b
        flops = launch["matrix_pairs"] * 2.0 * (2048.0 if L > 1024 else 1024.0)
        best = min(kernel)
        out.append({"ranks": world, "rank_played": rank, "rows": [begin, end], "scan_form": launch["form"],
                    "scan_ms": min(wall), "kernel_ms": best, "clock_ghz": launch["matrix_clock_ghz"] or None,
                    "ordered_pairs_per_s": rows * float(C) / (min(wall) * 1e-3),
                    "frac_of_fp4_peak": flops / (best * 1e-3) / 1e12 / MFMA_FP4_PEAK_TFLOPS if best > 0 else None,
                    "rows_equal_to_the_symmetric_result": rows, "rows_against_the_oracle": checked})
        del ws, pairs, used
        torch.cuda.empty_cache()
    return {"what": "one GPU playing rank r of P of the row-shard form (SURVEY 8e: contiguous rows x all columns), scan only; "
                    "a real node adds the rank's projection (1/P of the headline's) and one all_gather of the signatures",
            "shards": out}


def non_integer_projection_block(pipe, oracle, synthetic, torch, toc, data, vectors_host, repeats=3):
    """The projection of the same matrix with every count x 1.5 (normalised expression matrices are not integers): the 16-bit
    tier then sums its products in floating point instead of exactly in integers.  Signatures of 64 cells against the oracle."""
    G, L = pipe.gene_count, pipe.lsh_count
    scaled = data.clone()
    scaled.view(torch.float32)[1::2] *= 1.5          # em2_count = {gene: uint32, count: float32}
    original = pipe.data
    pipe.data = scaled
    try:
        times = []
        for _ in range(repeats + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipe.project()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        tier = pipe.projection_tier()
        sample = min(64, pipe.rows)
        t_h, g_h, c_h = synthetic.csr_to_host(toc[:sample + 1], scaled[:int(toc[sample].item())])
        expect = oracle.compute_signatures(t_h, g_h, c_h, G, vectors_host, L)
        got = pipe.local_sig[:sample].cpu().numpy().view(np.uint64).reshape(sample, -1)
        if not np.array_equal(expect, got):
            raise SystemExit("PARITY FAILURE: signatures of the non-integer matrix differ from the oracle")
    finally:
        pipe.data = original
        pipe.project()                                # (the pipeline's signatures are the integer matrix's again)
        torch.cuda.synchronize()
    return {"what": "em2_dev_compute_signatures on the headline's matrix with every count x 1.5", "tier": tier,
            "ms": sum(times[1:]) / repeats, "signature_cells_against_the_oracle": sample}


def facade_block(args, capi, synthetic, torch, toc, data, reference_pairs, reference_used):
    """The user-visible call: ExpressionMatrix.findSimilarPairs4(similarPairsName=...) on a data directory in the reference's
    formats (src/PythonModule.cpp:802-824) -- mmap'd CellExpressionCounts in, SimilarPairs-<name>-{Info,Pairs,CellInfo} out.
    Wall time of the call (second of two), and its stages from the library's own timers (a third call with EM2_TIMING=1, whose
    stages are synchronised: they do not overlap as they do in the timed call).  The reference's accounting separates the store
    as well ("excluding time to store similarities", src/ExpressionMatrixLsh.cpp:270-285).  The files' pairs must equal the
    device pipeline's result.  Never `value`: the inputs are not resident in HBM here."""
    import re
    import shutil
    import tempfile
    from expressionmatrix2_amd import ExpressionMatrix, files
    C, G, L, k, thr = args.cells, args.genes, args.lsh_count, args.k, args.threshold
    t_h, g_h, c_h = synthetic.csr_to_host(toc, data)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 8 * (1 << 30) else tempfile.gettempdir()
    directory = tempfile.mkdtemp(prefix="em2bench", dir=base)
    try:
        t0 = time.perf_counter()
        files.create_directory(directory, G, t_h, capi.make_counts(g_h, c_h))
        create_s = time.perf_counter() - t0
        del t_h, g_h, c_h
        e = ExpressionMatrix(directory)
        wall = []
        for _ in range(2):
            t0 = time.perf_counter()
            e.findSimilarPairs4(similarPairsName="Bench", k=k, similarityThreshold=thr, lshCount=L, seed=args.seed)
            wall.append(time.perf_counter() - t0)
        _, pairs, used = files.read_similar_pairs(directory, "Bench")
        same = bool(np.array_equal(used, reference_used) and np.array_equal(pairs["cell"], reference_pairs["cell"]) and
                    np.array_equal(pairs["similarity"].view(np.uint32), reference_pairs["similarity"].view(np.uint32)))
        if not same:
            raise SystemExit("PARITY FAILURE: SimilarPairs written by ExpressionMatrix.findSimilarPairs4 differ from the device pipeline's")
        # the split: the library's timers write to file descriptor 2
        sys.stderr.flush()
        saved = os.dup(2)
        stages = {}
        with tempfile.TemporaryFile(mode="w+b") as capture:
            os.dup2(capture.fileno(), 2)
            os.environ["EM2_TIMING"] = "1"
            try:
                t0 = time.perf_counter()
                e.findSimilarPairs4(similarPairsName="Bench", k=k, similarityThreshold=thr, lshCount=L, seed=args.seed)
                staged_wall = time.perf_counter() - t0
            finally:
                os.environ.pop("EM2_TIMING", None)
                os.dup2(saved, 2)
                os.close(saved)
            capture.seek(0)
            for line in capture.read().decode(errors="replace").splitlines():
                m = re.match(r"\[em2 timing\]\s+(.*?):\s+(.*) ([0-9.]+) ms$", line)
                if m and "symmetric scan" not in line and "matrix kernel" not in line:
                    stages["%s: %s" % (m.group(1).strip(), m.group(2).strip())] = float(m.group(3))
        return {"what": "ExpressionMatrix.findSimilarPairs4 on a data directory (mmap'd CSR in, SimilarPairs files out), %d cells x %d genes; "
                        "PCIe and file I/O included -- beside the headline, never `value`" % (C, G),
                "directory_on": base, "create_directory_s": create_s, "call_s": wall[1], "first_call_s": wall[0],
                "call_with_synchronised_stage_timers_s": staged_wall, "stages_ms": stages,
                "rows_equal_to_the_device_pipeline": int(C)}
    finally:
        shutil.rmtree(directory, ignore_errors=True)


def clustered_regime_block(args, capi, oracle, device, torch, threads, regimes=((64, 0.15), (8, 0.05)), repeats=3, oracle_rows=1024):
    """The scan alone on SURVEY.md 8(d)'s scan-only input -- cluster centres, every bit flipped with probability `flip`, cells
    dealt to the clusters at random -- at the headline's size: signatures whose same-cluster similarity (0.70 at flip 0.15) lies
    ABOVE the reference's default graph threshold, where the headline's projected signatures (similarity ~ 0.10 inside a cluster)
    lie below it.  Per regime: em2_dev_find_similar_pairs4 over all cells (best of `repeats` after one untimed call), the matrix
    kernel's own time by HIP events inside the library and its fraction of the dense FP4 peak, the deferred candidates against
    the pool that holds them, and `oracle_rows` rows in 16 places of the last call's result against the oracle.  (VERDICT r5:
    the scan's rate is a property of the data; this is the data on which it is lowest.)"""
    C, L, k, thr = args.cells, args.lsh_count, args.k, args.threshold
    stream = torch.cuda.current_stream().cuda_stream
    pairs = torch.zeros((C, k, 2), dtype=torch.int32, device=device)
    used = torch.zeros(C, dtype=torch.int32, device=device)
    ws_bytes = capi.dev_find_similar_pairs4_workspace(C, C, L, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    out = {"note": "scan only (no projection): synthetic_signatures(clusters, flip), threshold %g, k = %d; kernel_ms / clock from the "
                   "library's own events and counters (em2_dev_find_similar_pairs4_last_launch); frac = 2 x %d flop per pair on the "
                   "matrix cores / kernel_ms / the dense FP4 peak" % (thr, k, 2048 if L > 1024 else 1024), "regimes": []}
    for clusters, flip in regimes:
        sig = synthetic_signatures(torch, C, L, device, clusters, flip, 4321)
        times, kernel = [], []
        for i in range(repeats + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.dev_find_similar_pairs4(sig.data_ptr(), C, 0, C, L, k, thr, pairs.data_ptr(), used.data_ptr(), ws.data_ptr(), ws_bytes, stream)
            torch.cuda.synchronize()
            if i:
                times.append(time.perf_counter() - t0)
                kernel.append(capi.dev_find_similar_pairs4_last_launch()["matrix_kernel_ms"])
        capi.dev_find_similar_pairs4_status(ws.data_ptr(), C, k, stream)
        launch = capi.dev_find_similar_pairs4_last_launch()
        entry = {"clusters": clusters, "flip": flip, "scan_ms": min(times) * 1e3, "unordered_pairs_per_s": C * (C - 1) / 2.0 / min(times),
                 "scan_form": launch["form"], "kernel_ms": min(kernel), "clock_ghz": launch["matrix_clock_ghz"] or None,
                 "deferred_candidates": launch["inbox_entries"], "deferred_candidates_per_cell": launch["inbox_entries"] / C,
                 "deferred_pool_entries_per_cell": 1024}
        if launch["matrix_pairs"] > 0 and min(kernel) > 0:
            flops = launch["matrix_pairs"] * 2.0 * (2048.0 if L > 1024 else 1024.0)
            entry["roofline"] = {"bound": "mfma", "achieved": flops / (min(kernel) * 1e-3) / 1e12, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": flops / (min(kernel) * 1e-3) / 1e12 / MFMA_FP4_PEAK_TFLOPS}
        if not args.no_check:
            sig_host = sig.cpu().numpy().view(np.uint64)
            host_pairs = pairs.cpu().numpy().view(np.uint32)
            host_used = used.cpu().numpy().view(np.uint32)
            checked = 0
            for b, e, cell, sim, oused in oracle_rows_parallel(oracle, sig_host, L, k, thr, sample_ranges([(0, C)], oracle_rows), threads):
                if not (np.array_equal(host_used[b:e], oused) and np.array_equal(host_pairs[b:e, :, 0], cell) and
                        np.array_equal(host_pairs[b:e, :, 1], sim.view(np.uint32))):
                    raise SystemExit("PARITY FAILURE: clustered regime %d / %g, rows [%d, %d)" % (clusters, flip, b, e))
                checked += e - b
            entry["rows_equal_to_the_oracle"] = checked
            del sig_host, host_pairs, host_used
        out["regimes"].append(entry)
        del sig
    return out


def small_config(args, capi, sharded, synthetic, oracle, device, torch, cells=100000, genes=20000, steps=10, warmup=2):
    """BASELINE configs[1] on one GPU: ms per step and pairs/s, gated like the headline."""
    L, k, thr = args.lsh_count, args.k, args.threshold
    pipe = sharded.DevicePipeline(cells, genes, L, k, thr, world_size=1, rank=0, dist=None, device=device)
    toc, data = synthetic.expression_shard(0, cells, genes, density=args.density, device=device)
    vectors_host = capi.lsh_generate_vectors(genes, L, args.seed)
    pipe.set_inputs(toc, data, torch.from_numpy(vectors_host).to(device))
    pipe.step()
    torch.cuda.synchronize()
    sig_host = pipe.full_sig[:cells].cpu().numpy().view(np.uint64)
    # "the full CPU result for config B" -- every row against every column, before and after the timed steps
    before = parity_gate(pipe, oracle, synthetic, sig_host, toc, data, vectors_host, 0, args.check_threads)
    for _ in range(warmup):
        pipe.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pipe.check()
    after = parity_gate(pipe, oracle, synthetic, pipe.full_sig[:cells].cpu().numpy().view(np.uint64), toc, data, vectors_host,
                        0, args.check_threads)
    launch = capi.dev_find_similar_pairs4_last_launch()
    return {"workload": "%d cells x %d genes (%.3g nnz/cell), %d-bit signatures, findSimilarPairs4 k=%d threshold=%g, 1 GPU"
                        % (cells, genes, args.density * genes, L, k, thr),
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "value": cells * (cells - 1) / 2.0 * steps / elapsed, "unit": "pairs/s", "scan_form": launch["form"],
            "parity_check": {"signature_cells": before[0], "fsp4_rows": before[1], "after_timing_rows": after[1]}}


def synthetic_signatures(torch, cells, lsh_count, device, cluster_count=64, flip=0.15, seed=4321, chunk=65536):
    """SURVEY.md 8(d), scan-only variant: signature = its cluster's centre (random bits) with every bit flipped with
    probability `flip`; packed MSB-first into uint64 words like src/BitSet.hpp:48-62.  Returns int64 [cells, words]."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    words = (lsh_count - 1) // 64 + 1
    centres = torch.randint(0, 2, (cluster_count, words * 64), generator=gen, device=device, dtype=torch.int8)
    centres[:, lsh_count:] = 0
    weights = (1 << torch.arange(63, -1, -1, device=device, dtype=torch.int64))
    out = torch.empty((cells, words), dtype=torch.int64, device=device)
    for begin in range(0, cells, chunk):
        end = min(cells, begin + chunk)
        cluster = torch.randint(0, cluster_count, (end - begin,), generator=gen, device=device)
        flips = (torch.rand((end - begin, words * 64), generator=gen, device=device) < flip).to(torch.int8)
        flips[:, lsh_count:] = 0
        bits = (centres[cluster] ^ flips).to(torch.int64).view(end - begin, words, 64)
        out[begin:end] = (bits * weights).sum(dim=2)
    return out


def bench_fsp5(args, capi, oracle, device, torch):
    """BASELINE configs[3] on one GPU: bucketed findSimilarPairs5, 2048-bit signatures, lshSliceLength 20, bucketOverflow
    1000, k=100.  One step = em2_dev_find_similar_pairs5 over all cells (tables + candidate filter + selection)."""
    C, L, k, thr, q = args.cells, 2048 if args.lsh_count == 1024 else args.lsh_count, args.k, args.threshold, args.slice_length
    W = capi.word_count(L)
    sig = synthetic_signatures(torch, C, L, device)
    pairs = torch.zeros((C, k, 2), dtype=torch.int32, device=device)
    used = torch.zeros(C, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        capi.dev_find_similar_pairs5(sig.data_ptr(), C, 0, C, L, k, thr, q, args.bucket_overflow, pairs.data_ptr(), used.data_ptr(), stream)

    step()
    torch.cuda.synchronize()
    check = {"skipped": "--no-check"}
    if not args.no_check:
        # sampled cells in sixteen places of the id range against the oracle (one build of its tables over all cells)
        sig_host = sig.cpu().numpy().view(np.uint64)
        places = 16
        span = max(1, min(C, args.fsp5_check_cells) // places)
        listed = np.unique(np.concatenate([np.arange(b, b + span) for b in
                                           [(C - span) * i // (places - 1) for i in range(places)]]).clip(0, C - 1)).astype(np.uint32)
        cell, sim, oused = oracle.find_similar_pairs5_cells(sig_host, L, k, thr, q, args.bucket_overflow, listed)
        index = torch.from_numpy(listed.astype(np.int64)).to(device)
        got = pairs[index].cpu().numpy().view(np.uint32)
        ok = (np.array_equal(used[index].cpu().numpy().view(np.uint32), oused) and np.array_equal(got[:, :, 0], cell) and
              np.array_equal(got[:, :, 1], sim.view(np.uint32)))
        if not ok:
            raise SystemExit("PARITY FAILURE: findSimilarPairs5 differs from the oracle on the sampled cells")
        check = {"fsp5_cells": int(len(listed)), "places": places}
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    filter_ms = select_ms = 0.0
    for _ in range(args.steps):
        step()
        info = capi.dev_find_similar_pairs5_last_launch()
        filter_ms += info["filter_ms"]
        select_ms += info["select_ms"]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    info = capi.dev_find_similar_pairs5_last_launch()
    filter_ms /= args.steps
    select_ms /= args.steps
    # candidates x 8*W bytes of signature gathers + 4*N*sliceCount bytes of tables.  Candidates = what the
    # filter reads: the DISTINCT ids of every cell's union of buckets, counted by the library (the ids gathered with duplicates
    # are reported beside it).
    distinct = info["distinct_candidates"] if info["distinct_candidates"] >= 0 else info["gathered_candidates"]
    algorithmic = distinct * 8.0 * W + 4.0 * C * info["slice_count"]
    achieved = algorithmic / (filter_ms * 1e-3) / 1e9 if filter_ms > 0 else 0.0
    per_launch, launches = profiled_traffic(PROFILE_DIGESTS["fsp5"], profile_query("fsp5", C, L, k, slice_length=q, bucket_overflow=args.bucket_overflow),
                                            ("filterWideKernel", "filterCooperativeKernel"))
    traffic = per_launch * info["batches"] if per_launch is not None else None
    traffic_source = ("profiles/%s (%s of the filter, x the batches of one call)" % (PROFILE_DIGESTS["fsp5"], TRAFFIC_NOTE)) if traffic is not None else None
    return {
        "metric": "cells/sec through findSimilarPairs5 (bucketed LSH, tables + candidate filter + selection)",
        "value": C * args.steps / elapsed, "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 popcount", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %d synthetic cells, %d-bit signatures, findSimilarPairs5 lshSliceLength=%d "
                               "bucketOverflow=%d k=%d threshold=%g, 1 GPU" % (C, L, q, args.bucket_overflow, k, thr),
                   "cells": C, "lsh_count": L, "k": k, "slice_length": q, "bucket_overflow": args.bucket_overflow,
                   "slices": info["slice_count"], "batches": info["batches"]},
        "phases_ms": {"candidate_filter": filter_ms, "selection": select_ms,
                      "tables_and_candidate_unions": elapsed / args.steps * 1e3 - filter_ms - select_ms},
        # The bound that binds the filter since the grouped visiting order (round 4): its vector ALUs.  Essential work = one v_xor_b32
        # and one v_bcnt_u32_b32 per 32 bits of every distinct candidate's signature (SURVEY 8(d)'s own VALU ceiling for
        # xor/popcount: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz lane-operations per second, the issue rate measured in
        # profiles/r01_ubench_valu_xor_bcnt.txt); `bound_unit` "lane-op/s".  The HBM byte model of 8(d) rides along (hbm_view).
        "roofline": {"kernel": ("filterWideKernel<%s> (all batches)" % ("4, 4" if W == 32 else "2, 16" if W == 64 else "T, LPC")) if W % 2 == 0 and W <= 64
                               else "filterCooperativeKernel (all batches)", "kernel_ms": filter_ms, "bound": "valu",
                     "achieved": distinct * 2.0 * W * 2.0 / (filter_ms * 1e-3) / 1e12 if filter_ms > 0 else 0.0,
                     "peak": VALU_LANE_OPS_PER_S / 1e12, "unit": "T lane-op/s",
                     "frac": distinct * 2.0 * W * 2.0 / (filter_ms * 1e-3) / VALU_LANE_OPS_PER_S if filter_ms > 0 else 0.0,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "essential_lane_ops": distinct * 2.0 * W * 2.0, "distinct_candidates": distinct, "gathered_candidates": info["gathered_candidates"],
                     "hbm_view": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes": algorithmic},
                     "note": "essential lane-ops = distinct candidates x (2 x W 32-bit words) x 2 instructions (v_xor_b32 + v_bcnt_u32_b32); "
                             "candidates = the distinct ids of every cell's union of buckets (counted by the library); kernel_ms = HIP events "
                             "around the filter kernels on the launch stream.  hbm_view = SURVEY.md 8(d)'s byte model (candidates x 8*W bytes of "
                             "gathers + 4*N*sliceCount of tables) against the HBM peak: since the filter visits cells grouped by neighbourhood "
                             "those bytes come from the L2s (`traffic` is what crosses the fabric), its frac exceeds 1 and bounds nothing.  "
                             "The filter executes about 3.4 x the essential instructions (lane sums, rank bookkeeping, address arithmetic)"},
        # the other stages against what they must at least move (bytes over the HBM peak): none of them is near it -- the union
        # is bound by its own instructions (VALU busy 84 %), the selection by LDS latency at 4-5 waves per CU
        "stage_bounds": {
            "candidate_unions": {"least_bytes": info["gathered_candidates"] * 4.0 + distinct * 4.0,
                                 "least_ms_at_hbm_peak": (info["gathered_candidates"] * 4.0 + distinct * 4.0) / (HBM_PEAK_GBS * 1e9) * 1e3,
                                 "note": "every gathered bucket member read once (4 B), every distinct candidate written once (4 B)"},
            "selection": {"least_bytes": distinct * 8.0 + C * k * 8.0, "least_ms_at_hbm_peak": (distinct * 8.0 + C * k * 8.0) / (HBM_PEAK_GBS * 1e9) * 1e3,
                          "note": "every list entry read once (8 B), k pairs per cell written (8 B)"},
            "tables": {"least_bytes": C * info["slice_count"] * (12.0 + 12.0) * 3.0, "least_ms_at_hbm_peak": C * info["slice_count"] * 72.0 / (HBM_PEAK_GBS * 1e9) * 1e3,
                       "note": "one key (8 B) + cell id (4 B) per (slice, cell), read and written by each of ~3 radix passes"}},
        "parity_check": check,
    }


def chain_roofline(capi, args, times, edges, iterations, cells, per_vertex):
    """The chain's dominant kernel (the scan of its findSimilarPairs4, against the FP4 MFMA peak as on the headline) and the
    least the two consumers must move, against the HBM peak -- with what binds them instead."""
    launch = capi.dev_find_similar_pairs4_last_launch()
    contraction = 2048.0 if args.lsh_count > 1024 else 1024.0
    flops = launch["matrix_pairs"] * 2.0 * contraction
    tflops = flops / (launch["matrix_kernel_ms"] * 1e-3) / 1e12 if launch["matrix_kernel_ms"] > 0 else 0.0
    label_ms = times["labelPropagationClustering"] / args.steps * 1e3
    graph_ms = times["createCellGraph"] / args.steps * 1e3
    # label propagation: an iteration reads one 32-byte record per (vertex, neighbour) -- 2 x edges of them -- and a record's 128-byte
    # line is what a random gather moves; the turns of an iteration are ordered (a vertex waits for its neighbours at smaller
    # positions of the shuffle): the longest such chain, not bandwidth, sets the pace
    record_bytes = 2.0 * edges * 32.0 * max(1, iterations)
    line_bytes = 2.0 * edges * 128.0 * max(1, iterations)
    # createCellGraph: the first <= graph_k pairs of every cell (8 B each), the selections written and read twice (count, write),
    # and for every selected neighbour with a smaller id a gather of its list (graph_k x 4 B in one or two 128-byte lines)
    graph_bytes = cells * per_vertex * 8.0 + 3.0 * cells * per_vertex * 8.0 + 3.0 * edges * 12.0
    graph_gathers = 2.0 * cells * per_vertex * 0.5
    return {
        "kernel": "fsp4ScanMatrixPinnedKernel<true> (the chain's findSimilarPairs4)", "kernel_ms": launch["matrix_kernel_ms"], "bound": "mfma",
        "achieved": tflops, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_FP4_PEAK_TFLOPS, "traffic": None,
        "flop_per_launch": flops,
        "label_propagation": {
            "ms": label_ms, "iterations": iterations, "bound": "hbm", "least_bytes": record_bytes, "least_bytes_in_128_byte_lines": line_bytes,
            "frac_of_hbm_peak_by_records": record_bytes / (label_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if label_ms > 0 else None,
            "frac_of_hbm_peak_by_lines": line_bytes / (label_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if label_ms > 0 else None,
            "turns_per_s": cells * max(1, iterations) / (label_ms * 1e-3) if label_ms > 0 else None,
            "note": "bytes = 2 x edges x 32 B (one record per vertex and neighbour) per iteration, or x 128 B as the lines the gathers move "
                    "(profiles/r05_pmc_bench_chain_1Mcells.json: 5.2 GB fetched per launch against 0.97 GB of records).  What binds is the "
                    "ORDER: the reference's asynchronous updates (src/CellGraph.cpp:445-616) make a vertex wait for every neighbour at a "
                    "smaller position of the iteration's shuffle, and the kernel keeps that order bit for bit (DESIGN.md 3.6); ms includes "
                    "the host's renumbering of the clusters"},
        "create_cell_graph": {
            "ms": graph_ms, "bound": "hbm", "least_bytes": graph_bytes,
            "frac_of_hbm_peak": graph_bytes / (graph_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if graph_ms > 0 else None,
            "dependent_gathers": graph_gathers,
            "note": "bytes = the first graph_k pairs of every cell, the selections (written once, read by the count and the write pass) "
                    "and the edges; the mutual-selection test gathers the list of every selected neighbour with a smaller id (80 B in "
                    "one or two 128-byte lines, from an 80 MB array that no L2 holds): 11-12.6 GB of fabric reads per filterEdgesKernel "
                    "launch in profiles/r05_pmc_bench_chain_1Mcells.json -- the kernel is bound by those gathers' latency (VALU 2-3 %), "
                    "3 % of the chain"},
    }


def bench_chain(args, capi, sharded, synthetic, oracle, device, torch):
    """BASELINE configs[4] on one GPU: findSimilarPairs4 (k=100) -> createCellGraph (threshold 0.2, k=20) -> label
    propagation, the SimilarPairs content staying on the device between the first two."""
    C, G, L, k, thr = args.cells, args.genes, args.lsh_count, args.k, args.threshold
    pipe = sharded.DevicePipeline(C, G, L, k, thr, world_size=1, rank=0, dist=None, device=device)
    toc, data = synthetic.expression_shard(0, C, G, density=args.density, device=device)
    vectors_host = capi.lsh_generate_vectors(G, L, args.seed)
    pipe.set_inputs(toc, data, torch.from_numpy(vectors_host).to(device))
    cells = np.arange(C, dtype=np.uint32)
    times = {"findSimilarPairs4": 0.0, "createCellGraph": 0.0, "labelPropagationClustering": 0.0}

    per_vertex = min(args.graph_k, k) if args.graph_k else k
    d_v0 = torch.empty(C * per_vertex, dtype=torch.int32, device=device)
    d_v1 = torch.empty(C * per_vertex, dtype=torch.int32, device=device)
    d_sim = torch.empty(C * per_vertex, dtype=torch.float32, device=device)

    def step(record, fetch=False):
        # pairs and edges stay on the device from the scan to the clusters; what comes back is the cluster of every cell
        t0 = time.perf_counter()
        pipe.step()
        pipe.check()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        edges = capi.dev_cell_graph_edges_to_device(pipe.pairs.data_ptr(), pipe.used.data_ptr(), C, k, cells, cells, thr, args.graph_k,
                                                    d_v0.data_ptr(), d_v1.data_ptr(), d_sim.data_ptr())
        t2 = time.perf_counter()
        clusters, iterations = capi.dev_cell_graph_label_propagation(cells, d_v0.data_ptr(), d_v1.data_ptr(), d_sim.data_ptr(), edges)
        t3 = time.perf_counter()
        if record:
            times["findSimilarPairs4"] += t1 - t0
            times["createCellGraph"] += t2 - t1
            times["labelPropagationClustering"] += t3 - t2
        v0 = d_v0[:edges].cpu().numpy().view(np.uint32) if fetch else None          # (for the oracle check only)
        v1 = d_v1[:edges].cpu().numpy().view(np.uint32) if fetch else None
        sim = d_sim[:edges].cpu().numpy() if fetch else None
        return v0, v1, sim, clusters, iterations, edges

    v0, v1, sim, clusters, iterations, edge_count = step(False, fetch=True)
    check = {"skipped": "--no-check"}
    if not args.no_check:
        # the WHOLE edge list and EVERY label against the oracle (above 250000 cells its hash-table form: the same loop as the
        # literal std::map / std::set restatement, tests/test_cell_graph_cpu.py holds the two equal)
        p, u = pipe.results_for(0, C)
        ev0, ev1, es = oracle.cell_graph_edges(p["cell"], p["similarity"], u, cells, cells, thr, args.graph_k, hashed=C > 250000)
        if not (np.array_equal(ev0, v0) and np.array_equal(ev1, v1) and np.array_equal(es.view(np.uint32), sim.view(np.uint32))):
            raise SystemExit("PARITY FAILURE: cell graph edges differ from the oracle")
        oc, oit = oracle.label_propagation(cells, v0, v1, sim)
        if not (np.array_equal(oc, clusters) and oit == iterations):
            raise SystemExit("PARITY FAILURE: clusters differ from the oracle")
        check = {"edges": int(len(v0)), "labels": int(C), "iterations": int(iterations)}
        del p, u, ev0, ev1, es, oc
    for _ in range(args.warmup):
        step(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, _, _, clusters, iterations, edge_count = step(True)
    elapsed = time.perf_counter() - t0
    return {
        "metric": "cells/sec through findSimilarPairs4 -> createCellGraph -> labelPropagationClustering",
        "value": C * args.steps / elapsed, "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 popcount / f32 similarities", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: %d synthetic cells x %d genes, %d-bit signatures, findSimilarPairs4 k=%d "
                               "threshold=%g -> createCellGraph(threshold %g, k=%d) -> label propagation, 1 GPU"
                               % (C, G, L, k, thr, thr, args.graph_k),
                   "cells": C, "edges": int(edge_count), "iterations": int(iterations), "clusters": int(clusters.max()) + 1 if len(clusters) else 0},
        "phases_ms": {key: value / args.steps * 1e3 for key, value in times.items()},
        "roofline": chain_roofline(capi, args, times, int(edge_count), int(iterations), C, per_vertex),
        "note": "SimilarPairs and the graph's edges stay device-resident from findSimilarPairs4 through createCellGraph "
                "(em2_dev_cell_graph_edges) to the clusters (em2_dev_cell_graph_label_propagation)",
        "parity_check": check,
    }


def timed_steps_across_ranks(args, torch, dist, device, step):
    """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; the MAX over the ranks."""
    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_fsp5_ranks(args, capi, sharded, oracle, device, torch, dist, world, rank, watchdog, collective_check):
    """BASELINE configs[3] on N GPUs, partitioned as SURVEY.md 8(e) and src/ExpressionMatrixLsh.cpp:405-483 allow: the loop over
    the cells is a loop over independent rows, so every rank builds the (replicated) slice tables from the full signature array
    and queries only the cells of its contiguous id range -- em2_dev_find_similar_pairs5(rowBegin, rowEnd) -- with NO data-path
    collective.  The signatures are the scan-only synthetic ones, generated on every rank from the same seed (what an all-gather of
    the projection's shards would leave there, as on the headline's line).  Gate: cells of every rank's own range, in places spread
    over it, against the oracle.  value = cells of the whole job / the slowest rank's time."""
    C, L, k, thr, q = args.cells, 2048 if args.lsh_count == 1024 else args.lsh_count, args.k, args.threshold, args.slice_length
    begin, end = sharded.shard_range(C, world, rank)
    rows = end - begin
    watchdog.arm("fsp5: signatures, first pass and parity gate", 900)
    sig = synthetic_signatures(torch, C, L, device)
    pairs = torch.zeros((max(1, rows), k, 2), dtype=torch.int32, device=device)
    used = torch.zeros(max(1, rows), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    stage_ms = {"candidate_filter": 0.0, "selection": 0.0}

    def step():
        if rows:
            capi.dev_find_similar_pairs5(sig.data_ptr(), C, begin, end, L, k, thr, q, args.bucket_overflow, pairs.data_ptr(), used.data_ptr(), stream)

    step()
    torch.cuda.synchronize()
    check = {"skipped": "--no-check"}
    if not args.no_check and rows:
        sig_host = sig.cpu().numpy().view(np.uint64)
        places = 8
        span = max(1, min(rows, max(64, args.fsp5_check_cells // world)) // places)
        listed = np.unique(np.concatenate([np.arange(b, b + span) for b in
                                           [begin + (rows - span) * i // (places - 1) for i in range(places)]]).clip(begin, end - 1)).astype(np.uint32)
        cell, sim, oused = oracle.find_similar_pairs5_cells(sig_host, L, k, thr, q, args.bucket_overflow, listed)
        index = torch.from_numpy(listed.astype(np.int64) - begin).to(device)
        got = pairs[index].cpu().numpy().view(np.uint32)
        if not (np.array_equal(used[index].cpu().numpy().view(np.uint32), oused) and np.array_equal(got[:, :, 0], cell) and
                np.array_equal(got[:, :, 1], sim.view(np.uint32))):
            raise SystemExit("PARITY FAILURE: findSimilarPairs5 of rank %d differs from the oracle on the sampled cells" % rank)
        check = {"fsp5_cells_rank0": int(len(listed)), "places": places}
        del sig_host
    # every rank's gate has passed when the collective below returns (a rank that failed has exited: the others' watchdogs end the run)
    counted = torch.tensor([0 if args.no_check else 1], dtype=torch.int64, device=device)
    dist.all_reduce(counted)
    check["ranks_that_passed_their_gate"] = int(counted.item())
    watchdog.arm("fsp5: warmup and timed steps", 180 + 20 * (args.warmup + args.steps))
    elapsed = timed_steps_across_ranks(args, torch, dist, device, step)
    info = capi.dev_find_similar_pairs5_last_launch() if rows else {"filter_ms": 0.0, "select_ms": 0.0, "slice_count": 0, "batches": 0,
                                                                    "distinct_candidates": 0, "gathered_candidates": 0}
    stages = torch.tensor([info["filter_ms"], info["select_ms"]], dtype=torch.float64, device=device)
    dist.all_reduce(stages, op=dist.ReduceOp.MAX)
    stage_ms["candidate_filter"], stage_ms["selection"] = (float(x) for x in stages.tolist())
    watchdog.disarm()
    distinct = info["distinct_candidates"] if info["distinct_candidates"] >= 0 else info["gathered_candidates"]
    W = capi.word_count(L)
    return {
        "metric": "cells/sec through findSimilarPairs5 (bucketed LSH, tables + candidate filter + selection)",
        "value": C * args.steps / elapsed, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 popcount", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %d synthetic cells, %d-bit signatures, findSimilarPairs5 lshSliceLength=%d "
                               "bucketOverflow=%d k=%d threshold=%g, cells sharded by id range over %d GPU(s), slice tables replicated"
                               % (C, L, q, args.bucket_overflow, k, thr, world),
                   "cells": C, "lsh_count": L, "k": k, "slice_length": q, "bucket_overflow": args.bucket_overflow,
                   "rows_per_gpu": sharded.shard_size(C, world), "slices": info["slice_count"], "batches_rank0": info["batches"]},
        "phases_ms_max_over_ranks": dict(stage_ms, tables_and_candidate_unions=elapsed / args.steps * 1e3 - stage_ms["candidate_filter"] - stage_ms["selection"]),
        "collectives_in_a_step": "none: every rank builds the slice tables of ALL cells itself (they are what is replicated) and queries its own "
                                 "cells; the tables' share of a step (a radix sort over cells x slices keys) does not shrink with the ranks",
        "roofline": {"kernel": "filterWideKernel (all batches, rank 0)", "kernel_ms": info["filter_ms"], "bound": "valu",
                     "achieved": distinct * 2.0 * W * 2.0 / (info["filter_ms"] * 1e-3) / 1e12 if info["filter_ms"] > 0 else 0.0,
                     "peak": VALU_LANE_OPS_PER_S / 1e12, "unit": "T lane-op/s",
                     "frac": distinct * 2.0 * W * 2.0 / (info["filter_ms"] * 1e-3) / VALU_LANE_OPS_PER_S if info["filter_ms"] > 0 else 0.0,
                     "traffic": None, "essential_lane_ops": distinct * 2.0 * W * 2.0, "distinct_candidates_rank0": distinct,
                     "note": "rank 0's filter: essential v_xor_b32 + v_bcnt_u32_b32 lane-operations of its cells' distinct candidates against "
                             "the vector ALUs' issue rate (as on the 1-GPU line)"},
        "parity_check": check, "collective_check": collective_check,
    }


def bench_chain_ranks(args, capi, sharded, synthetic, oracle, device, torch, dist, world, rank, watchdog, collective_check):
    """BASELINE configs[4] on N GPUs: findSimilarPairs4 with north_star's partitioning (projection shards, all-gather of the
    signatures, every rank its contiguous rows against all columns on the matrix cores), an all-gather of the finished rows
    (k pairs per cell: 800 MB at 1M cells), then createCellGraph and label propagation on rank 0 -- both are serial by contract
    (insertion order of the edges; the turns of an iteration) and take 20 % of the 1-GPU chain.  Gate: the
    whole edge list and every label against the oracle, from the gathered pairs."""
    C, G, L, k, thr = args.cells, args.genes, args.lsh_count, args.k, args.threshold
    watchdog.arm("chain: inputs, first pass and parity gate", 1200)
    saved = os.environ.get("EM2_SHARDED_SCAN")
    os.environ["EM2_SHARDED_SCAN"] = "0"            # (contiguous rows per rank: the gathered buffer is the SimilarPairs table)
    try:
        pipe = sharded.DevicePipeline(C, G, L, k, thr, world_size=world, rank=rank, dist=dist, device=device)
    finally:
        if saved is None:
            os.environ.pop("EM2_SHARDED_SCAN", None)
        else:
            os.environ["EM2_SHARDED_SCAN"] = saved
    toc, data = synthetic.expression_shard(pipe.row_begin, pipe.row_end, G, density=args.density, device=device)
    vectors_host = capi.lsh_generate_vectors(G, L, args.seed)
    pipe.set_inputs(toc, data, torch.from_numpy(vectors_host).to(device))
    size = pipe.shard
    local_pairs = torch.zeros((size, k, 2), dtype=torch.int32, device=device)
    local_used = torch.zeros(size, dtype=torch.int32, device=device)
    all_pairs = torch.zeros((size * world, k, 2), dtype=torch.int32, device=device)
    all_used = torch.zeros(size * world, dtype=torch.int32, device=device)
    cells = np.arange(C, dtype=np.uint32)
    per_vertex = min(args.graph_k, k) if args.graph_k else k
    if rank == 0:
        d_v0 = torch.empty(C * per_vertex, dtype=torch.int32, device=device)
        d_v1 = torch.empty(C * per_vertex, dtype=torch.int32, device=device)
        d_sim = torch.empty(C * per_vertex, dtype=torch.float32, device=device)
    times = {"findSimilarPairs4": 0.0, "gather_pairs": 0.0, "createCellGraph": 0.0, "labelPropagationClustering": 0.0}
    outcome = {}

    def step(record=True, fetch=False):
        t0 = time.perf_counter()
        pipe.step()
        pipe.check()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if pipe.rows:
            local_pairs[:pipe.rows].copy_(pipe.pairs[:pipe.rows])
            local_used[:pipe.rows].copy_(pipe.used[:pipe.rows])
        dist.all_gather_into_tensor(all_pairs, local_pairs)
        dist.all_gather_into_tensor(all_used, local_used)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rank == 0:
            edges = capi.dev_cell_graph_edges_to_device(all_pairs.data_ptr(), all_used.data_ptr(), C, k, cells, cells, thr, args.graph_k,
                                                        d_v0.data_ptr(), d_v1.data_ptr(), d_sim.data_ptr())
            t3 = time.perf_counter()
            clusters, iterations = capi.dev_cell_graph_label_propagation(cells, d_v0.data_ptr(), d_v1.data_ptr(), d_sim.data_ptr(), edges)
            t4 = time.perf_counter()
            outcome.update(edges=int(edges), iterations=int(iterations), clusters=clusters)
            if fetch:
                outcome.update(v0=d_v0[:edges].cpu().numpy().view(np.uint32), v1=d_v1[:edges].cpu().numpy().view(np.uint32),
                               sim=d_sim[:edges].cpu().numpy())
            if record:
                times["createCellGraph"] += t3 - t2
                times["labelPropagationClustering"] += t4 - t3
        if record:
            times["findSimilarPairs4"] += t1 - t0
            times["gather_pairs"] += t2 - t1

    step(record=False, fetch=True)
    check = {"skipped": "--no-check"}
    if rank == 0 and not args.no_check:
        p = all_pairs[:C].cpu().numpy().view(np.uint32)
        u = all_used[:C].cpu().numpy().view(np.uint32)
        ev0, ev1, es = oracle.cell_graph_edges(p[:, :, 0], p[:, :, 1].view(np.float32), u, cells, cells, thr, args.graph_k, hashed=C > 250000)
        if not (np.array_equal(ev0, outcome["v0"]) and np.array_equal(ev1, outcome["v1"]) and
                np.array_equal(es.view(np.uint32), outcome["sim"].view(np.uint32))):
            raise SystemExit("PARITY FAILURE: cell graph edges differ from the oracle")
        oc, oit = oracle.label_propagation(cells, outcome["v0"], outcome["v1"], outcome["sim"])
        if not (np.array_equal(oc, outcome["clusters"]) and oit == outcome["iterations"]):
            raise SystemExit("PARITY FAILURE: clusters differ from the oracle")
        # the gathered rows themselves: sampled rows of every rank's range against the oracle's findSimilarPairs4
        sig_host = pipe.full_sig[:C].cpu().numpy().view(np.uint64)
        rows_checked = 0
        for b, e, cell, sim, oused in oracle_rows_parallel(oracle, sig_host, L, k, thr, sample_ranges([(0, C)], min(args.check_rows, 1024)), 0):
            if not (np.array_equal(u[b:e], oused) and np.array_equal(p[b:e, :, 0], cell) and np.array_equal(p[b:e, :, 1], sim.view(np.uint32))):
                raise SystemExit("PARITY FAILURE: gathered SimilarPairs rows [%d, %d) differ from the oracle" % (b, e))
            rows_checked += e - b
        check = {"edges": outcome["edges"], "labels": int(C), "iterations": outcome["iterations"], "gathered_rows_against_oracle": rows_checked}
        del p, u, ev0, ev1, es, oc, sig_host
    dist.barrier()
    watchdog.arm("chain: warmup and timed steps", 240 + 30 * (args.warmup + args.steps))
    elapsed = timed_steps_across_ranks(args, torch, dist, device, step)
    watchdog.disarm()
    steps_recorded = args.warmup + args.steps            # (step() records the warm-up steps too: the phases are means over all of them)
    gathered_bytes = float(size) * world * (k * 8 + 4)
    return {
        "metric": "cells/sec through findSimilarPairs4 -> createCellGraph -> labelPropagationClustering",
        "value": C * args.steps / elapsed, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "fp4 e2m1 0/1 products, f32 accumulate, exact (scan) / f32 similarities", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: %d synthetic cells x %d genes, %d-bit signatures, findSimilarPairs4 k=%d threshold=%g "
                               "(rows sharded over %d GPUs) -> all-gather of the pairs -> createCellGraph(threshold %g, k=%d) -> label "
                               "propagation on rank 0" % (C, G, L, k, thr, world, thr, args.graph_k),
                   "cells": C, "edges": outcome.get("edges"), "iterations": outcome.get("iterations"),
                   "clusters": int(outcome["clusters"].max()) + 1 if rank == 0 and len(outcome.get("clusters", [])) else None},
        "phases_ms_rank0": {key: value / steps_recorded * 1e3 for key, value in times.items()},
        "collectives_in_a_step": {"all_gather_signatures_bytes": float(pipe.shard) * world * capi.word_count(L) * 8,
                                  "all_gather_pairs_bytes": gathered_bytes, "backend": dist.get_backend()},
        "parity_check": check, "collective_check": collective_check,
        "note": "graph and labels run on rank 0 while the other ranks wait: their order is the contract (DESIGN.md 3.4, 3.6)",
    }


def main():
    args = parse_args()
    # (the pool's driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))           # (nothing GPU-related has been imported yet; the ranks are children)
    import torch
    import torch.distributed as dist
    from expressionmatrix2_amd import capi, sharded, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to benchmark)")
    # Dry-run aid for boxes with fewer GPUs than ranks (EM2_BENCH_SHARE_DEVICE=1 EM2_BENCH_BACKEND=gloo): all ranks
    # use cuda:0 and the exchange goes through gloo.  Never set by the driver; such a run is not a measurement.
    if os.environ.get("EM2_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    capi.load()
    watchdog = Watchdog(rank)
    collective_check = None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EM2_BENCH_BACKEND", "nccl")
        watchdog.arm("process group + first collective", 300)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=300))
        # what the communicator itself says: every rank adds a one, and the ranks' ids are gathered
        ones = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank))
        dist.all_reduce(ones)
        ids = torch.empty(world, dtype=torch.int64, device=torch.device("cuda", local_rank))
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=torch.device("cuda", local_rank)))
        torch.cuda.synchronize()
        collective_check = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_counted_by_all_reduce": int(ones.item()),
                            "rank_ids_gathered": [int(x) for x in ids.cpu()], "devices_visible_to_rank0": torch.cuda.device_count()}
        if collective_check["ranks_counted_by_all_reduce"] != world or collective_check["rank_ids_gathered"] != list(range(world)):
            raise SystemExit("bench.py: the communicator counts %s ranks, %d expected" % (collective_check, world))
        watchdog.disarm()

    C, G, L, k, thr = args.cells, args.genes, args.lsh_count, args.k, args.threshold
    W = capi.word_count(L)
    device = torch.device("cuda", local_rank)
    if args.workload != "fsp4":
        import oracle_binding
        oracle = oracle_binding.load_oracle()
        if world == 1:
            if args.workload == "fsp5":
                print(json.dumps(bench_fsp5(args, capi, oracle, device, torch)))
            else:
                print(json.dumps(bench_chain(args, capi, sharded, synthetic, oracle, device, torch)))
            return
        # BASELINE names 8 x MI355X for configs[3] and configs[4]: their legs on N ranks (rows by cell-id range)
        if args.workload == "fsp5":
            line = bench_fsp5_ranks(args, capi, sharded, oracle, device, torch, dist, world, rank, watchdog, collective_check)
        else:
            line = bench_chain_ranks(args, capi, sharded, synthetic, oracle, device, torch, dist, world, rank, watchdog, collective_check)
        if rank == 0:
            print(json.dumps(line), flush=True)
        watchdog.arm("destroy_process_group", 60)
        watchdog.exit_code = 0
        dist.destroy_process_group()
        watchdog.disarm()
        return

    # ---- synthetic inputs, resident in HBM ----
    vectors_host = capi.lsh_generate_vectors(G, L, args.seed)            # Lsh::generateLshVectors (host, once)
    vectors = torch.from_numpy(vectors_host).to(device)
    import oracle_binding
    oracle = oracle_binding.load_oracle()
    inputs = {}

    prepare_ms = []

    def make_pipe(sharded_scan):
        saved = os.environ.get("EM2_SHARDED_SCAN")
        if not sharded_scan:
            os.environ["EM2_SHARDED_SCAN"] = "0"
        try:
            pipe = sharded.DevicePipeline(C, G, L, k, thr, world_size=world, rank=rank, dist=dist if world > 1 else None, device=device)
        finally:
            if saved is None:
                os.environ.pop("EM2_SHARDED_SCAN", None)
            else:
                os.environ["EM2_SHARDED_SCAN"] = saved
        if not inputs:
            inputs["toc"], inputs["data"] = synthetic.expression_shard(pipe.row_begin, pipe.row_end, G, density=args.density, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.set_inputs(inputs["toc"], inputs["data"], vectors)
        torch.cuda.synchronize()
        # (em2_dev_prepare_vectors: the hyperplanes' column sums of src/Lsh.cpp:137-144 and their quantised copies -- once per
        # hyperplane matrix, outside the timed step, itemised on the line as phases_ms_rank0.prepare_vectors_once)
        prepare_ms.append((time.perf_counter() - t0) * 1e3)
        return pipe

    # With several ranks on one host the gates share its cores: every rank checks its share of the rows with its share of the
    # threads (the job as a whole still compares >= --check-rows rows per gate).
    rows_per_gate = args.check_rows if world == 1 or args.check_rows <= 0 else max(min(1024, args.check_rows), -(-args.check_rows // world))
    gate_threads = args.check_threads or max(1, min(64, usable_cpus() // world))

    def run_leg(pipe, label, golden_cases):
        """One measurement by the contract: correctness gate (an untimed pass against the CPU oracle), W warmup steps, exactly K
        timed steps between barrier + synchronize on both sides, MAX over the ranks, the gate again on the LAST timed step's
        result.  Every stage runs under the watchdog."""
        toc, data = inputs["toc"], inputs["data"]
        check = {"golden_cases": 0, "signature_cells": 0, "fsp4_rows": 0}
        watchdog.arm(label + ": first pass and parity gate", 900)
        if rank == 0 and not args.no_check and golden_cases:
            # the committed golden digests (tests/golden/, produced by the oracle) against this build's GPU path
            from golden.make_golden import digest, make_signatures, regression_cases
            with open(os.path.join(ROOT, "tests", "golden", "oracle_regression.json")) as f:
                golden = json.load(f)
            for case in regression_cases():
                g_pairs, g_used = capi.find_similar_pairs4(make_signatures(case), case["L"], case["k"], case["thr"])
                if digest(np.ascontiguousarray(g_pairs["cell"]), np.ascontiguousarray(g_pairs["similarity"]), g_used) != golden[case["name"]]["fsp4"]:
                    raise SystemExit("PARITY FAILURE: golden case %s" % case["name"])
                check["golden_cases"] += 1
        pipe.step()
        torch.cuda.synchronize()
        sig_host = pipe.full_sig[:C].cpu().numpy().view(np.uint64)
        if not args.no_check:
            check["signature_cells"], check["fsp4_rows"] = parity_gate(pipe, oracle, synthetic, sig_host, toc, data, vectors_host,
                                                                       rows_per_gate, gate_threads)
        # ---- warmup + timed steps ----
        watchdog.arm(label + ": warmup and timed steps", 180 + 20 * (args.warmup + args.steps))
        if label == "sharded symmetric" and os.environ.get("EM2_BENCH_TEST_FAIL_RANK") == str(rank):
            raise RuntimeError("injected failure (EM2_BENCH_TEST_FAIL_RANK: tests/test_gpu_sharded.py)")
        for _ in range(args.warmup):
            pipe.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pipe.scan_events = []
        proj_events = []
        state_before = device_state(local_rank) if rank == 0 else None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            pipe.project()
            e1.record()
            proj_events.append((e0, e1))
            pipe.exchange()
            pipe.scan(record_events=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        state_after = device_state(local_rank) if rank == 0 else None
        pipe.check()                 # the scan's inter-wave hand-offs all completed (raises otherwise)
        # The hand-off, speculation and inbox paths of the scan depend on timing, so the result of the LAST timed step is
        # put through the same gate again: same signatures as before, sampled rows bit-identical to the oracle.
        watchdog.arm(label + ": parity gate after the timed steps", 900)
        if args.no_check:
            check["skipped"] = "--no-check: diagnostic run, NOT a measurement"
        else:
            sig_after = pipe.full_sig[:C].cpu().numpy().view(np.uint64)
            if not np.array_equal(sig_after, sig_host):
                raise SystemExit("PARITY FAILURE: the signatures of the last timed step differ from the first pass")
            check["after_timing_signature_cells"], check["after_timing_rows"] = parity_gate(
                pipe, oracle, synthetic, sig_after, toc, data, vectors_host, rows_per_gate, gate_threads)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        watchdog.disarm()
        return {"pipe": pipe, "elapsed": elapsed, "check": check, "sig_host": sig_host,
                "scan_ms": float(np.mean([a.elapsed_time(b) for a, b in pipe.scan_events])) if pipe.scan_events else 0.0,
                "proj_ms": float(np.mean([a.elapsed_time(b) for a, b in proj_events])) if proj_events else 0.0,
                "launch": capi.dev_find_similar_pairs4_last_launch(),
                "device_state": {"before_timed_steps": state_before, "after_timed_steps": state_after}}

    def stage_times(pipe, label):
        """Diagnostics outside the timed region: wall ms per stage and per collective with a device synchronisation after each
        (so the stages do not overlap as they do in the measurement), MAX over the ranks."""
        watchdog.arm(label + ": per-stage diagnostic pass", 300)
        pipe.start_timing()
        for _ in range(2):
            pipe.step()
        stages = pipe.stop_timing()
        names = sorted(stages)
        t = torch.tensor([stages[n] for n in names], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        watchdog.disarm()
        return {n: float(v) for n, v in zip(names, t.tolist())}

    total_pairs = C * (C - 1) / 2.0

    def assemble(leg):
        """The JSON line of one leg."""
        pipe, elapsed, check, scan_ms, proj_ms, launch = (leg[key] for key in ("pipe", "elapsed", "check", "scan_ms", "proj_ms", "launch"))
        nnz_local = int(inputs["data"].numel())
        value = total_pairs * args.steps / elapsed
        # The scan of THIS rank handles rows*(C-1)/2 unordered pairs' worth of the job (its share of N(N-1)/2).  In the
        # ordered form that is rows*C comparisons by the kernel; in the symmetric form (1 GPU, all rows in one launch)
        # every unordered pair is evaluated once.  The library reports what it ran.
        matrix = launch["form"] in (3, 4)             # 3: the symmetric form with its triangle part on the matrix cores
        rows_matrix = launch["form"] == 4             # 4: the rows of the shard x all columns on the matrix cores
        symmetric = launch["form"] in (1, 3)
        sharded_symmetric = launch["form"] == 2
        kernel_ms = launch["scan_kernel_ms"] if (symmetric or rows_matrix) and launch["scan_kernel_ms"] > 0 else scan_ms
        launch_pairs = total_pairs / world if sharded_symmetric else pipe.rows * (C - 1) / 2.0
        algorithmic_bytes = launch_pairs * 16.0 * W
        achieved = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
        lane_ops = launch["wave_column_steps"] * 64.0 * 4.0 * W      # (v_xor + v_bcnt) per 32 bits per (lane, column)
        valu_frac = lane_ops / (kernel_ms * 1e-3) / VALU_LANE_OPS_PER_S if kernel_ms else 0.0

        result = {
            "metric": "cell-pair Hamming comparisons/sec (whole node), findSimilarPairs4 incl. signature projection",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64 popcount (scan) / f64 (projection)",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: %d synthetic cells x %d genes (%.3g nnz/cell), %d-bit signatures, "
                            "findSimilarPairs4 k=%d threshold=%g, rows sharded over %d GPU(s)"
                            % (C, G, args.density * G, L, k, thr, world),
                "cells": C, "genes": G, "lsh_count": L, "k": k, "similarity_threshold": thr,
                "rows_per_gpu": pipe.shard, "nnz_rank0": nnz_local,
                "scan": ("sharded-symmetric" if sharded_symmetric else "row-shards-matrix" if rows_matrix else "symmetric-matrix" if matrix
                         else "symmetric" if symmetric else "row-shards"),
            },
            "phases_ms_rank0": {"projection": proj_ms, "scan": scan_ms,
                                "prepare_vectors_once": prepare_ms[-1] if prepare_ms else None},
            "roofline": None,
        }
        hbm_roofline = {
                "kernel": ("fsp4ScanKernel<%d,...>" if os.environ.get("EM2_SCAN_MODE") == "simple"
                           else "fsp4ScanSymmetricKernel<%d,true>" if symmetric
                           else "fsp4ScanSymmetricKernel<%d,true> + fsp4TileKernel" if sharded_symmetric
                           else "fsp4ScanPersistentKernel<%d,true>") % (2 * W),
                "kernel_ms": kernel_ms,
                "form": "symmetric: every unordered pair evaluated once; inbox sort + replay follow the kernel"
                        if symmetric else
                        "sharded symmetric: every unordered pair evaluated once across the ranks (blocks dealt round-robin; "
                        "2 all_reduce + 1 all_gather inside the scan time)" if sharded_symmetric
                        else "ordered: every row of the shard against every column",
                "inbox_entries": launch["inbox_entries"] if (symmetric or sharded_symmetric) else None,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,            # (no PMC digest of the v_xor/v_bcnt forms at this build: the matrix form is the product)
                "traffic_source": None,
                "algorithmic_bytes": algorithmic_bytes,
                "comparisons_executed_per_s": launch["wave_column_steps"] * 64.0 / (kernel_ms * 1e-3) if kernel_ms else 0.0,
                "valu_frac": valu_frac,
                "note": "algorithmic bytes = 16*W per unordered pair; comparisons_executed_per_s = (row, column) mismatch counts the kernel "
                        "actually evaluated per second (about one per unordered pair in the symmetric forms, two in the ordered one); "
                        "operands are cache/SGPR resident so frac is not "
                        "bounded by 1; valu_frac = (v_xor+v_bcnt lane-ops actually executed)/(256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz), "
                        "the measured issue rate of these ops; kernel_ms = HIP events on the launch stream around the scan kernel "
                        "(recorded inside the library for the symmetric form, around the call otherwise)",
        }
        if matrix:
            result["dtype"] = ("fp4 e2m1 %s products, f32 accumulate, exact (scan on the matrix cores) / u32 popcount (band, full rows) / f64 (projection)"
                               % ("+-1" if L > 1024 else "0/1"))
        matrix_traffic = None
        matrix_traffic_source = None
        pinned_walk = True
        profile_config = profile_query("fsp4", C, L, k, world, genes=G)
        if matrix and pinned_walk:
            matrix_traffic, _ = profiled_traffic(PROFILE_DIGESTS["fsp4"], profile_config,
                                                 ("fsp4ScanMatrixPinnedKernel", "fsp4ScanMatrixWideKernel"))
            if matrix_traffic is not None:
                matrix_traffic_source = "profiles/%s (tools/profile_bench.sh: %s)" % (PROFILE_DIGESTS["fsp4"], TRAFFIC_NOTE)
        if matrix and launch["matrix_kernel_ms"] > 0:
            # Dominant kernel: fsp4ScanMatrixKernel, bound by the matrix cores.  One (row, column) pair = a 1024-long dot
            # product of FP4 0 / 1 values = 2 * 1024 flop on v_mfma_f32_32x32x64_f8f6f4; peak = the dense FP4 MFMA
            # figure of MI355X_MICROARCH.md (10 PFLOP/s).  The HBM / instruction view of the same launch rides along.
            # (signatures of 1025..2048 bits: fsp4ScanMatrixWideKernel, a 2048-long contraction per pair)
            contraction = 2048.0 if L > 1024 else 1024.0
            operand_note = ("+-1 operands: 2048 - 2 * mismatches" if L > 1024 else
                            "0 / 1 operands, the accumulator starts at -(popcount(row) + popcount(column)) / 2 and ends at -mismatches / 2")
            flops = launch["matrix_pairs"] * 2.0 * contraction
            tflops = flops / (launch["matrix_kernel_ms"] * 1e-3) / 1e12
            result["roofline"] = {
                "kernel": "fsp4ScanMatrixWideKernel<true>" if L > 1024 else
                          ("fsp4ScanMatrixPinnedKernel<true>" if pinned_walk else "fsp4ScanMatrixKernel<true>"),
                "kernel_ms": launch["matrix_kernel_ms"],
                "form": ("row shards on the matrix cores (north_star's partitioning): every row of the rank's contiguous shard walks ALL "
                         "columns in ascending order as FP4 dot products over %d bits (%s; exact in f32), i.e. each unordered "
                         "pair is evaluated once per side across the node; no inbox, no sort, no second kernel" % (int(contraction), operand_note))
                        if rows_matrix else
                        "symmetric, triangle part on the matrix cores: every unordered pair evaluated once as an FP4 dot "
                        "product over %d bits (%s; exact in f32); the first cells' full rows and each quad's own 256 columns "
                        "stay on v_xor/v_bcnt; inbox sort + replay follow" % (int(contraction), operand_note),
                "bound": "mfma",
                "achieved": tflops,
                "peak": MFMA_FP4_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": tflops / MFMA_FP4_PEAK_TFLOPS,
                "clock_ghz": launch["matrix_clock_ghz"] or None,
                "frac_of_peak_at_that_clock": (tflops / (MFMA_FP4_PEAK_TFLOPS * launch["matrix_clock_ghz"] / 2.4)
                                               if launch["matrix_clock_ghz"] else None),
                "traffic": matrix_traffic,
                "traffic_source": matrix_traffic_source,
                "flop_per_launch": flops,
                "pairs_on_matrix_cores": launch["matrix_pairs"],
                "pairs_per_s_on_matrix_cores": launch["matrix_pairs"] / (launch["matrix_kernel_ms"] * 1e-3),
                "inbox_entries": launch["inbox_entries"],
                "scan_launches_ms": kernel_ms,
                "hbm_view": {key: hbm_roofline[key] for key in ("achieved", "peak", "unit", "frac", "algorithmic_bytes")},
                "note": "flop = 2 * %d per (row, column) pair contracted by fsp4ScanMatrixKernel (pairs counted by the launcher: "
                        "64 rows x the columns below each quad); kernel_ms = HIP events on the launch stream around that kernel "
                        "alone, scan_launches_ms = around all launches of the scan (full-row blocks on v_xor/v_bcnt, fragment "
                        "expansion, the matrix kernel); hbm_view = the algorithmic 16*W bytes per unordered pair over "
                        "scan_launches_ms, kept for comparison with earlier rounds (operands are cache resident, so it is not "
                        "bounded by 1)" % int(contraction),
            }
        elif sharded_symmetric and launch["matrix_pairs"] > 0 and scan_ms > 0:
            # Sharded scan with phases 1 and 2 on the matrix cores: this rank's share of the FP4 contraction over the whole
            # scan time of the rank (phases 0..3 and the collectives between them; no per-kernel events here).
            flops = launch["matrix_pairs"] * 2.0 * 1024.0
            tflops = flops / (scan_ms * 1e-3) / 1e12
            result["dtype"] = "fp4 e2m1 0/1 products, f32 accumulate, exact (phases 1-2 on the matrix cores) / u32 popcount (phase 0, bands) / f64 (projection)"
            result["roofline"] = {
                "kernel": "fsp4ScanMatrixKernel<true> + fsp4TileMatrixKernel (rank 0)",
                "kernel_ms": scan_ms,
                "form": "sharded symmetric, phases 1 and 2 on the matrix cores: every unordered pair evaluated once across the ranks; "
                        "2 all_reduce + the entry exchange are inside kernel_ms",
                "bound": "mfma",
                "achieved": tflops,
                "peak": MFMA_FP4_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": tflops / MFMA_FP4_PEAK_TFLOPS,
                "traffic": None,
                "flop_per_launch": flops,
                "pairs_on_matrix_cores": launch["matrix_pairs"],
                "inbox_entries": launch["inbox_entries"],
                "hbm_view": {key: hbm_roofline[key] for key in ("achieved", "peak", "unit", "frac", "algorithmic_bytes")},
                "note": "per rank: flop = 2 * 1024 per (row, column) pair of this rank's share of phases 1 and 2; kernel_ms is the whole "
                        "scan of the rank including the collectives, so frac understates the kernels",
            }
        else:
            result["roofline"] = hbm_roofline
        result["parity_check"] = check
        # The other kernel of the step: the signature projection.  Algorithmic bytes per SURVEY.md 8(d): the CSR once
        # (8 B per expression count), the hyperplanes once (8 B x genes x bits), the signatures out; flop = 2 x counts x
        # bits (the reference's own formula, src/Lsh.cpp:222).  Bound: HBM.
        if proj_ms > 0 and pipe.rows:
            proj_bytes = 8.0 * nnz_local + 8.0 * G * L + pipe.rows * L / 8.0
            result["roofline_projection"] = {
                "kernel": "projectionScreenQuantizedKernel + projectionScreenItemsKernel + projectionExactItemsKernel (rank 0)",
                "kernel_ms": proj_ms, "bound": "hbm", "achieved": proj_bytes / (proj_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": proj_bytes / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": proj_bytes,
                "flop_per_launch": 2.0 * nnz_local * L, "tflops": 2.0 * nnz_local * L / (proj_ms * 1e-3) / 1e12,
                "tier": pipe.projection_tier(),
                "note": "tier = the first tier the library ran on THIS data (include/em2_lsh.h, EM2_TIER_*: the synthetic counts are "
                        "integers, so the 16-bit copy's products are summed exactly in integers; extra.projection_non_integer_counts "
                        "times the same matrix with every count x 1.5).  "
                        "kernel_ms = HIP events around em2_dev_compute_signatures on the launch stream (16-bit fixed-point screening "
                        "tier, float tier on the words and single bits it left, exact recomputation of the rest); traffic = fabric bytes "
                        "of all its kernels per launch from the committed PMC digest of this configuration: the tiers gather one "
                        "hyperplane line per (count, 64-bit word) from a copy that fills the L2s, so they move far more than the "
                        "algorithmic bytes (wasted_traffic_ratio)",
            }
            projection_traffic, _ = profiled_traffic(PROFILE_DIGESTS["fsp4"], profile_query("fsp4", C, L, k, world, genes=G),
                                                     ("projectionScreen", "projectionExact", "cellStatsKernel"))
            result["roofline_projection"]["traffic"] = projection_traffic
            result["roofline_projection"]["traffic_source"] = ("profiles/%s (%s, all projection kernels)" % (PROFILE_DIGESTS["fsp4"], TRAFFIC_NOTE)
                                                               if projection_traffic else None)
            result["roofline_projection"]["wasted_traffic_ratio"] = projection_traffic / proj_bytes if projection_traffic else None
            # What the tiers actually wait for: one 128-byte line of the 16-bit copy per (count, 64-bit word)
            # through the L2 -> L1 path of a CU, 64 bytes per clock.  A floor of the first tier's design, not of the problem.
            lines = float(nnz_local) * ((L + 63) // 64)
            floor_ms = lines * 128.0 / (256 * 64.0 * 2.4e9) * 1e3
            result["roofline_projection"]["gather_view"] = {
                "lines_per_launch": lines, "bytes_per_launch": lines * 128.0, "floor_ms": floor_ms, "frac": floor_ms / proj_ms,
                "note": "the first tier's gathers at 64 B/clk per CU x 256 CUs x 2.4 GHz; kernel_ms includes the later tiers"}
        result["device_state_rank0"] = leg["device_state"]
        if collective_check is not None:
            result["collective_check"] = collective_check
        return result

    # ---- the measurement(s) ----
    # One GPU: one leg.  Several GPUs: north_star's own partitioning FIRST (contiguous row shards, one all_gather of the
    # signatures, every rank scans its rows against all columns: the simplest collectives), measured and parity-gated by the
    # same contract, and only then the sharded symmetric form (every unordered pair once across the ranks: two all_reduce and
    # an all_to_all inside the scan).  If the second form fails or stops in a collective, the first leg's line is what this
    # run reports -- with the failure on it -- instead of nothing.
    result = None
    if world == 1:
        leg = run_leg(make_pipe(True), "single GPU", True)
        result = assemble(leg)
    else:
        rows_leg = run_leg(make_pipe(False), "row shards", True)
        rows_result = assemble(rows_leg)
        rows_result["stages_ms_max_over_ranks"] = {"form": "row-shards", "backend": dist.get_backend(),
                                                   "note": "diagnostic pass with a synchronisation after every stage; not the measurement",
                                                   **stage_times(rows_leg["pipe"], "row shards")}
        result = rows_result
        symmetric_pipe = make_pipe(True)
        if symmetric_pipe.sharded:
            def report_first_leg(stage):
                rows_result["sharded_symmetric_leg"] = {"status": "did not complete", "stage": stage,
                                                        "note": "the sharded symmetric form stopped in this stage; the line is the row-shard leg, measured before it"}
                if rank == 0:
                    print(json.dumps(rows_result), flush=True)
            watchdog.on_expiry = report_first_leg
            watchdog.exit_code = 0
            try:
                sym_leg = run_leg(symmetric_pipe, "sharded symmetric", False)
                if sym_leg["pipe"].sharded is None:
                    # (a pool overflow or a failed phase on some rank: all ranks agreed to fall back to row shards)
                    rows_result["sharded_symmetric_leg"] = {"status": "fell back to row shards by agreement of the ranks"}
                else:
                    result = assemble(sym_leg)
                    result["stages_ms_max_over_ranks"] = {"form": result["config"]["scan"], "backend": dist.get_backend(),
                                                          "note": "diagnostic pass with a synchronisation after every stage; not the measurement",
                                                          **stage_times(sym_leg["pipe"], "sharded symmetric")}
                    result["row_shard_leg"] = {key: rows_result[key] for key in ("value", "unit", "ms_per_step", "steps", "warmup", "parity_check",
                                                                                   "phases_ms_rank0", "stages_ms_max_over_ranks")}
                    result["row_shard_leg"]["scan"] = "row-shards"
                    result["row_shard_leg"]["note"] = ("the same job with north_star's partitioning, measured first by the same contract: every "
                                                       "rank scans its contiguous rows against all columns (each unordered pair evaluated twice "
                                                       "across the node)")
            except BaseException as error:                # noqa: BLE001 -- this rank must not leave the others in a collective alone
                if isinstance(error, SystemExit) and str(error).startswith("PARITY FAILURE"):
                    # wrong results are never papered over: the whole run fails
                    print("[bench] rank %d: %s" % (rank, error), file=sys.stderr, flush=True)
                    os._exit(1)
                print("[bench] rank %d: sharded symmetric leg failed: %r" % (rank, error), file=sys.stderr, flush=True)
                # the other ranks are (or will be) waiting in a collective this rank has left: all of them leave through their
                # watchdogs, rank 0 printing the first leg; so does this one
                watchdog.arm("sharded symmetric leg failed on rank %d (%s)" % (rank, type(error).__name__), 5)
                time.sleep(3600)
            watchdog.on_expiry = None
            watchdog.exit_code = 1
        leg = rows_leg if result is rows_result else sym_leg
    pipe, sig_host = leg["pipe"], leg["sig_host"]

    if rank == 0 and not args.no_cpu_baseline:
        m = min(args.cpu_baseline_cells, C)
        t1 = time.perf_counter()
        oracle.find_similar_pairs4(sig_host[:m], L, k, thr)
        dt = time.perf_counter() - t1
        result["cpu_baseline"] = {
            "value": (m * (m - 1) / 2.0) / dt,
            "unit": "pairs/s",
            "cores": 1,
            "kind": "port",
            "sample": "oracle literal findSimilarPairs4 pair loop + selection on the first %d cells of the same "
                      "signatures (%.3g unordered pairs, %.1f s, host has %d hardware threads and a quota of %d CPUs, 1 used)"
                      % (m, m * (m - 1) / 2.0, dt, os.cpu_count() or 0, usable_cpus()),
        }
        # Not the reference's way of running (it is single-threaded): the same sample with the per-cell contract's rows
        # spread over host threads (every row against all m columns, i.e. each pair from both sides), to show what the
        # whole host could do.  Reported beside the baseline, never used for any ratio.
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(64, usable_cpus()))
        bounds = [m * i // threads for i in range(threads + 1)]
        t2 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(lambda i: oracle.find_similar_pairs4_rows(sig_host[:m], L, k, thr, bounds[i], bounds[i + 1]),
                          range(threads)))
        dt2 = time.perf_counter() - t2
        result["cpu_all_threads"] = {
            "value": (m * (m - 1) / 2.0) / dt2, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "our row-parallel use of the oracle (not the reference's execution model): the same %d cells, rows "
                      "split over %d threads, %.1f s" % (m, threads, dt2),
        }
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0 and world == 1 and not args.no_extra and (C, G) == (1000000, 30000):
        # BASELINE configs[1] (100k cells x 20k genes, 1% nnz, 1024 bit, 1 GPU): a second, small measurement on the same
        # line -- never the headline.  Same pipeline, same gates (sampled rows against the oracle before and after).
        result["extra"] = {}
        if not args.no_check:
            result["extra"]["rows_form_one_gpu_as_rank_r_of_P"] = rows_form_block(args, capi, pipe, oracle, sig_host, torch, gate_threads)
            reference_pairs, reference_used = pipe.results_for(0, C)
            result["extra"]["projection_non_integer_counts"] = non_integer_projection_block(pipe, oracle, synthetic, torch, inputs["toc"], inputs["data"], vectors_host)
            result["extra"]["facade_e2e"] = facade_block(args, capi, synthetic, torch, inputs["toc"], inputs["data"], reference_pairs, reference_used)
            del reference_pairs, reference_used
        leg.clear()
        inputs.clear()
        del pipe, vectors
        torch.cuda.empty_cache()
        if L <= 2048:
            result["extra"]["clustered_regime"] = clustered_regime_block(args, capi, oracle, device, torch, gate_threads)
            torch.cuda.empty_cache()
        result["extra"]["configs[1]"] = small_config(args, capi, sharded, synthetic, oracle, device, torch)
        # BASELINE configs[3] (bucketed findSimilarPairs5, 2048 bit) and configs[4] (findSimilarPairs4 -> createCellGraph ->
        # label propagation) at their 1-GPU sizes, a few steps each: the secondary lines of --workload fsp5 / chain,
        # abridged, so that they are on the line the driver records (never the headline)
        import copy
        small = copy.copy(args)
        small.steps, small.warmup = 2, 1
        line = bench_fsp5(small, capi, oracle, device, torch)
        result["extra"]["configs[3]"] = {key: line[key] for key in ("metric", "value", "unit", "ms_per_step", "steps", "config",
                                                                     "phases_ms", "roofline", "stage_bounds", "parity_check")}
        torch.cuda.empty_cache()
        line = bench_chain(small, capi, sharded, synthetic, oracle, device, torch)
        result["extra"]["configs[4]"] = {key: line[key] for key in ("metric", "value", "unit", "ms_per_step", "steps", "config",
                                                                     "phases_ms", "roofline", "parity_check", "note")}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        watchdog.arm("destroy_process_group", 60)
        watchdog.exit_code = 0                 # (the line is out; a shutdown that hangs is not a failed measurement)
        dist.destroy_process_group()
        watchdog.disarm()


if __name__ == "__main__":
    main()
