#!/usr/bin/env python3
"""Scale checks on a real MI355X (not part of pytest: minutes of GPU time, GBs of memory).

    python3 tools/scale_check.py fsp5   # BASELINE configs[3] shape: 1M cells, 2048 bit, lshSliceLength 20
    python3 tools/scale_check.py fsp4w  # findSimilarPairs4 at 2048 bit, 200k cells
    CELLS=1000000 LSH=1024 SWEEP="EM2_SCAN_MODE=persistent;EM2_MIN_SEGMENT_COLUMNS=32768,EM2_FULL_ROW_CELLS=8192" \
        python3 tools/scale_check.py sweep   # scan-only timings under different EM2_* knobs, each parity-checked

Signatures are synthetic (64 cluster centres, each bit flipped with probability 0.15), generated in HBM with
torch.  Sampled cells are compared bit-for-bit with the CPU oracle; timings are printed as JSON."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding  # noqa: E402
from expressionmatrix2_amd import capi  # noqa: E402


def clustered_signatures_gpu(cells, lsh_count, clusters=64, flip=0.15, seed=1):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    words = (lsh_count - 1) // 64 + 1
    centre = torch.randint(0, 2, (clusters, words * 64), generator=gen, device="cuda", dtype=torch.uint8)
    cluster = torch.randint(0, clusters, (cells,), generator=gen, device="cuda")
    out = torch.empty((cells, words), dtype=torch.int64, device="cuda")
    weights = (1 << torch.arange(63, -1, -1, device="cuda", dtype=torch.int64))       # first bit most significant
    for begin in range(0, cells, 65536):
        end = min(cells, begin + 65536)
        bits = centre[cluster[begin:end]] ^ (torch.rand((end - begin, words * 64), generator=gen, device="cuda") < flip).to(torch.uint8)
        bits[:, lsh_count:] = 0
        out[begin:end] = (bits.view(end - begin, words, 64).to(torch.int64) * weights).sum(dim=2)
    return out


def compare(pairs_dev, used_dev, cell, sim, used, rows):
    p = pairs_dev[rows].cpu().numpy().view(np.uint32)
    u = used_dev[rows].cpu().numpy().view(np.uint32)
    return bool(np.array_equal(u, used) and np.array_equal(p[:, :, 0], cell) and np.array_equal(p[:, :, 1], sim.view(np.uint32)))


def main():
    what = sys.argv[1]
    oracle = oracle_binding.load_oracle()
    capi.load()
    stream = torch.cuda.current_stream().cuda_stream
    if what == "fsp5":
        cells, L, k, thr, q, ovf = int(os.environ.get("CELLS", 1000000)), 2048, 100, 0.2, 20, 1000
        sig = clustered_signatures_gpu(cells, L)
        d_pairs = torch.zeros((cells, k, 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(cells, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.dev_find_similar_pairs5(sig.data_ptr(), cells, 0, cells, L, k, thr, q, ovf, d_pairs.data_ptr(),
                                     d_used.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        host = sig.cpu().numpy().view(np.uint64)
        begin = cells // 3
        t1 = time.perf_counter()
        cell, sim, used = oracle.find_similar_pairs5_rows(host, L, k, thr, q, ovf, begin, begin + 32)
        t_oracle = time.perf_counter() - t1
        ok = compare(d_pairs, d_used, cell, sim, used, slice(begin, begin + 32))
        u = d_used.cpu().numpy()
        print(json.dumps({"check": "fsp5", "cells": cells, "lsh_count": L, "lsh_slice_length": q, "k": k,
                          "gpu_seconds": dt, "cells_with_k_neighbours": int((u == k).sum()),
                          "mean_used": float(u.mean()), "sampled_rows_bit_exact": ok,
                          "oracle_seconds_for_32_rows_incl_tables": t_oracle}))
        if not ok:
            raise SystemExit("PARITY FAILURE")
    elif what == "fsp4w":
        cells, L, k, thr = int(os.environ.get("CELLS", 200000)), 2048, 100, 0.2
        sig = clustered_signatures_gpu(cells, L)
        ws_bytes = capi.dev_find_similar_pairs4_workspace(cells, cells, L, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        d_pairs = torch.zeros((cells, k, 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(cells, dtype=torch.int32, device="cuda")
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.dev_find_similar_pairs4(sig.data_ptr(), cells, 0, cells, L, k, thr, d_pairs.data_ptr(),
                                         d_used.data_ptr(), ws.data_ptr(), ws_bytes, stream)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        host = sig.cpu().numpy().view(np.uint64)
        begin = cells // 2
        cell, sim, used = oracle.find_similar_pairs4_rows(host, L, k, thr, begin, begin + 32)
        ok = compare(d_pairs, d_used, cell, sim, used, slice(begin, begin + 32))
        print(json.dumps({"check": "fsp4 2048-bit", "cells": cells, "seconds": min(times),
                          "ordered_comparisons_per_s": cells * cells / min(times), "sampled_rows_bit_exact": ok}))
        if not ok:
            raise SystemExit("PARITY FAILURE")
    elif what == "fsp7":
        cells, L, k, thr = int(os.environ.get("CELLS", 1000000)), int(os.environ.get("LSH", 1024)), 100, 0.2
        lengths, max_check, log2b = [20, 14], int(os.environ.get("MAXCHECK", 1000)), 18
        sig = clustered_signatures_gpu(cells, L)
        host = sig.cpu().numpy().view(np.uint64)
        import ctypes
        lengths_arr = np.asarray(lengths, dtype=np.int32)
        d_pairs = torch.zeros((cells, k, 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(cells, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.check(capi.load().em2_dev_find_similar_pairs7(sig.data_ptr(), cells, 0, cells, L, k, thr, capi._ptr(lengths_arr),
                                                           len(lengths), max_check, log2b, d_pairs.data_ptr(), d_used.data_ptr(),
                                                           stream))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        u = d_used.cpu().numpy()
        print(json.dumps({"check": "fsp7", "cells": cells, "lsh_count": L, "lengths": lengths, "max_check": max_check,
                          "gpu_seconds": dt, "mean_used": float(u.mean())}), flush=True)
        # parity on a problem the literal oracle can hold: the first 20000 cells on their own
        small = 20000
        sub = np.ascontiguousarray(host[:small])
        cell, sim, used = oracle.find_similar_pairs7(sub, L, k, thr, lengths, max_check, log2b)
        pairs, gused = capi.find_similar_pairs7(sub, L, k, thr, lengths, max_check, log2b)
        ok = bool(np.array_equal(gused, used) and np.array_equal(pairs["cell"], cell) and
                  np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32)))
        print(json.dumps({"check": "fsp7 parity on the first 20000 cells", "bit_exact": ok}))
        if not ok:
            raise SystemExit("PARITY FAILURE")
    elif what == "rows":
        # North_star's partitioning on one GPU: rank r of P scans its contiguous rows against all columns
        # (em2_dev_find_similar_pairs4 with a row range).  RANKS="8:0,8:3,8:7,4:1,2:1", CHECK_ROWS rows per shard (16 places)
        # against the oracle on all host threads.  SWEEP as in `sweep` (knob settings separated by ';').
        from concurrent.futures import ThreadPoolExecutor
        cells, L, k, thr = int(os.environ.get("CELLS", 1000000)), int(os.environ.get("LSH", 1024)), 100, 0.2
        check_rows = int(os.environ.get("CHECK_ROWS", 10240))
        sig = clustered_signatures_gpu(cells, L)
        host = sig.cpu().numpy().view(np.uint64)
        threads = max(1, min(64, os.cpu_count() or 1))
        for config in os.environ.get("SWEEP", "").split(";"):
            knobs = dict(item.split("=", 1) for item in config.split(",") if "=" in item)
            os.environ.update(knobs)
            for item in os.environ.get("RANKS", "8:0,8:3,8:7,4:1,2:1").split(","):
                world, rank = (int(x) for x in item.split(":"))
                per = -(-cells // world)
                begin, end = min(cells, rank * per), min(cells, (rank + 1) * per)
                rows = end - begin
                ws_bytes = capi.dev_find_similar_pairs4_workspace(cells, rows, L, k)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
                d_pairs = torch.zeros((rows, k, 2), dtype=torch.int32, device="cuda")
                d_used = torch.zeros(rows, dtype=torch.int32, device="cuda")
                times, kernel = [], []
                for _ in range(int(os.environ.get("REPEATS", 3))):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    capi.dev_find_similar_pairs4(sig.data_ptr(), cells, begin, end, L, k, thr, d_pairs.data_ptr(),
                                                 d_used.data_ptr(), ws.data_ptr(), ws_bytes, stream)
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - t0)
                    kernel.append(capi.dev_find_similar_pairs4_last_launch()["matrix_kernel_ms"])
                capi.dev_find_similar_pairs4_status(ws.data_ptr(), rows, k, stream)
                span = max(1, check_rows // 16)
                starts = sorted(set(begin + (rows - span) * i // 15 for i in range(16))) if rows > span else [begin]
                pieces = [(s0 + j, min(end, s0 + j + 16)) for s0 in starts for j in range(0, span, 16) if s0 + j < end]
                with ThreadPoolExecutor(max_workers=threads) as pool:
                    expected = list(pool.map(lambda piece: oracle.find_similar_pairs4_rows(host, L, k, thr, piece[0], piece[1]), pieces))
                checked, ok = 0, True
                for (b, e), (c, s_, u) in zip(pieces, expected):
                    ok = ok and compare(d_pairs, d_used, c, s_, u, slice(b - begin, e - begin))
                    checked += e - b
                launch = capi.dev_find_similar_pairs4_last_launch()
                print(json.dumps({"check": "rows", "cells": cells, "lsh_count": L, "world": world, "rank": rank, "rows": [begin, end],
                                  "knobs": knobs, "ms": [round(t * 1e3, 2) for t in times], "kernel_ms": [round(t, 2) for t in kernel],
                                  "ordered_pairs_per_s": rows * cells / min(times), "rows_checked": checked, "rows_bit_exact": ok,
                                  "form": launch["form"], "clock_ghz": round(launch["matrix_clock_ghz"], 3)}), flush=True)
                del ws, d_pairs, d_used
                if not ok and not os.environ.get("IGNORE_PARITY"):
                    raise SystemExit("PARITY FAILURE")
            for key in knobs:
                os.environ.pop(key, None)
    elif what == "sweep":
        cells, L, k, thr = int(os.environ.get("CELLS", 1000000)), int(os.environ.get("LSH", 1024)), 100, 0.2
        sig = clustered_signatures_gpu(cells, L)
        host = sig.cpu().numpy().view(np.uint64)
        sample = [0, cells // 2, cells - 24]
        expected = [oracle.find_similar_pairs4_rows(host, L, k, thr, b, b + 24) for b in sample]
        d_pairs = torch.zeros((cells, k, 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(cells, dtype=torch.int32, device="cuda")
        for config in os.environ.get("SWEEP", "").split(";"):
            knobs = dict(item.split("=", 1) for item in config.split(",") if "=" in item)
            os.environ.update(knobs)
            ws_bytes = capi.dev_find_similar_pairs4_workspace(cells, cells, L, k)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
            times = []
            for _ in range(int(os.environ.get("REPEATS", 2))):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                capi.dev_find_similar_pairs4(sig.data_ptr(), cells, 0, cells, L, k, thr, d_pairs.data_ptr(),
                                             d_used.data_ptr(), ws.data_ptr(), ws_bytes, stream)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            ok = all(compare(d_pairs, d_used, c, s_, u, slice(b, b + 24)) for b, (c, s_, u) in zip(sample, expected))
            print(json.dumps({"check": "sweep", "cells": cells, "lsh_count": L, "knobs": knobs, "ms": [round(t * 1e3, 2) for t in times],
                              "unordered_pairs_per_s": cells * (cells - 1) / 2 / min(times), "sampled_rows_bit_exact": ok,
                              "last_launch": capi.dev_find_similar_pairs4_last_launch()}), flush=True)
            del ws
            for key in knobs:
                os.environ.pop(key, None)
            if not ok and not os.environ.get("IGNORE_PARITY"):
                raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
