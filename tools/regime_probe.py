#!/usr/bin/env python3
"""One data regime of the findSimilarPairs4 scan at a time (the generator of SURVEY.md 8(d): cluster centres, every bit
flipped with probability `flip`), under whatever EM2_* knobs the environment holds: scan ms (best of REPEATS), the kernel's ms
and clock, deferred candidates, CHECK_ROWS rows in 16 places against the oracle.  One JSON line per regime.

    REGIMES="64:0.15:0.2,8:0.05:0.2" CELLS=1000000 python3 tools/regime_probe.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import oracle_binding  # noqa: E402
from expressionmatrix2_amd import capi  # noqa: E402


def main():
    cells, L, k = int(os.environ.get("CELLS", 1000000)), int(os.environ.get("LSH", 1024)), int(os.environ.get("K", 100))
    check_rows, repeats = int(os.environ.get("CHECK_ROWS", 1024)), int(os.environ.get("REPEATS", 2))
    regimes = [tuple(r.split(":")) for r in os.environ.get("REGIMES", "64:0.15:0.2").split(",")]
    oracle = oracle_binding.load_oracle() if check_rows else None
    capi.load()
    device = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    pairs = torch.zeros((cells, k, 2), dtype=torch.int32, device=device)
    used = torch.zeros(cells, dtype=torch.int32, device=device)
    ws_bytes = capi.dev_find_similar_pairs4_workspace(cells, cells, L, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    threads = max(1, min(64, os.cpu_count() or 1))
    last = None
    for clusters, flip, thr in regimes:
        clusters, flip, thr = int(clusters), float(flip), float(thr)
        if last != (clusters, flip):
            sig = bench.synthetic_signatures(torch, cells, L, device, clusters, flip, 4321)
            sig_host = sig.cpu().numpy().view(np.uint64) if check_rows else None
            last = (clusters, flip)
        times, kernel = [], []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.dev_find_similar_pairs4(sig.data_ptr(), cells, 0, cells, L, k, thr, pairs.data_ptr(), used.data_ptr(), ws.data_ptr(),
                                         ws_bytes, stream)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            kernel.append(capi.dev_find_similar_pairs4_last_launch()["matrix_kernel_ms"])
        capi.dev_find_similar_pairs4_status(ws.data_ptr(), cells, k, stream)
        launch = capi.dev_find_similar_pairs4_last_launch()
        checked, ok = 0, True
        if check_rows:
            ranges = bench.sample_ranges([(0, cells)], check_rows)
            host_pairs = pairs.cpu().numpy().view(np.uint32)
            host_used = used.cpu().numpy().view(np.uint32)
            for b, e, cell, sim, oused in bench.oracle_rows_parallel(oracle, sig_host, L, k, thr, ranges, threads):
                ok = ok and bool(np.array_equal(host_used[b:e], oused) and np.array_equal(host_pairs[b:e, :, 0], cell) and
                                 np.array_equal(host_pairs[b:e, :, 1], sim.view(np.uint32)))
                checked += e - b
        print(json.dumps({"tag": os.environ.get("TAG", ""), "cells": cells, "lsh_count": L, "clusters": clusters, "flip": flip, "threshold": thr,
                          "scan_ms": round(min(times) * 1e3, 2), "kernel_ms": round(min(kernel), 2), "scan_form": launch["form"],
                          "clock_ghz": round(launch["matrix_clock_ghz"], 3), "deferred_candidates": launch["inbox_entries"],
                          "rows_checked": checked, "rows_bit_exact": ok}), flush=True)
        if not ok:
            raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
