#!/usr/bin/env python3
"""Data-regime sweep of the findSimilarPairs4 scan (measurement 7b): the scan-only input of SURVEY.md 8(d) --
cluster centres with every bit flipped with probability `flip` -- at CELLS cells x LSH bits for cluster counts 8 / 64 / 512,
flips 0.05 / 0.15 / 0.3 and thresholds 0.0 / 0.2 / 0.5: per case the scan's time (best of REPEATS), unordered pairs/s, the
form the launch took (3 = symmetric on the matrix cores, 4 = every row against all columns on the matrix cores: the fall-back
of a symmetric scan whose inbox overflowed), the kernel's time and clock, the deferred candidates, and CHECK_ROWS rows in 16
places against the oracle.  One JSON line per case (profiles/r05_data_regime_sweep.jsonl).

    python3 tools/regime_sweep.py > gpurun_out/regime.jsonl"""
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
    oracle = oracle_binding.load_oracle()
    capi.load()
    device = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    pairs = torch.zeros((cells, k, 2), dtype=torch.int32, device=device)
    used = torch.zeros(cells, dtype=torch.int32, device=device)
    ws_bytes = capi.dev_find_similar_pairs4_workspace(cells, cells, L, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    threads = max(1, min(64, os.cpu_count() or 1))
    for clusters in (8, 64, 512):
        for flip in (0.05, 0.15, 0.3):
            sig = bench.synthetic_signatures(torch, cells, L, device, clusters, flip, 4321)
            sig_host = sig.cpu().numpy().view(np.uint64)
            for thr in (0.0, 0.2, 0.5):
                times = []
                for _ in range(repeats):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    capi.dev_find_similar_pairs4(sig.data_ptr(), cells, 0, cells, L, k, thr, pairs.data_ptr(), used.data_ptr(), ws.data_ptr(),
                                                 ws_bytes, stream)
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - t0)
                capi.dev_find_similar_pairs4_status(ws.data_ptr(), cells, k, stream)
                launch = capi.dev_find_similar_pairs4_last_launch()
                ranges = bench.sample_ranges([(0, cells)], check_rows)
                host_pairs = pairs.cpu().numpy().view(np.uint32)
                host_used = used.cpu().numpy().view(np.uint32)
                checked, ok = 0, True
                for b, e, cell, sim, oused in bench.oracle_rows_parallel(oracle, sig_host, L, k, thr, ranges, threads):
                    ok = ok and bool(np.array_equal(host_used[b:e], oused) and np.array_equal(host_pairs[b:e, :, 0], cell) and
                                     np.array_equal(host_pairs[b:e, :, 1], sim.view(np.uint32)))
                    checked += e - b
                print(json.dumps({"cells": cells, "lsh_count": L, "k": k, "clusters": clusters, "flip": flip, "threshold": thr,
                                  "scan_ms": round(min(times) * 1e3, 2), "unordered_pairs_per_s": cells * (cells - 1) / 2.0 / min(times),
                                  "scan_form": launch["form"], "kernel_ms": round(launch["matrix_kernel_ms"], 2),
                                  "clock_ghz": round(launch["matrix_clock_ghz"], 3), "deferred_candidates": launch["inbox_entries"],
                                  "mean_used": float(host_used.mean()), "rows_checked": checked, "rows_bit_exact": ok}), flush=True)
                if not ok:
                    raise SystemExit("PARITY FAILURE")
            del sig


if __name__ == "__main__":
    main()
