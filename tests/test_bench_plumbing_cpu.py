"""bench.py's plumbing that needs no GPU: the PMC digests it takes `traffic` from really are of the configurations it asks for
(BENCH_r03.json cited a round-2 file and printed nulls because a digest had lost its `config`), the row sampling of the
parity gates, and `python bench.py --gpus N` starting its own ranks as children."""
import glob
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_named_digest_answers_for_its_configuration():
    queries = {
        "fsp4": (bench.profile_query("fsp4", 1000000, 1024, 100, 1, genes=30000),
                 [("fsp4ScanMatrixPinnedKernel", "fsp4ScanMatrixWideKernel"), ("projectionScreen", "projectionExact", "cellStatsKernel")]),
        "fsp5": (bench.profile_query("fsp5", 1000000, 2048, 100, slice_length=20, bucket_overflow=1000), [("filterWideKernel", "filterCooperativeKernel")]),
    }
    for workload, (query, prefix_sets) in queries.items():
        for prefixes in prefix_sets:
            traffic, launches = bench.profiled_traffic(bench.PROFILE_DIGESTS[workload], query, prefixes)
            assert traffic is not None and traffic > 0 and launches, (workload, prefixes)
    # another configuration is refused, never answered from a stale file
    assert bench.profiled_traffic(bench.PROFILE_DIGESTS["fsp4"], bench.profile_query("fsp4", 100000, 1024, 100, 1, genes=20000),
                                  ("fsp4ScanMatrixPinnedKernel",)) == (None, None)
    assert bench.profiled_traffic(bench.PROFILE_DIGESTS["fsp4"], bench.profile_query("fsp4", 1000000, 1024, 100, 2, genes=30000),
                                  ("fsp4ScanMatrixPinnedKernel",)) == (None, None)
    assert bench.profiled_traffic(bench.PROFILE_DIGESTS["fsp5"], bench.profile_query("fsp5", 1000000, 2048, 100, slice_length=16, bucket_overflow=1000),
                                  ("filterWideKernel",)) == (None, None)


def test_every_committed_bench_digest_names_its_configuration():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_pmc_bench_*.json")))
    assert files
    for path in files:
        with open(path) as f:
            digest = json.load(f)
        assert digest["config"].get("cells") and digest["config"].get("n_gpus"), path
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import profile_digest
        assert digest["config"] == profile_digest.config_of_line(digest["bench_line_under_trace"]), path
    for name in bench.PROFILE_DIGESTS.values():
        assert os.path.join(ROOT, "profiles", name) in files


def test_bench_source_names_no_older_digest():
    with open(os.path.join(ROOT, "bench.py")) as f:
        text = f.read()
    assert "r02_pmc" not in text and "r01_pmc_hbm" not in text and "r01_pmc_matrix" not in text


def test_sample_ranges():
    whole = [(0, 1000000)]
    ranges = bench.sample_ranges(whole, 10240)
    assert sum(e - b for b, e in ranges) >= 10240 and ranges[0][0] == 0 and ranges[-1][1] == 1000000
    assert all(b < e for b, e in ranges) and all(ranges[i][1] <= ranges[i + 1][0] for i in range(len(ranges) - 1))
    assert bench.sample_ranges(whole, 0) == whole and bench.sample_ranges([(5, 50)], 100) == [(5, 50)]
    blocks = [(64 * b, 64 * b + 64) for b in range(1, 15625, 2)]
    picked = bench.sample_ranges(blocks, 10240)
    assert sum(e - b for b, e in picked) >= 10240 and picked[0] == blocks[0] and picked[-1] == blocks[-1] and set(picked) <= set(blocks)


def test_oracle_rows_parallel_equals_one_call(oracle):
    import numpy as np
    rng = np.random.default_rng(3)
    sig = rng.integers(0, 2 ** 63, size=(700, 2), dtype=np.uint64)
    sig[100:400] = sig[100]
    sig[100:400, 1] ^= rng.integers(0, 255, size=300, dtype=np.uint64)
    ranges = [(0, 130), (300, 700)]
    pieces = bench.oracle_rows_parallel(oracle, sig, 128, 7, 0.2, ranges, threads=3)
    assert sum(e - b for b, e, *_ in pieces) == 530
    for begin, end, cell, sim, used in pieces:
        c1, s1, u1 = oracle.find_similar_pairs4_rows(sig, 128, 7, 0.2, begin, end)
        assert np.array_equal(cell, c1) and np.array_equal(sim.view(np.uint32), s1.view(np.uint32)) and np.array_equal(used, u1)


def test_plain_command_with_gpus_2_starts_its_own_ranks():
    """No launcher, no GPU here: the parent must start two ranks as children (which report that they need a GPU -- the
    launcher ends the other rank as soon as the first has failed, so one report is all that is certain) and hand back their
    non-zero exit code -- not stop with 'must be launched with torch.distributed.run'."""
    env = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "2048"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert "must be launched" not in run.stderr
    if bench_has_gpu():
        return
    assert run.returncode != 0
    assert run.stderr.count("bench.py needs a GPU") >= 1 and "torch.distributed" in run.stderr


def bench_has_gpu():
    from expressionmatrix2_amd import capi
    try:
        return capi.device_count() > 0
    except Exception:                    # noqa: BLE001
        return False
