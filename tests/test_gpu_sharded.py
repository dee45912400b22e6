"""The multi-GPU scan paths end to end through bench.py's pipeline with TWO ranks on the one GPU of the test box
(gloo for the collectives, EM2_BENCH_SHARE_DEVICE=1): projection shards -> all-gather -> scan -> parity gate of
every rank against the CPU oracle (bench.py exits non-zero on any difference).  Covers the sharded symmetric scan
(block-cyclic ownership, snapshot all_reduce, entry all_gather, replay) and the row-shard scan."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_two_ranks(extra_env, cells, port, ranks=2, extra_args=(), timeout=600):
    env = dict(os.environ)
    env.update({"EM2_BENCH_SHARE_DEVICE": "1", "EM2_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1",
                # two persistent kernels share one GPU here: keep them from oversubscribing it (hand-off waits of one
                # process must not keep the other's waves off the machine)
                "EM2_BLOCKS_PER_CU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1",
           "--warmup", "0", "--cells", str(cells), "--genes", "3000", "--no-cpu-baseline", "--check-rows", "96"] + list(extra_args)
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("ranks", [2, 3])
def test_sharded_symmetric_scan_two_ranks(ranks):
    """Two ranks exchange their deferred candidates by all_to_all (a power-of-two world), three gather the pools."""
    result = run_two_ranks({"EM2_SHARDED_MIN_CELLS": "1000"}, cells=30000, port=29631 if ranks == 2 else 29634, ranks=ranks)
    assert result["config"]["scan"] == "sharded-symmetric"
    assert result["n_gpus"] == ranks and result["parity_check"]["fsp4_rows"] > 0
    assert result["roofline"]["inbox_entries"] > 0
    # north_star's own partitioning was measured first, by the same contract, and rides along
    leg = result["row_shard_leg"]
    assert leg["scan"] == "row-shards" and leg["value"] > 0 and leg["steps"] == 1 and leg["parity_check"]["fsp4_rows"] > 0
    assert result["collective_check"]["ranks_counted_by_all_reduce"] == ranks and result["collective_check"]["rank_ids_gathered"] == list(range(ranks))


def test_sharded_symmetric_scan_four_ranks_ragged_size():
    """Four ranks (all_to_all exchange with more than one peer) on a cell count that is no multiple of anything."""
    result = run_two_ranks({"EM2_SHARDED_MIN_CELLS": "1000"}, cells=40003, port=29635, ranks=4)
    assert result["config"]["scan"] == "sharded-symmetric"
    assert result["n_gpus"] == 4 and result["parity_check"]["fsp4_rows"] > 0


def test_row_shard_scan_two_ranks():
    result = run_two_ranks({"EM2_SHARDED_SCAN": "0"}, cells=30000, port=29632)
    assert result["config"]["scan"] == "row-shards"
    assert result["parity_check"]["fsp4_rows"] > 0


def test_sharded_scan_overflow_falls_back_to_row_shards():
    result = run_two_ranks({"EM2_SHARDED_MIN_CELLS": "1000", "EM2_INBOX_CAPACITY": "2048"}, cells=30000, port=29633)
    assert result["config"]["scan"] == "row-shards"
    assert result["parity_check"]["fsp4_rows"] > 0


def test_a_rank_that_leaves_the_symmetric_leg_does_not_void_the_run():
    """one broken exchange must not cost the whole record.  Rank 1 fails at the start of the sharded
    symmetric leg's timed steps; rank 0 is left in a collective; every rank leaves through its watchdog within the stage limit
    (shortened here), exit code 0, and the line is the row-shard leg measured before -- with the failure on it."""
    result = run_two_ranks({"EM2_SHARDED_MIN_CELLS": "1000", "EM2_BENCH_TEST_FAIL_RANK": "1", "EM2_BENCH_STAGE_LIMIT": "25"}, cells=30000,
                           port=29636)
    assert result["config"]["scan"] == "row-shards" and result["value"] > 0 and result["parity_check"]["after_timing_rows"] > 0
    assert result["sharded_symmetric_leg"]["status"] == "did not complete"


def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher (the shape of the driver's 1-GPU command): the parent starts
    torch.distributed.run as a child before touching the GPU, relays rank 0's one line and returns the child's exit code."""
    env = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"EM2_BENCH_SHARE_DEVICE": "1", "EM2_BENCH_BACKEND": "gloo", "EM2_BLOCKS_PER_CU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                "EM2_SHARDED_MIN_CELLS": "1000"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "30000", "--genes", "3000",
           "--no-cpu-baseline", "--check-rows", "96"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    result = json.loads(lines[0])
    assert result["n_gpus"] == 2 and result["collective_check"]["ranks_counted_by_all_reduce"] == 2
    assert result["row_shard_leg"]["value"] > 0 and result["row_shard_leg"]["parity_check"]["fsp4_rows"] > 0


# ---- BASELINE configs[3] and configs[4] on several ranks (bench.py --workload fsp5 / chain --gpus N) ----

@pytest.mark.parametrize("ranks,cells", [(2, 20000), (3, 20011)])
def test_fsp5_leg_on_several_ranks(ranks, cells):
    """findSimilarPairs5 with the cells sharded by id range and the slice tables replicated: every rank's sampled cells equal
    the oracle's (the rank exits non-zero otherwise), no data-path collective."""
    result = run_two_ranks({}, cells=cells, port=29641 + ranks, ranks=ranks,
                           extra_args=["--workload", "fsp5", "--slice-length", "14", "--fsp5-check-cells", "512"])
    assert result["n_gpus"] == ranks and result["value"] > 0 and result["scaling"] == "strong"
    assert result["parity_check"]["ranks_that_passed_their_gate"] == ranks and result["parity_check"]["fsp5_cells_rank0"] > 0
    assert result["collective_check"]["ranks_counted_by_all_reduce"] == ranks
    assert result["collectives_in_a_step"].startswith("none")


@pytest.mark.parametrize("ranks,cells", [(2, 20000), (3, 20011)])
def test_chain_leg_on_several_ranks(ranks, cells):
    """findSimilarPairs4 by row shards -> all-gather of the pairs -> createCellGraph + label propagation on rank 0: the whole
    edge list, every label and sampled gathered rows equal the oracle's."""
    result = run_two_ranks({}, cells=cells, port=29651 + ranks, ranks=ranks, extra_args=["--workload", "chain", "--k", "30"])
    assert result["n_gpus"] == ranks and result["value"] > 0
    check = result["parity_check"]
    assert check["edges"] > 0 and check["labels"] == cells and check["gathered_rows_against_oracle"] > 0
    assert result["collectives_in_a_step"]["all_gather_pairs_bytes"] > 0
    assert result["phases_ms_rank0"]["gather_pairs"] > 0


def test_eight_ranks_dry_run_of_every_leg():
    """First-contact readiness: EIGHT ranks on the one GPU at 131 072 cells, gloo for the collectives -- the
    row-shard leg, the sharded symmetric leg (all_to_all among eight), and the fsp5 and chain legs -- every rank's gate against
    the oracle.  Not a measurement: eight processes share a device."""
    result = run_two_ranks({"EM2_SHARDED_MIN_CELLS": "1000"}, cells=131072, port=29661, ranks=8, timeout=1500)
    assert result["config"]["scan"] == "sharded-symmetric" and result["n_gpus"] == 8
    assert result["parity_check"]["fsp4_rows"] > 0 and result["row_shard_leg"]["parity_check"]["fsp4_rows"] > 0
    assert result["collective_check"]["rank_ids_gathered"] == list(range(8))
    result = run_two_ranks({}, cells=131072, port=29662, ranks=8, timeout=1500,
                           extra_args=["--workload", "fsp5", "--slice-length", "16", "--fsp5-check-cells", "512"])
    assert result["parity_check"]["ranks_that_passed_their_gate"] == 8
    result = run_two_ranks({}, cells=131072, port=29663, ranks=8, timeout=1500, extra_args=["--workload", "chain", "--k", "30"])
    assert result["parity_check"]["labels"] == 131072 and result["parity_check"]["gathered_rows_against_oracle"] > 0
