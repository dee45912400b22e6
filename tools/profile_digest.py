"""Turns one tools/profile_bench.sh output directory into the digest committed under profiles/: per kernel of this library
the launches, average duration (kernel trace), fabric bytes per launch (FETCH_SIZE x 2 for the 16-byte-per-lane loads these
kernels issue -- MI355X_MICROARCH.md, HBM section -- plus WRITE_SIZE) and the figures derived from the SQ / GRBM counters.

    python3 tools/profile_digest.py gpurun_out/<dir> > profiles/rNN_pmc_bench_<what>.json

`config` of the digest is what bench.py itself printed under the trace (the scalar entries of the line's `config` plus
`n_gpus`): bench.profiled_traffic() accepts a digest only for the configuration it is running, so the digest must say which
one it is of -- never an argument somebody may forget.  `--rewrite-config FILE...` re-derives `config` of committed digests
from their embedded line in place (no GPU needed).
"""
import csv
import json
import os
import sys

SIMDS = 1024.0          # 256 CUs x 4


def short(name):
    name = name.replace("em2::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def config_of_line(bench_line):
    """The configuration a bench.py line was measured on: the scalar entries of its `config` (cells, genes, lsh_count, k, ...;
    not the prose `workload`) and `n_gpus`."""
    if not isinstance(bench_line, dict):
        return {}
    config = {key: value for key, value in bench_line.get("config", {}).items()
              if key != "workload" and isinstance(value, (int, float, str, bool))}
    if "n_gpus" in bench_line:
        config["n_gpus"] = bench_line["n_gpus"]
    return config


def rewrite_config(paths):
    for path in paths:
        with open(path) as f:
            digest = json.load(f)
        config = config_of_line(digest.get("bench_line_under_trace"))
        if not config:
            raise SystemExit("%s: no bench line under trace to take the configuration from" % path)
        digest["config"] = config
        with open(path, "w") as f:
            json.dump(digest, f, indent=1)
            f.write("\n")
        print(path, config)


def main():
    if sys.argv[1] == "--rewrite-config":
        return rewrite_config(sys.argv[2:])
    directory = sys.argv[1]
    pmc = json.load(open(os.path.join(directory, "pmc_summary.json")))
    durations = {}
    for row in csv.DictReader(open(os.path.join(directory, "kernel_stats.csv"))):
        durations[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]))
    kernels = {}
    for name, data in pmc.items():
        if "rocprim" in name or name.startswith("at::") or name.startswith("__amd") or not name:
            continue
        t = data["totals"]
        launches = float(data["dispatches_per_pass"][0]) if data["dispatches_per_pass"] else 0.0
        if not launches:
            continue
        calls, average_ns = durations.get(name, (0, 0.0))
        entry = {"launches_per_pass": launches, "average_ms_in_kernel_trace": average_ns / 1e6 if average_ns else None}
        if "FETCH_SIZE" in t:
            entry["fabric_fetch_bytes_per_launch"] = 2.0 * t["FETCH_SIZE"] * 1024.0 / launches
            entry["fabric_fetch_bytes_per_launch_as_reported"] = t["FETCH_SIZE"] * 1024.0 / launches
        if "WRITE_SIZE" in t:
            entry["fabric_write_bytes_per_launch"] = t["WRITE_SIZE"] * 1024.0 / launches
        if "GRBM_GUI_ACTIVE" in t:
            cycles = t["GRBM_GUI_ACTIVE"] / 8.0 / launches
            entry["cycles_per_launch"] = cycles
            if average_ns:
                entry["clock_GHz_from_counters"] = cycles / average_ns
            if "SQ_VALU_MFMA_BUSY_CYCLES" in t and t["SQ_VALU_MFMA_BUSY_CYCLES"]:
                entry["mfma_busy_fraction"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / launches / (SIMDS * cycles)
            if "SQ_ACTIVE_INST_VALU" in t:
                entry["valu_busy_fraction"] = 4.0 * t["SQ_ACTIVE_INST_VALU"] / launches / (SIMDS * cycles)
        for counter, label in (("SQ_INSTS_VALU", "valu_instructions_per_launch"), ("SQ_INSTS_SALU", "salu_instructions_per_launch"),
                               ("SQ_INSTS_LDS", "lds_instructions_per_launch"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles_per_launch")):
            if counter in t:
                entry[label] = t[counter] / launches
        if t.get("SQ_WAVE_CYCLES"):
            entry["wave_cycles_waiting_fraction (s_waitcnt / barrier)"] = t.get("SQ_WAIT_ANY", 0.0) / t["SQ_WAVE_CYCLES"]
            entry["wave_cycles_issue_stall_fraction"] = t.get("SQ_WAIT_INST_ANY", 0.0) / t["SQ_WAVE_CYCLES"]
        kernels[name] = entry
    bench_line = None
    try:
        text = open(os.path.join(directory, "bench_under_trace.json")).read().strip().split("\n")[-1]
        bench_line = json.loads(text)
    except Exception:           # noqa: BLE001
        pass
    config = config_of_line(bench_line)
    if not config:
        raise SystemExit("profile_digest: %s holds no bench line under trace: a digest without its configuration is unusable" % directory)
    out = {
        "_what": "tools/profile_bench.sh: rocprofv3 --kernel-trace --stats of bench.py, then one --pmc pass per counter group "
                 "(FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY "
                 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY | GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT "
                 "SQ_ACTIVE_INST_LDS), the trace domains never in a --pmc pass; digest by tools/profile_digest.py.  MI355X, 1 GPU.",
        "_caveat": "FETCH_SIZE counts the L2's memory-side (fabric) read requests, Infinity-Cache hits included; for 16-byte-per-lane "
                   "loads gfx950 reports half of the bytes (MI355X_MICROARCH.md, HBM section): fabric_fetch_bytes_per_launch doubles "
                   "the reported figure.  Profiled passes run at a lower clock than unprofiled ones.",
        "config": config,
        "kernels": kernels,
        "bench_line_under_trace": bench_line,
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
