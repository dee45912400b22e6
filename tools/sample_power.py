#!/usr/bin/env python3
"""Samples socket power and shader clock of every GPU of the host from sysfs (hwmon: power1_input in microwatts, freq1_input in
Hz, power1_cap) every few milliseconds while a command runs, and prints per-card statistics: the evidence behind "the scan kernel
is power-bound" (DESIGN.md 3.1.4).  The card of the process is the one whose power moves.
    python3 tools/sample_power.py python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra"""
import glob
import json
import subprocess
import sys
import time


def read(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def main():
    cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    samples = {c: [] for c in cards}
    child = subprocess.Popen(sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    while child.poll() is None:
        now = time.time()
        for c in cards:
            samples[c].append((now, read(c + "/power1_input"), read(c + "/freq1_input")))
        time.sleep(0.004)
    out = child.stdout.read()
    report = {}
    for c in cards:
        power = [p / 1e6 for _, p, _ in samples[c] if p is not None]
        clock = [f / 1e9 for _, _, f in samples[c] if f is not None]
        if not power:
            continue
        top = sorted(power)[int(0.9 * len(power)):]                      # the busiest tenth of the samples
        busy = [f for (_, p, f) in samples[c] if p is not None and f is not None and p / 1e6 >= top[0]]
        report[c.split("/")[4]] = {"samples": len(power), "power_w_mean": round(sum(power) / len(power), 1), "power_w_max": round(max(power), 1),
                                   "power_w_busiest_tenth_mean": round(sum(top) / len(top), 1), "power_cap_w": (read(c + "/power1_cap") or 0) / 1e6,
                                   "sclk_ghz_in_busiest_tenth_mean": round(sum(busy) / len(busy) / 1e9, 3) if busy else None,
                                   "sclk_ghz_min": round(min(clock), 3) if clock else None, "sclk_ghz_max": round(max(clock), 3) if clock else None}
    print(json.dumps(report, indent=1))
    line = [l for l in out.splitlines() if l.startswith("{")]
    if line:
        d = json.loads(line[-1])
        print(json.dumps({"ms_per_step": d.get("ms_per_step"), "roofline": {k: d["roofline"].get(k) for k in ("kernel_ms", "frac", "clock_ghz")} if d.get("roofline") else None}))


if __name__ == "__main__":
    main()
