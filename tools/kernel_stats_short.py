"""Prints a rocprofv3 *kernel_stats.csv with short kernel names:  python3 tools/kernel_stats_short.py FILE [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:limit]:
    name = r["Name"].replace("em2::(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70] if "rocprim" not in name else "rocprim:" + name.split("detail::")[-1][:50]
    print("%-72s calls %5s total %10.3f ms avg %10.3f ms %6.2f%%" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                  float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
