"""Folds rocprofv3's `<name>_kernel_stats.csv` (--kernel-trace --stats --output-format csv) into the
compact per-kernel CSV kept under profiles/ (same columns as scripts/rocprof_summary.py writes):
    python scripts/kernel_stats_csv.py in_kernel_stats.csv out.csv"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main(src, out):
    with open(src, newline="") as f:
        rows = list(csv.DictReader(f))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], int(int(r["TotalDurationNs"]) / 1e3),
                        round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"]])
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
