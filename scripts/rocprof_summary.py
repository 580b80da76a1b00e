"""Turns a rocprofv3 results .db (rocpd sqlite, --kernel-trace --stats) into a small CSV that
can be committed under profiles/:  python scripts/rocprof_summary.py in.db out.csv"""

import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage "
                           "from top_kernels order by total_duration desc"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, total, avg, pct in rows:
            w.writerow([short(name), calls, int(total), int(avg), round(pct, 3)])
    print("wrote", out_path, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
