"""Joins the `runs` of several `python -m tests.psnr_ensemble reference ... --resume` processes that
computed disjoint ranges of ONE protocol's seeds in parallel (seed k re-seeds torch and numpy at its
start and shares nothing with the other seeds, so a seed's curve does not depend on the process it
ran in -- checked from the other side: the HIP halves run all 24 seeds in ONE process and stay
within 0.0011 dB of every reference seed at every report, the first seeds of the parts included).
Build container only; the output is the fixture a single process would have written.

    python tests/golden/merge_ensemble_parts.py OUT.json BASE.json PART.json:FIRST [PART.json:FIRST ...]

BASE holds the protocol and runs 0 .. n-1; every PART contributes its runs from index FIRST on (the
placeholder entries below FIRST that made `--resume` start there are dropped)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.psnr_ensemble import stats_of      # noqa: E402


def main():
    out_path, base_path, parts = sys.argv[1], sys.argv[2], sys.argv[3:]
    with open(base_path) as f:
        doc = json.load(f)
    planned = doc["protocol"]["seeds"]
    runs = {r["seed"]: r for r in doc["runs"]}
    for item in parts:
        path, first = item.rsplit(":", 1)
        with open(path) as f:
            part = json.load(f)
        for r in part["runs"][int(first):]:
            assert not r.get("placeholder") and r["seed"] in planned and r["seed"] not in runs, r["seed"]
            runs[r["seed"]] = r
        assert part.get("threads", doc.get("threads")) == doc.get("threads"), "thread counts differ"
    ordered = [runs[s] for s in planned if s in runs]
    assert [r["seed"] for r in ordered] == planned[:len(ordered)], "the runs must be a prefix of the planned seeds"
    doc["runs"] = ordered
    doc["complete"] = len(ordered) == len(planned)
    doc["computed_in_parallel_parts"] = [os.path.basename(base_path)] + [os.path.basename(p.rsplit(":", 1)[0]) + " from run " + p.rsplit(":", 1)[1] for p in parts]
    doc.update(stats_of(ordered))
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(len(ordered), "runs,", "complete" if doc["complete"] else "incomplete", doc["final_val_psnr"]["mean"], doc["final_val_psnr"]["stderr"])


if __name__ == "__main__":
    main()
