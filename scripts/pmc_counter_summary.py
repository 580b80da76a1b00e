"""Folds a rocprofv3 `--pmc ... --output-format csv` counter_collection file into a per-kernel
JSON (mean of each counter over the launches of every ffn:: kernel, plus mean duration):

    python scripts/pmc_counter_summary.py in_counter_collection.csv out.json "note" [commit]

Derived ratios (when the counters are present): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
SQ_BUSY_CYCLES (both in cycles, summed over the SQs that were busy), valu_inst_per_mfma_inst,
lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE or / SQ_WAVE_CYCLES*4.
"""

import collections
import csv
import json
import re
import sys


def main(src, out, note, commit=None):
    acc = collections.OrderedDict()
    with open(src, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            if not name.startswith("ffn::"):
                continue
            k = acc.setdefault(name, {"dispatches": set(), "ns": {}, "counters": collections.OrderedDict()})
            k["dispatches"].add(row["Dispatch_Id"])
            k["ns"][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            c = k["counters"].setdefault(row["Counter_Name"], [0, 0.0])
            c[0] += 1
            c[1] += float(row["Counter_Value"])
    kernels = collections.OrderedDict()
    for name, k in acc.items():
        n = len(k["dispatches"])
        mean = {c: v[1] / n for c, v in k["counters"].items()}
        entry = {"launches": n, "avg_us_under_pmc": round(sum(k["ns"].values()) / n / 1e3, 1),
                 "counters": {c: round(v, 1) for c, v in mean.items()}}
        if mean.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
            entry["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / mean["SQ_BUSY_CYCLES"], 4)
        if mean.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in mean:
            entry["valu_insts_per_mfma_inst"] = round(
                (mean["SQ_INSTS_VALU"] - mean["SQ_INSTS_MFMA"]) / mean["SQ_INSTS_MFMA"], 4)
        if mean.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in mean:
            entry["valu_active_frac_of_wave_cycles"] = round(
                mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"], 4)
        if mean.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in mean:
            entry["issue_stall_frac_of_wave_cycles"] = round(
                mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"], 4)
        if mean.get("SQ_WAVE_CYCLES") and "SQ_LDS_BANK_CONFLICT" in mean:
            # SQ_LDS_BANK_CONFLICT counts cycles, SQ_WAVE_CYCLES quad-cycles (MI355X_MICROARCH.md)
            entry["lds_bank_conflict_frac_of_wave_cycles"] = round(
                mean["SQ_LDS_BANK_CONFLICT"] / (4 * mean["SQ_WAVE_CYCLES"]), 5)
        kernels[name] = entry
    doc = {"note": note, "commit": commit, "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
