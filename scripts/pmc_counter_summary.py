"""Folds a rocprofv3 `--pmc ... --output-format csv` counter_collection file into a per-kernel
JSON (mean of each counter over the launches of every ffn:: kernel, plus mean duration):

    python scripts/pmc_counter_summary.py in_counter_collection.csv out.json "note" [commit]

Derived ratios (when the counters are present):
  mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines):
      the share of the kernel's busy cycles in which a SIMD's matrix pipe is executing.  The
      normalisation is checked on the data itself: MFMA_BUSY / 1024 equals SQ_INSTS_MFMA / 1024 x 64
      cycles for v_mfma_f32_32x32x2_f32, and SQ_BUSY / 32 / duration gives a 2.1-2.2 GHz clock.
  effective_clock_ghz = SQ_BUSY_CYCLES / 32 / duration
  valu_insts_per_mfma_inst = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA
  *_frac_of_wave_cycles: SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES
  lds_bank_conflict: SQ_LDS_BANK_CONFLICT cycles over 4 x SQ_WAVE_CYCLES (quad-cycles)
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
            entry["mfma_busy_frac"] = round((mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) /
                                            (mean["SQ_BUSY_CYCLES"] / 32), 4)
        if mean.get("SQ_BUSY_CYCLES"):
            entry["effective_clock_ghz"] = round(mean["SQ_BUSY_CYCLES"] / 32 /
                                                 (sum(k["ns"].values()) / n), 3)
        if mean.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in mean:
            entry["valu_insts_per_mfma_inst"] = round(
                (mean["SQ_INSTS_VALU"] - mean["SQ_INSTS_MFMA"]) / mean["SQ_INSTS_MFMA"], 4)
        if mean.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in mean:
            entry["valu_active_frac_of_wave_cycles"] = round(
                mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"], 4)
        if mean.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in mean:
            entry["issue_stall_frac_of_wave_cycles"] = round(
                mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"], 4)
        if mean.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in mean:
            entry["wait_any_frac_of_wave_cycles"] = round(mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"], 4)
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
