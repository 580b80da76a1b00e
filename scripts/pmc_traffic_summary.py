"""Folds the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu/profile.sh into profiles/*.json:

    python scripts/pmc_traffic_summary.py gpurun_out/traffic profiles/r02_hbm_traffic.json [commit]

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md,
HBM section): FETCH_SIZE counts TCC_EA0_RDREQ x 64 B while wide streaming reads are 128-B
requests, so read bytes = 2 x FETCH_SIZE; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024,
averaged over the launches of each kernel.
"""

import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.OrderedDict()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            if not name.startswith("ffn::"):
                continue
            s = acc.setdefault(name, [0, 0.0])
            s[0] += 1
            s[1] += float(row["Counter_Value"])
    return acc


def main(src, out, commit=None, rays=65536, samples=64, command=None):
    fetch = per_kernel(src + "/fetch_counter_collection.csv", "FETCH_SIZE")
    write = per_kernel(src + "/write_counter_collection.csv", "WRITE_SIZE")
    kernels = collections.OrderedDict()
    for name, (n, total) in fetch.items():
        wn, wtotal = write.get(name, (0, 0.0))
        f_kb = total / n
        w_kb = wtotal / wn if wn else 0.0
        kernels[name] = {"launches": n, "FETCH_SIZE_KB": round(f_kb, 1),
                         "WRITE_SIZE_KB": round(w_kb, 1),
                         "hbm_bytes": int((2 * f_kb + w_kb) * 1024)}
    command = command or ("python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-shape "
                          "--no-config3 --no-config5 --no-skip-leg --no-bf16-leg")
    doc = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on "
                   "`" + command + "` (scripts/gpu/profile.sh), MI355X; "
                   "KB per launch averaged over launches; hbm_bytes = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide "
                   "streaming reads, MI355X_MICROARCH.md section HBM)",
           "commit": commit, "config": {"rays": rays, "samples": samples}, "kernels": kernels}
    step = [k for k in kernels if "mlp_forward" in k or "mlp_backward" in k or "wgrad_unit" in k]
    doc["training_step_mlp_kernels_hbm_bytes"] = sum(kernels[k]["hbm_bytes"] for k in step)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None,
         command=sys.argv[4] if len(sys.argv) > 4 else None)
