# A/B of the last exact-f32 levers on ONE box, interleaved twice: the product library against
# timing-only variants (scripts/probes/f32_levers/): per-kernel ms of the headline step
mkdir -p gpurun_out/r4lev
python - <<'PY'
import json, os, subprocess, sys
libs = ["new", "f32_stores_last", "f32_stores_mid", "f32_nomask", "f32_nosign"]
rows = []
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "new":
            env["FFN_HIP_LIBRARY"] = os.path.join(os.getcwd(), "scripts/probes/variants/libffn_%s.so" % lib)
        r = subprocess.run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-config3",
                            "--no-config5", "--no-skip-leg", "--no-bf16-leg", "--no-target-shape", "--no-render"],
                           env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "failed", r.stderr[-300:]); continue
        b = json.loads(line[-1])
        row = {"library": lib, "rep": rep, "ms_per_step": round(b["ms_per_step"], 3)}
        row.update({k.split("_kernel")[0]: v["avg_ms"] for k, v in b["kernels"].items()})
        rows.append(row); print(row, flush=True)
json.dump({"what": "exact-f32 lever A/B, headline step (65 536 rays x 64 samples), one box, interleaved", "rows": rows,
           "commit": open(".git_head").read().split()[0] if os.path.exists(".git_head") else None},
          open("gpurun_out/r4lev/f32_lever_ab.json", "w"), indent=1)
PY
