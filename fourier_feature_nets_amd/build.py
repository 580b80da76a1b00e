"""Builds libffn_hip.so (gfx950) in-tree with hipcc.  No CMake, no JIT cache.

    python -m fourier_feature_nets_amd.build [--force]

The shared library lands next to this file so that it travels with the source tree.
"""

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "libffn_hip.so")

# name -> extra flags.  The sampling kernels must not contract a*b+c into an FMA: the
# reference's ATen ops round every multiply and add separately.
SOURCES = {
    "abi.hip": [],
    "rays.hip": ["-ffp-contract=off"],
    # (focus.hip: the op-by-op rounding is switched on per function in focus_terms.h -- a
    # file-wide -ffp-contract=off would also change how the math library's own code is compiled,
    # and the fused coarse-pass kernel in mlp.hip has to produce the same bits)
    "focus.hip": [],
    "composite.hip": [],
    "encode.hip": [],
    "optim.hip": ["-ffp-contract=off"],
    "mlp.hip": [],
    "mlp_bf16.hip": [],
    "mlp_bf16_bwd.hip": [],
    # (two waves per SIMD: packed-f32 VALU instructions collide with the other wave's matrix
    # instructions -- keep the compiler from re-packing the unpacked feature / split arithmetic)
    "mlp_bf16_ws.hip": ["-fno-slp-vectorize"],
    "mlp_bf16_mv.hip": ["-fno-slp-vectorize"],
    "wgrad.hip": [],
    "wgrad_bf16.hip": [],
    "wgrad_bf16x6.hip": [],
    "occupancy.hip": [],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
          "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    stamp = os.path.getmtime(target)
    return any(os.path.getmtime(d) > stamp for d in deps)


def build_library(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "ffn_hip.h"))
    obj_dir = os.path.join(CSRC, "build")
    os.makedirs(obj_dir, exist_ok=True)
    jobs, objects = [], []
    for name, extra in SOURCES.items():
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            # a kernel file that went missing must fail the BUILD, not surface later as an
            # AttributeError on a symbol the smaller library does not export
            raise RuntimeError("source file listed in SOURCES is missing: %s" % src)
        obj = os.path.join(obj_dir, name.replace(".hip", ".o"))
        objects.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + COMMON + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (res.stdout, res.stderr))
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if jobs or force or _stale(LIB_PATH, objects):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objects)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
