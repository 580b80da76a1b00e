"""Builds scripts/probes/variants/libffn_dbg.so: the product library with s_memtime stamps
injected into the fused MLP kernels (one wave of one workgroup logs cycle counters into a
device array read back through an extra entry point, ffn_dbg_read).

    python scripts/probes/make_dbg_library.py steps      # for step_timeline.py
    python scripts/probes/make_dbg_library.py epilogue   # for epilogue_timeline.py
    FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_dbg.so python scripts/probes/step_timeline.py

The instrumented source is derived from csrc/mlp.hip by text substitution (asserted), so it
follows the product kernel; nothing of this is part of the shipped library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fourier_feature_nets_amd", "csrc")
OUT = os.path.join(ROOT, "scripts", "probes", "variants")
STAMP = "if (dbg_on && w.lane == 0) { g_dbg[g_dbg_n++] = %s__builtin_readcyclecounter(); }"


def sub(text, old, new):
    assert old in text, old
    return text.replace(old, new)


def instrument(mode):
    s = open(os.path.join(CSRC, "mlp.hip")).read()
    s = sub(s, "namespace ffn {", "__device__ long long g_dbg[8192];\n__device__ int g_dbg_n;\nnamespace ffn {")
    head = "    f32x16 acc[OT];\n    if (MODE == kBackward) {"
    s = sub(s, head, "    const bool dbg_on = blockIdx.x == 3 && (threadIdx.x >> 6) == 1 && w.block < 3 * 1024 + 16;\n    "
            + STAMP % "" + "\n" + head)
    kloops_done = "    const int ot_next = next != nullptr ? next->out_tiles / TW : 0;"
    s = sub(s, kloops_done, "    " + STAMP % "" + "\n" + kloops_done)
    end = "    if (WIDE) team_barrier();        // the step's output is in the slab\n}"
    if mode == "steps":
        seg = "        const f32x4* xa = w.act + w.lane;\n        x0 = xa[0];"
        s = sub(s, seg, "        " + STAMP % "-" + "\n" + seg)      # negative = a K loop starts
        s = sub(s, end, "    " + STAMP % "" + "\n}")
    elif mode == "tiles":          # one stamp per output tile of the epilogue loop
        tile = "        if (o >= OT) continue;\n"
        s = sub(s, tile, tile + "        " + STAMP % "" + "\n")
        s = sub(s, end, "    " + STAMP % "" + "\n}")
    else:
        init_done = "    const f32x4* wp = reinterpret_cast<const f32x4*>(packed_w + L.w_off) + half * OT * 64;   // uniform"
        s = sub(s, init_done, "    " + STAMP % "" + "\n" + init_done)
        oloop = "    if (WIDE) team_barrier();        // every K loop of this step has finished reading the slab"
        s = sub(s, oloop, "    " + STAMP % "" + "\n" + oloop)
        masks = "    if (MODE == kTrainFwd && L.relu && L.mask_slot >= 0 && w.active)"
        s = sub(s, masks, "    " + STAMP % "" + "\n" + masks)
        s = sub(s, end, "    " + STAMP % "" + "\n}")
    s += '''
extern "C" int ffn_dbg_read(long long* out, int reset) {
    int n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_dbg_n), sizeof(int));
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(long long) * 8192);
    if (reset) { int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_n), &z, sizeof(int)); }
    return n;
}
'''
    return s


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "steps"
    assert mode in ("steps", "epilogue", "tiles")
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(OUT, "mlp_dbg.hip")
    with open(src, "w") as f:
        f.write(instrument(mode))
    hipcc = "/opt/rocm/bin/hipcc"
    obj = os.path.join(OUT, "mlp_dbg.o")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    subprocess.run([hipcc] + flags + ["-c", src, "-o", obj], check=True)
    others = [os.path.join(CSRC, "build", o) for o in os.listdir(os.path.join(CSRC, "build"))
              if o.endswith(".o") and o != "mlp.o"]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                    os.path.join(OUT, "libffn_dbg.so")] + others + [obj], check=True)
    print("built", os.path.join(OUT, "libffn_dbg.so"), "mode", mode)


if __name__ == "__main__":
    main()
