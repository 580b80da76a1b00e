"""Builds scripts/probes/variants/libffn_<name>.so from csrc/mlp.hip with text substitutions
applied -- timing experiments ("what does this piece of the epilogue cost?") without touching
the product source.  The variants may compute WRONG results; they are for timing only.

    python scripts/probes/make_variant.py <name> <python file with SUBS = [(old, new), ...]>
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fourier_feature_nets_amd", "csrc")
OUT = os.path.join(ROOT, "scripts", "probes", "variants")


def main(name, subs_path, source="mlp.hip"):
    scope = {}
    exec(open(subs_path).read(), scope)
    subs = scope["SUBS"]
    if not isinstance(subs, dict):          # one source file (argv) or several (a dict in the file)
        subs = {source: subs}
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from fourier_feature_nets_amd.build import SOURCES          # the product's per-file flags
    objs = []
    for src_name, pairs in subs.items():
        text = open(os.path.join(CSRC, src_name)).read()
        for old, new in pairs:
            assert old in text, (src_name, old)
            text = text.replace(old, new)
        src = os.path.join(OUT, "%s_%s" % (name, src_name))
        with open(src, "w") as f:
            f.write(text)
        obj = src.replace(".hip", ".o")
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
        flags += SOURCES.get(src_name, [])
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", src, "-o", obj], check=True)
        objs.append(obj)
    replaced = {s.replace(".hip", ".o") for s in subs}
    others = [os.path.join(CSRC, "build", o) for o in os.listdir(os.path.join(CSRC, "build"))
              if o.endswith(".o") and o not in replaced]
    lib = os.path.join(OUT, "libffn_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + others + objs,
                   check=True)
    print("built", lib)


if __name__ == "__main__":
    main(*sys.argv[1:])
