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
    text = open(os.path.join(CSRC, source)).read()
    for old, new in scope["SUBS"]:
        assert old in text, old
        text = text.replace(old, new)
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(OUT, "%s_%s" % (name, source))
    with open(src, "w") as f:
        f.write(text)
    obj = src.replace(".hip", ".o")
    sys.path.insert(0, ROOT)
    from fourier_feature_nets_amd.build import SOURCES          # the product's per-file flags
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    flags += SOURCES.get(source, [])
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", src, "-o", obj], check=True)
    others = [os.path.join(CSRC, "build", o) for o in os.listdir(os.path.join(CSRC, "build"))
              if o.endswith(".o") and o != source.replace(".hip", ".o")]
    lib = os.path.join(OUT, "libffn_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + others + [obj],
                   check=True)
    print("built", lib)


if __name__ == "__main__":
    main(*sys.argv[1:])
