"""Builds scripts/probes/variants/libffn_<name>.so from the csrc sources with text substitutions
applied to ANY of them -- timing experiments without touching the product source (variants may be
wrong; they are for timing unless the experiment says otherwise).

    python scripts/probes/make_variants.py <python file defining VARIANTS = {name: {source: [(old, new), ...]}}>
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fourier_feature_nets_amd", "csrc")
OUT = os.path.join(ROOT, "scripts", "probes", "variants")
sys.path.insert(0, ROOT)
from fourier_feature_nets_amd.build import COMMON, SOURCES, build_library  # noqa: E402


def build(name, subs):
    objs = []
    for source, extra in SOURCES.items():
        stock = os.path.join(CSRC, "build", source.replace(".hip", ".o"))
        if source not in subs:
            objs.append(stock)
            continue
        text = open(os.path.join(CSRC, source)).read()
        for old, new in subs[source]:
            assert old in text, (source, old)
            text = text.replace(old, new)
        src = os.path.join(OUT, "%s_%s" % (name, source))
        with open(src, "w") as f:
            f.write(text)
        obj = src.replace(".hip", ".o")
        subprocess.run(["/opt/rocm/bin/hipcc"] + COMMON + extra + ["-c", src, "-o", obj], check=True)
        objs.append(obj)
    lib = os.path.join(OUT, "libffn_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def main(path):
    scope = {}
    exec(open(path).read(), scope)
    os.makedirs(OUT, exist_ok=True)
    build_library(verbose=False)          # the stock objects
    with ThreadPoolExecutor(max_workers=4) as pool:
        for lib in pool.map(lambda kv: build(*kv), scope["VARIANTS"].items()):
            print("built", lib)


if __name__ == "__main__":
    main(sys.argv[1])
