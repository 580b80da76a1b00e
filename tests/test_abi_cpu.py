"""The C-ABI shared library loads without a GPU and exports every declared symbol."""

import ctypes
import os

import pytest

from fourier_feature_nets_amd import _lib


def test_library_is_built_and_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from fourier_feature_nets_amd.build import build_library
        build_library(verbose=False)
    names = _lib.declared_symbols()
    assert "ffn_mlp_forward" in names and "ffn_composite_fwd" in names and len(names) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), name
    lib.ffn_abi_version.restype = ctypes.c_int
    assert lib.ffn_abi_version() == _lib.ABI_VERSION


def test_plan_struct_sizes_match_header():
    """ctypes mirrors of the plan structs have the C layout (checked against a tiny C probe
    compiled with gcc from the real header)."""
    import subprocess
    import tempfile
    from fourier_feature_nets_amd import mlp_engine as me
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ffn_hip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu\n", sizeof(ffn_encoding), sizeof(ffn_layer),
             sizeof(ffn_mlp_plan), offsetof(ffn_mlp_plan, layer), offsetof(ffn_mlp_plan, num_layers),
             offsetof(ffn_mlp_plan, slot_offset));
      return 0; }'''
    with tempfile.TemporaryDirectory() as tmp:
        c_path = os.path.join(tmp, "probe.c")
        with open(c_path, "w") as f:
            f.write(src)
        exe = os.path.join(tmp, "probe")
        inc = os.path.dirname(_lib.HEADER_PATH)
        subprocess.run(["gcc", "-I", inc, c_path, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    got = [ctypes.sizeof(me.FfnEncoding), ctypes.sizeof(me.FfnLayer), ctypes.sizeof(me.FfnMlpPlan),
           me.FfnMlpPlan.layer.offset, me.FfnMlpPlan.num_layers.offset,
           me.FfnMlpPlan.slot_offset.offset]
    assert [int(v) for v in out] == got


def test_ops_refuse_cpu_tensors():
    import torch
    from fourier_feature_nets_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops._dev(torch.zeros(3))
