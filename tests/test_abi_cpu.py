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
    """ctypes mirrors of the ABI structs have the C layout (checked against a tiny C probe
    compiled with gcc from the real header)."""
    import subprocess
    import tempfile
    from fourier_feature_nets_amd import mlp_engine as me
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ffn_hip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu ", sizeof(ffn_render_rays),
             offsetof(ffn_render_rays, ray_base), offsetof(ffn_render_rays, num_samples),
             offsetof(ffn_render_rays, t_values), sizeof(ffn_occupancy),
             offsetof(ffn_occupancy, resolution), sizeof(ffn_render_out),
             offsetof(ffn_render_out, image), offsetof(ffn_render_out, pixel_offset));
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ffn_encoding),
             sizeof(ffn_step), sizeof(ffn_mlp_chain), offsetof(ffn_mlp_chain, step),
             offsetof(ffn_mlp_chain, num_steps), offsetof(ffn_mlp_chain, bias_floats),
             offsetof(ffn_mlp_chain, slot_offset), offsetof(ffn_step, w_off),
             offsetof(ffn_step, save_enc_slot), sizeof(ffn_wgrad_unit), sizeof(ffn_wgrad_segment),
             sizeof(ffn_reduce_job));
      printf("%zu %zu %zu\n", sizeof(ffn_pack_job), offsetof(ffn_pack_job, kind),
             offsetof(ffn_pack_job, dst_cs));
      return 0; }'''
    with tempfile.TemporaryDirectory() as tmp:
        c_path = os.path.join(tmp, "probe.c")
        with open(c_path, "w") as f:
            f.write(src)
        exe = os.path.join(tmp, "probe")
        inc = os.path.dirname(_lib.HEADER_PATH)
        subprocess.run(["gcc", "-I", inc, c_path, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    got = [ctypes.sizeof(me.FfnRenderRays), me.FfnRenderRays.ray_base.offset,
           me.FfnRenderRays.num_samples.offset, me.FfnRenderRays.t_values.offset,
           ctypes.sizeof(me.FfnOccupancy), me.FfnOccupancy.resolution.offset,
           ctypes.sizeof(me.FfnRenderOut), me.FfnRenderOut.image.offset,
           me.FfnRenderOut.pixel_offset.offset,
           ctypes.sizeof(me.FfnEncoding), ctypes.sizeof(me.FfnStep), ctypes.sizeof(me.FfnMlpChain),
           me.FfnMlpChain.step.offset, me.FfnMlpChain.num_steps.offset,
           me.FfnMlpChain.bias_floats.offset, me.FfnMlpChain.slot_offset.offset,
           me.FfnStep.w_off.offset, me.FfnStep.save_enc_slot.offset, ctypes.sizeof(me.FfnWgradUnit),
           ctypes.sizeof(me.FfnWgradSegment), ctypes.sizeof(me.FfnReduceJob),
           ctypes.sizeof(me.FfnPackJob), me.FfnPackJob.kind.offset, me.FfnPackJob.dst_cs.offset]
    assert [int(v) for v in out] == got, (out, got)


def test_ops_refuse_cpu_tensors():
    import torch
    from fourier_feature_nets_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops._dev(torch.zeros(3))


def _chain(steps, bias_floats=2000, wide=0, head=None):
    """A chain descriptor with only the fields `ffn_mlp_bf16x6_organisation` looks at: per step
    (act_groups, aux_groups, out_tiles); `head` = (lg_col, lg_n) of step 0 -- which logits columns a
    backward chain's first step reads."""
    from fourier_feature_nets_amd import mlp_engine as me
    chain = me.FfnMlpChain()
    chain.num_steps = len(steps)
    chain.bias_floats = bias_floats
    chain.wide = wide
    for i, (act, aux, tiles) in enumerate(steps):
        chain.step[i].act_groups, chain.step[i].aux_groups, chain.step[i].out_tiles = act, aux, tiles
    if head is not None:
        chain.step[0].lg_col, chain.step[0].lg_n = head
    return chain


def test_bf16x6_organisation_query_is_host_logic(monkeypatch):
    """`ffn_mlp_bf16x6_organisation` (include/ffn_hip.h) decides on the host which workgroup organisation a
    bf16x6 chain runs in: 1 = matrix / vector waves (csrc/mlp_bf16_mv.hip) for a features-only first step of
    16 j K blocks followed by 256 -> 256 steps with eight output tiles (groups are half K blocks), and for the
    backward chains of the same models (the d_logits term alone in step 0); 0 = two waves per SIMD for
    everything else, and for everything under FFN_BF16X6_ORG=ws."""
    import ctypes
    lib = _lib.load()
    fn = lib.ffn_mlp_bf16x6_organisation
    fn.restype = ctypes.c_int
    monkeypatch.delenv("FFN_BF16X6_ORG", raising=False)
    monkeypatch.delenv("FFN_BF16X6_PRODUCTS", raising=False)
    monkeypatch.delenv("FFN_BF16X6_FWD_ACCS", raising=False)

    def org(chain, backward=0):
        return fn(ctypes.byref(chain), ctypes.c_int(backward))

    tiny = _chain([(0, 64, 8), (32, 0, 8), (32, 0, 8)])            # 510 features -> 32 K blocks, two hidden steps
    assert org(tiny) == 1
    assert org(_chain([(0, 64, 8)] + [(32, 0, 8)] * 7)) == 1      # eight layers
    assert org(_chain([(0, 32, 8), (32, 0, 8)])) == 1             # 16 K blocks of features, one hidden step
    for other in ([(0, 64, 8)],                                    # a single step
                  [(0, 8, 8), (32, 0, 8)],                         # NeRF's 63-channel encoding: 4 K blocks
                  [(0, 48, 8), (32, 0, 8)],                        # 24 K blocks: not 16 j
                  [(0, 64, 8), (32, 8, 8), (32, 0, 8)],            # a skip connection (activations + features)
                  [(0, 64, 8), (32, 0, 4), (16, 0, 8)],            # a 128-wide layer
                  [(0, 4, 8), (32, 0, 8)]):                        # three raw inputs
        assert org(_chain(other)) == 0, other
    assert org(_chain([(0, 64, 8), (32, 0, 8)], bias_floats=5000)) == 0     # a bias buffer beyond its LDS copy
    assert org(_chain([(0, 64, 8), (32, 0, 8)], wide=2)) == 0
    # backward-data chains: step 0 is the d_logits term alone, then 256 -> 256
    # (the vector waves compute that term from the head's 1 .. 4 logits columns: lg_col, lg_n)
    assert org(_chain([(0, 4, 8), (32, 0, 8), (32, 0, 8)], head=(0, 4)), 1) == 1
    assert org(_chain([(0, 4, 8), (32, 0, 8), (32, 0, 8)], head=(3, 1)), 1) == 1
    assert org(_chain([(0, 4, 8), (32, 0, 8), (32, 0, 8)]), 1) == 0              # no logits columns named
    assert org(_chain([(0, 4, 8), (32, 0, 8), (32, 0, 8)], head=(2, 3)), 1) == 0   # columns 2 .. 4 of four
    assert org(_chain([(32, 4, 8), (32, 0, 8)], head=(0, 4)), 1) == 0           # a head fused mid-chain
    assert org(_chain([(0, 4, 8), (32, 0, 4)], head=(0, 4)), 1) == 0
    monkeypatch.setenv("FFN_BF16X6_ORG", "ws")
    assert org(tiny) == 0 and org(_chain([(0, 4, 8), (32, 0, 8)], head=(0, 4)), 1) == 0
