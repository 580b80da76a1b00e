"""Host side of the fused MLP kernels: turns a chain of nn.Linear layers + encodings into
the ``ffn_mlp_chain`` programs the kernels interpret (forward and backward-data), the
weight-gradient job list, and owns the packed-operand buffers and workspaces.

Gradient layout: one flat fp32 buffer holding, for every layer in chain order, the weight
gradient (out, in) row-major followed by the bias gradient -- the same order as the
models' flat parameter buffer, so one RCCL all-reduce and one fused clip+Adam launch cover
the whole model.
"""

import ctypes
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import c_i, c_i64
from .ops import _call, _dev

MAX_STEPS = 16
BIAS_LDS_FLOATS = 4096      # csrc/mlp.hip kBiasLdsFloats: the LDS copy of the bias buffer's head
WGRAD_GROUPS = 256          # one persistent workgroup per CU for the LDS-staged units
# relative cost of one 32-sample block of a unit, by the number of 128x128 quadrants it has
# (calibrated on MI355X with FFN_UNIT_COST sweeps: a full unit is ~18k cycles per block, of which
# 16.4k are MFMA issue), and of a head unit (32 4x4x1 MFMAs per wave, bound by the stream of its
# input slab: measured ~3.9; a cost of 3 puts too many of its blocks on one workgroup and costs 30 %)
UNIT_COST = {4: 24, 2: 13, 1: 8}
HEAD_COST = 5
# folded narrow input windows (quadrants, fold): half / a quarter of the matrix instructions of the
# unfolded unit, and by then as much operand streaming as matrix work
FOLD_COST = {(2, 2): 7, (2, 4): 5, (1, 2): 4, (1, 4): 3}
if os.environ.get("FFN_FOLD_COST"):       # "h2,h4,q2,q4" -- calibration experiments
    _c = [int(v) for v in os.environ["FFN_FOLD_COST"].split(",")]
    FOLD_COST = {(2, 2): _c[0], (2, 4): _c[1], (1, 2): _c[2], (1, 4): _c[3]}
# the split-bf16 kernel (wgrad_bf16.hip): a full unit's block costs ~2.8 us (LDS-DMA staging,
# conversions one step ahead of the matrix instructions); units with fewer quadrants run the
# unpipelined path and cost about as much; the f32 logits-head unit (register-staged, 4x4x1
# MFMAs) ~1.4 us per block (FFN_UNIT_COST16 sweeps on MI355X, tiny and full NeRF: best at 12,
# a cliff at 10)
UNIT_COST16 = {4: 24, 2: 24, 1: 20}
HEAD_COST16 = 13
if os.environ.get("FFN_UNIT_COST16"):
    _c = [int(v) for v in os.environ["FFN_UNIT_COST16"].split(",")]
    UNIT_COST16, HEAD_COST16 = {4: _c[0], 2: _c[1], 1: _c[2]}, _c[3]
# the f32-accurate split kernel (wgrad_bf16x6.hip): six matrix instructions per product and a
# three-way split -- a full unit's block ~3.5 us; narrower units run unpipelined; the f32 head
# unit as above (FFN_UNIT_COST_X6 = "full,half,quarter,head" for calibration)
UNIT_COST_X6 = {4: 24, 2: 17, 1: 17}
HEAD_COST_X6 = 10
if os.environ.get("FFN_UNIT_COST_X6"):
    _c = [int(v) for v in os.environ["FFN_UNIT_COST_X6"].split(",")]
    UNIT_COST_X6, HEAD_COST_X6 = {4: _c[0], 2: _c[1], 1: _c[2]}, _c[3]
if os.environ.get("FFN_UNIT_COST"):       # "full,half,quarter,head" -- calibration experiments
    _c = [int(v) for v in os.environ["FFN_UNIT_COST"].split(",")]
    UNIT_COST, HEAD_COST = {4: _c[0], 2: _c[1], 1: _c[2]}, _c[3]


# A training launch whose block count leaves the persistent grid a short last round (every one of
# the 4 x CUs resident wavefronts owns whole 32-sample blocks, so 3.1 blocks per wavefront cost 4
# rounds) is split: the full rounds on the one-wave-per-block kernels, the remainder on kernels
# that put a TEAM of waves on a block -- four waves per block (one team per CU: a round takes ~0.3
# of the time) when the remainder is at most one block per CU, else two waves per block (~0.57)
# when it is at most half a round.  At the reference's default batch (1024 rays x 128 samples,
# ~3170 blocks = 3 rounds + 98 blocks) that is 3.3 instead of 4 rounds for the forward and
# backward-data kernels.  Training launches only (inference keeps its bit-exact batch
# independence: the team kernels sum a fused head's partial products in different orders).
TAIL_PAIRS = os.environ.get("FFN_TAIL_PAIRS", "1") == "1"
TAIL_QUADS = os.environ.get("FFN_TAIL_QUADS", "1") == "1"
TAIL_MAX_FULL_ROUNDS = 16       # beyond that the tail is < 3 % of the launch


class FfnEncoding(ctypes.Structure):
    _fields_ = [("b", ctypes.c_void_p), ("a", ctypes.c_void_p), ("num_freq", ctypes.c_int32),
                ("include_input", ctypes.c_int32), ("scale", ctypes.c_float),
                ("width", ctypes.c_int32)]


class FfnStep(ctypes.Structure):
    _fields_ = [("act_groups", ctypes.c_int32), ("aux_groups", ctypes.c_int32),
                ("enc_id", ctypes.c_int32), ("lg_col", ctypes.c_int32), ("lg_n", ctypes.c_int32),
                ("out_tiles", ctypes.c_int32), ("relu", ctypes.c_int32), ("dst", ctypes.c_int32),
                ("out_col", ctypes.c_int32), ("out_n", ctypes.c_int32),
                ("save_in_slot", ctypes.c_int32), ("save_out_slot", ctypes.c_int32),
                ("mask_slot", ctypes.c_int32), ("save_enc_slot", ctypes.c_int32),
                ("head_off", ctypes.c_int32), ("out_slot", ctypes.c_int32),
                ("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64)]


class FfnMlpChain(ctypes.Structure):
    _fields_ = [("enc", FfnEncoding * 2), ("step", FfnStep * MAX_STEPS),
                ("num_steps", ctypes.c_int32), ("num_slots", ctypes.c_int32),
                ("bias_floats", ctypes.c_int32), ("wide", ctypes.c_int32),
                ("slot_channels", ctypes.c_int32 * MAX_STEPS),
                ("slot_offset", ctypes.c_int64 * MAX_STEPS)]


class FfnWgradUnit(ctypes.Structure):
    _fields_ = [("m_slot", ctypes.c_int32), ("m_cq0", ctypes.c_int32), ("m_quads", ctypes.c_int32),
                ("want_bias", ctypes.c_int32), ("n_slot", ctypes.c_int32), ("n_cq0", ctypes.c_int32),
                ("n_quads", ctypes.c_int32), ("kind", ctypes.c_int32)]


class FfnWgradSegment(ctypes.Structure):
    _fields_ = [("job", ctypes.c_int32), ("slot", ctypes.c_int32),
                ("blk_begin", ctypes.c_int64), ("blk_end", ctypes.c_int64)]


class FfnReduceJob(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("slot_begin", ctypes.c_int32),
                ("slot_end", ctypes.c_int32), ("m_ch0", ctypes.c_int32), ("rows", ctypes.c_int32),
                ("n_quad0", ctypes.c_int32), ("n_quads", ctypes.c_int32),
                ("k_base", ctypes.c_int32), ("ld", ctypes.c_int32), ("has_bias", ctypes.c_int32),
                ("lg_n", ctypes.c_int32), ("slot_stride", ctypes.c_int32),
                ("n_fold", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("w_grad_off", ctypes.c_int64), ("b_grad_off", ctypes.c_int64),
                ("col_map", ctypes.c_void_p)]


class FfnPackJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("col_map", ctypes.c_void_p),
                ("kind", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("ld", ctypes.c_int32), ("transpose", ctypes.c_int32), ("groups", ctypes.c_int32),
                ("tiles", ctypes.c_int32), ("dst_rs", ctypes.c_int32), ("dst_cs", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class FfnRenderRays(ctypes.Structure):
    _fields_ = [("starts", ctypes.c_void_p), ("directions", ctypes.c_void_p),
                ("near_far", ctypes.c_void_p), ("num_rays_total", ctypes.c_int64),
                ("ray_index", ctypes.c_void_p), ("ray_base", ctypes.c_int64),
                ("valid", ctypes.c_void_p), ("num_rays", ctypes.c_int32),
                ("num_samples", ctypes.c_int32), ("unit", ctypes.c_void_p),
                ("t_values", ctypes.c_void_p)]


class FfnOccupancy(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_void_p), ("box_min", ctypes.c_float * 3),
                ("box_size", ctypes.c_float * 3), ("resolution", ctypes.c_int32)]


class FfnRenderOut(ctypes.Structure):
    _fields_ = [("color", ctypes.c_void_p), ("alpha", ctypes.c_void_p), ("depth", ctypes.c_void_p),
                ("nan_flag", ctypes.c_void_p), ("image", ctypes.c_void_p),
                ("pixel_offset", ctypes.c_int64)]


def _struct_array_to_device(items, device):
    if not items:
        return torch.zeros((0,), dtype=torch.uint8, device=device)
    arr = (type(items[0]) * len(items))(*items)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


class EncodingSpec:
    """A sin/cos feature map of a 3-vector: [a cos(s x B), a sin(s x B)] (+ x)."""

    def __init__(self, b: Optional[torch.Tensor], a: Optional[torch.Tensor], scale: float,
                 include_input: bool, device: Optional[torch.device] = None):
        self.num_freq = 0 if b is None else int(b.shape[1])
        dev = device if device is not None else (b.device if b is not None else None)
        if b is None:
            b = torch.zeros((3, 1), dtype=torch.float32, device=dev)
        if a is None:
            a = torch.ones((max(self.num_freq, 1),), dtype=torch.float32, device=dev)
        self.b = b.contiguous()
        self.a = a.contiguous()
        self.scale = float(scale)
        self.include_input = bool(include_input) or self.num_freq == 0
        if self.num_freq > 256:
            raise NotImplementedError("encodings with more than 256 frequencies")
        natural = 2 * self.num_freq + (3 if self.include_input else 0)
        self.natural_width = natural
        self.width = ((natural + 31) // 32) * 32

    def natural_index(self, internal: int) -> int:
        """Natural column ([cos F][sin F][x 3]) of internal channel 2k+trig / 2F+d, or -1."""
        freq = self.num_freq
        if internal < 2 * freq:
            k = internal >> 1
            return k if (internal & 1) == 0 else freq + k
        if self.include_input and internal < 2 * freq + 3:
            return internal
        return -1


class DenseSpec:
    """One nn.Linear of the chain.  Input = [previous activations (act_in channels),
    encoding `enc_id` (its natural width)] in that column order."""

    def __init__(self, weight, bias, act_in: int, enc_id: Optional[int], relu: bool,
                 to_logits: Optional[tuple] = None):
        self.weight = weight
        self.bias = bias
        self.act_in = int(act_in)
        self.enc_id = enc_id
        self.relu = bool(relu)
        self.to_logits = to_logits     # (first column, count) or None
        self.out = int(weight.shape[0])
        self.ld = int(weight.shape[1])
        # widths the kernels run at (zero-padded to a supported tile count; MlpProgram sets them)
        self.out_p = self.out
        self.act_in_p = self.act_in


MAX_HIDDEN_CHANNELS = 1024


def _padded_width(channels: int, wide=False) -> int:
    """Width a hidden layer of ``channels`` outputs runs at: the kernels' tile counts are 1/2/4/8
    tiles of 32 channels (2/4/8/16 in a chain with a layer wider than 256: ``wide`` = 1 / True;
    4/8/16/32 in a chain with a layer wider than 512: ``wide`` = 3), so any other width is
    zero-padded to the next one -- zero rows / columns in the operand packs, zero bias, zero
    fused-head weights; the padding's activations, dZ and gradients are exactly 0 in f32 and the
    reducer drops them (reference: nn.Linear accepts any width, ``train_nerf.py:28-31``)."""
    widths = {0: (32, 64, 128, 256), 1: (64, 128, 256, 512), 3: (128, 256, 512, 1024)}[int(wide)]
    for width in widths:
        if channels <= width:
            return width
    raise NotImplementedError("fused MLP kernels support hidden layers of up to %d channels "
                              "(got %d)" % (MAX_HIDDEN_CHANNELS, channels))


def _tiles(channels: int, wide=False) -> int:
    return _padded_width(channels, wide) // 32


class Workspace:
    """Per-batch-size plans of the training path.  The dZ slabs themselves are ONE grow-only
    buffer per program shared by every batch size (3 KiB per sample for the tiny NeRF, 9.5 KiB
    for the full one: a buffer per size would multiply tens of GB by the number of sizes seen)."""

    def __init__(self, prog: "MlpProgram", n: int):
        self.n = n
        blocks = (n + 31) // 32
        self._dz_floats = prog.dz_channels * 32 * blocks
        self._prog, self._blocks = prog, blocks
        self._plans = {}
        self.partials = None
        self.use_plan("f32")

    @property
    def dz(self) -> torch.Tensor:
        return self._prog._dz_buffer(self._dz_floats)

    def use_plan(self, precision: str):
        """Selects the weight-gradient plan (segments balanced with the kernel's cost model:
        the exact-f32 and the split-bf16 kernel weigh head and hidden units differently)."""
        plan = self._plans.get(precision)
        if plan is None:
            dev = self._prog.device
            raw = self._prog._plan_wgrad(self._blocks, precision)
            launches = [(kind, _struct_array_to_device(segs, dev), torch.tensor(starts, dtype=torch.int32, device=dev))
                        for kind, segs, starts in raw["launches"]]
            plan = dict(unit_segments=launches[0][1], unit_seg_start=launches[0][2], launches=launches,
                        reduce_jobs=_struct_array_to_device(raw["reduce_jobs"], dev),
                        num_reduce_jobs=len(raw["reduce_jobs"]),
                        partial_floats=raw["slots"] * self._prog.partial_floats)
            self._plans[precision] = plan
        self.unit_segments = plan["unit_segments"]
        self.unit_seg_start = plan["unit_seg_start"]
        self.launches = plan["launches"]      # (kernel precision, segments, starts) per launch
        self.reduce_jobs = plan["reduce_jobs"]
        self.num_reduce_jobs = plan["num_reduce_jobs"]
        if self.partials is None or self.partials.numel() < plan["partial_floats"]:
            self.partials = torch.empty((plan["partial_floats"],), dtype=torch.float32,
                                        device=self._prog.device)


class MlpProgram:
    """Chains + device buffers for one model.  All tensors live on ``device`` (a GPU)."""

    def __init__(self, encodings: Sequence[EncodingSpec], layers: Sequence[DenseSpec],
                 device: torch.device, planning_only: bool = False):
        """``planning_only`` builds the chains / job lists on any device without touching
        the library (host-logic tests); every launch method then raises."""
        if device.type != "cuda" and not planning_only:
            raise RuntimeError("MlpProgram needs a GPU device; there is no CPU fallback")
        if len(layers) > MAX_STEPS:
            raise NotImplementedError("at most %d dense layers" % MAX_STEPS)
        self.partial_floats = 16 * 16 * 64 + 256
        if not planning_only:
            lib = _lib.load()
            lib.ffn_mlp_wgrad_partial_floats.restype = ctypes.c_int64
            assert int(lib.ffn_mlp_wgrad_partial_floats()) == self.partial_floats
        self.device = device
        self.encodings = list(encodings)
        self.layers = list(layers)
        self._workspaces: Dict[int, Workspace] = {}
        self._build_forward()
        self._build_backward()
        self._build_wgrad_jobs()
        self._build_forward16()
        self._build_backward16()
        self._build_x6()
        self._build_pair_chains()

    def _dz_buffer(self, floats: int) -> torch.Tensor:
        """The shared dZ workspace, grown (old buffer released first) when a larger batch shows up."""
        buf = getattr(self, "_dz", None)
        if buf is None or buf.numel() < floats:
            self._dz = buf = None
            self._dz = buf = torch.empty((floats + floats // 16,), dtype=torch.float32,
                                         device=self.device)
        return buf[:floats]

    def release_workspaces(self):
        """Drops the dZ buffer and the per-size plans (they are rebuilt on demand): for callers
        that switch from one large workload to another inside one process."""
        self._dz = None
        self._workspaces.clear()

    # ------------------------------------------------------------------ chains
    def _fill_encodings(self, chain):
        for i, enc in enumerate(self.encodings):
            e = chain.enc[i]
            e.b = enc.b.data_ptr()
            e.a = enc.a.data_ptr()
            e.num_freq = enc.num_freq
            e.include_input = 1 if enc.include_input else 0
            e.scale = enc.scale
            e.width = enc.width
        for i in range(len(self.encodings), 2):     # unused entries still need valid pointers
            src = chain.enc[0]
            chain.enc[i].b, chain.enc[i].a = src.b, src.a
            chain.enc[i].num_freq, chain.enc[i].include_input = src.num_freq, src.include_input
            chain.enc[i].scale, chain.enc[i].width = src.scale, src.width

    def _build_forward(self):
        """Forward chain.  Hidden layers (and heads that read an encoding or have no hidden
        producer) are MFMA steps; a logits head that reads a hidden layer is fused into that
        layer's epilogue (``head_off``) and has no step of its own."""
        fwd = FfnMlpChain()
        self._fill_encodings(fwd)
        # a layer wider than 256 channels switches the whole chain to the two-waves-per-block
        # kernels (64 KiB slab per pair)
        # ... and a layer wider than 512 (up to 1024) to a team of FOUR waves per block on the whole
        # 128 KiB slab area (ffn_mlp_chain.wide == 3; exact-f32 kernels only, three-pass renders)
        widest = max([sp.out for sp in self.layers if sp.to_logits is None] or [0])
        self.wide = widest > 256
        self.big = widest > 512
        self.wide_level = 3 if self.big else (1 if self.wide else 0)
        fwd.wide = self.wide_level
        # any nn.Linear width is accepted: hidden layers run zero-padded to a supported tile count
        producer = -1
        for i, sp in enumerate(self.layers):
            sp.act_in_p = self.layers[producer].out_p if (sp.act_in > 0 and producer >= 0) else sp.act_in
            if sp.to_logits is None:
                sp.out_p = _padded_width(sp.out, self.wide_level)
                producer = i
        # bias buffer = [fused-head blocks | per-step padded biases]: the kernels keep its first
        # BIAS_LDS_FLOATS floats in LDS (every head block must be there: the epilogues read them
        # per channel quad); a step whose bias block lies beyond reads it from global memory / L2
        w_off = b_off = h_off = 0
        self.col_maps: List[torch.Tensor] = []
        self.step_of: List[Optional[int]] = []   # layer index -> forward step (None = fused head)
        self.fwd_shapes: Dict[int, tuple] = {}
        self.fused_heads = []                    # (head layer, bias-buffer offset, channels)
        self.grad_w_off, self.grad_b_off = [], []
        self.slot_of: Dict[int, int] = {}      # producer layer index -> slab slot
        self.producer_of: List[int] = []       # layer index -> layer whose output it consumes
        slot_off = 0
        g_off = 0
        last_producer = -1
        # Who saves a hidden layer's output for the backward pass: EVERY step that consumes it
        # saves it "on consume" from inside its K loop (the training forward's K-loop trips all
        # save what they read: one straight-line trip); only an output that no step consumes
        # (it feeds fused logits heads only) is saved by its producer's epilogue.  Supported
        # models have at most one consuming step per output; a second one would store the same
        # values to the same place again.
        stepped_consumers = set()
        prev = -1
        for j, sp in enumerate(self.layers):
            fusable = sp.to_logits is not None and sp.enc_id is None and sp.act_in > 0
            if sp.act_in > 0 and not fusable:
                stepped_consumers.add(prev)
            if sp.to_logits is None:
                prev = j
        num_steps = 0
        for i, spec in enumerate(self.layers):
            enc = None if spec.enc_id is None else self.encodings[spec.enc_id]
            self.producer_of.append(last_producer if spec.act_in > 0 else -1)
            if spec.act_in > 0 and (last_producer < 0 or
                                    self.layers[last_producer].out != spec.act_in):
                raise ValueError("layer %d consumes %d channels but the previous producer "
                                 "wrote a different width" % (i, spec.act_in))
            act_groups = spec.act_in_p // 8
            cmap = [c if c < spec.act_in else -1 for c in range(8 * act_groups)]
            if enc is not None:
                for c in range(enc.width):
                    nat = enc.natural_index(c)
                    cmap.append(-1 if nat < 0 else spec.act_in + nat)
            self.col_maps.append(torch.tensor(cmap, dtype=torch.int32, device=self.device))
            self.grad_w_off.append(g_off)
            g_off += spec.out * spec.ld
            self.grad_b_off.append(g_off)
            g_off += spec.out
            fuse = (spec.to_logits is not None and enc is None and spec.act_in > 0
                    and self.step_of[last_producer] is not None)
            if fuse:
                P = fwd.step[self.step_of[last_producer]]
                if P.head_off >= 0:
                    raise NotImplementedError("a layer may feed at most one logits head")
                P.head_off = h_off
                self.fused_heads.append((i, h_off, spec.act_in))
                h_off += 4 + 4 * spec.act_in_p
                if last_producer not in stepped_consumers:
                    P.save_out_slot = self.slot_of[last_producer]
                self.step_of.append(None)
                continue
            if num_steps >= MAX_STEPS:
                raise NotImplementedError("at most %d forward steps" % MAX_STEPS)
            L = fwd.step[num_steps]
            self.step_of.append(num_steps)
            num_steps += 1
            L.act_groups = act_groups
            L.aux_groups = 0 if enc is None else enc.width // 8
            L.enc_id = 0 if spec.enc_id is None else spec.enc_id
            L.out_tiles = spec.out_p // 32 if spec.to_logits is None else 1
            L.relu = 1 if spec.relu else 0
            L.save_in_slot = L.save_out_slot = L.mask_slot = L.save_enc_slot = L.head_off = -1
            if spec.act_in > 0:
                L.save_in_slot = self.slot_of[last_producer]
            if spec.to_logits is None:
                L.dst, L.out_col, L.out_n = 0, 0, 0
                slot = len(self.slot_of)
                self.slot_of[i] = slot
                if spec.relu:
                    L.mask_slot = slot           # sign bits of this layer's output
                fwd.slot_channels[slot] = spec.out_p
                fwd.slot_offset[slot] = slot_off
                slot_off += spec.out_p
                last_producer = i
            else:
                if self.wide:
                    raise NotImplementedError("a 512-wide model needs its logits heads to read a "
                                              "hidden layer (fused heads only)")
                L.dst, L.out_col, L.out_n = 1, spec.to_logits[0], spec.to_logits[1]
            groups = L.act_groups + L.aux_groups
            L.w_off, L.b_off = w_off, b_off
            self.fwd_shapes[i] = (groups, L.out_tiles)
            w_off += groups * L.out_tiles * 256
            b_off += 32 * L.out_tiles
        if h_off > BIAS_LDS_FLOATS and not self.big:      # (big chains read head blocks from L2)
            raise NotImplementedError("fused logits heads need %d floats of LDS (limit %d)"
                                      % (h_off, BIAS_LDS_FLOATS))
        for k in range(num_steps):
            fwd.step[k].b_off += h_off
        b_off += h_off
        fwd.num_steps = num_steps
        fwd.num_slots = len(self.slot_of)
        fwd.bias_floats = b_off
        # encoding features are saved by every step that generates them (slabs after the
        # hidden-layer ones; a second user -- NeRF's skip layer -- stores the same 8 KiB per block
        # again), so that every weight-gradient window is an ordinary slab window
        self.dz_channels = slot_off
        self.enc_slot: Dict[int, int] = {}
        for i, spec in enumerate(self.layers):
            if spec.enc_id is None:
                continue
            if spec.enc_id in self.enc_slot:
                fwd.step[self.step_of[i]].save_enc_slot = self.enc_slot[spec.enc_id]
                continue
            slot = fwd.num_slots + len(self.enc_slot)
            if slot >= MAX_STEPS:
                raise NotImplementedError("too many activation slabs")
            self.enc_slot[spec.enc_id] = slot
            fwd.step[self.step_of[i]].save_enc_slot = slot
            fwd.slot_channels[slot] = self.encodings[spec.enc_id].width
            fwd.slot_offset[slot] = slot_off
            slot_off += self.encodings[spec.enc_id].width
        self.fwd = fwd
        self.saved_channels = slot_off
        # uint32 of ReLU sign bits per slot and block: 64 lanes x 128 bit per wave of a block's team
        self.mask_words = 1024 if self.big else (512 if self.wide else 256)
        self.num_grad_floats = g_off
        self.packed_fwd = torch.zeros((max(w_off, 1),), dtype=torch.float32, device=self.device)
        self.bias_buf = torch.zeros((b_off,), dtype=torch.float32, device=self.device)

    def _build_pair_chains(self):
        """Copies of the forward / backward-data chains flagged for the team kernels -- two waves
        per block (``wide = 1``) and four waves per block (``wide = 2``) -- on the same operand
        packs and slabs, for chains those kernels accept: every logits head fused, every step
        64..256 (four waves: 128..256) channels wide."""
        ok = quad = not self.wide
        for chain in (self.fwd, self.bwd):
            for k in range(chain.num_steps):
                st = chain.step[k]
                ok = ok and st.out_tiles in (2, 4, 8) and st.dst == 0
                quad = quad and st.out_tiles in (4, 8) and st.dst == 0
        self.pair_chain_ok = bool(ok and self.bwd.num_steps > 0)
        self.quad_chain_ok = bool(quad and self.pair_chain_ok and TAIL_QUADS)
        self.fwd_pair = self.bwd_pair = self.fwd_quad = self.bwd_quad = None
        if self.pair_chain_ok:
            self.fwd_pair = FfnMlpChain.from_buffer_copy(bytes(self.fwd))
            self.bwd_pair = FfnMlpChain.from_buffer_copy(bytes(self.bwd))
            self.fwd_pair.wide = self.bwd_pair.wide = 1
        if self.quad_chain_ok:
            self.fwd_quad = FfnMlpChain.from_buffer_copy(bytes(self.fwd))
            self.bwd_quad = FfnMlpChain.from_buffer_copy(bytes(self.bwd))
            self.fwd_quad.wide = self.bwd_quad.wide = 2

    def _build_forward16(self):
        """Chain + operand buffer of the OPT-IN split-bf16 kernels (mlp_bf16.hip, mlp_bf16_ws.hip):
        the forward chain with its weight offsets pointing into a bf16 (hi, lo) operand buffer
        whose K order is the register hand-off order of those kernels.  512-wide chains pack 16
        output tiles per K block (two-waves-per-SIMD kernels only).  ``self.fwd16`` stays None for
        chains with MFMA logits steps."""
        self.fwd16 = None
        self.packed16 = None
        self._packed16_dirty = True
        self._packed_x6_dirty = True
        self.tiles16 = 16 if self.wide else 8
        if self.device.type != "cuda" or self.big:      # (no split-bf16 kernels beyond 512 channels)
            return
        steps = [(i, self.fwd.step[self.step_of[i]]) for i in range(len(self.layers))
                 if self.step_of[i] is not None]
        if any(st.dst != 0 for _, st in steps):
            return
        chain = FfnMlpChain.from_buffer_copy(bytes(self.fwd))
        seen_enc = set()               # this kernel saves an encoding's features once
        for i, _ in steps:
            st16 = chain.step[self.step_of[i]]
            if st16.save_enc_slot >= 0:
                if st16.save_enc_slot in seen_enc:
                    st16.save_enc_slot = -1
                seen_enc.add(st16.save_enc_slot)
        self.pack16_jobs = []          # (layer index, kblocks, col map tensor, element offset)
        off = 0
        for i, st in steps:
            spec = self.layers[i]
            enc = None if spec.enc_id is None else self.encodings[spec.enc_id]
            kb_act = spec.act_in_p // 16
            kb_feat = 0 if enc is None else enc.width // 16
            cmap = []
            for g in range(kb_act):
                for h in range(2):
                    for j in range(8):
                        c = 16 * g + (4 * h + j if j < 4 else 8 + 4 * h + (j - 4))
                        cmap.append(c if c < spec.act_in else -1)
            for g in range(kb_feat):
                for h in range(2):
                    for j in range(8):
                        nat = enc.natural_index(16 * g + 8 * h + j)
                        cmap.append(-1 if nat < 0 else spec.act_in + nat)
            kblocks = kb_act + kb_feat
            chain.step[self.step_of[i]].w_off = off
            # training mode of the kernel: every step saves its own output (the f32 chain saves
            # some of them on consumption by the next step)
            chain.step[self.step_of[i]].out_slot = self.slot_of.get(i, -1)
            self.pack16_jobs.append((i, kblocks, torch.tensor(cmap, dtype=torch.int32, device=self.device), off))
            off += kblocks * self.tiles16 * 1024     # tiles x (hi, lo) x 64 lanes x 8 bf16
        self.fwd16 = chain
        self.packed16 = torch.zeros((max(off, 1),), dtype=torch.int16, device=self.device)

    def _build_backward(self):
        """Backward-data chain: one step per producer layer that has consumers, walking
        the network from the outputs to the first layer."""
        bwd = FfnMlpChain()
        self._fill_encodings(bwd)
        bwd.wide = self.fwd.wide
        for s in range(MAX_STEPS):
            bwd.slot_channels[s] = self.fwd.slot_channels[s]
            bwd.slot_offset[s] = self.fwd.slot_offset[s]
        bwd.num_slots = self.fwd.num_slots
        consumers: Dict[int, List[int]] = {}
        for i, prod in enumerate(self.producer_of):
            if prod >= 0:
                consumers.setdefault(prod, []).append(i)
        self.bwd_packs = []        # (layer index, groups, tiles, w_off) transposed packs
        steps = []
        wt_off = 0
        producers = sorted(consumers.keys(), reverse=True)
        for j in producers:
            hidden = [c for c in consumers[j] if self.layers[c].to_logits is None]
            heads = [c for c in consumers[j] if self.layers[c].to_logits is not None]
            if len(hidden) > 1 or len(heads) > 1:
                raise NotImplementedError("a layer may feed at most one hidden layer and one head")
            st = FfnStep()
            st.out_tiles = self.layers[j].out_p // 32
            st.act_groups = 0 if not hidden else self.layers[hidden[0]].out_p // 8
            st.aux_groups = 4 if heads else 0
            st.relu = 0
            st.mask_slot = self.slot_of[j] if self.layers[j].relu else -1
            st.save_in_slot = self.slot_of[hidden[0]] if hidden else -1
            st.save_out_slot = -1
            st.w_off = wt_off
            if hidden:
                c = hidden[0]
                self.bwd_packs.append((c, st.act_groups, st.out_tiles, wt_off))
                wt_off += st.act_groups * st.out_tiles * 256
            if heads:
                c = heads[0]
                st.lg_col, st.lg_n = self.layers[c].to_logits
                self.bwd_packs.append((c, 4, st.out_tiles, wt_off))
                wt_off += 4 * st.out_tiles * 256
            steps.append(st)
        if steps:
            # the last step's output (dZ of the first producer) has no consumer step
            steps[-1].save_out_slot = self.slot_of[producers[-1]]
        if len(steps) > MAX_STEPS:
            raise NotImplementedError("backward chain too long")
        for k, st in enumerate(steps):
            bwd.step[k] = st
        bwd.num_steps = len(steps)
        self.bwd = bwd
        self.packed_bwd = torch.zeros((max(wt_off, 1),), dtype=torch.float32, device=self.device)

    def _build_backward16(self):
        """Chain + transposed operand buffer of the OPT-IN split-bf16 backward-data kernel
        (mlp_bf16_bwd.hip).  ``self.bwd16`` stays None where the forward one does."""
        self.bwd16 = None
        self.packed16_bwd = None
        self.pack16_bwd_jobs = []      # (consumer layer, kblocks, K map tensor, element offset)
        if self.fwd16 is None or self.bwd.num_steps == 0:
            return
        chain = FfnMlpChain.from_buffer_copy(bytes(self.bwd))
        consumers: Dict[int, List[int]] = {}
        for i, prod in enumerate(self.producer_of):
            if prod >= 0:
                consumers.setdefault(prod, []).append(i)
        off = 0
        for k, j in enumerate(sorted(consumers.keys(), reverse=True)):
            st = chain.step[k]
            st.w_off = off
            st.out_slot = self.slot_of[j]
            hidden = [c for c in consumers[j] if self.layers[c].to_logits is None]
            heads = [c for c in consumers[j] if self.layers[c].to_logits is not None]
            if hidden:
                c = hidden[0]
                kb = self.layers[c].out_p // 16
                # K order = the register hand-off order of the consumer's dZ
                cmap = [16 * g + (4 * h + jj if jj < 4 else 8 + 4 * h + (jj - 4))
                        for g in range(kb) for h in range(2) for jj in range(8)]
                cmap = [k if k < self.layers[c].out else -1 for k in cmap]
                self.pack16_bwd_jobs.append((c, kb, torch.tensor(cmap, dtype=torch.int32, device=self.device), off))
                off += kb * self.tiles16 * 1024
            if heads:
                c = heads[0]
                rows = self.layers[c].to_logits[1]
                cmap = [(jj if (g == 0 and h == 0 and jj < rows) else -1)
                        for g in range(2) for h in range(2) for jj in range(8)]
                self.pack16_bwd_jobs.append((c, 2, torch.tensor(cmap, dtype=torch.int32, device=self.device), off))
                off += 2 * self.tiles16 * 1024
        self.bwd16 = chain
        self.packed16_bwd = torch.zeros((max(off, 1),), dtype=torch.int16, device=self.device)

    def _build_x6(self):
        """Chains + operand buffers of the F32-ACCURATE split mode ("bf16x6", mlp_bf16_ws.hip):
        the split-bf16 chains with every operand as THREE bf16 parts (hi, mid, lo -- the f32 value
        exactly) instead of two, i.e. the same K order and job tables with 1.5x the offsets
        (``ffn_mlp_pack_bf16_parts(parts=3)``).  Narrow chains only (<= 256 channels per layer:
        the three-part X image of a 512-wide chain leaves room for ONE 32-sample block per pass
        (96 KiB), i.e. every weight would stream from L2 once per 32 samples -- 4.7 MB per pass of
        a three-layer 512-wide chain, ~620 GB per 2^22 samples, ~40 ms at the L2's rate where the
        exact-f32 kernel takes 50: not built)."""
        self.fwd_x6 = self.bwd_x6 = None
        self.packed_x6 = self.packed_x6_bwd = None
        self._packed_x6_dirty = True
        if self.fwd16 is None or self.wide:
            return

        def scaled(chain16, total16):
            chain = FfnMlpChain.from_buffer_copy(bytes(chain16))
            for k in range(chain.num_steps):
                assert chain.step[k].w_off % 2 == 0
                chain.step[k].w_off = chain.step[k].w_off * 3 // 2
            return chain, torch.zeros((max(total16 * 3 // 2, 1),), dtype=torch.int16, device=self.device)

        self.fwd_x6, self.packed_x6 = scaled(self.fwd16, self.packed16.numel())
        if self.bwd16 is not None:
            self.bwd_x6, self.packed_x6_bwd = scaled(self.bwd16, self.packed16_bwd.numel())

    def _build_wgrad_jobs(self):
        """Weight-gradient work list: LDS-staged units of <=256 output x <=256 input channels
        for the hidden layers, head units (<=4 output rows) for the logits heads.  Inputs are
        slab windows: hidden activations or the saved encoding features."""
        self.wgrad_units: List[FfnWgradUnit] = []
        self.unit_meta = []     # per unit: reducer metadata
        for i, spec in enumerate(self.layers):
            enc = None if spec.enc_id is None else self.encodings[spec.enc_id]
            windows = []        # (slot, first quad, quads, k_base)
            if spec.act_in > 0:
                slot = self.slot_of[self.producer_of[i]]
                quads = spec.act_in_p // 4
                for q0 in range(0, quads, 64):
                    windows.append((slot, q0, min(64, quads - q0), 0))
            if enc is not None:
                quads = enc.width // 4
                for q0 in range(0, quads, 64):
                    windows.append((self.enc_slot[spec.enc_id], q0, min(64, quads - q0),
                                    spec.act_in_p))
            if spec.to_logits is None:
                m_slot = self.slot_of[i]
                out_quads = spec.out_p // 4
                for m0 in range(0, out_quads, 64):
                    for wi, (ns, q0, nq, kb) in enumerate(windows):
                        self.wgrad_units.append(FfnWgradUnit(m_slot, m0, min(64, out_quads - m0),
                                                             int(wi == 0), ns, q0, nq, 0))
                        self.unit_meta.append(dict(layer=i, m0=m0, m_quads=min(64, out_quads - m0),
                                                   n_quad0=q0, n_quads=nq, k_base=kb,
                                                   first=(wi == 0)))
            else:
                col, cnt = spec.to_logits
                for wi, (ns, q0, nq, kb) in enumerate(windows):
                    self.wgrad_units.append(FfnWgradUnit(col, cnt, 0, 0, ns, q0, nq, 1))
                    self.unit_meta.append(dict(layer=i, head=True, n_quad0=q0, n_quads=nq,
                                               k_base=kb, first=(wi == 0), lg_n=cnt))
        self.wgrad_units_dev = _struct_array_to_device(self.wgrad_units, self.device)

    @staticmethod
    def _split(costs: List[int], blocks: int, workers: int):
        """Contiguous, cost-balanced split of the job-major (job, block) sequence.
        Returns (segments as (job, blk_begin, blk_end), starts per worker).  Worker w ends
        where the running cost crosses (w+1)/workers of the total, rounded to the nearest
        block -- rounding errors do not pile up on the last worker."""
        total = sum(c * blocks for c in costs)
        segs, starts = [], [0]
        job, blk, cum = 0, 0, 0
        for w in range(workers):
            boundary = total * (w + 1) / workers
            while job < len(costs):
                left = blocks - blk
                if w == workers - 1:
                    take = left
                else:
                    take = min(left, int((boundary - cum) / costs[job] + 0.5))
                if take <= 0:
                    break
                segs.append((job, blk, blk + take))
                cum += take * costs[job]
                blk += take
                if blk == blocks:
                    job, blk = job + 1, 0
                else:
                    break
            starts.append(len(segs))
        assert job == len(costs), "work left unassigned"
        return segs, starts

    @staticmethod
    def _fold(n_quads: int, precision: str) -> int:
        """Fold of a narrow input window (csrc/wgrad_common.h ``ffn_wgrad_fold``: the exact-f32
        unit kernel packs <= 16 / <= 8 quads into 2 / 1 column tiles; the split-bf16 kernel
        does not fold)."""
        if precision != "f32":
            return 1
        return 4 if n_quads <= 8 else (2 if n_quads <= 16 else 1)

    @staticmethod
    def _quadrants(m_quads: int, n_quads: int):
        """(m halves, n halves) of a unit: the 128x128 quadrants that exist."""
        return (2 if m_quads > 32 else 1), (2 if n_quads > 32 else 1)

    def _plan_wgrad(self, blocks: int, precision: str = "f32"):
        """Segments of the weight-gradient kernel + the reducer's job table.  One
        workgroup-segment = 4 consecutive partial slots (one per wave).

        "bf16x6" is planned as TWO launches over one partial buffer and one reducer table: the
        three-part kernel takes the units with all four 128x128 quadrants (and the logits-head
        units), the exact-f32 kernel -- which FOLDS narrow input windows -- the units with fewer
        (NeRF's 63- / 27-channel encodings, its 128-channel view layer): in the three-part kernel
        those run unpipelined and cost as much as 0.7 of a full unit each (full NeRF weight
        gradients: 30.9 ms per 2^21 samples with every unit on it, 18.3 exact).  ``launches`` lists
        (kernel precision, segments, starts); every unit's reducer jobs carry the fold of the kernel
        that wrote its partials."""
        slot = 0
        reduce_jobs = []
        kernel_of = []              # per unit: the precision of the kernel that computes it
        unit_costs = []
        for u, meta in zip(self.wgrad_units, self.unit_meta):
            kind = precision
            if precision == "bf16x6" and u.kind != 1:
                mh, nh = self._quadrants(meta["m_quads"], meta["n_quads"])
                kind = "bf16x6" if mh * nh == 4 else "f32"
            kernel_of.append(kind)
            unit_cost, head_cost = {"bf16x3": (UNIT_COST16, HEAD_COST16),
                                    "bf16x6": (UNIT_COST_X6, HEAD_COST_X6)}.get(kind, (UNIT_COST, HEAD_COST))
            if u.kind == 1:
                unit_costs.append(head_cost)
            else:
                mh, nh = self._quadrants(meta["m_quads"], meta["n_quads"])
                fold = self._fold(meta["n_quads"], kind)
                unit_costs.append(unit_cost[mh * nh] if fold == 1 else FOLD_COST[(mh * nh, fold)])
        unit_slots = [[] for _ in self.wgrad_units]
        launches = []
        for kind in sorted(set(kernel_of), key=lambda k: k != precision):      # the mode's own kernel first
            ids = [u for u, k in enumerate(kernel_of) if k == kind]
            raw, starts = self._split([unit_costs[u] for u in ids], blocks, WGRAD_GROUPS)
            segments = []
            for (j, b0, b1) in raw:
                segments.append(FfnWgradSegment(ids[j], slot, b0, b1))
                unit_slots[ids[j]].append(slot)
                slot += 4
            launches.append((kind, segments, starts))
        unit_segments, unit_starts = launches[0][1], launches[0][2]
        for u, meta in enumerate(self.unit_meta):
            spec = self.layers[meta["layer"]]
            sl = unit_slots[u]
            assert sl == list(range(sl[0], sl[-1] + 4, 4))
            if meta.get("head"):
                for wave in range(4):           # wave w owns quads 16 w .. 16 w + 15, every sample
                    if meta["n_quads"] - 16 * wave <= 0:
                        continue
                    reduce_jobs.append(FfnReduceJob(
                        1, sl[0] + wave, sl[-1] + 4, 0, spec.out, meta["n_quad0"] + 16 * wave,
                        min(16, meta["n_quads"] - 16 * wave), meta["k_base"], spec.ld,
                        int(meta["first"] and wave == 0), meta["lg_n"], 4, 1, 0,
                        self.grad_w_off[meta["layer"]], self.grad_b_off[meta["layer"]],
                        self.col_maps[meta["layer"]].data_ptr()))
                continue
            mh, nh = self._quadrants(meta["m_quads"], meta["n_quads"])
            for mp in range(mh):
                for np_ in range(nh):
                    # wave w owns quadrant w % (mh*nh): its slots are qd, qd + mh*nh, ...
                    qd = mp * nh + np_
                    reduce_jobs.append(FfnReduceJob(
                        0, sl[0] + qd, sl[-1] + 4, 4 * (meta["m0"] + 32 * mp), spec.out,
                        meta["n_quad0"] + 32 * np_, min(32, meta["n_quads"] - 32 * np_),
                        meta["k_base"], spec.ld, int(meta["first"] and np_ == 0), 0, mh * nh,
                        self._fold(meta["n_quads"], kernel_of[u]), 0,
                        self.grad_w_off[meta["layer"]], self.grad_b_off[meta["layer"]],
                        self.col_maps[meta["layer"]].data_ptr()))
        return dict(unit_segments=unit_segments, unit_starts=unit_starts,
                    reduce_jobs=reduce_jobs, slots=slot, launches=launches)

    # ------------------------------------------------------------------ packing
    def pack16(self, parts: int = 2):
        """bf16 operand copies of the current weights for the split kernels: ``parts`` = 2 the
        (hi, lo) packs of the bf16x3 kernels, 3 the (hi, mid, lo) packs of the bf16x6 kernels."""
        fwd_buf, bwd_buf = (self.packed16, self.packed16_bwd) if parts == 2 else (self.packed_x6, self.packed_x6_bwd)
        unit = self.tiles16 * 512 * parts        # elements per K block: tiles x parts x 64 lanes x 8 bf16
        for (i, kblocks, cmap, off) in self.pack16_jobs:
            w = self.layers[i].weight.detach()
            off = off * parts // 2
            dst = fwd_buf[off:off + kblocks * unit]
            _call("ffn_mlp_pack_bf16_parts", _dev(w), c_i(w.shape[0]), c_i(w.shape[1]), c_i(w.stride(0)),
                  _dev(cmap, torch.int32), c_i(kblocks), c_i(self.tiles16), c_i(0), c_i(parts),
                  _dev(dst, torch.int16))
        for (c, kblocks, cmap, off) in (self.pack16_bwd_jobs if bwd_buf is not None else []):
            w = self.layers[c].weight.detach()
            off = off * parts // 2
            dst = bwd_buf[off:off + kblocks * unit]
            # operand rows = the consumer's input channels (its activation part), K = its rows
            _call("ffn_mlp_pack_bf16_parts", _dev(w), c_i(w.shape[0]), c_i(self.layers[c].act_in),
                  c_i(w.stride(0)), _dev(cmap, torch.int32), c_i(kblocks), c_i(self.tiles16), c_i(1),
                  c_i(parts), _dev(dst, torch.int16))
        if parts == 2:
            self._packed16_dirty = False
        else:
            self._packed_x6_dirty = False

    def covers(self, precision: str) -> bool:
        """Whether the opt-in arithmetic mode has kernels for this chain ("f32": always).  bf16x6:
        chains of <= 256 channels per layer whose logits heads are fused; bf16x3: fused heads."""
        if precision == "bf16x6":
            return self.fwd_x6 is not None
        if precision == "bf16x3":
            return self.fwd16 is not None
        return precision == "f32"

    def x6_organisation(self, backward: bool = False) -> str:
        """Which workgroup organisation a bf16x6 launch of this chain takes (``ffn_mlp_bf16x6_organisation``):
        "matrix/vector waves" (csrc/mlp_bf16_mv.hip: the tiny NeRF / Fourier MLP family) or "two waves
        per SIMD" (csrc/mlp_bf16_ws.hip).  Same slab, mask and dZ bits either way; FFN_BF16X6_ORG=ws
        keeps the second for every chain."""
        chain = self.bwd_x6 if backward else self.fwd_x6
        if chain is None:
            raise NotImplementedError("no bf16x6 kernels for this chain")
        fn = _lib.load().ffn_mlp_bf16x6_organisation
        fn.restype = ctypes.c_int
        return "matrix/vector waves" if fn(ctypes.byref(chain), ctypes.c_int(1 if backward else 0)) == 1 else "two waves per SIMD"

    def _x6_ready(self):
        if self.fwd_x6 is None:
            raise NotImplementedError("the bf16x6 kernels cover chains of <= 256 channels per layer "
                                      "whose logits heads are fused")
        if self._packed_x6_dirty:
            self.pack16(parts=3)

    def forward16(self, positions: torch.Tensor, views: Optional[torch.Tensor]) -> torch.Tensor:
        """Inference in the opt-in split-bf16 mode (3 bf16 matrix products per f32 product):
        positions (N,3) [views (N,3)] -> raw logits (N,4)."""
        if self.fwd16 is None:
            raise NotImplementedError("the split-bf16 kernels cover chains whose logits heads are fused")
        if self._packed16_dirty:
            self.pack16()
        n = positions.shape[0]
        logits = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        _call("ffn_mlp_forward_bf16x3", ctypes.byref(self.fwd16), _dev(self.packed16, torch.int16),
              _dev(self.bias_buf), _dev(positions, name="positions"), _dev(views, name="views"),
              c_i64(n), _dev(logits))
        return logits

    def _pack_jobs(self):
        """Device table of everything ``pack`` derives from the nn.Linear tensors: operand packs
        for the forward and backward-data chains, bias blocks, fused-head blocks.  Built once
        per program: the parameters' storage is fixed for its lifetime (``_FusedModel.program``
        rebuilds the program when a parameter's data pointer changes)."""
        jobs = []
        f32 = 4

        def copy(src, src_ld, rows, cols, dst, dst_rs, dst_cs):
            jobs.append(FfnPackJob(src, dst, 0, 1, rows, cols, src_ld, 0, 0, 0, dst_rs, dst_cs, 0))

        for i, spec in enumerate(self.layers):
            if self.step_of[i] is None:
                continue
            L = self.fwd.step[self.step_of[i]]
            groups, tiles = self.fwd_shapes[i]
            w = spec.weight.detach()
            assert w.stride(1) == 1
            jobs.append(FfnPackJob(w.data_ptr(), self.packed_fwd.data_ptr() + f32 * L.w_off,
                                   self.col_maps[i].data_ptr(), 0, w.shape[0], w.shape[1],
                                   w.stride(0), 0, groups, tiles, 0, 0, 0))
            copy(spec.bias.detach().data_ptr(), spec.out, 1, spec.out,
                 self.bias_buf.data_ptr() + f32 * L.b_off, 0, 1)
        for (i, off, channels) in self.fused_heads:
            spec = self.layers[i]
            col, cnt = spec.to_logits
            w = spec.weight.detach()
            copy(spec.bias.detach().data_ptr(), cnt, 1, cnt, self.bias_buf.data_ptr() + f32 * (off + col), 0, 1)
            # head rows as [channel][column]: dst[c*4 + r] = W[r][c]
            copy(w.data_ptr(), w.stride(0), cnt, channels,
                 self.bias_buf.data_ptr() + f32 * (off + 4 + col), 1, 4)
        for (c, groups, tiles, off) in self.bwd_packs:
            w = self.layers[c].weight.detach()
            # operand rows = input channels (act part), operand K = output rows of layer c
            jobs.append(FfnPackJob(w.data_ptr(), self.packed_bwd.data_ptr() + f32 * off, 0, 0,
                                   w.shape[0], self.layers[c].act_in, w.stride(0), 1, groups,
                                   tiles, 0, 0, 0))
        self._pack_job_count = len(jobs)
        self._pack_jobs_dev = _struct_array_to_device(jobs, self.device)

    def pack(self):
        """Re-derives the MFMA-operand copies (and the bias / fused-head blocks) from the current
        nn.Linear weights: one launch over the job table."""
        self._packed16_dirty = True
        self._packed_x6_dirty = True
        if getattr(self, "_pack_jobs_dev", None) is None:
            self._pack_jobs()
        _call("ffn_mlp_pack_jobs", _dev(self._pack_jobs_dev, torch.uint8), c_i(self._pack_job_count))

    # ------------------------------------------------------------------ launches
    @staticmethod
    def plan_blocks(n: int) -> int:
        """Block count the weight-gradient plan is made for: the real count rounded up to 6
        significant bits (<= 1.6 % more), so that batches whose valid-ray count wobbles from
        step to step share one plan; the kernel clamps the plan to the real count."""
        blocks = (n + 31) // 32
        if blocks <= 64:
            return blocks
        unit = 1 << (blocks.bit_length() - 6)
        return -(-blocks // unit) * unit

    def workspace(self, n: int) -> Workspace:
        key = self.plan_blocks(n)
        ws = self._workspaces.get(key)
        if ws is None:
            if len(self._workspaces) >= 8:
                self._workspaces.clear()
            ws = Workspace(self, key * 32)
            self._workspaces[key] = ws
        return ws

    def saved_floats(self, n: int) -> int:
        """Size (in floats) of the per-call training buffer: activation slabs followed by
        the ReLU sign masks (256 words per 32-sample block and slab) and, for chains that can
        run a launch's tail on the two-waves-per-block kernels, that tail's masks (512 words per
        block and slab, at most half a round of blocks)."""
        blocks = (n + 31) // 32
        return (self.saved_channels * 32 * blocks + self.fwd.num_slots * self.mask_words * blocks
                + self._tail_mask_floats())

    def _split_saved(self, saved: torch.Tensor, n: int):
        blocks = (n + 31) // 32
        acts = self.saved_channels * 32 * blocks
        return saved[:acts], saved[acts:acts + self.fwd.num_slots * self.mask_words * blocks]

    def slab_rows(self, saved: torch.Tensor, n: int, layer: int) -> torch.Tensor:
        """The output of hidden layer ``layer`` (an index into ``self.layers``) as an ``(n, out)``
        tensor, read back from the activation slab a training forward left in ``saved`` -- a copy
        for `FourierFeatureMLP.keep_activations` (fourier_feature_models.py:74-75), off the hot
        path.  Slab layout (section 3 of DESIGN.md): per 32-sample block ``C * 32`` floats as
        float4 ``[cq][pos]``, ``cq = channel / 4``, ``pos = sample ^ (cq & 15)``."""
        slot = self.slot_of.get(layer) if isinstance(self.slot_of, dict) else None
        if slot is None:
            raise ValueError("layer %d leaves no activation slab" % layer)
        blocks = (n + 31) // 32
        acts, _ = self._split_saved(saved, n)
        ch, off = int(self.fwd.slot_channels[slot]), int(self.fwd.slot_offset[slot])
        region = acts[off * blocks * 32:(off + ch) * blocks * 32].view(blocks, ch // 4, 32, 4)
        cq = torch.arange(ch // 4, device=saved.device)
        pos = torch.arange(32, device=saved.device)[None, :] ^ (cq[:, None] & 15)      # where sample s sits
        rows = region[:, cq[:, None], pos, :]                                          # (blocks, cq, s, 4)
        return rows.permute(0, 2, 1, 3).reshape(blocks * 32, ch)[:n, :self.layers[layer].out].contiguous()

    # ------------------------------------------------------------------ tail on wave pairs
    def _resident_waves(self) -> int:
        if getattr(self, "_waves", None) is None:
            self._waves = 4 * torch.cuda.get_device_properties(self.device).multi_processor_count
        return self._waves

    def _tail_mask_floats(self) -> int:
        """Mask region of a launch's tail: 512 words per slot and block on wave pairs (at most
        waves / 2 blocks), 1024 on quads (at most waves / 4): the same size."""
        if not (TAIL_PAIRS and self.pair_chain_ok) or self.device.type != "cuda":
            return 0
        return self.fwd.num_slots * 512 * (self._resident_waves() // 2)

    def _tail_plan(self, n: int):
        """(blocks the one-wave-per-block launch keeps, waves per block of the tail kernels: 2 | 4),
        or None: at least one full round, a remainder of at most half a round."""
        if not (TAIL_PAIRS and self.pair_chain_ok):
            return None
        blocks = (n + 31) // 32
        waves = self._resident_waves()
        full, rest = divmod(blocks, waves)
        if full < 1 or full > TAIL_MAX_FULL_ROUNDS or rest == 0 or 2 * rest > waves:
            return None
        team = 4 if (self.quad_chain_ok and 4 * rest <= waves) else 2
        return full * waves, team

    def _tail_split(self, n: int) -> Optional[int]:
        plan = self._tail_plan(n)
        return None if plan is None else plan[0]

    def _tail_masks(self, saved: torch.Tensor, n: int) -> torch.Tensor:
        blocks = (n + 31) // 32
        start = self.saved_channels * 32 * blocks + self.fwd.num_slots * self.mask_words * blocks
        return saved[start:start + self._tail_mask_floats()]

    def forward(self, positions: torch.Tensor, views: Optional[torch.Tensor],
                saved: Optional[torch.Tensor] = None, precision: str = "f32") -> torch.Tensor:
        """positions (N,3) [views (N,3)] -> raw logits (N,4).  ``saved`` (a flat float
        buffer of ``saved_floats(N)`` elements) receives what the backward pass needs.
        ``precision="bf16x3"`` (opt-in) runs the split-bf16 kernel, ``"bf16x6"`` (opt-in) the
        f32-accurate three-part split (six bf16 matrix products per f32 product)."""
        n = positions.shape[0]
        logits = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        acts, masks = (None, None) if saved is None else self._split_saved(saved, n)
        if saved is not None:
            # what `backward` must match: the f32 forward writes the tail blocks' sign masks into
            # their own region when the launch is split (`_tail_split`), the split-bf16 kernels
            # know one mask region only
            split = self._tail_plan(n) if precision == "f32" else None
            # (per buffer: the last few forwards are remembered, so that a backward on an older
            # buffer is checked against ITS forward)
            records = getattr(self, "_fwd_records", None)
            if records is None:
                records = self._fwd_records = {}
            records.pop(saved.data_ptr(), None)
            records[saved.data_ptr()] = (saved.data_ptr(), n, precision, split)
            while len(records) > 8:
                records.pop(next(iter(records)))
        if precision == "bf16x3":
            if saved is None:
                return self.forward16(positions, views)
            if self.fwd16 is None:
                raise NotImplementedError("the split-bf16 kernels cover chains whose logits heads are fused")
            if self._packed16_dirty:
                self.pack16()
            _call("ffn_mlp_forward_bf16x3_train", ctypes.byref(self.fwd16),
                  _dev(self.packed16, torch.int16), _dev(self.bias_buf),
                  _dev(positions, name="positions"), _dev(views, name="views"), c_i64(n),
                  _dev(logits), _dev(acts), _dev(masks))
            return logits
        if precision == "bf16x6":
            self._x6_ready()
            if saved is None:
                _call("ffn_mlp_forward_bf16x6", ctypes.byref(self.fwd_x6), _dev(self.packed_x6, torch.int16),
                      _dev(self.bias_buf), _dev(positions, name="positions"), _dev(views, name="views"),
                      c_i64(n), _dev(logits))
            else:
                _call("ffn_mlp_forward_bf16x6_train", ctypes.byref(self.fwd_x6),
                      _dev(self.packed_x6, torch.int16), _dev(self.bias_buf),
                      _dev(positions, name="positions"), _dev(views, name="views"), c_i64(n),
                      _dev(logits), _dev(acts), _dev(masks))
            return logits
        if precision != "f32":
            raise ValueError("precision is 'f32', 'bf16x3' or 'bf16x6'")
        plan = None if saved is None else self._tail_plan(n)
        head = None if plan is None else plan[0]
        if head is None:
            _call("ffn_mlp_forward", ctypes.byref(self.fwd), _dev(self.packed_fwd),
                  _dev(self.bias_buf), _dev(positions, name="positions"),
                  _dev(views, name="views"), c_i64(n), _dev(logits), _dev(acts), _dev(masks),
                  c_i64(0), c_i64(0))
            return logits
        # full rounds on the one-wave-per-block kernel, the short last round on teams of two / four
        # waves per block; both write the batch's slabs (block ids of the whole batch), each its
        # own mask region
        tail_chain = self.fwd_quad if plan[1] == 4 else self.fwd_pair
        blocks, cut = (n + 31) // 32, head * 32
        v_head = None if views is None else views[:cut]
        v_tail = None if views is None else views[cut:]
        _call("ffn_mlp_forward", ctypes.byref(self.fwd), _dev(self.packed_fwd), _dev(self.bias_buf),
              _dev(positions[:cut], name="positions"), _dev(v_head, name="views"), c_i64(cut),
              _dev(logits[:cut]), _dev(acts), _dev(masks), c_i64(0), c_i64(blocks))
        _call("ffn_mlp_forward", ctypes.byref(tail_chain), _dev(self.packed_fwd), _dev(self.bias_buf),
              _dev(positions[cut:], name="positions"), _dev(v_tail, name="views"), c_i64(n - cut),
              _dev(logits[cut:]), _dev(acts), _dev(self._tail_masks(saved, n)), c_i64(head),
              c_i64(blocks))
        return logits

    def render(self, starts: torch.Tensor, directions: torch.Tensor, near_far: torch.Tensor,
               ray_index, num_samples: int, unit: Optional[torch.Tensor],
               t_values: Optional[torch.Tensor] = None, occupancy=None, want_color=True,
               want_depth=False, nan_flag: Optional[torch.Tensor] = None,
               image: Optional[torch.Tensor] = None, pixel_offset: int = 0):
        """Fused inference render of the rays ``ray_index`` -- device int64 ids into the sampler
        state, or a ``(first id, count, valid mask)`` tuple for a contiguous range filtered in
        the kernel (a whole camera, no index list, no host sync): one launch, nothing but the
        per-ray results touches HBM.  Returns
        (color (R,3) | None, alpha (R) | None, depth (R) | None); ``image`` (H,W,3) uint8,
        zeroed by the caller, additionally receives the truncated u8 pixels at
        ``ray id - pixel_offset``.  ``occupancy`` = an OccupancyGrid switches on empty-space
        skipping.  512-wide chains run the kernel's pair-of-waves-per-ray variant."""
        base, valid = 0, None
        if isinstance(ray_index, tuple):
            base, rays, valid = int(ray_index[0]), int(ray_index[1]), ray_index[2]
            ray_index = None
        else:
            rays = int(ray_index.shape[0])
        dev = self.device
        color = torch.empty((rays, 3), dtype=torch.float32, device=dev) if want_color else None
        alpha = torch.empty((rays,), dtype=torch.float32, device=dev) if want_color else None
        depth = torch.empty((rays,), dtype=torch.float32, device=dev) if want_depth else None
        if rays == 0:
            return color, alpha, depth
        if t_values is not None and tuple(t_values.shape) != (rays, num_samples):
            raise ValueError("t_values must be (num_rays, num_samples)")
        # the struct fields below are raw integers, which the same-device check of ops._call
        # cannot see: a sampler / grid / output on another GPU than the model must fail HERE,
        # not as an illegal access inside the kernel
        named = dict(starts=starts, directions=directions, near_far=near_far, ray_index=ray_index,
                     valid=valid, unit=unit, t_values=t_values, nan_flag=nan_flag, image=image,
                     occupancy=None if occupancy is None else occupancy.bits)
        for name, tensor in named.items():
            if tensor is not None and tensor.device != dev:
                raise RuntimeError("ffn_render_fused_fwd: %s lives on %s but the model is on %s"
                                   % (name, tensor.device, dev))
        ptr = lambda t, dtype=torch.float32: _dev(t, dtype).value or 0     # noqa: E731
        rr = FfnRenderRays(ptr(starts), ptr(directions), ptr(near_far), int(near_far.shape[1]),
                           ptr(ray_index, torch.int64), base, ptr(valid, torch.uint8), rays,
                           int(num_samples), ptr(unit),
                           ptr(t_values))
        occ = None
        if occupancy is not None:
            occ = FfnOccupancy(ptr(occupancy.bits, torch.int32),
                               (ctypes.c_float * 3)(*occupancy.box_min),
                               (ctypes.c_float * 3)(*occupancy.box_size), int(occupancy.resolution))
        out = FfnRenderOut(ptr(color), ptr(alpha), ptr(depth), ptr(nan_flag, torch.int32),
                           ptr(image, torch.uint8), int(pixel_offset))
        _call("ffn_render_fused_fwd", ctypes.byref(self.fwd), _dev(self.packed_fwd),
              _dev(self.bias_buf), ctypes.byref(rr), None if occ is None else ctypes.byref(occ),
              ctypes.byref(out))
        return color, alpha, depth

    def focus_sample(self, starts, directions, near_far, ray_index, num_samples: int, n_focus: int,
                     unit_focus, u, t_io):
        """Fused live focus sampling with THIS chain as the opacity model (see
        ``ffn_focus_fused``): fills ``t_io`` (R,S) in place."""
        if self.big or n_focus > 64:
            raise NotImplementedError("fused focus sampling: chains of up to 512 channels, at most 64 probe points")
        _call("ffn_focus_fused", ctypes.byref(self.fwd), _dev(self.packed_fwd), _dev(self.bias_buf),
              _dev(starts), _dev(directions), _dev(near_far), c_i64(near_far.shape[1]),
              _dev(ray_index, torch.int64), c_i(ray_index.shape[0]), c_i(num_samples), c_i(n_focus),
              _dev(unit_focus), _dev(u), _dev(t_io))
        return t_io

    def backward(self, d_logits: torch.Tensor, positions: torch.Tensor,
                 views: Optional[torch.Tensor], saved: torch.Tensor, grads: torch.Tensor,
                 precision: str = "f32"):
        """Fills ``grads`` (flat, num_grad_floats) from d(loss)/d(logits) (N,4) and the
        activations ``saved`` by the matching forward call, in the SAME ``precision`` and with the
        same tail split: the slab formats are shared by the two modes, but an f32 forward whose
        launch was split (``_tail_split``) leaves the tail blocks' ReLU masks in a region only the
        f32 backward reads -- a mismatch raises instead of differentiating with stale masks.
        ``precision="bf16x3"`` (opt-in) runs the split-bf16 backward-data and weight-gradient
        kernels; ``"bf16x6"`` (opt-in) the f32-accurate three-part backward-data kernel and a
        TWO-LAUNCH weight-gradient plan over one partial buffer: every full unit (four 128 x 128
        quadrants) and the logits-head units on the three-part kernel
        (``ffn_mlp_wgrad_units_bf16x6``), units with fewer quadrants -- NeRF's 63- / 27-channel
        encodings, its 128-channel view layer -- on the exact-f32 kernel, which folds narrow
        windows (``_plan_wgrad``).  ``FFN_BF16X6_WGRAD=f32`` puts every unit on the exact-f32 kernel.
        Forward / backward pairs: the same mode on both sides, or an exact-f32 forward under any
        backward (the slab, mask and save-on-consume layouts are shared and both directions are
        tested against each other); any other mix raises."""
        n = positions.shape[0]
        if n == 0:                      # an empty batch contributes no gradient
            return grads.zero_()
        record = (getattr(self, "_fwd_records", None) or {}).get(saved.data_ptr())
        if record is not None and record[1] == n:
            if record[2] != precision and record[2] != "f32":
                # the layouts of the three kernel families are the same TODAY; only the pairs the
                # tests differentiate through (same mode, or an exact-f32 forward) are let through
                raise RuntimeError("MlpProgram.backward: `saved` was filled by a %s forward; a %s backward "
                                   "on it is not a tested pair (same mode, or an f32 forward)"
                                   % (record[2], precision))
            # the slab and dZ formats are shared by the modes; what must match is WHERE the ReLU
            # masks of the launch's tail blocks were written (an f32 forward with a tail split keeps
            # them in a region of their own that only the f32 backward with the same split reads)
            split = self._tail_plan(n) if precision == "f32" else None
            if record[3] != split:
                raise RuntimeError("MlpProgram.backward: `saved` was filled by a %s forward (tail "
                                   "split %s) but the backward was asked for %s (tail split %s)"
                                   % (record[2], record[3], precision, split))
        ws = self.workspace(n)
        # (units are <= 256 x 256 windows at any layer width)  bf16x6: the three-part units, or --
        # FFN_BF16X6_WGRAD=f32 -- the exact-f32 units on the slabs the bf16x6 chain kernels wrote
        wgrad_mode = precision if precision == "bf16x3" else "f32"
        if precision == "bf16x6" and os.environ.get("FFN_BF16X6_WGRAD", "bf16x6") != "f32":
            wgrad_mode = "bf16x6"
        ws.use_plan(wgrad_mode)
        whole = saved
        saved, masks = self._split_saved(saved, n)
        if precision == "bf16x6":
            self._x6_ready()
            if self.bwd_x6 is not None:
                _call("ffn_mlp_backward_data_bf16x6", ctypes.byref(self.bwd_x6),
                      _dev(self.packed_x6_bwd, torch.int16), _dev(d_logits), c_i64(n), _dev(masks),
                      _dev(ws.dz))
        elif precision == "bf16x3" and self.bwd16 is not None:
            if self._packed16_dirty:
                self.pack16()
            _call("ffn_mlp_backward_data_bf16x3", ctypes.byref(self.bwd16),
                  _dev(self.packed16_bwd, torch.int16), _dev(d_logits), c_i64(n), _dev(masks),
                  _dev(ws.dz))
        elif self.bwd.num_steps > 0:
            plan = self._tail_plan(n) if precision == "f32" else None
            head = None if plan is None else plan[0]
            if head is None:
                _call("ffn_mlp_backward_data", ctypes.byref(self.bwd), _dev(self.packed_bwd),
                      _dev(d_logits), c_i64(n), _dev(masks), _dev(ws.dz), c_i64(0), c_i64(0))
            else:           # the split of the matching forward call
                tail_chain = self.bwd_quad if plan[1] == 4 else self.bwd_pair
                blocks, cut = (n + 31) // 32, head * 32
                _call("ffn_mlp_backward_data", ctypes.byref(self.bwd), _dev(self.packed_bwd),
                      _dev(d_logits[:cut]), c_i64(cut), _dev(masks), _dev(ws.dz), c_i64(0),
                      c_i64(blocks))
                _call("ffn_mlp_backward_data", ctypes.byref(tail_chain), _dev(self.packed_bwd),
                      _dev(d_logits[cut:]), c_i64(n - cut), _dev(self._tail_masks(whole, n)),
                      _dev(ws.dz), c_i64(head), c_i64(blocks))
        for kind, segments, starts in ws.launches:
            _call({"bf16x3": "ffn_mlp_wgrad_units_bf16x3", "bf16x6": "ffn_mlp_wgrad_units_bf16x6"}.get(
                      kind, "ffn_mlp_wgrad_units"),
                  ctypes.byref(self.fwd),
                  _dev(self.wgrad_units_dev, torch.uint8), _dev(segments, torch.uint8),
                  _dev(starts, torch.int32), c_i(WGRAD_GROUPS), _dev(saved),
                  _dev(ws.dz), _dev(d_logits), c_i64(n), _dev(ws.partials))
        _call("ffn_mlp_wgrad_reduce", _dev(ws.reduce_jobs, torch.uint8),
                  c_i(ws.num_reduce_jobs), _dev(ws.partials), _dev(grads))
        return grads
