"""ControlNet3D (BEV-layout branch) and ControlledUNetModel3D on MI355X — host-side mirror of
`sgm/modules/diffusionmodules/controlmodel.py`.

  ControlNet3D.forward :86-142            hint stem (8 convs, SiLU fused into the conv epilogue, fp16
                                          between layers) -> encoder + middle on tokens -> 13 zero 1x1
                                          convs (GEMM, control_scales folded into the packed weights)
  ControlledUNetModel3D.forward :160-202  `h += control.pop()` and `cat([h, hs.pop() + control.pop()])`
                                          are one fused pass each (pnc_add_f32 / pnc_concat_add)
"""
from __future__ import annotations

import importlib
from typing import List

import torch
import torch.nn as nn

from .. import engine as E
from ..engine import Act, Runtime
from .openaimodel import TimestepEmbedSequential, UNetModel3D, conv_params, run_conv3x3
from .util import conv_nd, runtime_for, zero_module

HINT_STRIDES = (1, 1, 2, 1, 2, 1, 2, 1)


def instantiate_from_config(config):
    """sgm/util.py:168-185 — {"target": dotted.path, "params": {...}}.  `sgm.modules...` targets resolve to the
    mirror classes of this package (the `sgm` shim re-exports them under the reference's import paths)."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    module, cls = config["target"].rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)(**dict(config.get("params", dict())))


class ControlNet3D(UNetModel3D):
    """controlmodel.py:19-142"""

    def __init__(self, hint_channels, control_scales, dims=2, disable_temporal=False, *args, **kwargs):
        kwargs["out_channels"] = kwargs["in_channels"]
        if disable_temporal:
            raise NotImplementedError("disable_temporal is not used by the Panacea inference configuration")
        if hint_channels > 19:
            raise NotImplementedError("hint_channels > 19 (controlmodel.py:108-117) is a dead path at 19 channels")
        self.control_scales, self.hint_channels, self.disable_temporal = control_scales, hint_channels, disable_temporal
        super().__init__(*args, **kwargs)
        model_channels, channel_mult = kwargs["model_channels"], kwargs["channel_mult"]
        del self.output_blocks
        del self.out
        chans = [hint_channels, 16, 16, 32, 32, 96, 96, 256, model_channels]
        layers = []
        for i, s in enumerate(HINT_STRIDES):
            conv = conv_nd(dims, chans[i], chans[i + 1], 3, padding=1, stride=s)
            layers.append(conv if i < 7 else zero_module(conv))
            if i < 7:
                layers.append(nn.SiLU())
        self.input_hint_block = TimestepEmbedSequential(*layers)
        self.zero_convs = nn.ModuleList([self.make_zero_conv(model_channels, dims=2)])
        ch = model_channels
        for level, mult in enumerate(channel_mult):
            for _ in range(self.num_res_blocks[level]):
                ch = mult * model_channels
                self.zero_convs.append(self.make_zero_conv(ch, dims=2))
            if level != len(channel_mult) - 1:
                self.zero_convs.append(self.make_zero_conv(ch, dims=2))
        self.middle_block_out = self.make_zero_conv(ch, dims=2)

    def make_zero_conv(self, channels, dims=2):
        return TimestepEmbedSequential(zero_module(conv_nd(dims, channels, channels, 1, padding=0)))

    def _pack(self):
        pk = super()._pack()
        pk["hint"] = [conv_params(m) for m in self.input_hint_block if isinstance(m, nn.Conv2d)]
        s = float(self.control_scales)
        zc = []
        for z in list(self.zero_convs) + [self.middle_block_out]:
            w, b = conv_params(z[0])
            zc.append(((w.float() * s).half().contiguous(), b * s) if s != 1.0 else (w, b))
        pk["zero"] = zc
        return pk

    def _hint_stem(self, rt: Runtime, hint: torch.Tensor) -> Act:
        """input_hint_block (controlmodel.py:43-59) on the image-resolution BEV layout."""
        pk = self.packed()
        F, C, H, W = hint.shape
        cp = (C + 7) // 8 * 8
        t16 = rt.empty((F * H * W, cp), torch.float16)
        t16lo = rt.lo_plane((F * H * W, cp), "conv_mid")
        rt.be.nchw_to_tokens_f16(hint.detach().to(torch.float32).contiguous(), C, None, 0, F, H * W, cp, t16, t16lo)
        a = Act(F, H, W, cp, f16=t16, f16_lo=t16lo)
        for i, ((w, b), s) in enumerate(zip(pk["hint"], HINT_STRIDES)):
            last = i == len(HINT_STRIDES) - 1
            a = run_conv3x3(rt, a.f16, a.F, a.H, a.W, a.C, w, b, w.shape[0], stride=s, act_silu=not last,
                            out32=last, out16=not last, x16_lo=a.f16_lo, split_out="conv_mid")
        return a

    def _run_control(self, rt: Runtime, x16: Act, hint: torch.Tensor, emb32: torch.Tensor) -> List[Act]:
        pk = self.packed()
        self._project_text(rt)
        self._project_emb(rt, emb32)
        guided = rt.guided if rt.guided is not None else self._hint_stem(rt, hint)
        outs, h = [], x16
        for i, module in enumerate(self.input_blocks):
            h = module._run(rt, h, emb32, want_f16=(i != 0))
            if rt.trace is not None:
                rt.trace[f"controlnet.input_blocks.{i}"] = h.to_nchw()
            if i == 0:
                if (guided.H, guided.W, guided.C) != (h.H, h.W, h.C):
                    raise ValueError(f"hint stem output {guided.H}x{guided.W}x{guided.C} does not match the latent "
                                     f"{h.H}x{h.W}x{h.C} (the hint must be 8x the latent resolution)")
                if guided.F < 1 or h.F % guided.F:
                    raise ValueError(f"{guided.F} hint frames do not tile the batch of {h.F} frames")
                h.f16 = rt.empty((h.M, h.C), torch.float16)
                h.f16_lo = rt.lo_plane((h.M, h.C), "stream")
                # h += guided_hint.  A hint of F / k frames is shared by the k groups of the batch (the two CFG halves carry
                # the SAME BEV layout: the fused sampler step hands it over once and the stem runs on T frames, not 2 T)
                mg = guided.M * h.C
                for r in range(h.F // guided.F):
                    sl = slice(r * guided.M, (r + 1) * guided.M)
                    rt.be.add_f32(h.f32[sl], guided.f32, mg, h.f32[sl], h.f16[sl], None if h.f16_lo is None else h.f16_lo[sl])
            outs.append(self._zero_conv(rt, h, pk, i))
        h = self.middle_block._run(rt, h, emb32, want_f16=True)
        if rt.trace is not None:
            rt.trace["controlnet.middle_block"] = h.to_nchw()
        outs.append(self._zero_conv(rt, h, pk, len(pk["zero"]) - 1))
        return outs

    @staticmethod
    def _zero_conv(rt: Runtime, h: Act, pk: dict, idx: int) -> Act:
        w16, b = pk["zero"][idx]
        o = rt.empty((h.M, h.C), torch.float32)
        x16 = h.need_f16(rt)
        rt.be.gemm(x16, w16, M=h.M, N=h.C, K=h.C, lda=h.C, bias=b, out32=o, ldc32=h.C, a16_lo=h.f16_lo,
                   w_lo=E.wlo(pk, ("zero", idx), h.f16_lo, w16))
        return Act(h.F, h.H, h.W, h.C, f32=o)

    def forward(self, x, hint, timesteps=None, context=None, y=None, **kwargs):
        """controlmodel.py:86-142 — returns the 13 residuals (encoder order, middle last) as NCHW tensors."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        with torch.no_grad():
            rt = runtime_for(x, self.num_frames, self.frame_shard, self.view_shard)
            rt.prec = E.precision(self.precision)
            rt.set_context(context)
            emb = self._time_embedding(rt, timesteps)
            outs = self._run_control(rt, self._stem_tokens(rt, x), hint, emb)
        return [o.to_nchw().to(x.dtype) for o in outs]


class ControlledUNetModel3D(UNetModel3D):
    """controlmodel.py:146-202"""

    def __init__(self, controlnet_config=None, only_add_on_center_frame=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if controlnet_config is not None:
            self.controlnet = instantiate_from_config(controlnet_config)

    def _to_control_acts(self, rt: Runtime, control) -> list:
        from .util import act_from_nchw
        return [c if isinstance(c, Act) else act_from_nchw(rt, c) for c in control]

    def forward(self, x, timesteps=None, context=None, y=None, control=None, **kwargs):
        """controlmodel.py:160-202.  `control` is consumed from the end, like the reference's pop()s."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        with torch.no_grad():
            rt = runtime_for(x, self.num_frames, self.frame_shard, self.view_shard)
            rt.prec = E.precision(self.precision)
            rt.set_context(context)
            emb = self._time_embedding(rt, timesteps)
            acts = None
            if control is not None:
                acts = self._to_control_acts(rt, control)
                del control[:]
            out = self._run_unet(rt, self._stem_tokens(rt, x), emb, acts)
        return out.to(x.dtype)

    def prepare(self, context: torch.Tensor, hint: torch.Tensor) -> "StepInvariants":
        """Everything on the path that does not depend on (x, t): the guided hint (input_hint_block of the BEV layout,
        controlmodel.py:43-59,125-129) and the text K/V of the 138 cross-attention sites of both networks.  The
        reference recomputes them in every one of the 25/50 sampler steps; a sampler that keeps `cond` fixed may
        compute them once — `denoise(..., invariants=inv)` is then bit-identical to the plain call."""
        with torch.no_grad():
            sh = self.frame_shard
            t_local = self.num_frames // (sh.G if sh is not None else 1)
            # the batch size comes from the CONTEXT (one row of text per sample); the hint holds either every frame of the batch
            # or — shared by the CFG halves, as the fused sampler step hands it over — the frames of one sample
            B = context.shape[0]
            if hint.shape[0] not in (B * t_local, t_local):
                raise ValueError(f"hint holds {hint.shape[0]} frames; expected {B * t_local} (every frame of the {B}-sample batch) "
                                 f"or {t_local} (one layout shared by the samples)")
            rt = Runtime(hint.device, B, self.num_frames, sh, self.view_shard)
            rt.prec = E.precision(self.precision)
            rt.set_context(context)
            self._project_text(rt)
            self.controlnet._project_text(rt)
            guided = self.controlnet._hint_stem(rt, hint.detach().to(torch.float32).contiguous())
        return StepInvariants(rt.ctx16, rt.n_text, dict(rt.text_kv), guided, (context, hint), rt.prec)

    two_stream = True      # run the ControlNet branch on a second HIP stream, concurrently with the UNet encoder
    split_samples = False  # additionally run every sample of the batch (CFG half) as its own stream pair

    def denoise(self, x, timesteps, context, hint, trace=None, invariants=None) -> torch.Tensor:
        """ControlNet + UNet in one runtime (what OpenAIWrapperControlLDM3D.forward calls): the stem tokens and
        the text context are prepared once and the 13 residuals stay in the resident layout.  The ControlNet only
        feeds the UNet from its middle block on (controlmodel.py:191-195), so it runs on a side stream next to the
        UNet encoder: its small-grid kernels (L2/L3, 4x48 tokens) fill CUs the encoder's leave idle, and vice versa.
        Samples of the batch never interact inside the network, so they can be issued as independent stream pairs."""
        T = self.num_frames // (self.frame_shard.G if self.frame_shard is not None else 1)   # frames per sample on this rank
        B = x.shape[0] // T
        if invariants is not None:
            return self._denoise_one(x, timesteps, context, hint, trace, 0, invariants)
        if hint is not None and hint.shape[0] != x.shape[0]:
            # this entry slices the hint per sample (split_samples) and per frame: a layout shared by the CFG halves (F / 2
            # frames) belongs to denoise_tokens(), which broadcasts the hint stem's output instead
            raise ValueError(f"hint holds {hint.shape[0]} frames for a batch of {x.shape[0]}; pass one hint frame per latent frame")
        if self.split_samples and self.two_stream and x.is_cuda and trace is None and B > 1:
            main = torch.cuda.current_stream()
            outs = []
            for b in range(B):
                sb = _side_stream(x.device, 2 * b)
                sb.wait_stream(main)
                with torch.cuda.stream(sb):
                    sl = slice(b * T, (b + 1) * T)
                    outs.append(self._denoise_one(x[sl], timesteps[sl], context[b:b + 1], hint[sl], None, 2 * b + 1))
                for t in (x, timesteps, context, hint):
                    t.record_stream(sb)
            for b in range(B):
                main.wait_stream(_side_stream(x.device, 2 * b))
                outs[b].record_stream(main)
            return torch.cat(outs, dim=0)
        return self._denoise_one(x, timesteps, context, hint, trace, 0)

    def denoise_tokens(self, x, c_in, timesteps, context, concat, hint, invariants=None) -> Act:
        """Fused sampler entry: `x` is the UNSCALED latent of one CFG half, `c_in` the per-frame input scale of
        the whole batch, `concat` the batch's conditioning latents; returns eps as channels-last fp32 tokens (Act.f32
        [F*h*w, 4]) for pnc_cfg_euler_step.  Same network evaluation as `denoise(cat(x * c_in, concat), ...)`.  `hint` may hold
        the BEV layout of ONE CFG half (F / 2 frames) when both halves share it: the hint stem then runs once per step on those
        frames (controlmodel.py:125-129 evaluates it on the doubled batch: 2x the work for the same result)."""
        return self._denoise_one(x, timesteps, context, hint, None, 0, invariants, fused=(c_in, concat))

    def _denoise_one(self, x, timesteps, context, hint, trace, side_idx, inv=None, fused=None):
        with torch.no_grad():
            rt = runtime_for(x if fused is None else fused[1], self.num_frames, self.frame_shard, self.view_shard)
            rt.prec = E.precision(self.precision)
            rt.trace = trace
            if inv is not None:
                inv.check(rt, context, hint)
                rt.ctx16, rt.n_text, rt.text_kv, rt.text_frozen = inv.ctx16, inv.n_text, dict(inv.text_kv), True
                rt.guided = inv.guided
            else:
                rt.set_context(context)
            x16 = self._stem_tokens(rt, x) if fused is None else self._stem_tokens(rt, x, fused[1], fused[0])
            cn = self.controlnet
            hint32 = hint if inv is not None else hint.detach().to(torch.float32).contiguous()
            # frame- / view-sharded runs issue collectives from both networks.  With shard objects of its own (process groups of its
            # own: parallel.Groups(side=True), apply_*_shard(net, shard, side_shard)) the ControlNet's collectives have one order per
            # communicator whatever the interleaving, and it runs on its side stream as on one GPU (round 5); sharing the UNet's
            # shard objects it stays on the main stream (one stream keeps the order of ONE communicator identical on all ranks)
            if (rt.shard is not None and cn.frame_shard is None) or (rt.vshard is not None and cn.view_shard is None):
                raise ValueError("the UNet is frame- / view-sharded but its ControlNet is not: it would silently evaluate this rank's "
                                 "slice as if it were the whole clip (set both with parallel.apply_frame_shard / apply_view_shard)")
            own_groups = (rt.shard is None or cn.frame_shard is not rt.shard) and (rt.vshard is None or cn.view_shard is not rt.vshard)
            rt_cn = rt
            if own_groups and (rt.shard is not None or rt.vshard is not None):
                import copy
                rt_cn = copy.copy(rt)                  # same context / text K,V / invariants, the ControlNet's own shards
                rt_cn.shard, rt_cn.vshard = cn.frame_shard, cn.view_shard
            if self.two_stream and x.is_cuda and trace is None and own_groups:
                main = torch.cuda.current_stream()
                # per-device tables that both streams read are created HERE, on the main stream, before the fork: their
                # first use would otherwise be an H2D copy on the side stream that the main stream does not wait for
                E.timestep_freqs(self.model_channels, x.device)
                E.timestep_freqs(cn.model_channels, x.device)
                self.packed(), cn.packed()
                side = _side_stream(x.device, side_idx)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    control = cn._run_control(rt_cn, x16, hint32, cn._time_embedding(rt_cn, timesteps))
                for t in (x16.f16, rt.ctx16, hint32, timesteps):
                    t.record_stream(side)

                def join():
                    main.wait_stream(side)
                    for c in control:
                        c.f32.record_stream(main)
                    return control
                out = self._run_unet(rt, x16, self._time_embedding(rt, timesteps), join, tokens=fused is not None)
            else:
                control = cn._run_control(rt_cn, x16, hint32, cn._time_embedding(rt_cn, timesteps))
                if trace is not None:
                    for j, c in enumerate(control):
                        trace[f"control.{j}"] = c.to_nchw()
                out = self._run_unet(rt, x16, self._time_embedding(rt, timesteps), control, tokens=fused is not None)
            self._range_monitor_collect(rt)          # (both networks' kernels are ordered before this point of the current stream)
        return out if fused is not None else out.to(x.dtype)


class StepInvariants:
    """Result of ControlledUNetModel3D.prepare(): tensors that are constant over the sampler steps of one sample."""

    def __init__(self, ctx16, n_text, text_kv, guided, sources, prec=E.FAST):
        self.ctx16, self.n_text, self.text_kv, self.guided, self.prec = ctx16, n_text, text_kv, guided, prec
        self._src = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in sources)

    def check(self, rt: Runtime, context, hint):
        """The invariants belong to ONE (context, hint) pair: refuse anything else instead of silently reusing them."""
        now = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in (context, hint))
        if now != self._src:
            raise ValueError("StepInvariants were prepared for different (or since modified) context / hint tensors")
        if self.prec != rt.prec:
            raise ValueError(f"StepInvariants were prepared with precision {self.prec.name}, the network runs {rt.prec.name}")
        # the guided hint holds every frame of the batch, or — prepared from a layout shared by the CFG halves — the frames of one
        # sample, which _run_control broadcasts over the samples (prepare() accepts both, so check() must too)
        if self.guided.F != rt.F and (self.guided.F != rt.F // max(1, rt.B) or rt.F % self.guided.F):
            raise ValueError(f"StepInvariants hold {self.guided.F} frames, the batch has {rt.F} ({rt.B} samples)")


_SIDE_STREAMS = {}
# Round 6 experiment (bench.py --cu-split): name of a hip.CU_MASK_PATTERNS entry.  With split_samples the stream pair of sample b
# (indices 2b, 2b + 1) is then confined to half b % 2 of the CUs: two sample chains side by side on disjoint halves of the chip
CU_SPLIT = None


def _side_stream(device, idx: int = 0) -> "torch.cuda.Stream":
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), idx, CU_SPLIT)
    if key not in _SIDE_STREAMS:
        if CU_SPLIT is not None:
            _SIDE_STREAMS[key] = E._hip.masked_stream(E._hip.CU_MASK_PATTERNS[CU_SPLIT][(idx // 2) % 2])
        else:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]
