"""First-stage decoder — host-side mirror of `sgm/modules/diffusionmodules/model.py` (Decoder, ResnetBlock,
AttnBlock, Upsample, Normalize) plus the `post_quant_conv` of `sgm/models/autoencoder.py:AutoencoderKL` (SURVEY.md
§8 f2).  Same constructor keywords, module tree and state-dict names as the reference, so the `first_stage_model.*`
sub-tree of a checkpoint loads with `strict=False`; all arithmetic runs through the C-ABI kernels of the denoiser:

* resident layout = channels-last tokens `[F*H*W, C]`, fp32 stream + fp16 operands (engine.py);
* `Normalize` (GroupNorm 32, eps 1e-6) + swish (= SiLU, model.py:54-57)  -> pnc_groupnorm_stats/apply (silu flag);
* every 3x3 conv, incl. nearest-2x `Upsample` + conv (model.py:65-77)     -> pnc_gemm_f16 implicit GEMM (upsample flag);
* 1x1 `nin_shortcut`, q / k / v / proj_out                                 -> pnc_gemm_f16, residual adds in its epilogue;
* the single-head d = C attention of the mid block (model.py:374-415)     -> S = Q K^T (fp32) -> pnc_softmax_rows_f16
  -> O = P V against the channel-major V^T the v-projection writes; one frame at a time (12288^2 scores at 256x3072).

The per-module `forward`s take/return NCHW tensors like the reference; `Decoder.forward` keeps everything resident.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import engine as E
from ..engine import Act, Packable, Runtime
from .util import act_from_nchw


def Normalize(in_channels, num_groups=32):
    """model.py:59-62"""
    return torch.nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _conv3x3_params(conv: nn.Conv2d):
    return E.pk_conv3x3(conv.weight), E.pk_f32(conv.bias)


def _conv1x1_params(conv: nn.Conv2d):
    return E.pk_linear(conv.weight), E.pk_f32(conv.bias)


def _gn_params(gn: nn.GroupNorm):
    if gn.num_groups != 32:
        raise NotImplementedError("only GroupNorm(32, C) is on the path")
    return E.pk_f32(gn.weight), E.pk_f32(gn.bias)


def _conv3x3(rt: Runtime, x16, F, Hin, Win, Cin, w16, b, Cout, *, upsample=False, down_br=False, res32=None,
             out16=False) -> Act:
    if down_br and (Hin % 2 or Win % 2):
        raise NotImplementedError("the stride-2 Downsample needs even image sizes")
    Hout, Wout = (2 * Hin, 2 * Win) if upsample else ((Hin // 2, Win // 2) if down_br else (Hin, Win))
    M = F * Hout * Wout
    o32 = rt.empty((M, Cout), torch.float32)
    o16 = rt.empty((M, Cout), torch.float16) if out16 else None
    rt.be.gemm(x16, w16, M=M, N=Cout, K=9 * Cin, a_mode=E._hip.A_CONV3X3,
               conv=dict(Cin=Cin, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=2 if down_br else 1,
                         upsample=int(upsample), pad_br=int(down_br)),
               bias=b, res1=res32, ldr1=Cout, out32=o32, ldc32=Cout, out16=o16, ldc16=Cout)
    return Act(F, Hout, Wout, Cout, f32=o32, f16=o16)


class Upsample(nn.Module, Packable):
    """model.py:65-77: nearest x2, then conv3x3 (folded into the conv's input gather)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Upsample without conv is not on the path (resamp_with_conv=True)")
        self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self._init_packable()

    def _pack(self):
        return dict(c=_conv3x3_params(self.conv))

    def _run(self, rt: Runtime, x: Act) -> Act:
        w, b = self.packed()["c"]
        return _conv3x3(rt, x.need_f16(rt), x.F, x.H, x.W, x.C, w, b, x.C, upsample=True)

    def forward(self, x):
        rt = Runtime(x.device, x.shape[0], 1)
        return self._run(rt, act_from_nchw(rt, x)).to_nchw().to(x.dtype)


class Downsample(nn.Module, Packable):
    """model.py:98-113: F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2 padding 0 (zero padding bottom / right only)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Downsample without conv (avg_pool2d) is not on the path (resamp_with_conv=True)")
        self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)
        self._init_packable()

    def _pack(self):
        return dict(c=_conv3x3_params(self.conv))

    def _run(self, rt: Runtime, x: Act) -> Act:
        w, b = self.packed()["c"]
        return _conv3x3(rt, x.need_f16(rt), x.F, x.H, x.W, x.C, w, b, x.C, down_br=True)

    def forward(self, x):
        rt = Runtime(x.device, x.shape[0], 1)
        return self._run(rt, act_from_nchw(rt, x)).to_nchw().to(x.dtype)


class ResnetBlock(nn.Module, Packable):
    """model.py:139-196 with temb_channels = 0 (the autoencoder has no timestep embedding)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        if temb_channels > 0:
            raise NotImplementedError("ResnetBlock with a timestep embedding is not on the first-stage path")
        if in_channels % 64 or out_channels % 64:
            raise NotImplementedError("GroupNorm(32) kernels need channel counts that are multiples of 64")
        self.norm1, self.conv1 = Normalize(in_channels), _conv(in_channels, out_channels)
        self.norm2, self.conv2 = Normalize(out_channels), _conv(out_channels, out_channels)
        self.dropout = torch.nn.Dropout(dropout)
        if in_channels != out_channels:                              # the checkpoint names the shortcut by its kind
            name, k = ("conv_shortcut", 3) if conv_shortcut else ("nin_shortcut", 1)
            setattr(self, name, _conv(in_channels, out_channels, k))
        self._init_packable()

    def _pack(self):
        pk = dict(n1=_gn_params(self.norm1), c1=_conv3x3_params(self.conv1), n2=_gn_params(self.norm2),
                  c2=_conv3x3_params(self.conv2))
        if self.in_channels != self.out_channels:
            pk["sc"] = _conv3x3_params(self.conv_shortcut) if self.use_conv_shortcut else _conv1x1_params(self.nin_shortcut)
        return pk

    def _run(self, rt: Runtime, x: Act) -> Act:
        pk = self.packed()
        Ci, Co = self.in_channels, self.out_channels
        h16, _ = E.gn_spatial(rt, x.f32, x.F, x.N, Ci, *pk["n1"], 1e-6, True)
        h = _conv3x3(rt, h16, x.F, x.H, x.W, Ci, *pk["c1"], Co)
        h16, _ = E.gn_spatial(rt, h.f32, x.F, x.N, Co, *pk["n2"], 1e-6, True)
        skip = x.f32
        if Ci != Co:
            w, b = pk["sc"]
            if self.use_conv_shortcut:
                skip = _conv3x3(rt, x.need_f16(rt), x.F, x.H, x.W, Ci, w, b, Co).f32
            else:
                skip = rt.empty((x.M, Co), torch.float32)
                rt.be.gemm(x.need_f16(rt), w, M=x.M, N=Co, K=Ci, lda=Ci, bias=b, out32=skip, ldc32=Co)
        return _conv3x3(rt, h16, x.F, x.H, x.W, Co, *pk["c2"], Co, res32=skip)      # x (or shortcut(x)) + h

    def forward(self, x, temb=None):
        assert temb is None
        rt = Runtime(x.device, x.shape[0], 1)
        return self._run(rt, act_from_nchw(rt, x)).to_nchw().to(x.dtype)


class AttnBlock(nn.Module, Packable):
    """model.py:374-415 (== MemoryEfficientAttnBlock :417-477): single-head self-attention over the h*w tokens of a frame."""

    def __init__(self, in_channels):
        super().__init__()
        if in_channels % 64:
            raise NotImplementedError("AttnBlock needs a channel count that is a multiple of 64")
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self._init_packable()

    def _pack(self):
        return dict(n=_gn_params(self.norm), q=_conv1x1_params(self.q), k=_conv1x1_params(self.k),
                    v=_conv1x1_params(self.v), o=_conv1x1_params(self.proj_out))

    def _run(self, rt: Runtime, x: Act) -> Act:
        pk = self.packed()
        C, N, M, F = self.in_channels, x.N, x.M, x.F
        if N > 16384 or N % 8:
            raise NotImplementedError(f"AttnBlock over {N} tokens per frame (supported: multiples of 8 up to 16384)")
        h16, _ = E.gn_spatial(rt, x.f32, F, N, C, *pk["n"], 1e-6, False)
        q16, k16 = rt.empty((M, C), torch.float16), rt.empty((M, C), torch.float16)
        vt16 = rt.empty((F, C, N), torch.float16)                       # channel-major V^T per frame
        rt.be.gemm(h16, pk["q"][0], M=M, N=C, K=C, lda=C, bias=pk["q"][1], out16=q16, ldc16=C)
        rt.be.gemm(h16, pk["k"][0], M=M, N=C, K=C, lda=C, bias=pk["k"][1], out16=k16, ldc16=C)
        rt.be.gemm(h16, pk["v"][0], M=M, N=C, K=C, lda=C, bias=pk["v"][1], out16t=vt16, ldt=N, t_rows=N,
                   t_gstride=C * N, n_split=0)
        o16 = rt.empty((M, C), torch.float16)
        s32 = rt.empty((N, N), torch.float32)                           # one frame's scores at a time (reused)
        p16 = rt.empty((N, N), torch.float16)
        for f in range(F):
            rt.be.gemm(q16[f * N:], k16[f * N:(f + 1) * N], M=N, N=N, K=C, lda=C, out32=s32, ldc32=N)
            rt.be.softmax_rows(s32, N, N, N, float(C) ** -0.5, p16, N)
            rt.be.gemm(p16, vt16[f], M=N, N=C, K=N, lda=N, out16=o16[f * N:], ldc16=C)
        out = rt.empty((M, C), torch.float32)
        rt.be.gemm(o16, pk["o"][0], M=M, N=C, K=C, lda=C, bias=pk["o"][1], res1=x.f32, ldr1=C, out32=out, ldc32=C)
        return Act(F, x.H, x.W, C, f32=out)

    def forward(self, x, **kwargs):
        rt = Runtime(x.device, x.shape[0], 1)
        return self._run(rt, act_from_nchw(rt, x)).to_nchw().to(x.dtype)


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None, temporal=False):
    """model.py:551-585: "vanilla" and "vanilla-xformers" are the same single-head attention."""
    if attn_type in ("vanilla", "vanilla-xformers") and not temporal:
        assert attn_kwargs is None
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError(f"attn_type {attn_type!r} is not on the first-stage path")


def _conv(cin: int, cout: int, k: int = 3) -> nn.Conv2d:
    return torch.nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2)


def _level_widths(ch: int, ch_mult) -> List[int]:
    return [ch * m for m in ch_mult]


def _stage(widths_in_out, n_blocks: int, with_attn: bool, attn_type: str, dropout: float, resample=None) -> nn.Module:
    """One resolution level of the first-stage networks as the reference names it (`block`, `attn`, optional `upsample` /
    `downsample`; model.py:800-820, 920-945): n_blocks ResnetBlocks from widths_in_out[0] to widths_in_out[1] channels, each
    followed by an AttnBlock where the level's resolution is listed in attn_resolutions.  The state-dict names of a checkpoint
    (`up.2.block.0.conv1.weight`, `down.1.downsample.conv.bias`, ...) are these attribute names."""
    cin, cout = widths_in_out
    level = nn.Module()
    level.block = nn.ModuleList(ResnetBlock(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=0, dropout=dropout)
                                for i in range(n_blocks))
    level.attn = nn.ModuleList(make_attn(cout, attn_type=attn_type) for _ in range(n_blocks if with_attn else 0))
    if resample is not None:
        name, module = resample
        setattr(level, name, module)
    return level


def _mid(width: int, attn_type: str, dropout: float) -> nn.Module:
    mid = nn.Module()
    mid.block_1 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
    mid.attn_1 = make_attn(width, attn_type=attn_type)
    mid.block_2 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
    return mid


class Decoder(nn.Module, Packable):
    """model.py:882-1026.  `forward(z)`: z (F, z_channels, h, w) -> image (F, out_ch, 8h, 8w) for ch_mult of length 4."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn:
            raise NotImplementedError("linear attention is not on the first-stage path")
        widths = _level_widths(ch, ch_mult)
        L = len(widths)
        self.ch, self.temb_ch, self.num_resolutions, self.num_res_blocks = ch, 0, L, num_res_blocks
        self.resolution, self.in_channels, self.z_channels, self.out_ch = resolution, in_channels, z_channels, out_ch
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        # coarse -> fine: the resolution of the k-th level built — the coarsest is floored ONCE and then doubled, as model.py:917-918,
        # 960 do it (curr_res = resolution // 2^(L-1); curr_res *= 2 per level): for a resolution that is no multiple of 2^(L-1) a
        # per-level floor gives another list (100: 12, 25, 50, 100 instead of 12, 24, 48, 96) and with it other AttnBlock placements
        res = [(resolution // 2 ** (L - 1)) * 2 ** k for k in range(L)]
        self.z_shape = (1, z_channels, res[0], res[0])
        self.conv_in = _conv(z_channels, widths[-1])
        self.mid = _mid(widths[-1], attn_type, dropout)
        # levels are built coarse -> fine (that is the order the widths chain in) and stored fine -> coarse like the reference's
        # `self.up.insert(0, up)`: up[i] works at resolution / 2^i and upsamples unless it is the finest
        levels, cin = [None] * L, widths[-1]
        for k, i_level in enumerate(reversed(range(L))):
            up = (("upsample", Upsample(widths[i_level], resamp_with_conv)) if i_level != 0 else None)
            levels[i_level] = _stage((cin, widths[i_level]), num_res_blocks + 1, res[k] in attn_resolutions, attn_type, dropout, up)
            cin = widths[i_level]
        self.up = nn.ModuleList(levels)
        self.norm_out = Normalize(widths[0])
        self.conv_out = _conv(widths[0], out_ch)
        self._init_packable()

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    def _pack(self):
        zc = (self.z_channels + 7) // 8 * 8
        return dict(cin=(E.pk_conv3x3(self.conv_in.weight, zc), E.pk_f32(self.conv_in.bias)), zc=zc,
                    no=_gn_params(self.norm_out), cout=_conv3x3_params(self.conv_out))

    def _run(self, rt: Runtime, z16: torch.Tensor, F: int, H: int, W: int) -> Act:
        """z16: [F*H*W, zc] fp16 tokens (channels zero-padded to zc)."""
        pk = self.packed()
        h = _conv3x3(rt, z16, F, H, W, pk["zc"], *pk["cin"], self.conv_in.out_channels)
        h = self.mid.block_1._run(rt, h)
        if isinstance(self.mid.attn_1, AttnBlock):
            h = self.mid.attn_1._run(rt, h)
        h = self.mid.block_2._run(rt, h)
        for i_level in reversed(range(self.num_resolutions)):
            up = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = up.block[i_block]._run(rt, h)
                if len(up.attn) > 0:
                    h = up.attn[i_block]._run(rt, h)
            if i_level != 0:
                h = up.upsample._run(rt, h)
        if self.give_pre_end:
            return h
        h16, _ = E.gn_spatial(rt, h.f32, h.F, h.N, h.C, *pk["no"], 1e-6, True)
        w, b = pk["cout"]
        return _conv3x3(rt, h16, h.F, h.H, h.W, h.C, w, b, self.out_ch)

    def forward(self, z, **kwargs):
        with torch.no_grad():
            F, C, H, W = z.shape
            rt = Runtime(z.device, F, 1)
            zc = self.packed()["zc"]
            z16 = rt.empty((F * H * W, zc), torch.float16)
            rt.be.nchw_to_tokens_f16(z.detach().to(torch.float32).contiguous(), C, None, 0, F, H * W, zc, z16)
            out = self._run(rt, z16, F, H, W).to_nchw()
            if self.tanh_out:
                out = torch.tanh(out)
        return out.to(z.dtype)


class Encoder(nn.Module, Packable):
    """model.py:763-880.  `forward(x)`: image (F, in_channels, H, W) -> moments (F, 2*z_channels, H/8, W/8)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn:
            raise NotImplementedError("linear attention is not on the first-stage path")
        widths = _level_widths(ch, ch_mult)
        L = len(widths)
        self.ch, self.temb_ch, self.num_resolutions, self.num_res_blocks = ch, 0, L, num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.in_ch_mult = (1,) + tuple(ch_mult)
        self.conv_in = _conv(in_channels, ch)
        chain = [ch] + widths                                        # level i: chain[i] -> chain[i + 1] channels at resolution / 2^i
        self.down = nn.ModuleList(
            _stage((chain[i], chain[i + 1]), num_res_blocks, (resolution // 2 ** i) in attn_resolutions, attn_type, dropout,
                   ("downsample", Downsample(chain[i + 1], resamp_with_conv)) if i != L - 1 else None)
            for i in range(L))
        self.mid = _mid(widths[-1], attn_type, dropout)
        self.norm_out = Normalize(widths[-1])
        self.out_channels = 2 * z_channels if double_z else z_channels
        self.conv_out = _conv(widths[-1], self.out_channels)
        self._init_packable()

    def _pack(self):
        ic = (self.in_channels + 7) // 8 * 8
        return dict(cin=(E.pk_conv3x3(self.conv_in.weight, ic), E.pk_f32(self.conv_in.bias)), ic=ic,
                    no=_gn_params(self.norm_out), cout=_conv3x3_params(self.conv_out))

    def _run(self, rt: Runtime, x16: torch.Tensor, F: int, H: int, W: int) -> Act:
        """x16: [F*H*W, ic] fp16 tokens (channels zero-padded to ic).  Returns the moments as an Act (fp32)."""
        pk = self.packed()
        h = _conv3x3(rt, x16, F, H, W, pk["ic"], *pk["cin"], self.ch)
        for i_level in range(self.num_resolutions):
            down = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = down.block[i_block]._run(rt, h)
                if len(down.attn) > 0:
                    h = down.attn[i_block]._run(rt, h)
            if i_level != self.num_resolutions - 1:
                h = down.downsample._run(rt, h)
        h = self.mid.block_1._run(rt, h)
        if isinstance(self.mid.attn_1, AttnBlock):
            h = self.mid.attn_1._run(rt, h)
        h = self.mid.block_2._run(rt, h)
        h16, _ = E.gn_spatial(rt, h.f32, h.F, h.N, h.C, *pk["no"], 1e-6, True)
        w, b = pk["cout"]
        return _conv3x3(rt, h16, h.F, h.H, h.W, h.C, w, b, self.out_channels)

    def forward(self, x):
        with torch.no_grad():
            if x.dim() == 5:                                   # (b t c h w) video input (model.py:855-856)
                x = x.reshape(-1, *x.shape[2:])
            F, C, H, W = x.shape
            rt = Runtime(x.device, F, 1)
            ic = self.packed()["ic"]
            x16 = rt.empty((F * H * W, ic), torch.float16)
            rt.be.nchw_to_tokens_f16(x.detach().to(torch.float32).contiguous(), C, None, 0, F, H * W, ic, x16)
            out = self._run(rt, x16, F, H, W).to_nchw()
        return out.to(x.dtype)


class FirstStageEncoder(nn.Module, Packable):
    """The encode half of `AutoencoderKL` (autoencoder.py:333-362): moments = quant_conv(encoder(x)); the posterior is
    DiagonalGaussianDistribution(moments): mean, logvar clamped to [-30, 20], sample = mean + exp(logvar / 2) * eps
    (`AutoencoderKLInferenceWrapper.encode` returns `.sample()`, autoencoder.py:371-373)."""

    def __init__(self, embed_dim: int, ddconfig: dict):
        super().__init__()
        assert ddconfig["double_z"]
        self.encoder = Encoder(**ddconfig)
        self.quant_conv = torch.nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.embed_dim = embed_dim
        self._init_packable()

    def _pack(self):
        return dict(q=_conv1x1_params(self.quant_conv))

    def moments(self, x: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            if x.dim() == 5:
                x = x.reshape(-1, *x.shape[2:])
            enc = self.encoder
            F, C, H, W = x.shape
            rt = Runtime(x.device, F, 1)
            ic = enc.packed()["ic"]
            x16 = rt.empty((F * H * W, ic), torch.float16)
            rt.be.nchw_to_tokens_f16(x.detach().to(torch.float32).contiguous(), C, None, 0, F, H * W, ic, x16)
            h = enc._run(rt, x16, F, H, W)
            w, b = self.packed()["q"]
            n = self.quant_conv.out_channels
            m32 = rt.empty((h.M, n), torch.float32)
            rt.be.gemm(h.need_f16(rt), w, M=h.M, N=n, K=h.C, lda=h.C, bias=b, out32=m32, ldc32=n)
            return Act(h.F, h.H, h.W, n, f32=m32).to_nchw().to(x.dtype)

    def encode(self, x: torch.Tensor, generator=None, sample: bool = True) -> torch.Tensor:
        mom = self.moments(x)
        mean, logvar = torch.chunk(mom, 2, dim=1)
        if not sample:
            return mean
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        eps = torch.randn(mean.shape, generator=generator, device=mean.device, dtype=mean.dtype)
        return mean + std * eps

    forward = encode


class FirstStageDecoder(nn.Module, Packable):
    """The decode half of `AutoencoderKL` (autoencoder.py:333-368): `decoder(post_quant_conv(z))`.  Attribute names
    match the reference's, so `load_state_dict(first_stage_sd, strict=False)` fills it (encoder / quant_conv /
    loss keys are reported as unexpected, like any partial load)."""

    def __init__(self, embed_dim: int, ddconfig: dict):
        super().__init__()
        self.decoder = Decoder(**ddconfig)
        self.post_quant_conv = torch.nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self._init_packable()

    def _pack(self):
        # 1x1 conv embed_dim -> z_channels, both zero-padded to multiples of 8 so that its fp16 output is directly the
        # padded token matrix conv_in gathers from
        ec, zc = (self.embed_dim + 7) // 8 * 8, (self.decoder.z_channels + 7) // 8 * 8
        w = torch.zeros((zc, ec), dtype=torch.float16, device=self.post_quant_conv.weight.device)
        w[: self.decoder.z_channels, : self.embed_dim] = self.post_quant_conv.weight.detach().reshape(
            self.decoder.z_channels, self.embed_dim).to(torch.float16)
        b = torch.zeros(zc, dtype=torch.float32, device=w.device)
        b[: self.decoder.z_channels] = self.post_quant_conv.bias.detach().float()
        return dict(w=w.contiguous(), b=b, ec=ec, zc=zc)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            pk = self.packed()
            F, C, H, W = z.shape
            rt = Runtime(z.device, F, 1)
            M = F * H * W
            e16 = rt.empty((M, pk["ec"]), torch.float16)
            rt.be.nchw_to_tokens_f16(z.detach().to(torch.float32).contiguous(), C, None, 0, F, H * W, pk["ec"], e16)
            z16 = rt.empty((M, pk["zc"]), torch.float16)
            rt.be.gemm(e16, pk["w"], M=M, N=pk["zc"], K=pk["ec"], lda=pk["ec"], bias=pk["b"], out16=z16, ldc16=pk["zc"])
            out = self.decoder._run(rt, z16, F, H, W).to_nchw()
        return out.to(z.dtype)

    forward = decode
