"""Error budget of the fp16 operand roundings (CPU, torch emulation of the C-ABI — tests/emu.py).

    python tools/exp/error_budget.py [tiny|full16] [--only] [class=mode,class=mode ...]

Every activation rounding of the path goes through `emu.r16(value, entry_point)`.  This tool classifies each
rounding by (entry point, calling host function) into operand classes and re-runs the network with a chosen rounding
per class:
    h   = fp16 (what the kernels do today)
    s   = split fp16 pair hi + lo*2^-11 (22-bit operand, the "precise" mode of DESIGN §6)
    x   = exact (fp32 kept)
    m   = split pair whose lo plane is MX block-scaled fp8 (e4m3, one 2^k scale per 32 channels); m2 = one mantissa bit less;
          m4 = MX fp4 (e2m1) lo plane;  M / M4 = m / m4 with the WEIGHTS of the lo pass in the same MX format (the real thing)
Activations are kept in fp32 buffers so that a class set to s/x really carries the extra bits to its consumer.
Prints max-abs / mean-abs of eps against the oracle (tiny: reference golden) per experiment.
"""
import inspect
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import emu  # noqa: E402
from helpers import cond, err_stats, golden, oracle_cfg, product_network, step_inputs  # noqa: E402
from oracle import panacea_oracle as po  # noqa: E402
from panacea_amd import configs, engine as E  # noqa: E402

CLASSES = ["ln", "gn_res", "gn_stt", "gn_head", "gnt", "qkv_views", "qkv_temporal", "q_text", "kv_text", "ff_hidden", "ff_out", "attn_o",
           "stream", "conv_mid", "stem", "ctx"]


def classify(site: str) -> str:
    """operand class of the rounding that `site` performs, from the host function that called the backend"""
    fn = None
    for fr in inspect.stack()[2:12]:
        if "/panacea_amd/" in fr.filename:
            fn = fr.function
            break
    if site == "layernorm":
        return "ln"
    if site == "groupnorm_apply":
        for fr in inspect.stack()[2:12]:
            if "/panacea_amd/" in fr.filename and fr.function in ("_branch", "_head"):
                return "gn_stt" if fr.function == "_branch" else "gn_head"
        return "gn_res"
    if site == "groupnorm_temporal":
        return "gnt"
    if site in ("attn_views", "attn_temporal"):
        return "attn_o"
    if site in ("concat_add", "add_f32"):
        return "stream"
    if site == "nchw_to_tokens":
        return "stem"
    if site == "gemm.geglu":
        return "ff_hidden"
    if site in ("gemm.out16", "gemm.out16t"):
        if fn == "_run_views":
            return "qkv_views"
        if fn == "_run_temporal":
            return "qkv_temporal"
        if fn == "_run_text":
            return "q_text"
        if fn in ("_text_kv", "run"):
            return "kv_text"
        if fn == "_run" and site == "gemm.out16":
            # FeedForward._run (last block: fp16 copy for proj_out) or ResBlock3D / TimestepEmbedSequential stream copies
            for fr in inspect.stack()[2:12]:
                if "/panacea_amd/" in fr.filename and fr.function == "_run":
                    if "attention.py" in fr.filename:
                        return "ff_out"
                    return "stream"
        if fn == "_branch":
            return "stream"
        if fn == "run_conv3x3":
            # hint stem intermediate layers (fp16 between layers) or stream copies next to an fp32 output
            return "conv_mid"
        return "stream"
    raise KeyError(site)


def rounder(mode: str):
    if mode == "h":
        return lambda v: v.half().float()
    if mode == "s":
        def split(v):
            hi = v.half().float()
            lo = ((v - hi) * 2048.0).half().float() / 2048.0
            return hi + lo
        return split
    if mode in ("m", "m2", "m4", "M", "M4"):
        # split pair whose lo plane is an MX block-scaled fp8 (e4m3, one power-of-two scale per 32 channels) — the operand of a
        # lo pass on the scaled 32x32x64 f8f6f4 MFMA (2x the fp16 rate, half the operand bytes).  "m2" drops one more mantissa
        # bit of the lo plane: a stand-in for the fp8 rounding of the WEIGHTS in that pass (same relative size, independent).
        def split_mx(v):
            hi = v.half().float()
            lo = (v - hi) * 2048.0
            shp = lo.shape
            C = shp[-1]
            pad = (-C) % 32
            b = torch.nn.functional.pad(lo.reshape(-1, C), (0, pad)).reshape(-1, (C + pad) // 32, 32)
            amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
            if mode in ("m4", "M4"):                                    # MX fp4 (e2m1: 0, .5, 1, 1.5, 2, 3, 4, 6), block max in [4, 8) -> clamp 6
                e = torch.floor(torch.log2(amax)) - 2.0
                x = (b / torch.exp2(e)).clamp(-6.0, 6.0)
                grid = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
                idx = (x.abs().unsqueeze(-1) - grid).abs().argmin(dim=-1)
                q = grid[idx] * torch.sign(x)
            else:
                e = torch.floor(torch.log2(amax)) - 7.0                 # block max lands in [128, 256) < 448 = e4m3 max
                q = (b / torch.exp2(e)).to(torch.float8_e4m3fn).float()
            if mode == "m2":                                            # keep 2 of e4m3's 3 mantissa bits
                ex = torch.floor(torch.log2(q.abs().clamp_min(1e-30)))
                q = torch.round(q / torch.exp2(ex - 2.0)) * torch.exp2(ex - 2.0)
            lo_q = (q * torch.exp2(e)).reshape(-1, C + pad)[:, :C].reshape(shp)
            return hi + lo_q / 2048.0
        return split_mx
    return lambda v: v


def _mx_weights(Wm: torch.Tensor, fp4: bool) -> torch.Tensor:
    """Wm [N, K] as the lo pass of modes M / M4 sees it: MX block-scaled fp8 / fp4 along K (blocks of 32)"""
    N, K = Wm.shape
    pad = (-K) % 32
    b = torch.nn.functional.pad(Wm, (0, pad)).reshape(N, (K + pad) // 32, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    if fp4:
        e = torch.floor(torch.log2(amax)) - 2.0
        x = (b / torch.exp2(e)).clamp(-6.0, 6.0)
        grid = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
        q = grid[(x.abs().unsqueeze(-1) - grid).abs().argmin(dim=-1)] * torch.sign(x)
    else:
        e = torch.floor(torch.log2(amax)) - 7.0
        q = (b / torch.exp2(e)).to(torch.float8_e4m3fn).float()
    return (q * torch.exp2(e)).reshape(N, K + pad)[:, :K]


class Experiment:
    def __init__(self, modes: dict):
        self.modes = {c: modes.get(c, modes.get("*", "h")) for c in CLASSES}
        self.count = {c: 0 for c in CLASSES}

    def r16(self, v, site):
        c = classify(site)
        self.count[c] += v.numel()
        return rounder(self.modes[c])(v.float())

    def __enter__(self):
        self._r16, self._strict = emu.r16, emu.STRICT_DTYPES
        emu.r16, emu.STRICT_DTYPES = self.r16, False
        self._empty, self._zeros, self._ctx = E.Runtime.empty, E.Runtime.zeros, E.Runtime.set_context
        f32 = lambda d: torch.float32 if d == torch.float16 else d   # noqa: E731
        E.Runtime.empty = lambda rt, shape, dtype, tail_rows=0: torch.empty(shape, device=rt.device, dtype=f32(dtype))
        E.Runtime.zeros = lambda rt, shape, dtype: torch.zeros(shape, device=rt.device, dtype=f32(dtype))
        exp = self

        def set_context(rt, context):
            B, n, D = context.shape
            rt.n_text = n
            c = torch.zeros((B, E.TEXT_PAD, D), dtype=torch.float32)
            exp.count["ctx"] += context.numel()
            c[:, :n] = rounder(exp.modes["ctx"])(context.float())
            rt.ctx16 = c.view(B * E.TEXT_PAD, D)
        E.Runtime.set_context = set_context
        # modes M / M4: the WEIGHTS of the lo pass are MX fp8 / fp4 too.  The A operand arrives as hi + lo_q (fp32 buffer); its lo
        # part is what fp16 rounding leaves, and the lo pass contributes lo_q . (W_mx - W) on top of the exact product.
        big = {m for m in self.modes.values() if m in ("M", "M4")}
        self._hook = emu.ACC_HOOK
        if big:
            fp4 = "M4" in big
            cache = {}

            def hook(acc, a16, Wm, kw):
                a = a16.float()
                lo = a - a.half().float()
                if not bool(lo.abs().max() > 0):
                    return acc
                key = (Wm.data_ptr(), tuple(Wm.shape))
                if key not in cache:
                    cache[key] = _mx_weights(Wm, fp4) - Wm
                return acc + emu._contract(lo, cache[key], kw["M"], kw["N"], kw["K"], kw["lda"], kw["a_mode"], kw["conv"], kw["tconv"])
            emu.ACC_HOOK = hook
        return self

    def __exit__(self, *a):
        emu.r16, emu.STRICT_DTYPES = self._r16, self._strict
        emu.ACC_HOOK = self._hook
        E.Runtime.empty, E.Runtime.zeros, E.Runtime.set_context = self._empty, self._zeros, self._ctx


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if which == "tiny":
        w, sd, kw = product_network("tiny")
        inp = step_inputs("tiny", kw)
        ref = torch.from_numpy(golden("tiny")["eps"])
    else:
        kw = configs.with_frames(configs.get("full"), 2)
        w, sd, _ = product_network("full", "cpu", kw=kw)
        tail = 0.0
        if "--tail" in sys.argv:           # heavy-tail weight set (round 5; synth.synth_tensor): which classes carry the error THERE
            i = sys.argv.index("--tail")
            tail = float(sys.argv[i + 1])
            del sys.argv[i:i + 2]
            from helpers import manifest
            from panacea_amd import synth
            sd = synth.synth_state_dict(manifest("full"), tail=tail)
            w.diffusion_model.load_state_dict(sd, strict=True)
        inp = step_inputs("full", kw, "cpu", shape=(1, 2, 16, 192))
        ref = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))

    w.diffusion_model.precision = "fast"      # the hook decides every rounding: the product's own split planes stay off

    def run(modes, label):
        with Experiment(modes) as ex, E.use_backend(emu):
            eps = w(inp["x"], inp["t"], cond(inp))
        st = err_stats(eps, ref)
        print(f"{label:60s} max {st['max_abs']:.3e}  mean {st['mean_abs']:.3e}", flush=True)
        return st, ex

    only = "--only" in sys.argv          # just the mixes given on the command line (+ the all-fp16 line)
    if only:
        sys.argv.remove("--only")
    st, ex = run({"*": "h"}, "all fp16 (today)")
    print("elements rounded per class:", {k: v for k, v in ex.count.items() if v})
    if not only:
        run({"*": "x"}, "no activation rounding")
        run({"*": "s"}, "all split")
    base = st["mean_abs"] ** 2
    for c in CLASSES:
        if ex.count[c] and not only:
            s1, _ = run({"*": "h", c: "x"}, f"fp16 except {c} exact")
            print(f"    -> {c}: {100 * (1 - s1['mean_abs'] ** 2 / base):5.1f} % of the error variance")
    for extra in sys.argv[2:]:
        modes = {"*": "h"}
        for kv in extra.split(","):
            k, v = kv.split("=")
            modes[k] = v
        run(modes, extra)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
