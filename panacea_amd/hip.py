"""ctypes binding of libpanacea_hip.so (the C-ABI declared in include/panacea_hip.h).

This is the ONLY compute backend of the package: there is no CPU or eager fallback.  Importing the
module is cheap; the shared object is loaded on first use and a missing / stale library raises
`HipLibraryError` (run `python -m panacea_amd.build`).  Every wrapper takes torch tensors that already
live in HBM, enqueues on the CURRENT torch stream and returns immediately (hipGraph-capturable).
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path
from typing import Optional

import torch

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libpanacea_hip.so"
HEADER = PKG.parent / "include" / "panacea_hip.h"

A_PLAIN, A_CONV3X3, A_CONV1D_T = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
ABI_VERSION = 7          # PNC_ABI_VERSION of include/panacea_hip.h this binding was written against
LO_F16, LO_E4M3 = 0, 1   # PNC_LO_*: storage format of the lo plane of a precise operand
# dtype of a lo-plane tensor <-> format: an fp16 tensor holds fp16(r), a uint8 tensor OCP e4m3 bytes (one per element)
LO_DTYPE = {LO_F16: torch.float16, LO_E4M3: torch.uint8}


def lo_fmt(t: Optional[torch.Tensor]) -> int:
    """PNC_LO_* of a lo-plane tensor, from its dtype (None -> PNC_LO_F16, the argument is ignored by the kernels then)"""
    if t is None or t.dtype == torch.float16:
        return LO_F16
    if t.dtype == torch.uint8:
        return LO_E4M3
    raise PncError(f"a lo plane is fp16 (PNC_LO_F16) or uint8 e4m3 bytes (PNC_LO_E4M3), got {t.dtype}")


class HipLibraryError(RuntimeError):
    pass


class PncError(RuntimeError):
    pass


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("a_mode", C.c_int32),
        ("Cin", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32),
        ("Wout", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32),
        ("T", C.c_int32), ("Npix", C.c_int32),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("rb_rows", C.c_int32), ("rb_mod", C.c_int32),
        ("res1", C.c_void_p), ("ldr1", C.c_int32),
        ("res2", C.c_void_p), ("ldr2", C.c_int32),
        ("out32", C.c_void_p), ("ldc32", C.c_int32),
        ("out16", C.c_void_p), ("ldc16", C.c_int32),
        ("out16t", C.c_void_p), ("ldt", C.c_int32), ("t_rows", C.c_int32), ("t_gstride", C.c_int64),
        ("n_split", C.c_int32), ("act", C.c_int32), ("geglu", C.c_int32),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("conv_pad_br", C.c_int32), ("struct_bytes", C.c_int32),
        ("A_lo", C.c_void_p), ("out16_lo", C.c_void_p),
        ("ldw", C.c_int32), ("ln_eps", C.c_float),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_out16", C.c_void_p),
        ("ldln", C.c_int32), ("a_lo_fmt", C.c_int32),
        ("out_lo_fmt", C.c_int32), ("ldw_lo", C.c_int32),
        ("W_lo", C.c_void_p), ("w_lo_exp", C.c_int32), ("t_halo", C.c_int32),
        ("x_halo_off", C.c_int64), ("gn_part", C.c_void_p),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", C.c_int32),
        ("k", C.c_void_p), ("ldk", C.c_int32),
        ("vt", C.c_void_p), ("ldvt", C.c_int32), ("vt_gstride", C.c_int64),
        ("o", C.c_void_p), ("ldo", C.c_int32),
        ("groups", C.c_int32), ("heads", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("views", C.c_int32),
        ("kvH", C.c_int32), ("kvW", C.c_int32), ("kv_views", C.c_int32),
        ("kv_rows_per_group", C.c_int32), ("q_per_kv", C.c_int32), ("kv_valid", C.c_int32),
        ("nseg", C.c_int32 * 8), ("seg", (C.c_int32 * 2) * 8),
        ("scale", C.c_float), ("causal", C.c_int32),
        ("k_halo", C.c_void_p * 2), ("vt_halo", C.c_void_p * 2),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGNATURES = {
    "pnc_version": (C.c_char_p, []),
    "pnc_abi_version": (_I, []),
    "pnc_build_digest": (C.c_char_p, []),
    "pnc_set_option": (_I, [_I, _I]),
    "pnc_gemm_f16": (_I, [C.POINTER(GemmParams), _P]),
    "pnc_gemm_workspace_floats": (_L, [C.POINTER(GemmParams)]),
    "pnc_gemm_fuses_layernorm": (_I, [C.POINTER(GemmParams)]),
    "pnc_attn_views_f16": (_I, [C.POINTER(AttnParams), _P]),
    "pnc_softmax_rows_f16": (_I, [_P, _L, _I, _I, _F, _I, _I, _P, _L, _P]),
    "pnc_attn_temporal_f16": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "pnc_groupnorm_stats": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "pnc_groupnorm_apply": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _F, _I, _P, _I, _P, _I, _I, _P]),
    "pnc_groupnorm_combine": (_I, [_P, _I, _I, _I, _P, _P]),
    "pnc_groupnorm_temporal_part": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _P, _I, _I, _P, _P, _I, _I, _P]),
    "pnc_groupnorm_temporal_silu": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _I, _P]),
    "pnc_layernorm": (_I, [_P, _I, _I, _I, _P, _P, _F, _P, _I, _P, _P]),
    "pnc_linear_smallm": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pnc_linear_smallm_segments": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "pnc_timestep_embedding": (_I, [_P, _I, _I, _P, _P, _P]),
    "pnc_nchw_to_tokens_f16": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P]),
    "pnc_cfg_euler_step": (_I, [_P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "pnc_tokens_to_nchw_f32": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "pnc_concat_add": (_I, [_P, _I, _P, _P, _I, _L, _P, _P, _P, _I, _P]),
    "pnc_concat_add_stats": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "pnc_add_f32": (_I, [_P, _P, _L, _P, _P, _P, _I, _P]),
    "pnc_cast_f16": (_I, [_P, _L, _P, _P, _I, _P]),
    "pnc_range_monitor_collect": (_I, [_P, _P]),
}

_lib = None


def header_symbols() -> list[str]:
    """Every function name declared in include/panacea_hip.h."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pnc_[a-z0-9_]+)\s*\(", txt)))


def load():
    """Load the shared library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    global LIB_PATH
    if os.environ.get("PANACEA_HIP_LIB"):          # A/B runs of two builds of the SAME ABI (tools/exp); never a fallback
        LIB_PATH = Path(os.environ["PANACEA_HIP_LIB"])
    if not LIB_PATH.exists():
        raise HipLibraryError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of panacea_amd. "
            "Build it with `python -m panacea_amd.build` (hipcc --offload-arch=gfx950).")
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    if not os.environ.get("PANACEA_HIP_LIB"):
        # a library built from other sources than the ones next to it is an ABI hazard (argument lists change): refuse it, do
        # not guess.  The digest is compiled INTO the library (pnc_build_digest), so no side file can vouch for a stale build.
        from . import build as _build
        try:
            lib.pnc_build_digest.restype = C.c_char_p
            have = lib.pnc_build_digest().decode()
        except AttributeError:
            have = "(a library older than pnc_build_digest)"
        if _build.CSRC.exists() and have != _build._digest():
            raise HipLibraryError(f"{LIB_PATH} was built from other sources ({have[:12]}) than panacea_amd/csrc holds now "
                                  f"({_build._digest()[:12]}); rebuild it with `python -m panacea_amd.build`")
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    if lib.pnc_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"{LIB_PATH} implements C-ABI version {lib.pnc_abi_version()}, this binding needs "
                              f"{ABI_VERSION}; rebuild the extension (`python -m panacea_amd.build --force`)")
    _lib = lib
    return lib


OPT_GEMM_TAIL_SPLIT, OPT_GEMM_TILE, OPT_ATTN_VARIANT, OPT_ATTN_DMA, OPT_GEMM_FUSE_LN, OPT_GEMM_GROUP_M, OPT_STENCIL_TILES = 0, 1, 2, 3, 4, 5, 6
OPT_GEMM_PERSIST = 7
OPT_ATTN_DEFER_MAX = 8
OPT_GEMM_GN_STATS = 9
OPT_GEMM_STAGGER = 10
OPT_ATTN_SUM_TRIGGER = 11


def build_digest() -> str:
    """pnc_build_digest() of the loaded library (what bench.py / tools/pmc_traffic.py stamp their records with)"""
    return load().pnc_build_digest().decode()


def set_option(option: int, value: int) -> int:
    """pnc_set_option: process-global tuning / test switch of the library; returns the previous value."""
    if not 0 <= option <= OPT_ATTN_SUM_TRIGGER:
        raise PncError(f"unknown library option {option}")
    return load().pnc_set_option(option, value)


CU_MASK_PATTERNS = {            # complementary halves of the 256 CUs, as 8 x 32-bit words (hipExtStreamCreateWithCUMask)
    "even-odd": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "nibbles": ([0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8),
    "halves": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
}


def masked_stream(words) -> "torch.cuda.Stream":
    """A HIP stream whose kernels run on the CUs of `words` only (hipExtStreamCreateWithCUMask of the HIP runtime this process
    already uses), wrapped for torch.  Round 6 experiment: two sample chains on complementary halves of
    the chip.  The stream lives as long as the process."""
    rt = None
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            rt = C.CDLL(line.split()[-1])
            break
    if rt is None:
        raise PncError("the HIP runtime is not loaded")
    rt.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    rt.hipExtStreamCreateWithCUMask.restype = C.c_int
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = rt.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    if rc != 0:
        raise PncError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    return torch.cuda.ExternalStream(st.value)


class Profiler:
    """Per-launch HIP-event timing by kernel family (bench.py's instrumented pass; off by default).
    Events are recorded on the current torch stream, which is the stream every kernel is launched on."""

    def __init__(self):
        self.records = []          # (family, ev0, ev1, flops, bytes)

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out = {}
        for fam, e0, e1, fl, by in self.records:
            d = out.setdefault(fam, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


_PROF: Optional[Profiler] = None


def set_profiler(p: Optional[Profiler]):
    global _PROF
    _PROF = p


def _timed(family: str, flops: float, nbytes: float, fn, *args):
    if _PROF is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _PROF.records.append((family, e0, e1, flops, nbytes))
    return rc


def _check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "PNC_EINVAL (unsupported shape/argument)", -2: "PNC_EALIGN (alignment)",
                -3: "PNC_EABI (parameter struct of another header version)"}.get(rc, f"hipError {rc}")
        raise PncError(f"{what} failed: {kind}")


def _stream() -> int:
    """the CURRENT torch stream of the CURRENT device: every wrapper launches there, and `_ptr` refuses operands that
    live on another device (a launch on device A's stream with device B's pointers reads garbage without an error)"""
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], dtype=None, name: str = "operand") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise PncError("panacea_amd kernels need tensors resident in HBM (got a CPU tensor); "
                       "there is no CPU fallback")
    if t.device.index != torch.cuda.current_device():
        raise PncError(f"{name} lives on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: "
                       "select the tensors' device (torch.cuda.device / set_device) before calling the kernels")
    if dtype is not None and t.dtype != dtype:
        raise PncError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.data_ptr()


# ----------------------------------------------------------------------------------------------
# wrappers (names mirror the C entry points)
# ----------------------------------------------------------------------------------------------
def gemm(a16: torch.Tensor, w16: torch.Tensor, *, M: int, N: int, K: int, lda: int = 0,
         a_mode: int = A_PLAIN, conv: Optional[dict] = None, tconv: Optional[dict] = None,
         bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
         rb_rows: int = 0, rb_mod: int = 0,
         res1: Optional[torch.Tensor] = None, ldr1: int = 0,
         res2: Optional[torch.Tensor] = None, ldr2: int = 0,
         out32: Optional[torch.Tensor] = None, ldc32: int = 0,
         out16: Optional[torch.Tensor] = None, ldc16: int = 0,
         out16t: Optional[torch.Tensor] = None, ldt: int = 0, t_rows: int = 0, t_gstride: int = 0,
         n_split: int = 0, act: int = ACT_NONE, geglu: bool = False,
         a16_lo: Optional[torch.Tensor] = None, out16_lo: Optional[torch.Tensor] = None, w_ld: int = 0,
         w_lo: Optional[tuple] = None,
         ln_gamma: Optional[torch.Tensor] = None, ln_beta: Optional[torch.Tensor] = None,
         ln_out16: Optional[torch.Tensor] = None, ldln: int = 0, ln_eps: float = 1e-5, ln_in_library: bool = False,
         gn_part: Optional[torch.Tensor] = None):
    """`gn_part` (temporal conv only): receives the GroupNorm(32) records of the fp32 output, ceil(Npix / 64) per frame
    (PncGemmParams.gn_part).  `a16_lo` / `out16_lo`: lo planes of precise (split) operands, see PncGemmParams.A_lo in the header; their dtype names the
    format (fp16, or uint8 = e4m3 bytes).  `w_lo` = (W_lo e4m3 bytes [N, K], w_lo_exp E8M0 byte of the tensor) — engine.pk_lo8 —
    is the weight side of an e4m3 lo pass."""
    p = GemmParams()
    p.struct_bytes = C.sizeof(GemmParams)
    f16, f32 = torch.float16, torch.float32
    p.A, p.W = _ptr(a16, f16, "a16"), _ptr(w16, f16, "w16")
    p.a_lo_fmt, p.out_lo_fmt = lo_fmt(a16_lo), lo_fmt(out16_lo)
    p.A_lo, p.out16_lo = _ptr(a16_lo, LO_DTYPE[p.a_lo_fmt], "a16_lo"), _ptr(out16_lo, LO_DTYPE[p.out_lo_fmt], "out16_lo")
    if p.a_lo_fmt == LO_E4M3:
        if w_lo is None:
            raise PncError("an e4m3 lo plane needs the e4m3 copy of the weights (w_lo = engine.pk_lo8(w16))")
        p.W_lo, p.w_lo_exp = _ptr(w_lo[0], torch.uint8, "w_lo"), int(w_lo[1])
        p.ldw_lo = w_lo[0].shape[-1]
    p.M, p.N, p.K, p.lda, p.a_mode, p.ldw = M, N, K, lda, a_mode, w_ld
    if ln_out16 is not None:       # LayerNorm of the fp32 output rows, fused into the GEMM where a workgroup owns whole rows
        p.ln_gamma, p.ln_beta, p.ln_out16 = _ptr(ln_gamma, torch.float32, "ln_gamma"), _ptr(ln_beta, torch.float32, "ln_beta"), \
            _ptr(ln_out16, torch.float16, "ln_out16")
        p.ldln, p.ln_eps = ldln, ln_eps
    if conv:
        p.Cin, p.Hin, p.Win = conv["Cin"], conv["Hin"], conv["Win"]
        p.Hout, p.Wout = conv["Hout"], conv["Wout"]
        p.stride, p.upsample = conv.get("stride", 1), int(conv.get("upsample", 0))
        p.conv_pad_br = int(conv.get("pad_br", 0))
        p.x_halo_off = int(conv.get("x_halo_off", 0))
    if tconv:
        p.Cin, p.T, p.Npix = tconv["C"], tconv["T"], tconv["Npix"]
        p.t_halo = int(tconv.get("halo", 0))
    p.gn_part = _ptr(gn_part, f32, "gn_part")
    p.bias, p.rowbias, p.rb_rows, p.rb_mod = _ptr(bias, f32, "bias"), _ptr(rowbias, f32, "rowbias"), rb_rows, rb_mod
    p.res1, p.ldr1, p.res2, p.ldr2 = _ptr(res1, f32, "res1"), ldr1, _ptr(res2, f32, "res2"), ldr2
    p.out32, p.ldc32, p.out16, p.ldc16 = _ptr(out32, f32, "out32"), ldc32, _ptr(out16, f16, "out16"), ldc16
    p.out16t, p.ldt, p.t_rows, p.t_gstride = _ptr(out16t, f16, "out16t"), ldt, t_rows, t_gstride
    p.n_split = n_split if out16t is not None else N
    p.act, p.geglu = act, int(geglu)
    lib = load()
    nws = lib.pnc_gemm_workspace_floats(C.byref(p))      # > 0: the library wants to split K (small-M shapes)
    ws = None
    if nws > 0:
        ws = torch.empty(nws, device=a16.device, dtype=torch.float32)
        p.ws, p.ws_floats = _ptr(ws), nws
    fam = ("gemm_plain", "gemm_conv3x3", "gemm_conv1d_t")[a_mode]
    trailing_ln = ln_out16 is not None and not ln_in_library and not lib.pnc_gemm_fuses_layernorm(C.byref(p))
    if trailing_ln:
        # rows span several workgroups: the library would launch its LayerNorm kernel after the GEMM.  Issue the two launches
        # from here instead (the same two kernels) so that the per-family timing of bench.py sees them separately.
        p.ln_gamma = p.ln_beta = p.ln_out16 = None
    _check(_timed(fam, 2.0 * M * N * K, 0.0, lib.pnc_gemm_f16, C.byref(p), _stream()), "pnc_gemm_f16")
    if trailing_ln:
        layernorm(out32, ldc32, M, N, ln_gamma, ln_beta, ln_eps, ln_out16, ldln)


def attn_views(q, ldq, k, ldk, vt, ldvt, vt_gstride, o, ldo, *, groups, heads, H, W, views,
               kvH, kvW, kv_views, kv_rows_per_group, q_per_kv, kv_valid, segs, scale, causal=False, k_halo=None, vt_halo=None):
    """`k_halo` / `vt_halo`: pairs (left, right) of buffers with the geometry of k / vt whose view column 0 holds a neighbour rank's
    view — kv view id -1 / kv_views in `segs` (PncAttnParams.k_halo)"""
    p = AttnParams()
    if k_halo is not None:
        for i in range(2):
            p.k_halo[i], p.vt_halo[i] = _ptr(k_halo[i]), _ptr(vt_halo[i])
    p.q, p.ldq, p.k, p.ldk = _ptr(q), ldq, _ptr(k), ldk
    p.vt, p.ldvt, p.vt_gstride, p.o, p.ldo = _ptr(vt), ldvt, vt_gstride, _ptr(o), ldo
    p.groups, p.heads, p.H, p.W, p.views = groups, heads, H, W, views
    p.kvH, p.kvW, p.kv_views = kvH, kvW, kv_views
    p.kv_rows_per_group, p.q_per_kv, p.kv_valid = kv_rows_per_group, q_per_kv, kv_valid
    for v, s in enumerate(segs):
        p.nseg[v] = len(s)
        for j, u in enumerate(s):
            p.seg[v][j] = u
    p.scale, p.causal = scale, int(causal)
    nq = H * (W // views)
    nkeys = sum(len(sv) for sv in segs) * kv_valid
    _check(_timed("attn_views", 4.0 * groups * heads * nq * nkeys * 64, 0.0, load().pnc_attn_views_f16,
                  C.byref(p), _stream()), "pnc_attn_views_f16")


def softmax_rows(s32, lds, M, N, scale, p16, ldp, causal=False, n_valid=0):
    _check(_timed("softmax_rows", 0.0, 6.0 * M * N, load().pnc_softmax_rows_f16, _ptr(s32), lds, M, N, scale, int(causal),
                  n_valid, _ptr(p16), ldp, _stream()), "pnc_softmax_rows_f16")


def attn_temporal(q, ldq, k, ldk, v, ldv, o, ldo, *, B, T, Npix, heads, scale):
    nb = 8.0 * B * T * Npix * heads * 64
    _check(_timed("attn_temporal", 0.0, nb, load().pnc_attn_temporal_f16, _ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv,
                  _ptr(o), ldo, B, T, Npix, heads, scale, _stream()), "pnc_attn_temporal_f16")


def _lo_bytes(t) -> float:
    return 0.0 if t is None else float(t.element_size())


def groupnorm_stats(x32, ldx, F, Npix, Cch, ppc, partial):
    _check(_timed("groupnorm", 0.0, 4.0 * F * Npix * Cch, load().pnc_groupnorm_stats, _ptr(x32), ldx, F, Npix, Cch,
                  ppc, _ptr(partial), _stream()), "pnc_groupnorm_stats")


def groupnorm_combine(parts_in, parts, F, nchunk, out):
    _check(load().pnc_groupnorm_combine(_ptr(parts_in), parts, F, nchunk, _ptr(out), _stream()), "pnc_groupnorm_combine")


def groupnorm_apply(x32, ldx, F, Npix, Cch, ppc, partial, gamma, beta, eps, silu, y16, ldy, y16_lo=None, n_records=0):
    nb = (6.0 + _lo_bytes(y16_lo)) * F * Npix * Cch
    _check(_timed("groupnorm", 0.0, nb, load().pnc_groupnorm_apply, _ptr(x32), ldx, F, Npix, Cch,
                  ppc, _ptr(partial), _ptr(gamma), _ptr(beta), eps, int(silu), _ptr(y16), ldy, _ptr(y16_lo), lo_fmt(y16_lo),
                  int(n_records), _stream()),
           "pnc_groupnorm_apply")


def groupnorm_temporal_silu(x32, B, T, Npix, Cch, gamma, beta, eps, y16, y16_lo=None):
    nb = (6.0 + _lo_bytes(y16_lo)) * B * T * Npix * Cch
    _check(_timed("groupnorm_temporal", 0.0, nb, load().pnc_groupnorm_temporal_silu,
                  _ptr(x32), B, T, Npix, Cch, _ptr(gamma), _ptr(beta), eps, _ptr(y16), _ptr(y16_lo), lo_fmt(y16_lo), _stream()),
           "pnc_groupnorm_temporal_silu")


def groupnorm_temporal_part(x32, B, T, Npix, Cch, gamma, beta, eps, stats, mode, T_total, y16=None, y16_lo=None, t_pad=0):
    """mode 1: this rank's {sum, sum of squares} per (pixel, group) over its T frames -> stats [B*Npix, 32, 2]; mode 2: normalise +
    SiLU with the sums of the whole frame group (T_total frames per sample), into the (T + 2 t_pad)-frame layout"""
    nb = (4.0 if mode == 1 else 6.0 + _lo_bytes(y16_lo)) * B * T * Npix * Cch
    _check(_timed("groupnorm", 0.0, nb, load().pnc_groupnorm_temporal_part, _ptr(x32), B, T, Npix, Cch, _ptr(gamma), _ptr(beta), eps,
                  _ptr(stats), mode, T_total, _ptr(y16), _ptr(y16_lo), lo_fmt(y16_lo), t_pad, _stream()), "pnc_groupnorm_temporal_part")


def layernorm(x32, ldx, M, Cch, gamma, beta, eps, y16, ldy, y16_lo=None):
    nb = (8.0 if y16_lo is not None else 6.0) * M * Cch
    _check(_timed("layernorm", 0.0, nb, load().pnc_layernorm, _ptr(x32), ldx, M, Cch, _ptr(gamma),
                  _ptr(beta), eps, _ptr(y16), ldy, _ptr(y16_lo), _stream()), "pnc_layernorm")


def linear_smallm(a32, lda, w16, bias, out32, ldo, M, N, K, silu_in=False, silu_out=False):
    _check(_timed("linear_smallm", 2.0 * M * N * K, 2.0 * N * K, load().pnc_linear_smallm, _ptr(a32), lda, _ptr(w16),
                  _ptr(bias), _ptr(out32), ldo, M, N, K, int(silu_in), int(silu_out), _stream()), "pnc_linear_smallm")


def linear_smallm_segments(a32, lda, w16, bias, out32, M, m0, Mtot, N, K, seg_start, silu_in=False, silu_out=False):
    """the same linear for len(seg_start) - 1 sites in one launch: out32 = one contiguous [Mtot, width] block per site, blocks back
    to back (include/panacea_hip.h); `seg_start` = the sites' first columns + [N], a host sequence"""
    segs = (C.c_int32 * len(seg_start))(*seg_start)
    _check(_timed("linear_smallm", 2.0 * M * N * K, 2.0 * N * K, load().pnc_linear_smallm_segments, _ptr(a32), lda, _ptr(w16),
                  _ptr(bias), _ptr(out32), M, m0, Mtot, N, K, C.cast(segs, C.c_void_p), len(seg_start) - 1, int(silu_in),
                  int(silu_out), _stream()), "pnc_linear_smallm_segments")


def timestep_embedding(t_i64, F, dim, freqs, out32):
    _check(load().pnc_timestep_embedding(_ptr(t_i64), F, dim, _ptr(freqs), _ptr(out32), _stream()),
           "pnc_timestep_embedding")


def nchw_to_tokens_f16(a32, C1, b32, C2, F, Npix, Cpad, out16, out16_lo=None, a_scale=None, a_frames=0):
    _check(_timed("layout", 0.0, F * Npix * (4.0 * (C1 + C2) + 2.0 * Cpad), load().pnc_nchw_to_tokens_f16, _ptr(a32),
                  C1, _ptr(a_scale), a_frames or F, _ptr(b32), C2, F, Npix, Cpad, _ptr(out16), _ptr(out16_lo), _stream()),
           "pnc_nchw_to_tokens_f16")


def cfg_euler_step(eps_tok, ld, T, Npix, Cch, cfg, scale, x, c_out, sigma, sigma_next, x_next):
    _check(_timed("elementwise", 0.0, T * Npix * Cch * (16.0 if cfg else 12.0), load().pnc_cfg_euler_step, _ptr(eps_tok), ld, T,
                  Npix, Cch, int(cfg), float(scale), _ptr(x), _ptr(c_out), _ptr(sigma), _ptr(sigma_next), _ptr(x_next),
                  _stream()), "pnc_cfg_euler_step")


def tokens_to_nchw_f32(x32, ld, F, Npix, Cch, out32):
    _check(load().pnc_tokens_to_nchw_f32(_ptr(x32), ld, F, Npix, Cch, _ptr(out32), _stream()),
           "pnc_tokens_to_nchw_f32")


def concat_add(a32, C1, s32, c32, C2, M, out32, out16, out16_lo=None, gn_part=None, frames=0, ppc=64):
    """`gn_part` (+ `frames`: M = frames * Npix; `ppc`: pixels per record): also write the GroupNorm(32) records of the output
    (pnc_concat_add_stats; [frames][ceil(Npix / ppc)][32][3] floats) — the GroupNorm that reads the concat then launches no statistics"""
    nb = M * (4.0 * C1 + (8.0 if c32 is not None else 4.0) * C2 + (6.0 + _lo_bytes(out16_lo)) * (C1 + C2))
    if gn_part is not None:
        if frames < 1 or M % frames:
            raise PncError(f"concat_add with gn_part: M = {M} rows are not {frames} whole frames")
        _check(_timed("elementwise", 0.0, nb, load().pnc_concat_add_stats, _ptr(a32), C1, _ptr(s32), _ptr(c32), C2, frames, M // frames, ppc,
                      _ptr(out32), _ptr(out16), _ptr(out16_lo), lo_fmt(out16_lo), _ptr(gn_part, torch.float32, "gn_part"), _stream()),
               "pnc_concat_add_stats")
        return
    _check(_timed("elementwise", 0.0, nb, load().pnc_concat_add, _ptr(a32), C1, _ptr(s32), _ptr(c32), C2, M,
                  _ptr(out32), _ptr(out16), _ptr(out16_lo), lo_fmt(out16_lo), _stream()), "pnc_concat_add")


def add_f32(x32, a32, n, y32, y16, y16_lo=None):
    _check(_timed("elementwise", 0.0, 12.0 * n, load().pnc_add_f32, _ptr(x32), _ptr(a32), n, _ptr(y32), _ptr(y16),
                  _ptr(y16_lo), lo_fmt(y16_lo), _stream()), "pnc_add_f32")


def range_monitor_collect(out_i32: torch.Tensor):
    """adds the number of e4m3 lo-plane quads that CLAMPED since the previous call to out_i32[0] (a device int32 word) and resets the
    library's counters (include/panacea_hip.h: pnc_range_monitor_collect)"""
    _check(load().pnc_range_monitor_collect(_ptr(out_i32, torch.int32, "out"), _stream()), "pnc_range_monitor_collect")


def cast_f16(x32, n, y16, y16_lo=None):
    _check(load().pnc_cast_f16(_ptr(x32), n, _ptr(y16), _ptr(y16_lo), lo_fmt(y16_lo), _stream()), "pnc_cast_f16")
