"""Plain-torch emulation of every entry point of panacea_amd.hip (same names, same arguments,
results written into the caller's output tensors).

TEST INFRASTRUCTURE ONLY.  Two uses:
  * `-m gpu` kernel tests: HIP kernel vs this emulation on the same device tensors;
  * `-m "not gpu"` host-logic tests: the engine (weight packing, layouts, quirks, graph order) runs
    on CPU against this module injected in place of `panacea_amd.hip`, and is compared with the oracle.
The product never imports it: `panacea_amd.hip` has no fallback and raises without the HIP library.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as TF

A_PLAIN, A_CONV3X3, A_CONV1D_T = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2


class PncError(RuntimeError):
    pass


def _r16_default(v: torch.Tensor, site: str) -> torch.Tensor:
    return v.half()


# every fp16 rounding of an activation goes through this hook: r16(value_fp32, producing_entry_point).  The default is
# the kernels' rounding; tools/exp/error_budget.py swaps it to measure what each operand class contributes.
r16 = _r16_default
STRICT_DTYPES = True
LO_SCALE = 2048.0


def _lo(v: torch.Tensor, hi: torch.Tensor, like: torch.Tensor = None) -> torch.Tensor:
    """lo plane of a precise (split) operand, exactly like the kernels: fp16((v - hi) * 2^11), or — when the destination `like`
    is a uint8 tensor (PNC_LO_E4M3) — the OCP e4m3 byte of the same residual clamped to +-448"""
    r = (v.float() - hi.float()) * LO_SCALE
    if like is not None and like.dtype == torch.uint8:
        global _LO_CLAMPED
        _LO_CLAMPED += int((r.abs() > 448.0).reshape(-1, 4).any(dim=1).sum()) if r.numel() % 4 == 0 else int((r.abs() > 448.0).sum())
        return r.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    return r.half()


_LO_CLAMPED = 0      # e4m3 lo-plane quads that clamped since the last range_monitor_collect (the kernels' per-quad count)


def range_monitor_collect(out_i32: torch.Tensor):
    """pnc_range_monitor_collect: adds the clamped-quad count since the previous call to out_i32[0] and resets it"""
    global _LO_CLAMPED
    out_i32[0] += _LO_CLAMPED
    _LO_CLAMPED = 0


def _lo_value(lo: torch.Tensor) -> torch.Tensor:
    """fp32 value of a lo plane in either storage format"""
    return lo.view(torch.float8_e4m3fn).float() if lo.dtype == torch.uint8 else lo.float()


def _join(hi: torch.Tensor, lo) -> torch.Tensor:
    return hi.float() if lo is None else hi.float() + _lo_value(lo) / LO_SCALE


# ---- MX block-scaled fp8 lo plane (the format of the NEXT round's lo pass; nothing in the library
# produces or consumes it yet — this is the reference the producers / the scaled-MFMA lo pass will be checked against).
# A block is 32 consecutive channels of one row; its scale is one E8M0 byte (2^(byte - 127)); elements are OCP e4m3 bytes.
MX_BLOCK = 32


def mx8_quant(r: torch.Tensor):
    """fp32 [..., C] (C % 32 == 0) -> (e4m3 bytes uint8 [..., C], E8M0 scale bytes uint8 [..., C / 32]).  The scale puts the block
    maximum in [128, 256) (e4m3 tops out at 448: no saturation, at most one binade of head room lost); an all-zero block gets
    scale byte 0 and zero elements."""
    shp = r.shape
    b = r.float().reshape(-1, shp[-1] // MX_BLOCK, MX_BLOCK)
    amax = b.abs().amax(dim=-1, keepdim=True)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-38))) - 7.0, torch.full_like(amax, -127.0)).clamp(-127.0, 127.0)
    q = (b / torch.exp2(e)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(shp), (e + 127.0).to(torch.uint8).reshape(*shp[:-1], shp[-1] // MX_BLOCK)


def mx8_dequant(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    shp = q.shape
    v = q.view(torch.float8_e4m3fn).float().reshape(-1, shp[-1] // MX_BLOCK, MX_BLOCK)
    return (v * torch.exp2(scale.float().reshape(-1, shp[-1] // MX_BLOCK, 1) - 127.0)).reshape(shp)


def _lo_mx8(v: torch.Tensor, hi: torch.Tensor):
    """MX fp8 lo plane of a split operand: (v - hi) * 2^11, block-scaled"""
    return mx8_quant((v.float() - hi.float()) * LO_SCALE)


def _mat(t: torch.Tensor, rows: int, cols: int, ld: int) -> torch.Tensor:
    """[rows, cols] strided view (row stride ld) of the flat storage of t."""
    return torch.as_strided(t.reshape(-1), (rows, cols), (ld, 1))


def load():
    return None


ACC_HOOK = None


def _contract(a16, Wm, M, N, K, lda, a_mode, conv, tconv):
    """gatherA[M, K] @ Wm[N, K]^T in fp32: the contraction of every GEMM mode (Wm already a float [N, K] matrix)"""
    if a_mode == A_PLAIN:
        A = _mat(a16, M, K, lda).float()
        acc = A @ Wm.t()
    elif a_mode == A_CONV3X3:
        Cin, Hin, Win, Hout, Wout = conv["Cin"], conv["Hin"], conv["Win"], conv["Hout"], conv["Wout"]
        stride, up = conv.get("stride", 1), conv.get("upsample", 0)
        Fr = M // (Hout * Wout)
        x = a16.reshape(-1)[: Fr * Hin * Win * Cin].view(Fr, Hin, Win, Cin).permute(0, 3, 1, 2).float()
        xh = int(conv.get("x_halo_off", 0))      # PncGemmParams.x_halo_off: columns -1 and Win of a view band, [2][Fr][Hin][Cin]
        if xh:
            cols = a16.reshape(-1)[xh: xh + 2 * Fr * Hin * Cin].view(2, Fr, Hin, Cin).permute(0, 1, 3, 2).float()
            x = torch.cat([cols[0][..., None], x, cols[1][..., None]], dim=3)
        if up:
            x = TF.interpolate(x, scale_factor=2, mode="nearest")
            if xh:
                x = x[..., 1:-1]
        if Cin % 64 == 0:     # K order (ci/64, ky, kx, ci%64)
            w = Wm.view(N, Cin // 64, 3, 3, 64).permute(0, 1, 4, 2, 3).reshape(N, Cin, 3, 3)
        else:                 # K order (ky, kx, ci)
            w = Wm.view(N, 3, 3, Cin).permute(0, 3, 1, 2)
        if conv.get("pad_br", 0):
            y = TF.conv2d(TF.pad(x, (0, 1, 0, 1)), w, stride=stride, padding=0)
        elif xh:
            y = TF.conv2d(TF.pad(x, (0, 0, 1, 1)), w, stride=stride, padding=0)
        else:
            y = TF.conv2d(x, w, stride=stride, padding=1)
        assert y.shape[2] == Hout and y.shape[3] == Wout, (y.shape, Hout, Wout)
        acc = y.permute(0, 2, 3, 1).reshape(M, N)
    else:
        Cc, T, Npix = tconv["C"], tconv["T"], tconv["Npix"]
        B = M // (T * Npix)
        halo = int(tconv.get("halo", 0))       # PncGemmParams.t_halo: A holds T + 2 frames per sample, no temporal padding
        Ta = T + 2 * halo
        x = a16.reshape(-1)[: B * Ta * Npix * Cc].view(B, Ta, Npix, Cc).permute(0, 2, 3, 1).reshape(B * Npix, Cc, Ta).float()
        if Cc % 64 == 0:      # K order (ci/64, dt, ci%64)
            w = Wm.view(N, Cc // 64, 3, 64).permute(0, 1, 3, 2).reshape(N, Cc, 3)
        else:                 # K order (dt, ci)
            w = Wm.view(N, 3, Cc).permute(0, 2, 1)
        y = TF.conv1d(x, w, padding=0 if halo else 1)        # [B*Npix, N, T]
        acc = y.view(B, Npix, N, T).permute(0, 3, 1, 2).reshape(M, N)
    return acc


def gemm(a16, w16, *, M, N, K, lda=0, a_mode=A_PLAIN, conv=None, tconv=None, bias=None, rowbias=None,
         rb_rows=0, rb_mod=0, res1=None, ldr1=0, res2=None, ldr2=0, out32=None, ldc32=0, out16=None,
         ldc16=0, out16t=None, ldt=0, t_rows=0, t_gstride=0, n_split=0, act=ACT_NONE, geglu=False,
         a16_lo=None, out16_lo=None, w_ld=0, ln_gamma=None, ln_beta=None, ln_out16=None, ldln=0, ln_eps=1e-5,
         ln_in_library=False, w_lo=None, gn_part=None):
    assert (not STRICT_DTYPES) or (a16.dtype == torch.float16 and w16.dtype == torch.float16)
    assert K % 8 == 0
    Wm = _mat(w16, N, K, w_ld or K).float()
    acc_lo = None
    if a16_lo is not None and a16_lo.dtype == torch.uint8:
        # e4m3 lo plane: its pass runs against the e4m3 copy of the weights (engine.pk_lo8), products weighted 2^-11
        if w_lo is None:
            raise PncError("an e4m3 lo plane needs the e4m3 copy of the weights (w_lo)")
        cin = lda if a_mode == A_PLAIN else (conv["Cin"] if a_mode == A_CONV3X3 else tconv["C"])
        if K % 16 or cin % (16 if a_mode == A_PLAIN else 64):      # the library's PNC_EALIGN
            raise PncError("e4m3 lo pass: K % 16 == 0 and lda % 16 == 0 (plain) / Cin % 64 == 0 (conv gathers)")
        Wl = _mat(w_lo[0], N, K, w_lo[0].shape[-1]).view(torch.float8_e4m3fn).float() * (2.0 ** (int(w_lo[1]) - 127))
        acc_lo = _contract(_lo_value(a16_lo.reshape(-1)[: a16.numel()]).reshape(a16.shape), Wl, M, N, K, lda, a_mode, conv, tconv) / LO_SCALE
    elif a16_lo is not None:     # fp16 lo plane: A = hi + lo * 2^-11 (the kernel sums the two planes' products in fp32)
        a16 = _join(a16.reshape(-1), a16_lo.reshape(-1)[: a16.numel()])
    acc = _contract(a16, Wm, M, N, K, lda, a_mode, conv, tconv)
    if acc_lo is not None:
        acc = acc + acc_lo
    if ACC_HOOK is not None:     # numerics experiments (tools/exp/error_budget.py): e.g. the weight side of an MX lo pass
        acc = ACC_HOOK(acc, a16, Wm, dict(M=M, N=N, K=K, lda=lda, a_mode=a_mode, conv=conv, tconv=tconv))
    v = acc
    if bias is not None:
        v = v + bias.reshape(-1)[:N].float()
    if geglu:
        assert N % 64 == 0
        vb = v.view(M, N // 64, 2, 32)
        v = (vb[:, :, 0, :] * TF.gelu(vb[:, :, 1, :])).reshape(M, N // 2)
        if out32 is not None:
            _mat(out32, M, N // 2, ldc32).copy_(v)
        if out16 is not None:
            h = r16(v, 'gemm.geglu')
            _mat(out16, M, N // 2, ldc16).copy_(h)
            if out16_lo is not None:
                _mat(out16_lo, M, N // 2, ldc16).copy_(_lo(v, h, out16_lo))
        return
    if rowbias is not None:
        idx = (torch.arange(M, device=v.device) // rb_rows) % rb_mod
        v = v + rowbias.reshape(-1)[: rb_mod * N].view(rb_mod, N).float()[idx]
    if act == ACT_SILU:
        v = TF.silu(v)
    if act == ACT_GELU:
        v = TF.gelu(v)
    if res1 is not None:
        v = v + _mat(res1, M, N, ldr1)
    if res2 is not None:
        v = v + _mat(res2, M, N, ldr2)
    ns = n_split if out16t is not None else N
    if out32 is not None:
        _mat(out32, M, ns, ldc32).copy_(v[:, :ns])
    if out16 is not None:
        h = r16(v[:, :ns], 'gemm.out16')
        _mat(out16, M, ns, ldc16).copy_(h)
        if out16_lo is not None:
            _mat(out16_lo, M, ns, ldc16).copy_(_lo(v[:, :ns], h, out16_lo))
    if ln_out16 is not None:
        assert out32 is not None and out16t is None
        layernorm(out32, ldc32, M, N, ln_gamma, ln_beta, ln_eps, ln_out16, ldln)
    if gn_part is not None:      # PncGemmParams.gn_part: the records pnc_groupnorm_stats writes for 64-pixel chunks
        assert a_mode == A_CONV1D_T and out32 is not None and N % 64 == 0
        groupnorm_stats(out32, ldc32, M // tconv["Npix"], tconv["Npix"], N, 64, gn_part)
    if out16t is not None:
        G = M // t_rows
        assert G * t_rows == M
        dst = torch.as_strided(out16t.reshape(-1), (G, N - ns, t_rows), (t_gstride, ldt, 1))
        dst.copy_(r16(v[:, ns:], 'gemm.out16t').view(G, t_rows, N - ns).permute(0, 2, 1))


PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def _unfrag(fr: torch.Tensor) -> torch.Tensor:
    """inverse of engine.mfma_a_fragments: [R, S, 64, 8] -> [32 R, 16 S]"""
    R, S = fr.shape[0], fr.shape[1]
    wp = fr.reshape(R, S, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(R, 32, S, 16)
    inv = [PERM16.index(k) for k in range(16)]
    return wp[..., inv].reshape(R * 32, S * 16)


def attn_views(q, ldq, k, ldk, vt, ldvt, vt_gstride, o, ldo, *, groups, heads, H, W, views, kvH, kvW,
               kv_views, kv_rows_per_group, q_per_kv, kv_valid, segs, scale, causal=False, k_halo=None, vt_halo=None):
    Cc = heads * 64
    Wv, kvWv = W // views, kvW // kv_views
    Q = _mat(q, groups * H * W, Cc, ldq).float().view(groups, H, W, heads, 64)
    n_kvg = (groups + q_per_kv - 1) // q_per_kv

    def kv_maps(kb, vb):
        Ka = _mat(kb, n_kvg * kv_rows_per_group, Cc, ldk).float().view(n_kvg, kv_rows_per_group, heads, 64)
        Ka = Ka[:, : kvH * kvW].reshape(n_kvg, kvH, kvW, heads, 64)
        Va = torch.as_strided(vb.reshape(-1), (n_kvg, Cc, kvH * kvW), (vt_gstride, ldvt, 1)).float()
        return Ka, Va.view(n_kvg, heads, 64, kvH, kvW)
    Kband, Vband = kv_maps(k, vt)
    # PncAttnParams.k_halo: view id -1 / kv_views = view column 0 of a buffer laid out like the band's
    halos = None if k_halo is None else [kv_maps(k_halo[i], vt_halo[i]) for i in range(2)]
    O = _mat(o, groups * H * W, Cc, ldo).view(groups, H, W, Cc)
    gidx = torch.arange(groups, device=q.device) // q_per_kv
    for v in range(views):
        qv = Q[:, :, v * Wv:(v + 1) * Wv].reshape(groups, H * Wv, heads, 64).permute(0, 2, 1, 3)
        ks, vs = [], []
        for u in segs[v]:
            Kall, Vall = Kband, Vband
            if u == -1 or u == kv_views:
                Kall, Vall = halos[0 if u < 0 else 1]
                u = 0
            kk = Kall[:, :, u * kvWv:(u + 1) * kvWv].reshape(n_kvg, kvH * kvWv, heads, 64)[:, :kv_valid]
            vv = Vall[:, :, :, :, u * kvWv:(u + 1) * kvWv].reshape(n_kvg, heads, 64, kvH * kvWv)[..., :kv_valid]
            ks.append(kk.permute(0, 2, 1, 3))          # [g, heads, keys, 64]
            vs.append(vv.permute(0, 1, 3, 2))
        kc = torch.cat(ks, dim=2)[gidx]
        vc = torch.cat(vs, dim=2)[gidx]
        s = torch.einsum("ghqd,ghkd->ghqk", qv, kc) * scale
        if causal:          # view-local key index <= view-local query index, in every key segment
            nk = min(kv_valid, kvH * kvWv)
            kidx = torch.arange(nk, device=s.device).repeat(len(segs[v]))
            qidx = torch.arange(H * Wv, device=s.device)
            s = s.masked_fill(kidx[None, :] > qidx[:, None], float("-inf"))
        pr = torch.softmax(s, dim=-1)
        ov = torch.einsum("ghqk,ghkd->ghqd", pr, vc)   # [g, heads, q, 64]
        ov = ov.permute(0, 2, 1, 3).reshape(groups, H, Wv, Cc)
        O[:, :, v * Wv:(v + 1) * Wv] = r16(ov, 'attn_views')


def softmax_rows(s32, lds, M, N, scale, p16, ldp, causal=False, n_valid=0):
    sc = _mat(s32, M, N, lds).float() * scale
    col, row = torch.arange(N, device=sc.device)[None, :], torch.arange(M, device=sc.device)[:, None]
    keep = col < (n_valid or N)
    if causal:
        keep = keep & (col <= row)
    pr = torch.softmax(sc.masked_fill(~keep, float("-inf")), dim=-1)
    _mat(p16, M, N, ldp).copy_(r16(pr, 'softmax_rows'))


def attn_temporal(q, ldq, k, ldk, v, ldv, o, ldo, *, B, T, Npix, heads, scale):
    Cc = heads * 64
    M = B * T * Npix

    def g(t, ld):
        return _mat(t, M, Cc, ld).float().view(B, T, Npix, heads, 64).permute(0, 2, 3, 1, 4)  # b p h t d
    s = torch.einsum("bphtd,bphsd->bphts", g(q, ldq), g(k, ldk)) * scale
    pr = torch.softmax(s, dim=-1)
    ov = torch.einsum("bphts,bphsd->bphtd", pr, g(v, ldv))
    _mat(o, M, Cc, ldo).copy_(r16(ov.permute(0, 3, 1, 2, 4).reshape(M, Cc), 'attn_temporal'))


def groupnorm_stats(x32, ldx, F, Npix, Cch, ppc, partial):
    nchunk = (Npix + ppc - 1) // ppc
    X = _mat(x32, F * Npix, Cch, ldx).view(F, Npix, 32, Cch // 32)
    P = partial.reshape(-1)[: F * nchunk * 32 * 3].view(F, nchunk, 32, 3)
    for c in range(nchunk):
        xs = X[:, c * ppc:(c + 1) * ppc].permute(0, 2, 1, 3).reshape(F, 32, -1)
        n = xs.shape[-1]
        mean = xs.mean(-1)
        P[:, c, :, 0] = n
        P[:, c, :, 1] = mean
        P[:, c, :, 2] = ((xs - mean[..., None]) ** 2).sum(-1)


def groupnorm_combine(parts_in, parts, F, nchunk, out):
    P = parts_in.reshape(-1)[: parts * F * nchunk * 96].view(parts, F, nchunk, 32, 3).double()
    P = P.permute(1, 0, 2, 3, 4).reshape(F, parts * nchunk, 32, 3)
    n = P[..., 0].sum(1)
    mean = (P[..., 0] * P[..., 1]).sum(1) / n
    m2 = (P[..., 2] + P[..., 0] * (P[..., 1] - mean[:, None]) ** 2).sum(1)
    O = out.reshape(-1)[: F * nchunk * 96].view(F, nchunk, 32, 3)
    O.zero_()
    O[:, 0] = torch.stack([n, mean, m2], dim=-1).to(O.dtype)


def groupnorm_apply(x32, ldx, F, Npix, Cch, ppc, partial, gamma, beta, eps, silu, y16, ldy, y16_lo=None, n_records=0):
    nchunk = n_records or (Npix + ppc - 1) // ppc
    P = partial.reshape(-1)[: F * nchunk * 32 * 3].view(F, nchunk, 32, 3).double()
    n = P[..., 0].sum(1)
    mean = (P[..., 0] * P[..., 1]).sum(1) / n
    m2 = (P[..., 2] + P[..., 0] * (P[..., 1] - mean[:, None]) ** 2).sum(1)
    rstd = 1.0 / torch.sqrt(m2 / n + eps)
    X = _mat(x32, F * Npix, Cch, ldx).view(F, Npix, 32, Cch // 32)
    y = (X - mean.float()[:, None, :, None]) * rstd.float()[:, None, :, None]
    y = y.reshape(F * Npix, Cch) * gamma.reshape(-1)[:Cch] + beta.reshape(-1)[:Cch]
    if silu:
        y = TF.silu(y)
    h = r16(y, 'groupnorm_apply')
    _mat(y16, F * Npix, Cch, ldy).copy_(h)
    if y16_lo is not None:
        _mat(y16_lo, F * Npix, Cch, ldy).copy_(_lo(y, h, y16_lo))


def groupnorm_temporal_part(x32, B, T, Npix, Cch, gamma, beta, eps, stats, mode, T_total, y16=None, y16_lo=None, t_pad=0):
    X = x32.reshape(-1)[: B * T * Npix * Cch].view(B, T, Npix, 32, Cch // 32).float()
    S = stats.reshape(-1)[: B * Npix * 64].view(B, Npix, 32, 2)
    if mode == 1:
        S[..., 0] = X.sum(dim=(1, 4))
        S[..., 1] = (X * X).sum(dim=(1, 4))
        return
    n = float((Cch // 32) * T_total)
    mean = S[..., 0].double() / n
    var = (S[..., 1].double() / n - mean * mean).clamp_min(0.0)
    rstd = (1.0 / torch.sqrt(var + eps)).float()
    y = (X - mean.float()[:, None, :, :, None]) * rstd[:, None, :, :, None]
    y = TF.silu(y.reshape(B, T, Npix, Cch) * gamma.reshape(-1)[:Cch] + beta.reshape(-1)[:Cch])
    h = r16(y, 'groupnorm_temporal')
    Y = y16.reshape(-1)[: B * (T + 2 * t_pad) * Npix * Cch].view(B, T + 2 * t_pad, Npix, Cch)
    Y[:, t_pad:t_pad + T] = h
    if y16_lo is not None:
        L = y16_lo.reshape(-1)[: B * (T + 2 * t_pad) * Npix * Cch].view(B, T + 2 * t_pad, Npix, Cch)
        L[:, t_pad:t_pad + T] = _lo(y, h, y16_lo)


def groupnorm_temporal_silu(x32, B, T, Npix, Cch, gamma, beta, eps, y16, y16_lo=None):
    X = x32.reshape(-1)[: B * T * Npix * Cch].view(B, T, Npix, Cch).permute(0, 2, 3, 1).reshape(B * Npix, Cch, T)
    y = TF.silu(TF.group_norm(X, 32, gamma.reshape(-1)[:Cch], beta.reshape(-1)[:Cch], eps))
    y = y.view(B, Npix, Cch, T).permute(0, 3, 1, 2).reshape(-1)
    h = r16(y, 'groupnorm_temporal')
    y16.reshape(-1)[: y.numel()].copy_(h)
    if y16_lo is not None:
        y16_lo.reshape(-1)[: y.numel()].copy_(_lo(y, h, y16_lo))


def layernorm(x32, ldx, M, Cch, gamma, beta, eps, y16, ldy, y16_lo=None):
    X = _mat(x32, M, Cch, ldx)
    y = TF.layer_norm(X, (Cch,), gamma.reshape(-1)[:Cch], beta.reshape(-1)[:Cch], eps)
    h = r16(y, 'layernorm')
    _mat(y16, M, Cch, ldy).copy_(h)
    if y16_lo is not None:
        _mat(y16_lo, M, Cch, ldy).copy_(_lo(y, h, y16_lo))


def linear_smallm(a32, lda, w16, bias, out32, ldo, M, N, K, silu_in=False, silu_out=False):
    A = _mat(a32, M, K, lda)
    if silu_in:
        A = TF.silu(A)
    y = A @ w16.reshape(-1)[: N * K].view(N, K).float().t()
    if bias is not None:
        y = y + bias.reshape(-1)[:N]
    if silu_out:
        y = TF.silu(y)
    _mat(out32, M, N, ldo).copy_(y)


def linear_smallm_segments(a32, lda, w16, bias, out32, M, m0, Mtot, N, K, seg_start, silu_in=False, silu_out=False):
    A = _mat(a32, M, K, lda)
    if silu_in:
        A = TF.silu(A)
    y = A @ w16.reshape(-1)[: N * K].view(N, K).float().t()
    if bias is not None:
        y = y + bias.reshape(-1)[:N]
    if silu_out:
        y = TF.silu(y)
    flat = out32.reshape(-1)
    for s0, s1 in zip(seg_start[:-1], seg_start[1:]):
        flat[s0 * Mtot:s1 * Mtot].view(Mtot, s1 - s0)[m0:m0 + M].copy_(y[:, s0:s1])


def timestep_embedding(t_i64, F, dim, freqs, out32):
    args = t_i64.reshape(-1)[:F, None].float() * freqs.reshape(-1)[None, : dim // 2]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    out32.reshape(-1)[: F * dim].view(F, dim)[:, : 2 * (dim // 2)].copy_(emb)


def nchw_to_tokens_f16(a32, C1, b32, C2, F, Npix, Cpad, out16, out16_lo=None, a_scale=None, a_frames=0):
    v = torch.zeros((F, Npix, Cpad), dtype=torch.float32, device=a32.device)
    af = a_frames or F
    a = a32.reshape(af, C1, Npix).float().repeat(F // af, 1, 1)
    if a_scale is not None:
        a = a * a_scale.reshape(F, 1, 1).float()
    v[:, :, :C1] = a.permute(0, 2, 1)
    if C2:
        v[:, :, C1:C1 + C2] = b32.reshape(F, C2, Npix).permute(0, 2, 1)
    h = r16(v, 'nchw_to_tokens')
    out16.reshape(-1)[: F * Npix * Cpad].view(F, Npix, Cpad).copy_(h)
    if out16_lo is not None:
        out16_lo.reshape(-1)[: F * Npix * Cpad].view(F, Npix, Cpad).copy_(_lo(v, h))


def cfg_euler_step(eps_tok, ld, T, Npix, Cch, cfg, scale, x, c_out, sigma, sigma_next, x_next):
    """the reference's op sequence (denoiser.py:22-28, guiders.py:25-29, sampling.py:96-133) on channels-last eps"""
    E = _mat(eps_tok, (2 if cfg else 1) * T * Npix, Cch, ld).view(-1, T, Npix, Cch).permute(0, 1, 3, 2)   # [halves, T, C, Npix]
    X = x.reshape(T, Cch, Npix)
    sg = sigma.reshape(T, 1, 1)
    D = E * c_out.reshape(T, 1, 1) + X
    Dn = D[0] + scale * (D[1] - D[0]) if cfg else D[0]
    d = (X - Dn) / sg
    x_next.reshape(T, Cch, Npix).copy_(X + (sigma_next.reshape(T, 1, 1) - sg) * d)


def tokens_to_nchw_f32(x32, ld, F, Npix, Cch, out32):
    X = _mat(x32, F * Npix, Cch, ld).view(F, Npix, Cch)
    out32.reshape(-1)[: F * Cch * Npix].view(F, Cch, Npix).copy_(X.permute(0, 2, 1))


def concat_add(a32, C1, s32, c32, C2, M, out32, out16, out16_lo=None, gn_part=None, frames=0, ppc=64):
    a = a32.reshape(-1)[: M * C1].view(M, C1)
    s = s32.reshape(-1)[: M * C2].view(M, C2)
    if c32 is not None:
        s = s + c32.reshape(-1)[: M * C2].view(M, C2)
    y = torch.cat([a, s], dim=1)
    if out32 is not None:
        out32.reshape(-1)[: y.numel()].copy_(y.reshape(-1))
    if out16 is not None:
        h = r16(y.reshape(-1), 'concat_add')
        out16.reshape(-1)[: y.numel()].copy_(h)
        if out16_lo is not None:
            out16_lo.reshape(-1)[: y.numel()].copy_(_lo(y.reshape(-1), h, out16_lo))
    if gn_part is not None:       # pnc_concat_add_stats: the records pnc_groupnorm_stats writes for 64-pixel chunks
        assert frames >= 1 and M % frames == 0 and (C1 + C2) % 64 == 0
        groupnorm_stats(y.contiguous(), C1 + C2, frames, M // frames, C1 + C2, ppc, gn_part)


def add_f32(x32, a32, n, y32, y16, y16_lo=None):
    y = x32.reshape(-1)[:n]
    if a32 is not None:
        y = y + a32.reshape(-1)[:n]
    if y16 is not None:
        h = r16(y, 'add_f32')
        y16.reshape(-1)[:n].copy_(h)
        if y16_lo is not None:
            y16_lo.reshape(-1)[:n].copy_(_lo(y, h, y16_lo))
    if y32 is not None and (a32 is not None or y32.data_ptr() != x32.data_ptr()):
        y32.reshape(-1)[:n].copy_(y)


def cast_f16(x32, n, y16, y16_lo=None):
    add_f32(x32, None, n, None, y16, y16_lo)
