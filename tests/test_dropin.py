"""The drop-in boundary: (1) the C-ABI library loads on a CPU-only box and exports every symbol declared in
include/panacea_hip.h; (2) with `panacea_amd.dropin.install()` the REFERENCE's own `instantiate_from_config`
builds the mirror classes from the reference's YAML target strings (needs /root/reference: build container)."""
import ctypes
import sys

import pytest
import torch

from panacea_amd import configs, hip


def test_cabi_library_loads_and_exports_header_symbols():
    lib = hip.load()
    syms = hip.header_symbols()
    assert len(syms) >= 16 and set(syms) == set(hip._SIGNATURES)
    for s in syms:
        assert getattr(lib, s) is not None
    assert lib.pnc_version().decode().startswith("panacea_hip")
    assert lib.pnc_abi_version() == hip.ABI_VERSION


def test_ctypes_structs_match_the_header_as_gcc_lays_it_out(tmp_path):
    """Every field offset and the size of both parameter blocks, as a C compiler sees include/panacea_hip.h, against
    the ctypes mirrors (a reference-side binding is written against the header, not against hip.py)."""
    import subprocess
    fields = {"PncGemmParams": [f[0] for f in hip.GemmParams._fields_], "PncAttnParams": [f[0] for f in hip.AttnParams._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{hip.HEADER}"', 'int main(void) {']
    for st, fs in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        src += [f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));' for f in fs]
    src.append('printf("abi %d\\n", PNC_ABI_VERSION); return 0; }')
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(c), "-o", str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, cls in (("PncGemmParams", hip.GemmParams), ("PncAttnParams", hip.AttnParams)):
        assert int(got[st]) == ctypes.sizeof(cls), st
        for f in fields[st]:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, (st, f)
    assert int(got["abi"]) == hip.ABI_VERSION
    assert ctypes.sizeof(hip.GemmParams) == 320


def test_enum_constants_match_the_header_as_gcc_reads_it(tmp_path):
    """PNC_OPT_* / PNC_A_* / PNC_ACT_* / error codes: the values a C caller gets from the header against the Python constants,
    and every option index accepted (and restored) by pnc_set_option on a box without a GPU."""
    import subprocess
    names = {"PNC_OPT_GEMM_TAIL_SPLIT": hip.OPT_GEMM_TAIL_SPLIT, "PNC_OPT_GEMM_TILE": hip.OPT_GEMM_TILE,
             "PNC_OPT_ATTN_VARIANT": hip.OPT_ATTN_VARIANT, "PNC_OPT_ATTN_DMA": hip.OPT_ATTN_DMA,
             "PNC_OPT_GEMM_FUSE_LN": hip.OPT_GEMM_FUSE_LN, "PNC_OPT_GEMM_GROUP_M": hip.OPT_GEMM_GROUP_M,
             "PNC_OPT_STENCIL_TILES": hip.OPT_STENCIL_TILES, "PNC_OPT_GEMM_PERSIST": hip.OPT_GEMM_PERSIST,
             "PNC_OPT_ATTN_DEFER_MAX": hip.OPT_ATTN_DEFER_MAX, "PNC_OPT_GEMM_GN_STATS": hip.OPT_GEMM_GN_STATS, "PNC_OPT_GEMM_STAGGER": hip.OPT_GEMM_STAGGER, "PNC_OPT_ATTN_SUM_TRIGGER": hip.OPT_ATTN_SUM_TRIGGER, "PNC_A_PLAIN": hip.A_PLAIN, "PNC_A_CONV3X3": hip.A_CONV3X3,
             "PNC_A_CONV1D_T": hip.A_CONV1D_T, "PNC_ACT_NONE": hip.ACT_NONE, "PNC_ACT_SILU": hip.ACT_SILU, "PNC_ACT_GELU": hip.ACT_GELU,
             "PNC_LO_F16": hip.LO_F16, "PNC_LO_E4M3": hip.LO_E4M3}
    src = ['#include <stdio.h>', f'#include "{hip.HEADER}"', 'int main(void) {']
    src += [f'printf("{n} %d\\n", (int){n});' for n in list(names) + ["PNC_OPT_COUNT"]]
    src.append('return 0; }')
    c = tmp_path / "enums.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "enums"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(c), "-o", str(exe)])
    got = {k: int(v) for k, v in (ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())}
    for n, v in names.items():
        assert got[n] == v, n
    n_opt = got["PNC_OPT_COUNT"]
    assert n_opt == 1 + max(v for n, v in names.items() if n.startswith("PNC_OPT_"))      # every option has a Python name
    hip.load()
    for opt in range(n_opt):
        before = hip.set_option(opt, 0)
        assert hip.set_option(opt, before) == 0
    with pytest.raises(Exception):
        hip.set_option(n_opt, 1)


def test_argument_validation_without_gpu():
    """Entry points validate before launching: bad arguments return PNC_E* codes even with no device."""
    lib = hip.load()
    p = hip.GemmParams()
    assert lib.pnc_gemm_f16(ctypes.byref(p), None) == -3                  # PNC_EABI: struct_bytes not set (short / old struct)
    assert lib.pnc_gemm_workspace_floats(ctypes.byref(p)) == 0
    p.struct_bytes = ctypes.sizeof(hip.GemmParams)
    assert lib.pnc_gemm_f16(ctypes.byref(p), None) == -1                  # PNC_EINVAL: null operands
    # split-K is offered only to small-M / long-K problems (the 4x48 level), never to GEGLU or V^T outputs
    p.M, p.N, p.K = 3072, 1280, 11520
    assert lib.pnc_gemm_workspace_floats(ctypes.byref(p)) == 4 * 3072 * 1280
    p.geglu = 1
    assert lib.pnc_gemm_workspace_floats(ctypes.byref(p)) == 0
    p.geglu, p.M = 0, 12288
    assert lib.pnc_gemm_workspace_floats(ctypes.byref(p)) == 0
    assert lib.pnc_attn_temporal_f16(None, 0, None, 0, None, 0, None, 0, 1, 9, 1, 1, 0.125, None) == -1
    assert lib.pnc_layernorm(None, 0, 0, 0, None, None, 1e-5, None, 0, None, None) == -1


def test_product_refuses_to_run_without_gpu_tensors():
    with pytest.raises(hip.PncError):
        hip.cast_f16(torch.zeros(8), 8, torch.zeros(8, dtype=torch.float16))


@pytest.mark.skipif(not __import__("pathlib").Path("/root/reference/sgm").exists(), reason="reference tree not present")
def test_dropin_rebinds_reference_targets():
    sys.path.insert(0, "/root/repo")
    from oracle import ref_import
    from panacea_amd import dropin, nn as mirror
    for m in list(sys.modules):
        if m.startswith("sgm"):
            del sys.modules[m]
    dropin._installed = False
    dropin.install(lazy=True, first_stage=True)                 # arm first, import the reference afterwards
    ns = ref_import.import_reference()
    import importlib
    from panacea_amd.nn import model as first_stage
    ref_model = importlib.import_module("sgm.modules.diffusionmodules.model")
    assert ref_model.Decoder is first_stage.Decoder and ref_model._reference_Decoder is not None
    assert ref_model.Encoder is not None and not issubclass(ref_model.Encoder, first_stage.Decoder)   # untouched
    assert ns.cm.ControlledUNetModel3D is mirror.ControlledUNetModel3D
    assert ns.cm.ControlNet3D is mirror.ControlNet3D
    assert ns.wr.OpenAIWrapperControlLDM3D is mirror.OpenAIWrapperControlLDM3D
    assert ns.cm._reference_ControlNet3D is not None            # the original stays reachable
    kw = configs.get("tiny")
    cfg = {"target": "sgm.modules.diffusionmodules.controlmodel.ControlledUNetModel3D",
           "params": dict(kw, out_channels=4, controlnet_config={
               "target": "sgm.modules.diffusionmodules.controlmodel.ControlNet3D",
               "params": dict(kw, hint_channels=19, control_scales=1.0)})}
    net = ns.util.instantiate_from_config(cfg)                  # the reference's own factory (sgm/util.py:168-185)
    assert isinstance(net, mirror.ControlledUNetModel3D) and isinstance(net.controlnet, mirror.ControlNet3D)
    wrapper = ns.util.get_obj_from_str(ns.wr.OPENAIUNETWRAPPERCONTROLLDM3D)(net, compile_model=False)
    assert isinstance(wrapper, mirror.OpenAIWrapperControlLDM3D)
    assert wrapper.diffusion_model.controlnet.input_hint_block[0].weight.dtype == torch.float32
    for m in list(sys.modules):
        if m.startswith("sgm") or m in ("omegaconf", "omegaconf.listconfig", "xformers", "xformers.ops"):
            del sys.modules[m]
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, dropin._Finder)]
    dropin._installed = False
    for k in dropin.FIRST_STAGE_TARGETS:
        dropin.TARGETS.pop(k, None)


def test_dropin_rebinds_the_conditioner_classes_on_request():
    """install(conditioner=True): the YAML's conditioner targets resolve to the mirror classes (needs /root/reference and the
    import shims of the golden generator: build container only)."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("/root/reference not present")
    from oracle.gen_golden_conditioner import import_conditioner
    from panacea_amd import conditioner as C, dropin
    m = import_conditioner()
    saved = {k: getattr(m, k) for k in dropin.CONDITIONER_TARGETS["sgm.modules.encoders.modules"]}
    # the bindings other modules of the reference hold BY VALUE when sgm was imported before install(): the YAML names
    # `sgm.modules.GeneralConditioner` (a re-export of sgm/modules/__init__.py), diffusion.py keeps its own `VAEEmbedder`
    import types
    pkg = sys.modules.get("sgm.modules") or types.ModuleType("sgm.modules")
    diff = sys.modules.get("sgm.models.diffusion") or types.ModuleType("sgm.models.diffusion")
    had = {n: n in sys.modules for n in ("sgm.modules", "sgm.models.diffusion")}
    sys.modules["sgm.modules"], sys.modules["sgm.models.diffusion"] = pkg, diff
    pkg.GeneralConditioner, diff.VAEEmbedder = saved["GeneralConditioner"], saved["VAEEmbedder"]
    try:
        dropin.install(lazy=True, conditioner=True)
        assert m.GeneralConditioner is C.GeneralConditioner and m.FrozenOpenCLIPEmbedder is C.FrozenOpenCLIPEmbedder
        assert m._reference_GeneralConditioner is saved["GeneralConditioner"]
        # resolved the way instantiate_from_config does it: getattr(import_module("sgm.modules"), "GeneralConditioner")
        import importlib
        assert getattr(importlib.import_module("sgm.modules"), "GeneralConditioner") is C.GeneralConditioner
        assert diff.VAEEmbedder is C.VAEEmbedder
        assert issubclass(C.VAEEmbedder, C.AbstractEmbModel)        # the isinstance check of GeneralConditioner.__init__
    finally:
        for k, v in saved.items():
            setattr(m, k, v)
        for k in dropin.CONDITIONER_TARGETS:
            dropin.TARGETS.pop(k, None)
        for n, was in had.items():
            if not was:
                sys.modules.pop(n, None)
        if had["sgm.modules"]:
            pkg.__dict__.pop("GeneralConditioner", None)
        if had["sgm.models.diffusion"]:
            diff.__dict__.pop("VAEEmbedder", None)
