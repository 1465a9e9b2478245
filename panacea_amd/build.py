"""Build libpanacea_hip.so (gfx950) in-tree with hipcc.

    python -m panacea_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object lands in panacea_amd/lib/ (git-ignored,
travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libpanacea_hip.so"
SOURCES = ["gemm.hip", "gemm_plain.hip", "gemm_conv3x3.hip", "gemm_stencil_tile.hip", "gemm_conv1d.hip", "attn.hip", "norm.hip", "misc.hip"]
HEADERS = [CSRC / "common.h", CSRC / "gemm_kernel.h", ROOT / "include" / "panacea_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", str(ROOT / "include"), "-I", str(CSRC)]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _digest() -> str:
    h = hashlib.sha256()
    for f in [CSRC / s for s in SOURCES] + HEADERS:
        h.update(f.read_bytes())
    # flags enter WITHOUT the checkout's absolute path: the digest must be the same wherever the tree is copied to
    h.update(" ".join(f.replace(str(ROOT), "<root>") for f in FLAGS).encode())
    return h.hexdigest()


def library_digest(lib: Path = LIB):
    """The source digest compiled INTO a built library (pnc_build_digest()), or None when it cannot be read.  The digest
    travels inside the .so, so a checkout that pulls new sources next to an old (git-ignored) build is detected whatever
    side files say (a tracked build.stamp could match the new sources next to the old library)."""
    try:
        blob = Path(lib).read_bytes()
    except OSError:
        return None
    i = blob.find(_DIGEST_MARK)          # read from the file, never by loading it (a loaded image would shadow a rebuild)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(_DIGEST_MARK):j].decode(errors="replace")


_DIGEST_MARK = b"pnc-build-digest:"


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    LIBDIR.mkdir(exist_ok=True)
    dig = _digest()
    if not force and LIB.exists() and library_digest() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        obj = LIBDIR / (Path(s).stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(CSRC / s), "-o", str(obj)]
        if s == "misc.hip":          # pnc_build_digest() lives there
            cmd.insert(1, f'-DPNC_BUILD_DIGEST="{dig}"')
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(obj))
    failed = False
    for s, pr in procs:
        out, _ = pr.communicate()
        if out.strip() and verbose:
            print(out)
        if pr.returncode:
            failed = True
            print(f"hipcc failed on {s}:\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
