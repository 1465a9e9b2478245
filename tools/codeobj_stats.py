"""Per-kernel resource table of the gfx950 code objects: VGPRs, SGPRs, spills, scratch, LDS, machine-code bytes.

    python tools/codeobj_stats.py [source.hip ...]        (default: every source of panacea_amd/build.py)

Compiles each source device-only with the build's own flags (hipcc cross-compiles without a GPU), reads the
AMDGPU metadata notes and the symbol table with llvm-readelf.  The output committed under profiles/ is the evidence
for "no scratch, fits the instruction cache" claims.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from panacea_amd import build as B  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def device_elf(src: Path, tmp: Path) -> Path:
    co = tmp / (src.stem + ".co")
    subprocess.check_call([B._hipcc(), *B.FLAGS, "--offload-device-only", "-c", str(src), "-o", str(co)])
    data = co.read_bytes()
    elf = tmp / (src.stem + ".elf")
    elf.write_bytes(data[data.find(b"\x7fELF"):])
    return elf


def kernels(elf: Path):
    notes = subprocess.check_output([READELF, "--notes", str(elf)], text=True)
    syms = subprocess.check_output([READELF, "-sW", str(elf)], text=True)
    size = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    out = []
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)   # noqa: E731
        name = g("name").group(1)
        out.append(dict(name=name, vgpr=int(g("vgpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                        spill=int(g("vgpr_spill_count").group(1)), scratch=int(g("private_segment_fixed_size").group(1)),
                        lds=int(g("group_segment_fixed_size").group(1)), code=size.get(name, 0)))
    return out


def short(name: str) -> str:
    d = subprocess.check_output([CXXFILT, name], text=True).strip()
    d = re.sub(r"\(.*$", "", d).replace("pnc_gemm::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"^void ", "", d)


def main():
    srcs = [Path(a) for a in sys.argv[1:]] or [B.CSRC / s for s in B.SOURCES]
    worst = dict(spill=0, scratch=0, code=0)
    with tempfile.TemporaryDirectory() as td:
        print(f"{'kernel':86s} {'vgpr':>5s} {'sgpr':>5s} {'spill':>5s} {'scratch':>7s} {'code B':>8s}")
        for s in srcs:
            for k in sorted(kernels(device_elf(s, Path(td))), key=lambda k: -k["code"]):
                print(f"{short(k['name'])[:86]:86s} {k['vgpr']:5d} {k['sgpr']:5d} {k['spill']:5d} {k['scratch']:7d} {k['code']:8d}")
                for w in worst:
                    worst[w] = max(worst[w], k[w])
    print(f"max over all kernels: spill {worst['spill']}, scratch {worst['scratch']} B, code {worst['code']} B")


if __name__ == "__main__":
    main()
