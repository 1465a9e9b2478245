"""HBM-side traffic of one denoising step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass:
TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"), calibrated on the LayerNorm launches whose byte count is known
(the guide: FETCH_SIZE under-counts wide loads by 2 on gfx950, WRITE_SIZE uncalibrated).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <evaluations> <precision> <command string>

Reads the counter_collection CSVs under the two rocprofv3 output directories, prints per-kernel tables and merges a record
    records[precision] = {build_stamp, fetch/write factors, traffic_GB_calibrated, ...}
into profiles/round6/pmc_traffic.json, which bench.py reports as roofline.traffic ONLY while the build stamp matches.
`--mfma` adds the SQ_VALU_MFMA_BUSY_CYCLES record of a third pass — stamped with the build digest too (round 6: round 5's record
was carried over from another build; bench.py now refuses an `mfma` record whose stamp differs, as it does for `traffic`).
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
OUT = ROOT / "profiles" / "round6" / "pmc_traffic.json"


def library_kernel_names():
    """the __global__ functions of panacea_amd/csrc: what "the library's launches" means (round 5 filtered by namespace spelling,
    which also matched torch's at::native::(anonymous namespace) kernels — ADVICE r5)"""
    import re
    names = set()
    for f in list((ROOT / "panacea_amd" / "csrc").glob("*.hip")) + list((ROOT / "panacea_amd" / "csrc").glob("*.h")):
        names.update(re.findall(r"__global__.{0,160}?\bvoid\s+(\w+)\s*\(", f.read_text(), flags=re.S))
    return names


_LIB_KERNELS = library_kernel_names()


def is_library_kernel(name: str) -> bool:
    if name.startswith("at::") or "at::native" in name:
        return False
    return any(k in name for k in _LIB_KERNELS)
# Calibration kernel: LayerNorm moves exactly 6 B per element (4 B fp32 read + 2 B fp16 write).  The elements of each dispatch
# follow from its grid: layernorm_kernel<J> runs 16 rows per 256-thread block, J = ceil(C / 256) names the width class.
LN_WIDTH = {2: 320, 3: 640, 5: 1280}


def layernorm_bytes(d):
    """bytes all LayerNorm dispatches under directory d move (summed over the run), from Grid_Size and the template width"""
    import re
    total, seen = 0.0, set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"layernorm_kernelILi(\d+)E|layernorm_kernel<(\d+)>", r["Kernel_Name"])
            if not m or r["Dispatch_Id"] in seen:
                continue
            seen.add(r["Dispatch_Id"])
            j = int(m.group(1) or m.group(2))
            rows = int(r["Grid_Size"]) // 256 * 16
            total += 6.0 * rows * LN_WIDTH[j]
    return total


def per_kernel(d, counter):
    agg = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]] += float(r["Counter_Value"])
    return agg


def mfma_record(mdir, evals, step_ms):
    """SQ_VALU_MFMA_BUSY_CYCLES summed over all kernels of one evaluation vs the cycles 1024 SIMDs offer in a step"""
    busy = per_kernel(mdir, "SQ_VALU_MFMA_BUSY_CYCLES")
    per_eval = sum(busy.values()) / evals
    top = sorted(busy.items(), key=lambda kv: -kv[1])[:12]
    from panacea_amd import build as _build
    return {"build_stamp": _build.library_digest(), "mfma_busy_cycles_per_step": per_eval,
            "note": "SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is occupied, summed over the 1024 SIMDs.  96.21 algorithmic "
                    "TFLOP at 1024 flop/cycle/SIMD (fp16 32x32x16) = 9.40e10 cycles; the e4m3 lo passes of the precise policy run at "
                    "2048 flop/cycle/SIMD",
            "utilisation_at_2p1GHz": per_eval / (1024 * 2.1e9 * step_ms * 1e-3),
            "top_kernels": [[k[:90], v / evals] for k, v in top]}


def main():
    if sys.argv[1] == "--mfma":
        mdir, evals, prec, step_ms = sys.argv[2], int(sys.argv[3]), sys.argv[4], float(sys.argv[5])
        out = OUT
        out.parent.mkdir(parents=True, exist_ok=True)
        doc = json.loads(out.read_text()) if out.exists() else {"records": {}}
        doc.setdefault("mfma", {})[prec] = mfma_record(mdir, evals, step_ms)
        out.write_text(json.dumps(doc, indent=1))
        print(json.dumps(doc["mfma"][prec], indent=1)[:1500])
        return
    fdir, wdir, evals, prec, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")      # KB
    ln_f = sum(v for k, v in fetch.items() if "layernorm" in k) * 1024 / evals
    ln_w = sum(v for k, v in write.items() if "layernorm" in k) * 1024 / evals
    LN_BYTES_PER_EVAL = layernorm_bytes(fdir) / evals
    ff, wf = (LN_BYTES_PER_EVAL * 4 / 6) / ln_f, (LN_BYTES_PER_EVAL * 2 / 6) / ln_w
    # Per STEP means the step's kernels: every launch of a step is one of the library's (torch kernels inside a step: 0.1 ms,
    # profiles/round5/per_step_torch_kernels_r5v.txt).  The torch / runtime kernels of the run are its ONE-TIME setup — synthetic
    # weights cast and packed into fp16 + e4m3 planes, 241 tensors — and were divided over the run's steps until round 5.
    step_kernel = is_library_kernel
    tf_all, tw_all = sum(fetch.values()) * 1024 / evals, sum(write.values()) * 1024 / evals
    tf = sum(v for k, v in fetch.items() if step_kernel(k)) * 1024 / evals
    tw = sum(v for k, v in write.items() if step_kernel(k)) * 1024 / evals
    from panacea_amd import build as _build
    rec = {"build_stamp": _build.library_digest(), "command": cmd,
           "evaluations": evals, "fetch_factor": round(ff, 3), "write_factor": round(wf, 3),
           "fetch_GB_raw": round(tf / 1e9, 1), "write_GB_raw": round(tw / 1e9, 1),
           "fetch_GB_calibrated": round(tf * ff / 1e9, 1), "write_GB_calibrated": round(tw * wf / 1e9, 1),
           "traffic_GB_calibrated": round((tf * ff + tw * wf) / 1e9, 1),
           "kernels": "the library's launches (a step launches nothing else)",
           "setup_GB_whole_run": round(((tf_all - tf) * ff + (tw_all - tw) * wf) * evals / 1e9, 1),
           "traffic_GB_calibrated_rounds_2_to_4_definition": round((tf_all * ff + tw_all * wf) / 1e9, 1),
           "calibration": f"LayerNorm launches: {LN_BYTES_PER_EVAL / 1e9:.1f} GB known per evaluation vs counters "
                          f"{ln_f / 1e9:.2f} GB fetched / {ln_w / 1e9:.2f} GB written"}
    out = OUT
    out.parent.mkdir(parents=True, exist_ok=True)
    doc = json.loads(out.read_text()) if out.exists() else {"records": {}}
    doc["records"][prec] = rec
    out.write_text(json.dumps(doc, indent=1))
    print(json.dumps(rec, indent=1))
    for name, agg, fac in (("fetch", fetch, ff), ("write", write, wf)):
        lines = [f"{k[:100]:100s} {v * 1024 * fac / evals / 1e9:9.2f} GB/eval" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:50]]
        (out.parent / f"pmc_{prec}_{name}_by_kernel.txt").write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
