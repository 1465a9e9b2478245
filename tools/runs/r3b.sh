#!/bin/bash
# round 3, second GPU pass: whole GPU suite on the hint-dedupe / module-precision / async-exchange tree, default bench line,
# per-shape GEMM profile in both operand policies
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/test_measurements.log
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -25 > $O/gpu_all.log
tail -4 $O/gpu_all.log
cp gpurun_out/test_measurements.log $O/ 2>/dev/null
timeout 500 python bench.py --steps 8 --warmup 2 --cpu-baseline none > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3b/bench_default.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 2), d["parity"]["eps_max_abs_err"], {k: round(v["ms_per_step"], 2) for k, v in d.get("modes", {}).items() if isinstance(v, dict)})
print({k: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
timeout 300 python tools/shape_profile.py precise > $O/shape_precise.log 2>&1
timeout 300 python tools/shape_profile.py fast > $O/shape_fast.log 2>&1
head -3 $O/shape_precise.log; head -3 $O/shape_fast.log
