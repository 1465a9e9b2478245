#!/bin/bash
# round 3, first GPU pass: the e4m3 lo pass (kernel tests, whole suite, A/B of the two lo formats in bench.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lo8_gpu.py -q --timeout=600 2>&1 | tail -40 > $O/lo8.log
tail -5 $O/lo8.log
timeout 900 python -m pytest tests -m gpu -q --timeout=900 -x --deselect tests/test_lo8_gpu.py 2>&1 | tail -25 > $O/gpu_all.log
tail -4 $O/gpu_all.log
timeout 500 python bench.py --steps 8 --warmup 2 --cpu-baseline none > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --steps 8 --warmup 2 --cpu-baseline none --precision precise-f16lo --no-modes --no-kernel-breakdown > $O/bench_f16lo.json 2> $O/bench_f16lo.err
python - <<'PY'
import json
for n in ("default", "f16lo"):
    try:
        d = json.loads(open(f"gpurun_out/r3a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"], 2), d["parity"]["eps_max_abs_err"], [round(q["eps_max_abs_err"] * 1e4, 2) for q in d["parity"]["pins"]],
              {k: round(v["ms_per_step"], 2) for k, v in d.get("modes", {}).items() if isinstance(v, dict)})
        if "kernels" in d.get("roofline", {}):
            print({k: v["ms"] for k, v in d["roofline"]["kernels"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
