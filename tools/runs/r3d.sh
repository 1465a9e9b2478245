#!/bin/bash
# round 3, fourth GPU pass: the fused feed-forward launch — kernel tests, level-0 timing, whole-network tests, whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ff_chain_gpu.py -q --timeout=120 -s 2>&1 | grep -v amdgpu.ids | tail -25 > $O/ffchain_tests.log
tail -12 $O/ffchain_tests.log
if ! grep -q "passed" $O/ffchain_tests.log || grep -q "failed" $O/ffchain_tests.log; then echo "ff_chain tests did not pass: stopping"; exit 0; fi
timeout 200 python tools/runs/r3d_ffchain.py 2>&1 | grep -v amdgpu.ids | tee $O/ffchain_kbench.log
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_modules_vs_oracle.py -q --timeout=600 -x 2>&1 | tail -8 > $O/model_tests.log
tail -4 $O/model_tests.log
B="python bench.py --steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown"
for rep in 1 2; do
  timeout 300 $B > $O/bench_chain_$rep.json 2> $O/bench_chain_$rep.err
  timeout 300 $B --no-ff-chain > $O/bench_nochain_$rep.json 2> $O/bench_nochain_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3d/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 2), d["parity"]["eps_max_abs_err"], [round(q["eps_max_abs_err"] * 1e4, 2) for q in d["parity"]["pins"]])
    except Exception as e:
        print(f, "failed", e)
PY
