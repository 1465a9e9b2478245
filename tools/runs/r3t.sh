#!/bin/bash
# GEGLU gate: conflict-free piecewise-cubic Phi table (512 B) instead of the 2048-entry linear one: tests (bit-equality with the staged path), per-shape and whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 -k "geglu or grouped or tail_row" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.log
PREV=$GRAFT_REPO_ROOT/panacea_amd/lib/libpanacea_hip_prev.so
for r in 1 2; do
  echo "== new $r"; timeout 200 python tools/kbench.py ff1-geglu 2>&1 | grep "ff1-geglu"
  echo "== prev $r"; PANACEA_HIP_LIB=$PREV timeout 200 python tools/kbench.py ff1-geglu 2>&1 | grep "ff1-geglu"
done | tee $O/kbench_geglu_ab.log
B="--steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown --no-parity"
for r in 1 2; do
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench_new_$r.json
  PANACEA_HIP_LIB=$PREV timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench_prev_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3t/bench_*.json')):
    print(f.split('/')[-1], round(json.loads(open(f).read())['ms_per_step'],2))
PY
