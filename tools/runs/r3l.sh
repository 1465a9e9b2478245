#!/bin/bash
# register epilogues (GEGLU on the accumulators; fp16-only outputs): kernel tests, per-shape and whole-step A/B against the previous build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=300 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests.log
PREV=$GRAFT_REPO_ROOT/panacea_amd/lib/libpanacea_hip_prev.so
for r in 1 2; do
  echo "== new $r"; timeout 200 python tools/kbench.py gemm 2>&1 | grep "qkv\|ff1-geglu"
  echo "== prev $r"; PANACEA_HIP_LIB=$PREV timeout 200 python tools/kbench.py gemm 2>&1 | grep "qkv\|ff1-geglu"
done | tee $O/kbench_ab.log
B="--steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown --no-parity"
for r in 1 2; do
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench_new_$r.json
  PANACEA_HIP_LIB=$PREV timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench_prev_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l/bench_*.json')):
    print(f.split('/')[-1], round(json.loads(open(f).read())['ms_per_step'],2))
PY
