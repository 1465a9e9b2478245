#!/bin/bash
# attention: K / V fragments requested per tile up front, max exchange by v_permlane32_swap instead of ds_bpermute: tests, per-shape and
# whole-step A/B against the previous build (prev = the shipped library of r3m)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout=600 -k "attn or attention or golden or tiny or plain" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.log
PREV=$GRAFT_REPO_ROOT/panacea_amd/lib/libpanacea_hip_prev.so
for r in 1 2; do
  echo "== new $r"; timeout 200 python tools/kbench.py attn 2>&1 | grep "attn"
  echo "== prev $r"; PANACEA_HIP_LIB=$PREV timeout 200 python tools/kbench.py attn 2>&1 | grep "attn"
done | tee $O/kbench_attn_ab.log
B="--steps 6 --warmup 2 --cpu-baseline none --no-modes --no-kernel-breakdown --no-parity"
for r in 1 2; do
  timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench_new_$r.json
  PANACEA_HIP_LIB=$PREV timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/bench_prev_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3o/bench_*.json')):
    print(f.split('/')[-1], round(json.loads(open(f).read())['ms_per_step'],2))
PY
