#!/bin/bash
# side benches on the final round-2 build: config 2 (1 frame), config 5 set-up on one GPU, hoisted invariants, first stage
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--cpu-baseline none --no-modes --no-kernel-breakdown"
timeout 300 python bench.py --steps 8 --warmup 2 --frames 1 $B > gpurun_out/r2w_frames1.json 2>/dev/null
timeout 300 python bench.py --steps 6 --warmup 2 --yaml-exact $B > gpurun_out/r2w_yaml_exact.json 2>/dev/null
timeout 300 python bench.py --steps 6 --warmup 2 --hoist $B > gpurun_out/r2w_hoist.json 2>/dev/null
timeout 300 python bench.py --stage vae-decode --steps 5 --warmup 2 > gpurun_out/r2w_vae_decode.json 2>/dev/null
timeout 300 python bench.py --stage vae-encode --steps 5 --warmup 2 > gpurun_out/r2w_vae_encode.json 2>/dev/null
for f in frames1 yaml_exact hoist vae_decode vae_encode; do python -c "
import json,sys; d=json.loads(open('gpurun_out/r2w_$f.json').read().strip().splitlines()[-1]); print('$f', d['metric'], round(d['value'],3), d['unit'], round(d['ms_per_step'],2), 'ms', (d.get('roofline') or {}).get('frac'))"; done
