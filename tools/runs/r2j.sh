#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "oracle or yaml or config1" -s 2>&1 | grep -E "ResBlock3D|STT C|config-5|plain 64|passed|failed|Error|error" | tail -30 > gpurun_out/r2j_pytest.log
cat gpurun_out/r2j_pytest.log
