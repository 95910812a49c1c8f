#!/bin/bash
# Runs GPU steps one after the other from the repo root (through gpurun); a step that times out or is killed stops the whole call
# (no further GPU step after a hang), a step that merely fails does not.   tools/gpu_steps.sh TAG "cmd1" "cmd2" ...
# Each step: timeout -k 10 <limit> bash -c "<cmd>"; limit from STEP_TIMEOUT (default 900 s).  Output of step i -> gpurun_out/TAG_i.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$1; shift
mkdir -p "$R/gpurun_out"
i=0
for c in "$@"; do
  i=$((i+1))
  echo "=== step $i: $c" | tee -a "$R/gpurun_out/${T}_steps.txt"
  s=$(date +%s)
  timeout -k 10 ${STEP_TIMEOUT:-900} bash -c "$c" > "$R/gpurun_out/${T}_$i.log" 2>&1
  rc=$?
  echo "    rc=$rc  $(( $(date +%s) - s )) s" | tee -a "$R/gpurun_out/${T}_steps.txt"
  tail -3 "$R/gpurun_out/${T}_$i.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $i timed out / was killed: stopping"; exit 1; fi
done
exit 0
