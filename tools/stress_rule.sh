#!/bin/bash
# Second runs of the device decoder on the stress streams under several rules for the range length (TIC_DECODE_RULE = "<average blocks>,<least words>")
export TIC_TEST_HOOKS=1
for rule in "$@"; do
  export TIC_DECODE_RULE=$rule
  timeout -k 10 400 python tools/stress_decoder.py ${TIC_STRESS_DEC:-300} > gpurun_out/stress_rule_$rule.txt 2>&1 || { echo "rule $rule: stress run failed"; tail -3 gpurun_out/stress_rule_$rule.txt; exit 1; }
  echo "rule $rule: $(tail -1 gpurun_out/stress_rule_$rule.txt)"
  grep "second run" gpurun_out/stress_rule_$rule.txt | grep "damage [03]" | sed 's/^/     /'
done
