#!/bin/bash
# first fit + predict of K fresh processes per configuration: one hash each (tools/first_fit_hash.py); a second hash
# in a configuration is a first-launch hazard (profiles/r02h_packed_mean_bug.txt).  Run on the GPU box.
K=${1:-25}
for cfg in "2048 12 float32 auto" "2048 12 mixed auto" "2048 12 mixed f16x3" "2048 12 mixed bf16x3" "2048 12 mixed bf16x6" "512 3 float32 auto" "256 6 mixed f16x3" "4096 6 float32 auto"; do
  for i in $(seq 1 $K); do python tools/first_fit_hash.py $cfg 2>/dev/null | tail -n 1; done | sort | uniq -c | awk -v c="$cfg" '{printf "%-28s %s x %s\n", c, $2, $1}'
done
