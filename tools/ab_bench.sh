# A/B of two builds of the library through bench.py itself (same box, alternating fresh processes):
#   bash tools/ab_bench.sh OUT_FILE ROUNDS LIB_A LIB_B -- <bench.py arguments>
OUT=$1; ROUNDS=$2; A=$3; B=$4; shift 5
: > $OUT
for r in $(seq $ROUNDS); do for L in $A $B; do
  GPSO_HIP_LIB=$PWD/$L GPSO_HIP_LIB_OLDER=1 python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$L', 'value %.5g' % d['value'], 'ms_per_step %.5g' % d['ms_per_step'], 'kernel_ms %.5g' % r['kernel_ms'], 'frac %.4f' % r['frac'])" >> $OUT
done; done
cat $OUT
