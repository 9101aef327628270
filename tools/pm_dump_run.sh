#!/bin/bash
# packed-mean hunt (profiles/r02h_packed_mean_bug.txt).  The failure depends on the box: first a CONTROL -- N fresh
# processes of the packed build in its original protocol -- and only on a box that shows it the dump build
# (tools/micro/packed_mean_probe.hip -DGPSO_PROBE_PACKED_MEAN -DGPSO_PROBE_DUMP_MACC): which lanes / links of the
# first launch differ from a warm launch.
# (round 4: git apply tools/attic/predict_hooks.patch first -- the GPSO_PROBE_* hooks left predict.hip)
N=${1:-40}
M=${2:-200}
mkdir -p gpurun_out
OUT=gpurun_out/pm_dump.log
CTL=gpurun_out/pm_control.log
: > $OUT; : > $CTL
for i in $(seq 1 $N); do
  timeout -k 5 60 tools/micro/packed_mean_probe_packed.bin 3 d >> $CTL 2>&1 || echo "exit $?" >> $CTL
done
# the minimal probe (tools/micro/pk_hazard_probe.hip): dependent v_pk_fma_f32 chains beside a partner wave's MFMAs
timeout -k 5 120 tools/micro/pk_hazard_probe.bin 2000 256 3 > gpurun_out/pk_hazard.log 2>&1 || echo "exit $?" >> gpurun_out/pk_hazard.log
cat gpurun_out/pk_hazard.log
BAD=$(grep -c DIFFERS $CTL)
echo "control: processes whose first launch differs: $BAD of $N"
if [ "$BAD" -eq 0 ]; then echo "this box does not show the failure"; exit 0; fi
for i in $(seq 1 $M); do
  echo "== process $i" >> $OUT
  timeout -k 5 60 tools/micro/packed_mean_probe_dump.bin dump >> $OUT 2>&1 || echo "exit $?" >> $OUT
done
grep "dump:" $OUT | sort | uniq -c
