TAG=r04; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "2048 12 c3" "8192 20 c4" "16384 40 c5"; do
  set -- $cfg
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_posterior $1 $2 > /dev/null 2>&1
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_grad $1 $2 grad > /dev/null 2>&1
  echo "fit timeline $3 done"
done
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c4 c5; do python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
for a in "2 5 50" "4 7 80" "6 9 60"; do set -- $a; python3 $R/tools/loop_bench.py --dim $1 --depth $2 --budget $3; done > $O/loop_bench.jsonl 2>/dev/null
echo done
