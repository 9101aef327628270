# Round-2 evidence bundle (run on the GPU box: bash tools/collect_profiles_r02.sh TAG): bench lines of every
# workload, kernel stats + PMC of the default bench, fit timelines at C3 / C4 / C5, the caller-level loop
# bench, the small-fit phase stamps and the generation probe.  Outputs under gpurun_out/TAG; the files to
# keep are copied into profiles/ by hand (see profiles/README.md).
TAG=${1:-r02e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c2 c4 c5 c3f64; do python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_c3_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc1 rc $?"
rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc2 rc $?"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc3 rc $?"
for pat in leaf_tiles_v2 potrf_step gram_kernel; do echo "## $pat (largest-grid dispatches only)"; for p in pmc1 pmc2 pmc3; do python3 $R/tools/pmc_summary.py $O/$p $pat; done; done > $O/pmc_summary.txt 2>&1
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
echo "bench profile done"
for cfg in "2048 12 c3" "8192 20 c4" "16384 40 c5"; do
  set -- $cfg
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_posterior $1 $2 > /dev/null 2>&1
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_grad $1 $2 grad > /dev/null 2>&1
  echo "fit timeline $3 done"
done
for a in "2 5 50" "4 7 80" "6 9 60"; do set -- $a; python3 $R/tools/loop_bench.py --dim $1 --depth $2 --budget $3; done > $O/loop_bench.jsonl 2>/dev/null
python3 $R/tools/host_overhead.py 52 2 2>/dev/null | head -1 > $O/host_overhead.txt
python3 $R/tools/host_overhead.py 100 6 2>/dev/null | head -1 >> $O/host_overhead.txt
if [ -x $R/tools/micro/small_phases.bin ]; then for a in "12 2" "52 2" "100 6"; do $R/tools/micro/small_phases.bin $a | tail -1; done > $O/small_fit_phases.txt; fi
python3 $R/tools/gen_probe.py > $O/gen_probe.jsonl 2>/dev/null
python3 $R/tools/explore_overhead.py > $O/explore_overhead.txt 2>/dev/null
for s in 11 12 13; do FUZZ_CASES=80 FUZZ_SEED=$s python3 $R/tools/fuzz_gpu.py > $O/fuzz_seed$s.log 2>&1; tail -1 $O/fuzz_seed$s.log; done
if [ -x $R/tools/micro/overlap_probe.bin ]; then timeout -k 5 120 $R/tools/micro/overlap_probe.bin 1024 6144 > $O/overlap_probe.txt 2>&1; fi
bash $R/tools/pmc_sq.sh > $O/pmc_sq_leaf_tiles.txt 2>&1
ls -la $O
