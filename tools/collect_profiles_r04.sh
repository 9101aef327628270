# Round-4 evidence bundle on the final code (run on the GPU box: bash tools/collect_profiles_r04.sh [TAG] [a|b|all]).
# Outputs under gpurun_out/TAG; the files to keep are copied into profiles/ (see profiles/README.md).
TAG=${1:-r04}
PART=${2:-all}   # a = bench lines + kernel stats + PMC, b = fit timelines, loop bench, host overhead, fuzz
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
KPAT="leaf_tiles_bf16_kernel<2, float, 0, true, true, 1>"   # the fp16 split instantiation (first rung of GPSO_MATH_AUTO)
if [ $PART != b ]; then
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c2 c4 c5 c3f64; do python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
for m in native bf16x6; do python3 $R/bench.py --math $m --no-cpu-baseline > $O/bench_c3_$m.json 2>/dev/null; done
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_c3_kernel_stats.csv
python3 $R/tools/rocprof_fullsize.py $O/prof "$KPAT" $O/bench_c3_rocprof_fullsize.json
rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "pmc1 rc $?"
rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "pmc2 rc $?"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "pmc3 rc $?"
for pat in "$KPAT" "leaf_tiles_bf16_kernel<3" "leaf_tiles_v2_kernel" potrf_step gram_kernel; do echo "## $pat (largest-grid dispatches only)"; for p in pmc1 pmc2 pmc3; do python3 $R/tools/pmc_summary.py $O/$p "$pat"; done; done > $O/pmc_summary.txt 2>&1
python3 $R/tools/pmc_traffic_json.py $O/pmc_leaf_tiles_f16x3_c3.json c3 "$KPAT" "leaf_tiles_bf16_kernel<2, float, 0, true, true, 1> (fp16 split x3, fused step, fp16 contraction: first rung of GPSO_MATH_AUTO)" $O/pmc1 $O/pmc2 $O/pmc3 > /dev/null
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
echo "bench profile done"
rocprofv3 --kernel-trace --output-format csv -d $O/steptrace -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/step_timeline.py $O/steptrace > $O/step_timeline.txt 2>&1; rm -rf $O/steptrace
python3 $R/tools/c16_check.py > $O/c16_check.jsonl 2>/dev/null
python3 $R/tools/sweep.py 2>/dev/null | grep "^{" > $O/sweep.jsonl
echo "timeline, contraction check, sweep done"
fi
if [ $PART = a ]; then exit 0; fi
for cfg in "2048 12 c3" "8192 20 c4" "16384 40 c5"; do
  set -- $cfg
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_posterior $1 $2 > /dev/null 2>&1
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_grad $1 $2 grad > /dev/null 2>&1
  echo "fit timeline $3 done"
done
for a in "2 5 50" "4 7 80" "6 9 60"; do set -- $a; python3 $R/tools/loop_bench.py --dim $1 --depth $2 --budget $3; done > $O/loop_bench.jsonl 2>/dev/null
python3 $R/tools/host_overhead.py 52 2 2>/dev/null | grep -E "wall|device" > $O/host_overhead.txt
for s in 41 42 43; do FUZZ_CASES=80 FUZZ_SEED=$s python3 $R/tools/fuzz_gpu.py > $O/fuzz_seed$s.log 2>&1; tail -1 $O/fuzz_seed$s.log; done
(cd $R && python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -q -m gpu -s 2>/dev/null | grep "|d mean|" > $O/float_errors_raw.txt)
ls -la $O
