# The evidence bundle of a round on the final code -- ONE parameterised collector (rounds 1-4 had a script per round).
# Run on the GPU box: bash tools/collect_profiles.sh [TAG] [a|b|c|all]; outputs under gpurun_out/TAG, the files to keep are
# copied into profiles/ as TAG_* (see profiles/README.md).
#   a = bench lines + kernel stats + PMC passes + step timeline      b = fit timelines, loop bench, host overhead, fuzz, float errors
#   c = round-5 additions: append bench, fit-plane modes, vendor yardstick
TAG=${1:-r06}
PART=${2:-all}   # a = bench lines + kernel stats + PMC, b = fit timelines, loop bench, host overhead, fuzz
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
KPAT="leaf_tiles_bf16_kernel<2, float, 0, true, true, 1, false, 1>"   # the fp16 split instantiation at C3 (first rung of GPSO_MATH_AUTO; fused step, one chunk of the fp16 contraction, 16x16x32, packed contraction)
if [ $PART = a ] || [ $PART = all ]; then
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c2 c4 c5 c3f64; do python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
# (round 6) the reference's own dtype and "mixed" at the large shares; SURVEY 8(d) family A leaves (grown on the device)
for w in c4 c5; do for t in float64 mixed; do python3 $R/bench.py --workload $w --dtype $t --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_${w}_$t.json 2>/dev/null; done; done
for cfg in "c2 8" "c3 11" "c4 12"; do set -- $cfg; python3 $R/bench.py --workload $1 --leaves grow --depth $2 --steps 100 --no-hyperopt > $O/bench_grow_d$2.json 2>/dev/null; done
for sd in 1 2 3 4; do python3 $R/bench.py --seed $sd --no-cpu-baseline --no-hyperopt > $O/bench_c3_seed$sd.json 2>/dev/null; done
for m in native bf16x6; do python3 $R/bench.py --math $m --no-cpu-baseline > $O/bench_c3_$m.json 2>/dev/null; done
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_c3_kernel_stats.csv
python3 $R/tools/rocprof_fullsize.py $O/prof "$KPAT" $O/bench_c3_rocprof_fullsize.json
rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "pmc1 rc $?"
rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "pmc2 rc $?"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "pmc3 rc $?"
for pat in "$KPAT" "leaf_tiles_bf16_kernel<3" "leaf_tiles_v2_kernel" potrf_step gram_kernel; do echo "## $pat (largest-grid dispatches only)"; for p in pmc1 pmc2 pmc3; do python3 $R/tools/pmc_summary.py $O/$p "$pat"; done; done > $O/pmc_summary.txt 2>&1
python3 $R/tools/pmc_traffic_json.py $O/pmc_leaf_tiles_f16x3_c3.json c3 "$KPAT" "leaf_tiles_bf16_kernel<2, float, 0, true, true, 1, false, 1> (fp16 split x3, fused step, fp16 contraction with its small products packed: first rung of GPSO_MATH_AUTO)" $O/pmc1 $O/pmc2 $O/pmc3 > /dev/null
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
echo "bench profile done"
rocprofv3 --kernel-trace --output-format csv -d $O/steptrace -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/step_timeline.py $O/steptrace > $O/step_timeline.txt 2>&1; rm -rf $O/steptrace
python3 $R/tools/c16_check.py > $O/c16_check.jsonl 2>/dev/null
python3 $R/tools/sweep.py 2>/dev/null | grep "^{" > $O/sweep.jsonl
echo "timeline, contraction check, sweep done"
fi
if [ $PART = b ] || [ $PART = all ]; then
for cfg in "2048 12 c3" "8192 20 c4" "16384 40 c5"; do
  set -- $cfg
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_posterior $1 $2 > /dev/null 2>&1
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_grad $1 $2 grad > /dev/null 2>&1
  echo "fit timeline $3 done"
done
python3 $R/tools/fit_overlap_ab.py n4096 c4 c5 --modes 0 2 > $O/fit_overlap_ab.jsonl 2>/dev/null
python3 $R/tools/loop_large_n.py --refit-every 1 5 > $O/loop_large_n.jsonl 2>/dev/null
python3 $R/tools/loop_large_n.py --refit-every 1 5 --dtype float32 >> $O/loop_large_n.jsonl 2>/dev/null
for a in "2 5 50" "4 7 80" "6 9 60"; do set -- $a; python3 $R/tools/loop_bench.py --dim $1 --depth $2 --budget $3; done > $O/loop_bench.jsonl 2>/dev/null
python3 $R/tools/host_overhead.py 52 2 2>/dev/null | grep -E "wall|device" > $O/host_overhead.txt
for s in 41 42 43; do FUZZ_CASES=80 FUZZ_SEED=$s python3 $R/tools/fuzz_gpu.py > $O/fuzz_seed$s.log 2>&1; tail -1 $O/fuzz_seed$s.log; done
(cd $R && python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_precision.py tests/test_gpu_append.py -q -m gpu -s 2>/dev/null | grep -E "\|d mean\||float winner rule|fit planes mode" > $O/float_errors_raw.txt)
fi
if [ $PART = c ] || [ $PART = all ]; then
python3 $R/tools/append_bench.py c3 c4 c5 > $O/append_bench.jsonl 2>/dev/null
python3 $R/tools/append_bench.py c3 c4 --dtype float64 --k 1 7 >> $O/append_bench.jsonl 2>/dev/null
python3 $R/tools/fit_bench.py n4096 c4 c5 --modes 2 1 --oracle > $O/fit_bench.jsonl 2>/dev/null
python3 $R/tools/vendor_yardstick.py c3 c4 c5 > $O/vendor_yardstick.json 2>/dev/null
echo "append / fit modes / yardstick done"
fi
ls -la $O
