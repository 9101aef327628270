# Round-2 (late) evidence bundle, after the split-bf16 fit path, the look-ahead and GPSO_MATH_AUTO (r02g), and again
# on the final code of the round (r02h: float pivot chain, pipelined panels)
# (run on the GPU box: bash tools/collect_profiles_r02g.sh [TAG]).  Outputs under gpurun_out/TAG; the files to
# keep are copied into profiles/ by hand (see profiles/README.md).
TAG=${1:-r02g}
PART=${2:-all}   # a = bench lines + kernel stats + PMC, b = fit timelines, loop bench, micro tools, fuzz
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ $PART != b ]; then
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c2 c4 c5 c3f64; do python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
python3 $R/bench.py --math native --no-cpu-baseline > $O/bench_c3_native.json 2>/dev/null
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_c3_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc1 rc $?"
rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc2 rc $?"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "pmc3 rc $?"
for pat in "leaf_tiles_bf16_kernel<3" leaf_tiles_v2 potrf_step gram_kernel; do echo "## $pat (largest-grid dispatches only)"; for p in pmc1 pmc2 pmc3; do python3 $R/tools/pmc_summary.py $O/$p "$pat"; done; done > $O/pmc_summary.txt 2>&1
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
echo "bench profile done"
fi
if [ $PART = a ]; then exit 0; fi
for cfg in "2048 12 c3" "8192 20 c4" "16384 40 c5"; do
  set -- $cfg
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_posterior $1 $2 > /dev/null 2>&1
  bash $R/tools/collect_fit_timeline.sh $TAG/fit_$3_grad $1 $2 grad > /dev/null 2>&1
  echo "fit timeline $3 done"
done
# PMC of the fit's split-bf16 GEMM at C5 (one pass: matrix-pipe occupancy)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcf -- python3 $R/tools/fit_trace.py 16384 40 float32 > /dev/null 2>&1; echo "pmcf rc $?"
{ echo "## gemm_bf16_kernel at the C5 fit (largest-grid dispatches only)"; python3 $R/tools/pmc_summary.py $O/pmcf gemm_bf16_kernel; } > $O/pmc_fit_c5_gemm_bf16.txt 2>&1
rm -rf $O/pmcf
for a in "2 5 50" "4 7 80" "6 9 60"; do set -- $a; python3 $R/tools/loop_bench.py --dim $1 --depth $2 --budget $3; done > $O/loop_bench.jsonl 2>/dev/null
for m in "6144 1024" "14336 1024"; do timeout -k 5 60 $R/tools/micro/syrk_bench.bin $m | tail -1; done > $O/syrk_bench.txt 2>&1
timeout -k 5 60 $R/tools/micro/syrk_bench_stamps.bin 14336 1024 | tail -8 > $O/syrk_bench_tile_stamps.txt 2>&1
timeout -k 5 120 $R/tools/micro/lookahead_probe.bin > $O/lookahead_probe.txt 2>&1
timeout -k 5 100 $R/tools/micro/leaf_bf16_phases.bin | tail -3 > $O/leaf_bf16_phases.txt 2>&1
timeout -k 5 100 $R/tools/micro/diag_phases.bin 1024 | tail -1 > $O/diag_phases.txt 2>&1
timeout -k 5 60 $R/tools/micro/lat_probe.bin > $O/lat_probe.txt 2>&1
{ timeout -k 5 60 $R/tools/micro/small_phases.bin 52 2 | tail -1; timeout -k 5 60 $R/tools/micro/small_phases.bin 100 2 | tail -1; } > $O/small_fit_phases.txt 2>&1
python3 $R/tools/host_overhead.py 2>/dev/null | grep -E "wall|device" > $O/host_overhead.txt
for s in 21 22 23; do FUZZ_CASES=80 FUZZ_SEED=$s python3 $R/tools/fuzz_gpu.py > $O/fuzz_seed$s.log 2>&1; tail -1 $O/fuzz_seed$s.log; done
ls -la $O
