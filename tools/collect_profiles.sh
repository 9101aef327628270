R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c2 c4 c5 c3f64; do python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_c3_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for pat in leaf_tiles_v2 potrf_step gemm128 gram_kernel; do echo "## $pat"; for p in pmc1 pmc2 pmc3; do python3 $R/tools/pmc_summary.py $O/$p $pat; done; done > $O/pmc_summary.txt 2>&1
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
tail -c 600 $O/bench_c3.json; echo; for w in c2 c4 c5 c3f64; do python3 -c "
import json,sys
d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
print('$w', d['value'], d['ms_per_step'], d['fit_ms'], d['roofline']['frac'], d['roofline']['kernel_ms'])"; done
