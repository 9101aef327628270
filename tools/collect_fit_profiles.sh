# Round 2: kernel stats + PMC of the FIT kernels (potrf_step_kernel, gemm128_kernel) at the C4 / C5 shapes.
# usage (on the GPU box): bash tools/collect_fit_profiles.sh r02a
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "8192 20 c4" "16384 40 c5"; do
  set -- $cfg; N=$1; D=$2; W=$3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$W -- python3 $R/tools/fit_trace.py $N $D float32 grad > $O/fit_${W}_under_rocprof.txt 2>&1
  cp $(ls $O/ks_$W/*/*kernel_stats.csv | head -1) $O/fit_${W}_kernel_stats.csv
  python3 $R/tools/fit_trace.py analyse $(ls $O/ks_$W/*/*kernel_trace.csv | head -1) > $O/fit_${W}_timeline.txt 2>&1
  rm -rf $O/ks_$W
  timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/fit_trace.py $N $D float32 > /dev/null 2>&1; echo "$W p1 rc $?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/fit_trace.py $N $D float32 > /dev/null 2>&1; echo "$W p2 rc $?"
  timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p3 -- python3 $R/tools/fit_trace.py $N $D float32 > /dev/null 2>&1; echo "$W p3 rc $?"
  for pat in potrf_step gemm128 gram_kernel; do echo "## $W $pat"; for p in p1 p2 p3; do python3 $R/tools/pmc_summary.py $O/$p $pat; done; done > $O/fit_${W}_pmc_summary.txt 2>&1
  rm -rf $O/p1 $O/p2 $O/p3
done
ls -la $O
