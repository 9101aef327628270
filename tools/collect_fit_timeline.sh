# kernel-trace timeline of the posterior fit (and of the NLML+gradient evaluation) at one size.
# usage (on the GPU box): bash tools/collect_fit_timeline.sh TAG N D [grad]
TAG=$1; N=$2; D=$3; G=${4:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/tools/fit_trace.py $N $D float32 $G > $O/fit_under_rocprof.txt 2>&1
cp $(ls $O/ks/*/*kernel_stats.csv | head -1) $O/fit_kernel_stats.csv
python3 $R/tools/fit_trace.py analyse $(ls $O/ks/*/*kernel_trace.csv | head -1) > $O/fit_timeline.txt 2>&1
rm -rf $O/ks
cat $O/fit_timeline.txt
