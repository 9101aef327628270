# LDS bank-conflict share per kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, separate --pmc pass): fit with gradient at two
# sizes and the bench's predict step.  Run on the GPU box: bash tools/lds_conflicts.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/c3 -- python3 $R/tools/fit_trace.py 2048 12 float32 grad > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/c4 -- python3 $R/tools/fit_trace.py 8192 40 float32 grad > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/b3 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<P
import csv, glob, collections
for tag in ("c3", "c4", "b3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void gpso::", "")[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", tag)
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
        a, b = c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_BANK_CONFLICT", 0)
        if a > 0: print(f"  {k:62s} lds cycles {a:12.4g}  conflict share {b / a:5.2f}")
P
rm -rf $O
