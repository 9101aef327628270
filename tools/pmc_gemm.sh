# PMC passes for the big SYRK GEMM of a C4 fit (known-good counter groups only, each under a timeout)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/fit_trace.py 8192 20 float32 > /dev/null 2>&1; echo "p1 rc $?"
timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/fit_trace.py 8192 20 float32 > /dev/null 2>&1; echo "p2 rc $?"
timeout 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p3 -- python3 $R/tools/fit_trace.py 8192 20 float32 > /dev/null 2>&1; echo "p3 rc $?"
python3 - <<PY
import csv,glob,collections
for p in ["p1","p2","p3"]:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/"+p+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n=row["Kernel_Name"]
            if "gemm128_kernel<float, 128, true, true>" in n and int(row["Grid_Size_Y"])>=56:
                acc["syrk_big"][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "gemm128_kernel<float, 128, false, false>" in n:
                acc["kinv"][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,cs in acc.items():
        for c,v in sorted(cs.items()): print(p,k,c,f"{sum(v)/len(v):.5g}",len(v))
PY
