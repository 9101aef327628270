R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "p1 rc $?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/p2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "p2 rc $?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $O/p3 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "p3 rc $?"
python3 - <<PY
import csv,glob,collections
for p in ["p1","p2","p3"]:
    acc=collections.defaultdict(list)
    for f in glob.glob("$O/"+p+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "leaf_tiles_v2_kernel<float, float, 256, 2, 0>" in row["Kernel_Name"] and int(row["Grid_Size"])>=512*256*8:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c,v in sorted(acc.items()): print(p,c,f"{sum(v)/len(v):.5g}",len(v))
PY
