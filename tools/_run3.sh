mkdir -p gpurun_out/r02c
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r02c/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02c/pytest_gpu.log
grep -E "passed|failed|FAILED|rc " gpurun_out/r02c/pytest_gpu.log | tail -40
for s in 1 2 3; do FUZZ_SEED=$s FUZZ_CASES=80 python3 tools/fuzz_gpu.py > gpurun_out/r02c/fuzz_$s.log 2>&1; tail -1 gpurun_out/r02c/fuzz_$s.log; grep "^BAD" gpurun_out/r02c/fuzz_$s.log | cut -c1-400 | head -6; done
python3 bench.py --steps 10 --warmup 3 > gpurun_out/r02c/bench_c3.json 2> gpurun_out/r02c/bench_c3.err; tail -c 2500 gpurun_out/r02c/bench_c3.json; tail -3 gpurun_out/r02c/bench_c3.err
