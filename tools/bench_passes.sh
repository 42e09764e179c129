cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --host-witness --witness-scatter > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err &&
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --lookups --outer > gpurun_out/bench_r2b_full.json 2> gpurun_out/bench_r2b_full.err &&
COZK_MSM_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o b -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
python tools/db_summary.py gpurun_out/prof_bench/b_results.db 40 > gpurun_out/kernel_stats_r2b.txt
cat gpurun_out/bench_r2b.json | cut -c1-600; cat gpurun_out/bench_r2b_full.json | cut -c1-1200
