# the round's measurement pass (run on the GPU box): default bench line (+ host-witness / witness-scatter legs), the full step
# with the 8(f)1 / 8(f)2 phases, the kernel-trace summary with serial MSM phases, the lookups pipeline timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --host-witness --witness-scatter > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err &&
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --lookups --outer > gpurun_out/bench_r2d_full.json 2> gpurun_out/bench_r2d_full.err &&
COZK_MSM_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o b -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 &&
timeout -k 10 300 python tools/run_lookups.py --mode plain --log-n 20 --primary > gpurun_out/lk_plain.json 2> gpurun_out/lk_plain.err &&
timeout -k 10 400 python tools/run_lookups.py --mode rep3 --log-n 20 --primary > gpurun_out/lk_rep3.json 2> gpurun_out/lk_rep3.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lk -o lk -- python3 tools/run_lookups.py --mode plain --log-n 20 --primary --steps 1 > gpurun_out/lk_prof.log 2>&1
python tools/db_summary.py gpurun_out/prof_bench/b_results.db 40 > gpurun_out/kernel_stats_r2d.txt
python tools/db_summary.py gpurun_out/prof_lk/lk_results.db 40 > gpurun_out/kernel_stats_lk_r2d.txt
python -c "
import json
for f in ('bench_r2d','bench_r2d_full'):
    d=json.load(open('gpurun_out/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['phases_ms_per_step'])
"
cat gpurun_out/lk_plain.json gpurun_out/lk_rep3.json | cut -c1-700
