# round 3 measurement pass (run on the GPU box): default bench line (full_step + cpu_baseline), north_star's target size,
# kernel-trace summaries of the bench command and of the chained flow, the flow / Spartan / lookups timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3p &&
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --host-witness > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof_bench -o b -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/r3p/prof_bench.log 2>&1 &&
COZK_MSM_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof_bench_serial -o b -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/r3p/prof_bench_serial.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof_flow -o f -- python3 tools/run_flow.py --log-n 20 --steps 1 > gpurun_out/r3p/prof_flow.log 2>&1 &&
timeout -k 10 300 python tools/run_flow.py --log-n 20 --steps 3 > gpurun_out/r3p/flow_plain_2p20.json 2> gpurun_out/r3p/flow.err &&
timeout -k 10 300 python tools/run_outer.py --log-steps 20 --steps 3 > gpurun_out/r3p/spartan_plain_2p20.json 2> gpurun_out/r3p/spartan.err &&
timeout -k 10 400 python tools/run_flow.py --log-n 16 --mode rep3 --steps 2 > gpurun_out/r3p/flow_rep3_2p16.json 2> gpurun_out/r3p/flow_rep3.err
python tools/db_summary.py gpurun_out/r3p/prof_bench/b_results.db 40 > gpurun_out/r3p/kernel_stats_bench.txt
python tools/db_summary.py gpurun_out/r3p/prof_bench_serial/b_results.db 40 > gpurun_out/r3p/kernel_stats_bench_serial.txt
python tools/db_summary.py gpurun_out/r3p/prof_flow/f_results.db 50 > gpurun_out/r3p/kernel_stats_flow.txt
rm -rf gpurun_out/r3p/prof_bench gpurun_out/r3p/prof_bench_serial gpurun_out/r3p/prof_flow
python -c "
import json
d=json.load(open('gpurun_out/r3p/bench.json')); print(d['value'], d['ms_per_step'], d['phases_ms_per_step']); print(d.get('full_step')); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['commit_fr_scalar_muls_per_s_per_core_lower_bound'])
"
cat gpurun_out/r3p/flow_plain_2p20.json gpurun_out/r3p/spartan_plain_2p20.json gpurun_out/r3p/flow_rep3_2p16.json | cut -c1-900
head -30 gpurun_out/r3p/kernel_stats_bench.txt
