# round 3 measurement pass (run on the GPU box): default bench line (full_step + cpu_baseline), north_star's target size,
# kernel-trace summaries of the bench command and of the chained flow, the flow / Spartan / lookups timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3z &&
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --host-witness > gpurun_out/r3z/bench.json 2> gpurun_out/r3z/bench.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3z/prof_bench -o b -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/r3z/prof_bench.log 2>&1 &&
COZK_MSM_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3z/prof_bench_serial -o b -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/r3z/prof_bench_serial.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3z/prof_flow -o f -- python3 tools/run_flow.py --log-n 20 --steps 1 > gpurun_out/r3z/prof_flow.log 2>&1 &&
timeout -k 10 300 python tools/run_flow.py --log-n 20 --steps 3 > gpurun_out/r3z/flow_plain_2p20.json 2> gpurun_out/r3z/flow.err &&
timeout -k 10 300 python tools/run_outer.py --log-steps 20 --steps 3 > gpurun_out/r3z/spartan_plain_2p20.json 2> gpurun_out/r3z/spartan.err &&
timeout -k 10 400 python tools/run_flow.py --log-n 16 --mode rep3 --steps 2 > gpurun_out/r3z/flow_rep3_2p16.json 2> gpurun_out/r3z/flow_rep3.err &&
timeout -k 10 400 python tools/run_flow.py --log-n 18 --mode rep3 --steps 2 > gpurun_out/r3z/flow_rep3_2p18.json 2> gpurun_out/r3z/flow_rep3_18.err &&
timeout -k 10 400 python bench.py --log-n 22 --steps 3 --warmup 1 --no-full-step > gpurun_out/r3z/bench_2p22.json 2> gpurun_out/r3z/bench_2p22.err &&
timeout -k 10 600 python tools/run_flow.py --log-n 22 --steps 1 > gpurun_out/r3z/flow_plain_2p22.json 2> gpurun_out/r3z/flow_22.err &&
timeout -k 10 300 python tools/run_spartan_lookup.py --log-n 18 --pub-workers 0 1 2 > gpurun_out/r3z/spartan_lookup_pub.json 2> gpurun_out/r3z/spartan_lookup_pub.err &&
timeout -k 10 120 python tools/probe_layer.py 24 > gpurun_out/r3z/probe_layer.log 2>&1 &&
for m in uniform sha2; do timeout -k 10 200 python tools/run_lookups.py --log-n 20 --primary --mix $m --steps 3 > gpurun_out/r3z/lookups_$m.json 2>/dev/null; done
python tools/db_summary.py gpurun_out/r3z/prof_bench/b_results.db 40 > gpurun_out/r3z/kernel_stats_bench.txt
python tools/db_summary.py gpurun_out/r3z/prof_bench_serial/b_results.db 40 > gpurun_out/r3z/kernel_stats_bench_serial.txt
python tools/db_summary.py gpurun_out/r3z/prof_flow/f_results.db 50 > gpurun_out/r3z/kernel_stats_flow.txt
rm -rf gpurun_out/r3z/prof_bench gpurun_out/r3z/prof_bench_serial gpurun_out/r3z/prof_flow
python -c "
import json
d=json.load(open('gpurun_out/r3z/bench.json')); print(d['value'], d['ms_per_step'], d['phases_ms_per_step']); print(d.get('full_step')); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['commit_fr_scalar_muls_per_s_per_core_lower_bound'])
"
for f in flow_plain_2p20 spartan_plain_2p20 flow_rep3_2p16 flow_rep3_2p18 flow_plain_2p22 lookups_uniform lookups_sha2; do cut -c1-800 gpurun_out/r3z/$f.json; echo; done
cat gpurun_out/r3z/probe_layer.log
head -30 gpurun_out/r3z/kernel_stats_bench.txt
