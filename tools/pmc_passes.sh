cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/pmc_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/pmc_write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/pmc_sq.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-step > gpurun_out/pmc_sq2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_hbm.json gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_hbm.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_sq.json gpurun_out/pmc_sq gpurun_out/pmc_sq2 > gpurun_out/pmc_sq.txt 2>&1
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/pmc_sq2 -name "*.csv" -size +2000k -delete
tail -3 gpurun_out/pmc_sq.log; tail -3 gpurun_out/pmc_sq2.log
