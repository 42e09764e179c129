# SQ counters of the layer round kernels on one 2^24-element layer (tools/probe_layer.py), two passes; run on the GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_layer &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_layer/sq -- python3 tools/probe_layer.py 24 > gpurun_out/pmc_layer/sq.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/pmc_layer/sq2 -- python3 tools/probe_layer.py 24 > gpurun_out/pmc_layer/sq2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_layer/fetch -- python3 tools/probe_layer.py 24 > gpurun_out/pmc_layer/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_layer/write -- python3 tools/probe_layer.py 24 > gpurun_out/pmc_layer/write.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_layer/pmc_layer.json gpurun_out/pmc_layer/sq gpurun_out/pmc_layer/sq2 gpurun_out/pmc_layer/fetch gpurun_out/pmc_layer/write > gpurun_out/pmc_layer/summary.txt 2>&1
find gpurun_out/pmc_layer -name "*.csv" -size +2000k -delete
tail -3 gpurun_out/pmc_layer/sq2.log
