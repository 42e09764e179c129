# SQ counters of the instruction-lookups kernels (primary sumcheck + toggled grand product) at 2^20 cycles; run on the GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_lookups &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_lookups/sq -- python3 tools/run_lookups.py --log-n 20 --primary --steps 1 > gpurun_out/pmc_lookups/sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/pmc_lookups/sq2 -- python3 tools/run_lookups.py --log-n 20 --primary --steps 1 > gpurun_out/pmc_lookups/sq2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_lookups/raw.json gpurun_out/pmc_lookups/sq gpurun_out/pmc_lookups/sq2 > /dev/null 2>&1
python tools/pmc_curate.py sq gpurun_out/pmc_lookups/raw.json gpurun_out/pmc_lookups/pmc_lookups.json "round 3" "tools/pmc_probe_lookups.sh"
find gpurun_out/pmc_lookups -name "*.csv" -size +2000k -delete
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_lookups/pmc_lookups.json"))["kernels"]
rows=[]
for k,v in d.items():
    der=v.get("derived")
    if not der: continue
    rows.append((v.get("GRBM_GUI_ACTIVE",0)*v["launches"],k,v["launches"],der))
for t,k,n,der in sorted(rows,reverse=True)[:24]:
    print("%-40s launches %4d cyc/xcd %9d util %.2f wait_mem %.2f instr/wave %.0f"%(k[:40],n,der["cycles_per_xcd"],der["valu_issue_utilisation"],der.get("wait_mem_share",0),der.get("valu_instructions_per_wave",0)))
PY
