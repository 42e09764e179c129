"""Curate the raw per-kernel PMC averages of tools/pmc_passes.sh / pmc_probe_layer.sh (tools/pmc_summary.py output) into the
committed profiles/ format bench.py reads: metadata + derived utilisation figures per kernel.
Usage: python tools/pmc_curate.py hbm|sq RAW.json OUT.json "measured_at text" ["command text"]"""
import json
import sys

kind, raw, out, when = sys.argv[1:5]
cmd = sys.argv[5] if len(sys.argv) > 5 else "tools/pmc_passes.sh"
d = json.load(open(raw))["kernels"]
res = {"command": cmd, "measured_at": when}
if kind == "hbm":
    res["units"] = ("KB as reported by rocprofv3 (FETCH_SIZE = TCC_EA0_RDREQ x 64 B; reads exactly 1/2 of wide coalesced streams on gfx950, "
                    "uncalibrated for the 64-byte gathers of k_msm_accum0_f9 -- MI355X_MICROARCH.md HBM section)")
    res["kernels"] = {k: {"launches": v["launches"], "FETCH_SIZE_KB_per_launch": v.get("FETCH_SIZE_per_launch", 0.0),
                          "WRITE_SIZE_KB_per_launch": v.get("WRITE_SIZE_per_launch", 0.0)} for k, v in d.items()}
else:
    res["units"] = ("per-launch averages; SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over all SIMDs "
                    "(MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs")
    res["derived"] = ("valu_issue_utilisation = SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 / 4 x 1024 SIMDs); lane_utilisation = "
                      "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); *_same_pass uses SQ_BUSY_CYCLES / 32 shader engines of the SAME pass; "
                      "wait_mem_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave cycles spent in s_waitcnt)")
    ks = {}
    for k, v in d.items():
        e = {n.replace("_per_launch", ""): x for n, x in v.items()}
        g, act = e.get("GRBM_GUI_ACTIVE"), e.get("SQ_ACTIVE_INST_VALU")
        if g and act:
            der = {"cycles_per_xcd": int(g / 8), "valu_issue_utilisation": round(act * 4 / (g * 128), 4)}
            if e.get("SQ_THREAD_CYCLES_VALU"):
                der["lane_utilisation"] = round(e["SQ_THREAD_CYCLES_VALU"] / (64 * act), 4)
            if e.get("SQ_WAVES") and e.get("SQ_INSTS_VALU"):
                der["valu_instructions_per_wave"] = round(e["SQ_INSTS_VALU"] / e["SQ_WAVES"], 1)
            if e.get("SQ_BUSY_CYCLES"):
                der["cycles_per_se_same_pass"] = int(e["SQ_BUSY_CYCLES"] / 32)
                der["valu_issue_utilisation_same_pass"] = round(act * 4 / (e["SQ_BUSY_CYCLES"] / 32 * 1024), 4)
            if e.get("SQ_WAIT_ANY") and e.get("SQ_WAVE_CYCLES"):
                der["wait_mem_share"] = round(e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], 4)
            e["derived"] = der
        ks[k] = e
    res["kernels"] = ks
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res["kernels"]), "kernels")
