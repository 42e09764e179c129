"""Time the co-jolt Spartan harness and print one JSON line: the WHOLE Rep3UniformSpartanProver::prove (outer + inner
+ shift sumchecks, two opening appends) on the reference's constraint set (128 rows per step, 78 inputs), or its parts.
  python tools/run_outer.py --mode plain --log-steps 20 [--system jolt|toy] [--outer-only] [--steps 3]"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["plain", "rep3"], default="plain")
ap.add_argument("--log-steps", type=int, default=20)
ap.add_argument("--system", choices=["jolt", "toy"], default="jolt")
ap.add_argument("--outer-only", action="store_true")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--no-verify", action="store_true")
args = ap.parse_args()
OU = importlib.import_module("co-zkvms_amd.outer")
ngpu = torch.cuda.device_count()
devs = (0, 1, 2) if ngpu >= 3 else (0, 0, 0)
t0 = time.time()
h = OU.OuterHarness(mode=args.mode, log_steps=args.log_steps, seed=2026, devices=devs, system=args.system, full=not args.outer_only)
setup_s = time.time() - t0
t0 = time.time()
r = h.prove(verify=not args.no_verify)
assert args.no_verify or r.verified == 1, h.last_error()
verify_s = time.time() - t0
t0 = time.perf_counter()
for _ in range(args.steps):
    r = h.prove(verify=False)
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"what": "co-jolt Spartan: " + ("outer sumcheck" if args.outer_only else "whole worker (outer + inner + shift sumchecks + 2 opening appends)"),
                  "system": args.system, "mode": args.mode, "log_steps": args.log_steps, "rows_per_step": 128 if args.system == "jolt" else 8,
                  "devices": list(devs), "verified": int(r.verified) if not args.no_verify else None, "ms_per_proof": round(dt * 1e3, 2),
                  "steps_per_s": round((1 << args.log_steps) / dt, 1),
                  "phases_ms": {"build_AzBzCz": round(r.t_build_ms, 2), "outer": round(r.t_outer_ms, 2), "inner": round(r.t_inner_ms, 2),
                                "shift": round(r.t_shift_ms, 2), "openings": round(r.t_openings_ms, 2), "prove_total": round(r.t_prove_ms, 2)},
                  "star_messages": int(r.star_messages), "proof_bytes": int(r.proof_len), "setup_s": round(setup_s, 1), "first_prove_and_verify_s": round(verify_s, 1),
                  "hbm_gib_in_use": round((torch.cuda.mem_get_info(0)[1] - torch.cuda.mem_get_info(0)[0]) / 2**30, 1)}), flush=True)
h.close()
