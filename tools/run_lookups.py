"""Time the instruction-lookups harness (toggled / sparse grand product of Lasso's read / write memory
checking) at Jolt's shape -- 54 memories (108 circuits), ~10 % flag density -- and print one JSON line.
  python tools/run_lookups.py --mode plain --log-n 20 [--pairs 54] [--density 10] [--steps 3]"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["plain", "rep3"], default="plain")
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--pairs", type=int, default=54)
ap.add_argument("--density", type=int, default=10)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--primary", action="store_true", help="also run Lasso's primary sumcheck (8f1b)")
ap.add_argument("--mix", choices=["uniform", "sha2"], default="uniform", help="instruction mix of the synthetic trace (sha2: trace-shaped, ~6 %% multiplicative)")
args = ap.parse_args()
LK = importlib.import_module("co-zkvms_amd.lookups")
ngpu = torch.cuda.device_count()
devs = (0, 1, 2) if ngpu >= 3 else (0, 0, 0)
t0 = time.time()
h = LK.LookupsHarness(mode=args.mode, log_n=args.log_n, n_pairs=args.pairs, density_pct=args.density, seed=2026, devices=devs, primary=args.primary, mix=args.mix)
setup_s = time.time() - t0
r = h.prove(verify=True)
assert r.verified == 1, h.last_error()
t0 = time.perf_counter()
for _ in range(args.steps):
    r = h.prove(verify=False)
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"what": ("Lasso primary sumcheck + " if args.primary else "") + "toggled grand product (instruction lookups read/write memory checking)", "mode": args.mode, "mix": args.mix, "log_n": args.log_n,
                  "memories": args.pairs, "circuits": 2 * args.pairs, "density_pct": args.density, "devices": list(devs), "verified": 1,
                  "ms_per_proof": round(dt * 1e3, 2), "cycles_per_s": round((1 << args.log_n) / dt, 1),
                  "phases_ms": {"primary_sumcheck": round(r.t_primary_ms, 2), "gp_construct": round(r.t_construct_ms, 2), "gp_prove": round(r.t_prove_ms, 2)},
                  "ring_bytes_all_parties": int(r.bytes_ring), "star_messages": int(r.star_messages), "proof_bytes": int(r.proof_len),
                  "setup_s": round(setup_s, 1), "hbm_gib_in_use": round((torch.cuda.mem_get_info(0)[1] - torch.cuda.mem_get_info(0)[0]) / 2**30, 1)}), flush=True)
h.close()
