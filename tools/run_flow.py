"""Time the ONE chained co-jolt worker flow (cozk_flow_*: commit-all -> bytecode -> instruction lookups -> read-write memory ->
Spartan -> one reduce_and_prove; co-jolt/src/jolt/vm/jolt/worker.rs:175-266) at Jolt's shape and print one JSON line.
  python tools/run_flow.py --mode plain --log-n 20 [--n-mem 54] [--steps 3]"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["plain", "rep3"], default="plain")
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--n-mem", type=int, default=54)
ap.add_argument("--n-subtables", type=int, default=26)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--small-witness", action="store_true", help="counters < 2^log_n, E / memory values 32 bits (a plain prover's real widths) instead of uniform field elements")
args = ap.parse_args()
FL = importlib.import_module("co-zkvms_amd.flow")
ngpu = torch.cuda.device_count()
devs = (0, 1, 2) if ngpu >= 3 else (0, 0, 0)
n = args.log_n
t0 = time.time()
h = FL.FlowHarness(mode=args.mode, log_n=n, log_m=min(16, n), log_b=min(14, n), log_mem=min(17, n), n_mem=args.n_mem, n_subtables=args.n_subtables,
                   seed=2026, devices=devs, small_witness=1 if args.small_witness else 0)
setup_s = time.time() - t0
t0 = time.time()
r = h.prove(verify=True)
assert r.verified == 1, h.last_error()
verify_s = time.time() - t0
t0 = time.perf_counter()
for _ in range(args.steps):
    r = h.prove(verify=False)
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"what": "one chained co-jolt worker flow (commit-all, bytecode, instruction lookups, read-write memory, Spartan, one batched opening)",
                  "mode": args.mode, "witness": "real widths (counters < 2^log_n, E / memory values 32 bits)" if args.small_witness else "uniform field elements", "log_n": n, "memories": args.n_mem, "subtables": args.n_subtables, "polys_committed": int(r.n_polys), "openings": int(r.n_openings),
                  "devices": list(devs), "verified": 1, "ms_per_proof": round(dt * 1e3, 2), "cycles_per_s": round((1 << n) / dt, 1),
                  "phases_ms": {"commit": round(r.t_commit_ms, 2), "bytecode": round(r.t_bytecode_ms, 2), "lookups_primary_sumcheck": round(r.t_primary_ms, 2),
                                "lookups_memory_checking": round(r.t_lookups_gp_ms, 2), "read_write_memory": round(r.t_rw_ms, 2), "spartan": round(r.t_spartan_ms, 2),
                                "spartan_build_AzBzCz": round(r.t_spartan_build_ms, 2), "reduce_and_prove": round(r.t_open_ms, 2)},
                  "ring_bytes_all_parties": int(r.bytes_ring), "star_messages": int(r.star_messages), "proof_bytes": int(r.proof_len), "setup_s": round(setup_s, 1),
                  "first_prove_and_verify_s": round(verify_s, 1),
                  "hbm_gib_in_use": round((torch.cuda.mem_get_info(0)[1] - torch.cuda.mem_get_info(0)[0]) / 2**30, 1)}), flush=True)
h.close()
