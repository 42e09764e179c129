"""Run one of BASELINE.json's configurations (SURVEY 8d's synthetic restatements) through the in-process
harness on the visible GPU(s) and print one JSON line.
  --config 1 : 2^14, 3-party Rep3 (all three parties time-sliced on GPU 0)
  --config 2 : 2^20, plain prover, 128 polys (= bench.py's N=1 workload)
  --config 3 : 2^22, 3-party Rep3, 137 shared polys per party; one GPU per party when >= 3 GPUs are visible,
               otherwise the three parties share GPU 0 (110 GB of shares + 3 window tables: fits 288 GB)
  --config 4 : co-noir-spartan, 2^18 constraints, 3-party Rep3 (SpartanHarness: zero_round, PST commit, both
               sumchecks, z(ry), distributed_open)
  --config 5 : BASELINE's 8-party Shamir run has no reference prover; the substitute SURVEY 8d names: the plain prover with
               the worker-sub-net split -- ONE proof of a 2^23-cycle trace as 8 workers (all eight time-sliced on GPU 0 when
               fewer than 8 GPUs are visible; 2^24 needs the 8-GPU node)
Each prove() is checked by the harness verifier on the first pass."""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, required=True, choices=[1, 2, 3, 4, 5])
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--log-n", type=int, default=None, help="override the configuration's trace length")
args = ap.parse_args()
cozk = importlib.import_module("co-zkvms_amd")
ngpu = torch.cuda.device_count()
devs = (0, 1, 2) if ngpu >= 3 else (0, 0, 0)
if args.config == 4:
    log_n = args.log_n or 18
    t0 = time.time()
    h = cozk.SpartanHarness(mode="rep3", log_n=log_n, seed=2026, devices=devs)
    setup_s = time.time() - t0
    r = h.prove(verify=True)
    assert r.verified == 1, h.last_error()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = h.prove(verify=False)
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"config": 4, "mode": "rep3", "log_n": log_n, "devices": list(devs), "verified": 1, "ms_per_proof": round(dt * 1e3, 2),
                      "constraints_per_s": round((1 << log_n) / dt, 1),
                      "phases_ms": {"zero_round": round(r.t_zero_round_ms, 2), "commit": round(r.t_commit_ms, 2), "sumcheck1": round(r.t_sumcheck1_ms, 2),
                                    "matrix_build": round(r.t_matrix_build_ms, 2), "sumcheck2": round(r.t_sumcheck2_ms, 2), "open": round(r.t_open_ms, 2)},
                      "star_messages": int(r.star_messages), "proof_bytes": int(r.proof_len), "setup_s": round(setup_s, 1)}), flush=True)
    h.close()
    sys.exit(0)
if args.config == 5:
    log_n = args.log_n or 23
    wd = list(range(8)) if ngpu >= 8 else [0] * 8
    t0 = time.time()
    h = cozk.Harness(mode="plain", log_n=log_n, n_fr=64, n_u16=32, n_u32=16, n_flags=16, n_small=0, gp_batch=8, gp_log_leaves=log_n + 1, seed=2026,
                     log_workers=3, worker_devices=wd)
    setup_s = time.time() - t0
    r = h.prove(verify=True)
    assert r.verified == 1, h.last_error()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = h.prove(verify=False)
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"config": "5 (substitute: plain prover, 8 worker sub-nets)", "log_n": log_n, "worker_devices": wd, "verified": 1,
                      "ms_per_proof": round(dt * 1e3, 1), "cycles_per_s": round((1 << log_n) / dt, 1),
                      "phases_ms": {"commit": round(r.t_commit_ms, 1), "gp_prove": round(r.t_gp_prove_ms, 1), "evaluate": round(r.t_eval_ms, 1),
                                    "open": round(r.t_open_ms, 1)}, "setup_s": round(setup_s, 1)}), flush=True)
    h.close()
    sys.exit(0)
if args.config == 1:
    kw = dict(mode="rep3", log_n=14, n_fr=64, n_u16=32, n_u32=16, n_flags=16, gp_batch=8)
elif args.config == 2:
    kw = dict(mode="plain", log_n=20, n_fr=64, n_u16=32, n_u32=16, n_flags=16, gp_batch=8)
else:
    kw = dict(mode="rep3", log_n=22, n_fr=137, n_u16=0, n_u32=0, n_flags=0, gp_batch=8)
if args.log_n:
    kw["log_n"] = args.log_n
t0 = time.time()
h = cozk.Harness(seed=2026, devices=devs, n_small=0, **kw)
setup_s = time.time() - t0
print("setup done in %.1f s" % setup_s, flush=True)
r = h.prove(verify=True)
assert r.verified == 1, h.last_error()
print("verified; first pass %.1f ms" % r.wall_ms, flush=True)
t0 = time.perf_counter()
for _ in range(args.steps):
    r = h.prove(verify=False)
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"config": args.config, **{k: v for k, v in kw.items()}, "devices": list(devs), "verified": 1,
                  "ms_per_proof": round(dt * 1e3, 2), "cycles_per_s": round((1 << kw["log_n"]) / dt, 1),
                  "phases_ms": {"commit": round(r.t_commit_ms, 2), "gp_construct": round(r.t_gp_construct_ms, 2),
                                "gp_prove": round(r.t_gp_prove_ms, 2), "evaluate": round(r.t_eval_ms, 2), "open": round(r.t_open_ms, 2)},
                  "ring_bytes_all_parties": int(r.bytes_ring), "star_messages": int(r.star_messages), "proof_bytes": int(r.proof_len),
                  "setup_s": round(setup_s, 1), "hbm_gib_in_use": round((torch.cuda.mem_get_info(0)[1] - torch.cuda.mem_get_info(0)[0]) / 2**30, 1)}), flush=True)
h.close()
