"""3-party Rep3 proof with one party per process (config 3's mapping: one GPU per party).
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port P tools/rep3_dist.py [--log-n 16]
COZK_DIST_BACKEND=gloo stages the ring through host memory so that 3 ranks can share one GPU (rehearsal);
otherwise the ring rides RCCL (nccl backend) device-to-device."""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=14)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--polys", type=int, default=16)
args = ap.parse_args()
rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
assert world == 3, "Rep3 needs exactly 3 parties"
backend = os.environ.get("COZK_DIST_BACKEND", "nccl")
dev = local_rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend, rank=rank, world_size=world)
hub_group = dist.new_group(backend="gloo")  # star messages: tiny, CPU
P = importlib.import_module("co-zkvms_amd.party_dist")
k = args.polys
party = P.DistributedParty(rank, device=dev, log_n=args.log_n, n_fr=k // 2, n_u16=k // 4, n_u32=k // 8, n_flags=k // 8, n_small=0,
                           gp_batch=8, seed=2026)
hub = P.TorchHub(rank, world, hub_group)
ring = P.TorchRing(party.ctx_handle(), rank, world, group=None, device=None if backend == "gloo" else dev)
res = party.prove(hub, ring, verify=True)
assert res.verified == 1, party.last_error()
dist.barrier()
t0 = time.perf_counter()
for _ in range(args.steps):
    r = party.prove(hub, ring, verify=False)
    assert bytes(r.proof_digest) == bytes(res.proof_digest)
dist.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    print(json.dumps({"mode": "rep3, one party per process", "backend": backend, "log_n": args.log_n, "polys": k, "verified": 1,
                      "ms_per_proof": round(dt * 1e3 / args.steps, 2), "cycles_per_s": round((1 << args.log_n) * args.steps / dt, 1),
                      "ring_bytes_per_party": int(r.bytes_ring), "star_messages": int(r.star_messages),
                      "proof_sha256": bytes(res.proof_digest).hex()[:16]}), flush=True)
party.close()
dist.destroy_process_group()
