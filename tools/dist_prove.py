"""One distributed proof, one (party, worker) participant per process / GPU.
  --mode rep3 --log-workers 0 : 3 ranks, one Rep3 party each (BASELINE config 3's mapping)
  --mode plain --log-workers k: 2^k ranks, one worker sub-net (high-variable chunk) each
  --mode rep3 --log-workers k : 3 * 2^k ranks, rank = worker * 3 + party (each worker index has its own ring)
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P tools/dist_prove.py ...
The Rep3 ring: --ring native (default when every rank has its own GPU) = libcozk's RCCL ring (cozk_ring_*: ncclSend /
ncclRecv on the party's stream, GPU to GPU over xGMI, asynchronous); --ring torch = torch.distributed P2P through
callbacks (nccl backend device-to-device, or -- COZK_DIST_BACKEND=gloo / fewer GPUs than ranks -- staged through host
memory so that several ranks can share one GPU for rehearsal).  The control plane (rendezvous, barriers, naming the
shared-memory segment) is gloo.  The per-round star messages go through libcozk's shared-memory hub (--hub shm, single
node) or a gloo all-gather (--hub gloo)."""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["rep3", "plain"], default="rep3")
ap.add_argument("--log-workers", type=int, default=0)
ap.add_argument("--hub", choices=["shm", "gloo"], default="shm")
ap.add_argument("--log-n", type=int, default=14)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--polys", type=int, default=16)
ap.add_argument("--ring", choices=["auto", "native", "torch"], default="auto")
args = ap.parse_args()
rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
nparties = 3 if args.mode == "rep3" else 1
assert world == nparties << args.log_workers, "need nparties * 2^log_workers ranks"
party_id, worker_id = rank % nparties, rank // nparties
own_gpu = torch.cuda.device_count() >= world
ring_kind = args.ring if args.ring != "auto" else ("native" if own_gpu and os.environ.get("COZK_DIST_BACKEND") != "gloo" else "torch")
if ring_kind == "native" and not own_gpu:
    raise SystemExit("--ring native needs one GPU per rank (RCCL refuses two ranks of a communicator on one device)")
# torch P2P ring: nccl when every rank has a GPU of its own, else host-staged gloo
backend = "gloo" if ring_kind == "native" else os.environ.get("COZK_DIST_BACKEND", "nccl" if own_gpu else "gloo")
dev = local_rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend, rank=rank, world_size=world)
hub_group = dist.new_group(backend="gloo")  # CPU group: star messages (gloo hub) / naming the shm segment
P = importlib.import_module("co-zkvms_amd.party_dist")
k = args.polys
party = P.DistributedParty(party_id, device=dev, worker=worker_id, mode=args.mode, log_workers=args.log_workers, log_n=args.log_n,
                           n_fr=k // 2, n_u16=k // 4, n_u32=k // 8, n_flags=k // 8, n_small=0, gp_batch=8, seed=2026)
hub = P.ShmHub(rank, world, hub_group) if args.hub == "shm" else P.TorchHub(rank, world, hub_group)
ring = None
if nparties == 3:
    base = worker_id * 3
    if ring_kind == "native":
        # one ring (= one RCCL communicator of 3 ranks) per worker index; every rank creates every group, as
        # torch.distributed requires, and keeps its own
        groups = [dist.new_group(ranks=[3 * w, 3 * w + 1, 3 * w + 2], backend="gloo") for w in range(world // 3)]
        ring = P.NativeRing(party.ctx_handle(), party_id, 3, groups[worker_id])
    else:
        ring = P.TorchRing(party.ctx_handle(), rank, world, group=None, device=None if backend == "gloo" else dev,
                           next_rank=base + (party_id + 1) % 3, prev_rank=base + (party_id + 2) % 3)
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
    print(json.dumps({"mode": "%s, %d worker sub-net(s), one participant per process" % (args.mode, 1 << args.log_workers),
                      "backend": backend, "ring": ring_kind if nparties == 3 else None, "hub": args.hub, "log_n": args.log_n, "polys": k, "verified": 1,
                      "ms_per_proof": round(dt * 1e3 / args.steps, 2), "cycles_per_s": round((1 << args.log_n) * args.steps / dt, 1),
                      "ring_bytes_per_party": int(r.bytes_ring), "star_messages": int(r.star_messages),
                      "proof_sha256": bytes(res.proof_digest).hex()[:16]}), flush=True)
if ring is not None and hasattr(ring, "close"):
    ring.close()
party.close()
hub.close() if hasattr(hub, "close") else None
dist.destroy_process_group()
