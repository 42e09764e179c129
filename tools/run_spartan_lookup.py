"""co-noir-spartan config 4 with the public lookup round on one GPU: 1, 2 and 4 public workers
(--pub-workers, log2 values), plain and rep3.  The proof digest must not depend on the worker count."""
import argparse
import importlib
import json
import os
import sys

sys.path.insert(0, os.getcwd())
m = importlib.import_module("co-zkvms_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=18)
ap.add_argument("--pub-workers", type=int, nargs="*", default=[0], help="log2 of the public worker counts to run")
ap.add_argument("--modes", nargs="*", default=["plain", "rep3"])
a = ap.parse_args()
out = {}
digests = set()
for mode in a.modes:
    for k in a.pub_workers:
        h = m.SpartanHarness(mode=mode, log_n=a.log_n, seed=4, lookup_round=True, log_pub_workers=k)
        r = h.prove(verify=True)
        assert r.verified == 1, h.last_error()
        r = h.prove(verify=False)
        digests.add(bytes(r.proof_digest).hex())
        out[f"{mode}_k{k}"] = dict(wall_ms=round(r.wall_ms, 2), zero_round=round(r.t_zero_round_ms, 2), commit=round(r.t_commit_ms, 2),
                                   sumcheck1=round(r.t_sumcheck1_ms, 2), matrix_build=round(r.t_matrix_build_ms, 2),
                                   sumcheck2=round(r.t_sumcheck2_ms, 2), open=round(r.t_open_ms, 2), lookup_round=round(r.t_lookup_ms, 2),
                                   pub_workers=int(r.pub_workers), pub_star_messages=int(r.pub_star_messages),
                                   proof_bytes=int(r.proof_len), verified=1)
        h.close()
assert len(digests) == 1, "proof digest depends on the mode / worker count"
print(json.dumps({"what": f"co-noir-spartan config 4 (2^{a.log_n} constraints, 3 x 2^{a.log_n} entries) incl. the public lookup round (8(f)4), one GPU",
                  "proof_digest": digests.pop(), **out}))
