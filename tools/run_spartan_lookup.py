import importlib, sys, os, json
sys.path.insert(0, os.getcwd())
m = importlib.import_module("co-zkvms_amd")
out = {}
for mode in ("plain", "rep3"):
    h = m.SpartanHarness(mode=mode, log_n=18, seed=4, lookup_round=True)
    r = h.prove(verify=True)
    assert r.verified == 1, h.last_error()
    r = h.prove(verify=False)
    out[mode] = dict(wall_ms=round(r.wall_ms, 2), zero_round=round(r.t_zero_round_ms, 2), commit=round(r.t_commit_ms, 2), sumcheck1=round(r.t_sumcheck1_ms, 2),
                     matrix_build=round(r.t_matrix_build_ms, 2), sumcheck2=round(r.t_sumcheck2_ms, 2), open=round(r.t_open_ms, 2),
                     lookup_round=round(r.t_lookup_ms, 2), proof_bytes=int(r.proof_len), verified=1)
    h.close()
print(json.dumps({"what": "co-noir-spartan config 4 (2^18 constraints, 3 x 2^18 entries) incl. the public lookup round (8(f)4), one GPU", **out}))
