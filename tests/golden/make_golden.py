"""Generates tests/golden/*.json from the exact big-int oracle (oracle/pyref.py, oracle/pyharness.py).
The reference is Rust and cannot run here, so these are the build's own fixtures;
they pin the three implementations (Python big-int, plain C, HIP) to one another and to the few
constants the reference holds.  Run:  python tests/golden/make_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyharness  # noqa: E402
import pyref as O  # noqa: E402


def hx(x):
    return hex(x)


def main():
    rng = O.SplitMix64(20260101)
    edge = [0, 1, 2, O.R - 1, O.R_MONT_ONE, O.TWO_INV, (1 << 253) + 5]
    fr = []
    for _ in range(12):
        a, b = rng.field(), rng.field()
        fr.append(dict(a=hx(a), b=hx(b), add=hx((a + b) % O.R), sub=hx((a - b) % O.R), mul=hx(a * b % O.R)))
    for a in edge:
        for b in edge[:4]:
            fr.append(dict(a=hx(a), b=hx(b), add=hx((a + b) % O.R), sub=hx((a - b) % O.R), mul=hx(a * b % O.R)))
    fq = []
    for _ in range(12):
        a, b = rng.field(O.P), rng.field(O.P)
        fq.append(dict(a=hx(a), b=hx(b), add=hx((a + b) % O.P), sub=hx((a - b) % O.P), mul=hx(a * b % O.P)))
    # G1 / MSM
    pts = [O.g1_mul(O.G1_GEN, rng.field()) for _ in range(17)]
    sc = [rng.field() for _ in range(17)]
    sc[0], sc[1], sc[2], sc[3] = 0, 1, O.R - 1, 65536
    msm = dict(points=[[hx(p[0]), hx(p[1])] for p in pts], scalars=[hx(s) for s in sc],
               result=[hx(v) for v in O.msm_naive(pts, sc)])
    g1 = dict(two_g=[hx(v) for v in O.g1_add(O.G1_GEN, O.G1_GEN)], three_g=[hx(v) for v in O.g1_mul(O.G1_GEN, 3)])
    # pipeline digests
    pipes = []
    for cfg in (dict(log_n=5, n_fr=3, n_u16=1, n_u32=1, n_flags=1, n_small=2, gp_batch=2, gp_log_leaves=6, seed=42),
                dict(log_n=4, n_fr=2, n_u16=0, n_u32=0, n_flags=1, n_small=0, gp_batch=3, gp_log_leaves=4, seed=7)):
        for mode in ("plain", "rep3"):
            r = pyharness.run(dict(cfg, mode=mode))
            pipes.append(dict(cfg=dict(cfg, mode=mode), digest=r["digest"], proof_len=len(r["proof_bytes"])))
    json.dump(dict(fr=fr, fq=fq, g1=g1, msm=msm, pipelines=pipes), open(os.path.join(HERE, "vectors.json"), "w"), indent=1)
    print("wrote vectors.json")


if __name__ == "__main__":
    main()
