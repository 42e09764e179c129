"""Time the GKR layer round kernels on one big layer (run on the GPU box): k_layer_cubic (round 0) and k_layer_bind_cubic
(later rounds) of a 2^log_len-element interleaved layer, plain and Rep3.  python tools/probe_layer.py [log_len]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("co-zkvms_amd")
P = importlib.import_module("co-zkvms_amd.poly")
log_len = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = m.Context(0)
n = 1 << log_len
for mode in ("plain", "rep3"):
    a = m.Vec.random(ctx, n, seed=11)
    b = m.Vec.random(ctx, n, seed=12) if mode == "rep3" else None
    base = P.Rep3DenseInterleavedPolynomial.from_vecs(ctx, a, b)
    w = [1234567 + 7 * i for i in range(log_len - 1)]
    for rep in range(2):
        layer = base.clone()
        eq = P.SplitEqPolynomial(ctx, w)
        ctx.synchronize()
        times = []
        r = None
        for rnd in range(6):
            t0 = time.perf_counter()
            layer.round(eq, r, 5)
            times.append((time.perf_counter() - t0) * 1e3)
            r = 987654321 + rnd
        if rep == 1:
            # round j streams len / 2^j elements of 32 B (x2 for Rep3): reads them once, writes half (bind rounds)
            out = []
            for j, t in enumerate(times):
                ln = n >> j
                comps = 2 if mode == "rep3" else 1
                byts = ln * 32 * comps * (1.0 if j == 0 else 1.5) * (2 if j else 1)  # round j >= 1 reads the unbound layer of 2 ln
                chunks = ln // 4
                out.append("r%d %.3f ms (%.0f GB/s, %.2f G chunks/s)" % (j, t, byts / t / 1e6, chunks / t / 1e6))
            print(mode, "2^%d:" % log_len, "; ".join(out), flush=True)
        layer.free() if hasattr(layer, "free") else None
        eq.free()
