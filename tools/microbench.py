"""GPU micro-benchmarks (run on the GPU box): mont-mul peak and single-MSM timing."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("co-zkvms_amd")
ctx = m.Context(0)
for variant in (0, 1):
    for lanes in (256 * 256 * 4, 256 * 256 * 8, 256 * 256 * 16):
        iters = 2000
        ms = ctx.bench_montmul(lanes, iters, variant)
        print(f"montmul variant={variant} lanes={lanes} iters={iters}: {ms:.3f} ms -> {lanes*iters/ms/1e6:.2f} Gmul/s", flush=True)
for logn in (16, 18, 20):
    n = 1 << logn
    t0 = time.time()
    B = m.Bases.from_scalars(ctx, m.Vec.random(ctx, n, seed=7), precompute=True)
    ctx.synchronize()
    print(f"srs 2^{logn} build+table: {time.time()-t0:.3f}s", flush=True)
    vs = [m.Vec.random(ctx, n, seed=100 + i) for i in range(8)]
    B.batch_msm_raw(vs[:1])
    for k in (1, 8):
        ctx.prof_enable(True)
        t0 = time.time()
        B.batch_msm_raw(vs[:k])
        dt = time.time() - t0
        nl, kms, adds, _ = ctx.prof_read()
        print(f"msm 2^{logn} x{k}: {dt*1e3:.2f} ms total, {dt*1e3/k:.2f} ms/msm; accum0 {kms:.2f} ms for {adds} adds -> {adds/kms/1e6:.2f} Gadd/s", flush=True)
    for kind, bits in ((m.SCALAR_U16, 0), (m.SCALAR_U32, 0), (m.SCALAR_U8, 1)):
        v = m.Vec.random(ctx, n, seed=5, kind=kind, max_bits=bits)
        B.batch_msm_raw([v])
        t0 = time.time(); B.batch_msm_raw([v] * 4); dt = time.time() - t0
        print(f"msm 2^{logn} kind={kind} bits={bits}: {dt*1e3/4:.2f} ms/msm", flush=True)
    del B, vs
