import importlib, sys, os
sys.path.insert(0, os.getcwd())
m = importlib.import_module("co-zkvms_amd")
ctx = m.Context(0)
lanes = 256*256*16
for v in (2, 3):
    for kib in (0, 26, 30, 40, 53, 64):
        best = 1e9
        for rep in range(3):
            ms = ctx.bench_montmul(lanes, 2000, v | (kib << 8))
            best = min(best, ms)
        print(f"variant={v} lds_kib={kib}: {best:.3f} ms -> {lanes*2000/best/1e6:.1f} G products/s", flush=True)
