cd $GRAFT_REPO_ROOT
for g in 320 256 224 192 160 128; do
COZK_EVAL_GX=$g timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
python -c "
import json
d=json.load(open('gpurun_out/ab.json')); k=d['roofline']['kernels']; print('gx $g', k['k_poly_eval_chi']['achieved'], k['k_poly_eval_chi']['avg_launch_ms'], d['phases_ms_per_step']['evaluate'], d['ms_per_step'])"
done
