# same-box A/B of environment switches: bash tools/ab_env.sh "VAR=1" "VAR2=x VAR3=y" ...  ("" = defaults)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/ab.json')); k=d['roofline']['kernels']['k_msm_scatter_lds']
print('[$cfg]: step', d['ms_per_step'], 'commit', d['phases_ms_per_step']['commit'], 'scatter avg', k['avg_launch_ms'], 'gather avg', d['roofline']['avg_launch_ms'])
"
done
done
