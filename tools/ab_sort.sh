cd $GRAFT_REPO_ROOT
for cfg in "512 64" "256 64" "256 32" "256 16" "256 48" "512 64" "256 64"; do
  set -- $cfg
  COZK_MSM_LTPB=$1 COZK_MSM_WGS=$2 timeout -k 10 200 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/ab.json')); k=d['roofline']['kernels']['k_msm_scatter_lds']
print('ltpb $1 wgs $2: step', d['ms_per_step'], 'commit', d['phases_ms_per_step']['commit'], 'scatter avg', k['avg_launch_ms'], 'share', k['share_of_step'], 'accum avg', d['roofline']['avg_launch_ms'])
"
done
