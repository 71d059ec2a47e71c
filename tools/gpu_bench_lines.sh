cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c2.json
timeout 600 python3 bench.py --config c3 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c3.json
timeout 600 python3 bench.py --config c4 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c4.json
timeout 600 python3 bench.py --mode sgs --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c2_sgs.json
CUADMM_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --sharding allreduce --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c2_allreduce_forced_1rank.json
for f in c2 c3 c4 c2_sgs c2_allreduce_forced_1rank; do python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench_$f.json')); r=d['roofline']
print('$f', round(d['value'],1), round(d['ms_per_step'],4), 'psd', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'issued', round(r['mfma_issued_tflops'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, d.get('cpu_baseline',{}).get('value'))"; done
