cd $GRAFT_REPO_ROOT
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['breakdown_ms_per_iter']['psd_project'])"; }
for b in 4096 8192 10000 12288 16384 20000 40000; do
  timeout 300 python bench.py --no-cpu-baseline --blocks-per-gpu $b --steps 100 2>&1 | grep '^{' | pl "c2 blocks $b"
done
