cd $GRAFT_REPO_ROOT
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, round(d['roofline']['frac'],4))"; }
for e in "$@"; do
  env $e timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 $e"
  env $e timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4 $e"
done
