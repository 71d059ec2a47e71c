cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_psd.py tests/test_gpu_fused.py -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, round(d['roofline']['frac'],4), d.get('newton_schulz_steps',{}).get('mean'))"; }
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2"
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4"
for e in "$@"; do
  env $e timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 $e"
  env $e timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4 $e"
done
export CUADMM_PSD_DEBUG=1 CUADMM_PSD_DEBUG_GEN=1; for n in 15 32 45 64; do python tools/probe_w32_occ.py $n 16667 1 2>&1 | grep "psd debug" | tail -1; done
