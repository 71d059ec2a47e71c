cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_psd.py -x -q 2>&1 | tail -5
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, d.get('newton_schulz_steps'))"; }
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4 wave48"
CUADMM_PSD_MID=lds timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4 lds48"
for n in 45 48 40 64 56; do
  python tools/probe_w32_occ.py $n 16667 3 2>&1 | tail -1
  CUADMM_PSD_MID=lds python tools/probe_w32_occ.py $n 16667 3 2>&1 | sed 's/^/   lds: /' | tail -1
  CUADMM_PSD_W64=1 python tools/probe_w32_occ.py $n 16667 3 2>&1 | sed 's/^/   w64: /' | tail -1
done
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 tuned32"
CUADMM_PSD_W32_GEN=3 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 gen occ3"
CUADMM_PSD_W32_GEN=4 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 gen occ4"
for pad in 9000 29000; do
  CUADMM_PSD_W32_PAD=$pad CUADMM_PSD_DEBUG=1 python3 tools/probe_w32_occ.py 32 10000 1 2>&1 | grep "psd debug" | tail -1
done
