cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_psd.py tests/test_gpu_fused.py -x -q > gpurun_out/cross_pytest.log 2>&1; grep -a "passed\|failed" gpurun_out/cross_pytest.log | tail -2
for n in 45 64; do for cnt in 128 512 1024 2048 4096 8192; do
  a=$(python tools/probe_w32_occ.py $n $cnt 5 2>&1 | tail -1 | sed 's/.*: \([0-9.]*\) us.*/\1/')
  b=$(CUADMM_PSD_MID=lds python tools/probe_w32_occ.py $n $cnt 5 2>&1 | tail -1 | sed 's/.*: \([0-9.]*\) us.*/\1/')
  echo "n=$n count=$cnt wave $a us  lds $b us"
done; done
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, round(d['roofline']['frac'],4))"; }
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2"
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4"
CUADMM_FUSE_ROWS=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 norows"
