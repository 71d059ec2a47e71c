#!/bin/bash
# static instruction mix of one kernel of psd_kernels.hip per phase (CPU only): tools/isa_stats.sh [mangled-name-regex]
set -e
K="${1:-psd_sign_closed_cu_kernelILi2ELi16ELi4ELb1}"
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -I/root/repo/cuadmm_amd/csrc -c /root/repo/cuadmm_amd/csrc/psd_kernels.hip --save-temps -o psd.o 2>/dev/null
S=psd_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v k="$K" '$0 ~ "^_ZN6cuadmm[0-9]+"k".*:" {p=1} p{print} /\.end_amdhsa_kernel/{if(p) exit}' $S > k.s
python3 - <<'PY'
import re,collections
lines=open('/tmp/isa/k.s').read().split('\n')
def cat(m):
    if m.startswith('v_mfma'): return 'mfma'
    if m.startswith(('v_readlane','v_writelane','v_readfirstlane')): return 'lane'
    if m.startswith('v_') and ('f64' in m): return 'valu64'
    if m.startswith('v_'): return 'valu'
    if m.startswith(('s_waitcnt','s_nop')): return 'wait'
    if m.startswith('s_'): return 'salu'
    if m.startswith('ds_bpermute'): return 'bperm'
    if m.startswith('ds_'): return 'lds'
    if m.startswith(('global_','scratch_','flat_','buffer_')): return 'vmem'
    return 'other'
def count(a,b):
    c=collections.Counter()
    for i in range(a,b+1):
        l=lines[i]
        if l.startswith('\t') and not l.strip().startswith(('.',';')) and l.strip():
            c[cat(l.strip().split()[0])]+=1
    return dict(sorted(c.items()))
mf=[i for i,l in enumerate(lines) if '\tv_mfma' in l]
# the Newton-Schulz loop: the depth-2 loop that holds the first MFMA cluster
hdrs=[i for i,l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:',l)]
first=mf[0]
lab=None
for i in reversed(hdrs):
    if i<first:
        name=lines[i].split(':')[0]
        back=[j for j,l in enumerate(lines) if re.search(r'\bs_c?branch\w*\s+'+re.escape(name)+r'\b',l) and j>first]
        if back: lab=(i,max(back)); break
print('lines',len(lines),'scratch ops',sum('scratch_' in l for l in lines),'vgprs',[l.split()[-1] for l in lines if 'next_free_vgpr' in l], 'private',[l.split()[-1] for l in lines if 'private_segment_fixed_size' in l])
if lab:
    print('prologue (before the sign loop):',count(0,lab[0]-1))
    print('sign loop (static)             :',count(lab[0],lab[1]))
    print('epilogue                       :',count(lab[1]+1,len(lines)-1))
PY
