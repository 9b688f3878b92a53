#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing (VALU, quarter-rate integer multiplies, LDS, vector loads /
stores, scalar, scratch), by demangled-name regex:

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/opt/rocm/include -S --offload-device-only \
        mpifft4py_amd/csrc/kernels_b_d.hip -o /tmp/kernels_b_d.s
    scripts/isa_count.py /tmp/kernels_b_d.s 'ColFft<mfft::Spec<1024, 8, 8, 4, 4>, double'

Used for profiles/r05_plain_rows.txt (how much of a strided kernel's VALU time is address arithmetic)."""
import sys,re,subprocess,collections
path=sys.argv[1]; pat=sys.argv[2]
cur=None; stats={}
for line in open(path):
    mm=re.match(r'(_Z\S+):',line)
    if mm:
        cur=mm.group(1); stats[cur]=collections.Counter(); continue
    if cur is None: continue
    s=line.strip()
    if s.startswith('.end_amdhsa') or s.startswith('.Lfunc_end'): cur=None; continue
    m=re.match(r'([a-z_0-9]+)\s',s)
    if not m: continue
    op=m.group(1)
    c=stats[cur]
    if op.startswith('v_'):
        c['valu']+=1
        if 'f64' in op: c['f64']+=1
        if op.startswith('v_pk_'): c['pk']+=1
        if 'mul_lo' in op or 'mul_hi' in op or 'mad_u' in op or 'mad_i' in op: c['imul']+=1
    elif op.startswith('s_'):
        c['salu']+=1
        if op.startswith('s_waitcnt'): c['wait']+=1
        if op.startswith('s_barrier'): c['barrier']+=1
    elif op.startswith('ds_'): c['lds']+=1
    elif op.startswith('global_load') or op.startswith('buffer_load'): c['vld']+=1
    elif op.startswith('global_store') or op.startswith('buffer_store'): c['vst']+=1
    elif op.startswith('scratch_'): c['scratch']+=1
names=list(stats)
dem=subprocess.run(['c++filt'],input='\n'.join(names),capture_output=True,text=True).stdout.split('\n')
for n,d in zip(names,dem):
    if re.search(pat,d):
        c=stats[n]
        print(d[:150]); print('   ',dict(c))
