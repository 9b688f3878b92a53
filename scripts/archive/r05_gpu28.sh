#!/bin/bash
# round 5, session 28: HBM traffic (PMC, separate passes) and kernel trace of the 1440^3 and 1200^3 pairs on the final build:
# do the 64-byte y-pass tiles (two tiles per 128-byte line) and the 16-column tiles keep the traffic at the algorithmic bytes?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r05_tiles
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 1440 1200; do
  B="$R/bench.py --size $n --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -- python3 $B > $O/trace_$n.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$n -- python3 $B > $O/fetch_$n.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$n -- python3 $B > $O/write_$n.log 2>&1
done
cd $R
find $O -name "*.db" -delete
python3 - <<'PY'
import csv,glob,os,collections,re
O=os.path.join(os.environ.get('GRAFT_REPO_ROOT',os.getcwd()),'gpurun_out/prof_r05_tiles')
def short(k):
    m=re.search(r'(ColFft3?S?|R2CFft|C2RFft|RowFft)\w*<mfft::Spec<[^>]*>, \w+, \d+, (true|false)',k)
    return m.group(0) if m else None
for n in (1440,1200):
    N=n; alg=N*N*(N//2+1)*16/1e9
    print('== %d^3 fp64: algorithmic bytes per strided launch %.2f GB each way (real side of r2c / c2r: %.2f GB)'%(n,alg,N**3*8/1e9))
    for kind in ('fetch','write'):
        acc=collections.defaultdict(list)
        for f in glob.glob(O+'/%s_%d/**/*counter_collection.csv'%(kind,n),recursive=True):
            for row in csv.DictReader(open(f)):
                k=short(row['Kernel_Name'])
                if k: acc[k].append(float(row['Counter_Value']))
        for k,v in sorted(acc.items()):
            x=sum(v)/len(v)
            gb=x*(2 if kind=='fetch' else 1)*1024/1e9      # FETCH_SIZE: 2 KB units on gfx950 (guide), WRITE_SIZE: KB
            print('   %-6s %-75s %8.2f GB per launch (n=%d)'%(kind,k,gb,len(v)))
    st=glob.glob(O+'/trace_%d/**/*kernel_stats.csv'%n,recursive=True)
    for f in st:
        for row in csv.DictReader(open(f)):
            k=short(row['Name'])
            if k: print('   trace  %-75s calls %s  avg %.3f ms'%(k,row['Calls'],float(row['AverageNs'])/1e6))
PY
