#!/bin/bash
# round 5, session 17: what bounds the strided kernel of 1200 (3.7 - 3.9 TB/s against 5.0 - 5.1 at 1024)?  PMC passes of the
# 1200^3 and, beside them, the 1024^3 pair (separate passes: traffic, LDS, waits)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r05_1200
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 1200 1024; do
  B="$R/bench.py --size $n --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$n -- python3 $B > $O/fetch_$n.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$n -- python3 $B > $O/write_$n.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1_$n -- python3 $B > $O/sq1_$n.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2_$n -- python3 $B > $O/sq2_$n.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU --output-format csv -d $O/sq3_$n -- python3 $B > $O/sq3_$n.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/tcc_$n -- python3 $B > $O/tcc_$n.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TD_BUSY_avr --output-format csv -d $O/tcp_$n -- python3 $B > $O/tcp_$n.log 2>&1
done
cd $R
tail -2 $O/*.log | cut -c1-300
find $O -name "*.db" -delete
python3 - <<'PY'
import csv,glob,os,collections,re
O=os.path.join(os.environ.get('GRAFT_REPO_ROOT',os.getcwd()),'gpurun_out/prof_r05_1200')
for d in sorted(glob.glob(O+'/*_1*')):
    if not os.path.isdir(d): continue
    files=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not files: print(os.path.basename(d),'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name']
            m=re.search(r'(ColFft3?S?|R2CFft|C2RFft|RowFft)\w*<mfft::Spec<[^>]*>',k)
            k=m.group(0) if m else k[:60]
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==',os.path.basename(d))
    for k,v in acc.items():
        print('  ',k,{c:round(sum(x)/len(x),1) for c,x in v.items()},'n',len(next(iter(v.values()))))
PY
du -sh $O
