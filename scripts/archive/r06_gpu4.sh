#!/bin/bash
# PMC counters of the fused nonlinear z stage: where do the cycles of the 3/2-rule kernels go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/nlz_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, env, counters
  env $2 rocprofv3 --pmc $3 --output-format csv -d $O/$1 -- python3 $R/scripts/nlz_bench.py 768 257 73728 double 512 257 65536 double 1536 513 36864 double > $O/$1.log 2>&1
}
for v in 0 11; do
  export MFFT_NLZ_VARIANT=$v
  [ $v = 0 ] && export MFFT_NLZ3=0 || unset MFFT_NLZ3
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/a$v -- python3 $R/scripts/nlz_bench.py 768 257 73728 double 512 257 65536 double 1536 513 36864 double > $O/a$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/b$v -- python3 $R/scripts/nlz_bench.py 768 257 73728 double 512 257 65536 double 1536 513 36864 double > $O/b$v.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $O/c$v -- python3 $R/scripts/nlz_bench.py 768 257 73728 double 512 257 65536 double 1536 513 36864 double > $O/c$v.log 2>&1
done
cd $R
python3 - <<'PY' | tee gpurun_out/r06/nlz_pmc.txt
import csv, glob, os, collections
O = "gpurun_out/r06/nlz_pmc"
for d in sorted(glob.glob(O + "/[abc]*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        print(d, "no csv;", open(d + ".log").read()[-400:]); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if "Nlz" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("Nlz")[1][:48]
        acc.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("##", os.path.basename(d))
    for k, cs in acc.items():
        print("  ", k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in cs.items()))
PY
rm -rf $O
