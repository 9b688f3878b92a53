#!/bin/bash
# rocprofv3 passes of round 6 (run on the GPU box through gpurun; every command under its own timeout): kernel trace + separate
# PMC passes of the bench command, and FETCH / WRITE passes of the fused solver (the traffic of the nonlinear z kernel).
# Summaries are written by scripts/summarize_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $BENCH > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $BENCH > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $BENCH > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- python3 $BENCH > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $BENCH > $O/sq2.log 2>&1
DNS="$R/examples/spectral_dns_device.py --M 9 --steps 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dns_trace -- python3 $DNS > $O/dns_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dns_fetch -- python3 $DNS > $O/dns_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dns_write -- python3 $DNS > $O/dns_write.log 2>&1
cd $R
tail -2 $O/*.log
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
