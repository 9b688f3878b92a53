#!/bin/bash
# rocprofv3 passes of round 5 (run on the GPU box through gpurun): kernel trace + separate PMC passes of the bench
# command, and of the data-movement workload.  Summaries are written by scripts/summarize_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r05
MODE=${1:-all}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" != "pack" ]; then
BENCH="$R/bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $BENCH > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $BENCH > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $BENCH > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- python3 $BENCH > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $BENCH > $O/sq2.log 2>&1
fi
PACK="$R/scripts/pack_workload.py 512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pack_trace -- python3 $PACK > $O/pack_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pack_fetch -- python3 $PACK > $O/pack_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pack_write -- python3 $PACK > $O/pack_write.log 2>&1
cd $R
tail -2 $O/*.log
# keep only the small files (csv summaries)
find $O -name "*.db" -delete
du -sh $O
