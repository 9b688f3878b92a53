#!/bin/bash
# rocprofv3 passes (kernel trace, FETCH_SIZE, WRITE_SIZE, two SQ counter sets) of one python script of this repository,
# run on the GPU box through gpurun:   scripts/profile_cmd.sh <tag> <script relative to the repo> [args...]
# Every pass runs under `timeout` (PROFILE_TIMEOUT seconds, default 600): a hung profiler must not eat the lease.
# Output under gpurun_out/prof_<tag>/; scripts/summarize_profiles.py turns it into the files kept in profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
SCRIPT=$R/$1; shift
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout ${PROFILE_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $SCRIPT "$@" > $O/trace.log 2>&1
timeout ${PROFILE_TIMEOUT:-600} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $SCRIPT "$@" > $O/fetch.log 2>&1
timeout ${PROFILE_TIMEOUT:-600} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $SCRIPT "$@" > $O/write.log 2>&1
timeout ${PROFILE_TIMEOUT:-600} rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- python3 $SCRIPT "$@" > $O/sq1.log 2>&1
timeout ${PROFILE_TIMEOUT:-600} rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $SCRIPT "$@" > $O/sq2.log 2>&1
cd $R
tail -2 $O/trace.log
find $O -name "*.db" -delete
du -sh $O
