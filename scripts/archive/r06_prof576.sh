#!/bin/bash
# rocprofv3 passes (kernel trace, FETCH_SIZE, WRITE_SIZE; every pass under its own timeout) of the plain and 3/2-rule pairs of 576^3 fp64:
# the 864-point kernels of plans.h group T and the column-limited c2r kernel with its mirrors through LDS
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r06_576
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scripts/pitchprof.py 576 double none > $O/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/pitchprof.py 576 double none > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/pitchprof.py 576 double none > $O/write.log 2>&1
cd $R
python3 scripts/summarize_profiles.py r06_576 $O/trace $O/fetch $O/write "scripts/pitchprof.py 576 double none: plain and 3/2-rule pairs of 576^3 fp64 (padded 864^3) on one MI355X" > /dev/null
mkdir -p gpurun_out/r06/profiles_out; cp profiles/r06_576_* gpurun_out/r06/profiles_out/
find $O -name "*.db" -delete; rm -rf $O/trace $O/fetch $O/write
tail -2 $O/trace.log; ls gpurun_out/r06/profiles_out | grep 576
