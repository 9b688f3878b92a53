#!/bin/bash
# round 5, closing run (the end-of-round script): 1. the perf gate -- a size sweep of THIS binary compared stage by stage with
# the last committed sweep; its output becomes profiles/r05_size_sweep.txt, the only source of the numbers README / DESIGN
# quote -- 2. rocprofv3 passes of the headline (kernel trace + PMC), 3. the whole -m gpu suite with durations, the slow
# parametrisations, smoke, 4. the default bench line and a 2-rank line.
STEP=${1:-all}
if [ "$STEP" = all ] || [ "$STEP" = gate ]; then
  python3 scripts/perf_gate.py --baseline profiles/r04_size_sweep.txt --out gpurun_out/r05_size_sweep.txt > gpurun_out/r05_perf_gate.log 2>&1
  echo "perf gate rc $?" >> gpurun_out/r05_perf_gate.log
  python3 scripts/perf_gate.py --baseline profiles/r04_radix7_sweep.txt --sizes 1792 --precisions fp32 --out gpurun_out/r05_size_sweep_extra.txt >> gpurun_out/r05_perf_gate.log 2>&1
  tail -45 gpurun_out/r05_perf_gate.log
  for prec in double single; do python3 scripts/padprof.py 1024 slab $prec; done > gpurun_out/r05_padded_pair.txt 2>&1
  python3 scripts/padprof.py 1024 X double >> gpurun_out/r05_padded_pair.txt 2>&1
  cat gpurun_out/r05_padded_pair.txt
fi
if [ "$STEP" = all ] || [ "$STEP" = prof ]; then
  bash scripts/profile_r05.sh bench > gpurun_out/r05_profile.log 2>&1
  python3 scripts/summarize_profiles.py r05_final gpurun_out/prof_r05/trace gpurun_out/prof_r05/fetch gpurun_out/prof_r05/write "bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off: 1024^3 fp64 slab R2C forward+inverse on one MI355X" > /dev/null
  python3 scripts/summarize_profiles.py sq r05_final gpurun_out/prof_r05/sq1 gpurun_out/prof_r05/sq2 > /dev/null
  mkdir -p gpurun_out/r05_profiles_out; cp profiles/r05_final_* gpurun_out/r05_profiles_out/
  rm -rf gpurun_out/prof_r05
  ls gpurun_out/r05_profiles_out
fi
if [ "$STEP" = all ] || [ "$STEP" = tests ]; then
  python3 -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05_gputests_final.log 2>&1
  echo "pytest rc $?" >> gpurun_out/r05_gputests_final.log
  tail -18 gpurun_out/r05_gputests_final.log
  MFFT_TEST_SLOW=1 timeout 900 python3 -m pytest tests -m "gpu and slow" -x -q > gpurun_out/r05_gputests_slow.log 2>&1
  echo "pytest (slow) rc $?" >> gpurun_out/r05_gputests_slow.log
  tail -3 gpurun_out/r05_gputests_slow.log
  python3 -c "import __graft_entry__ as g; g.smoke()"
fi
if [ "$STEP" = all ] || [ "$STEP" = bench ]; then
  python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
  python3 scripts/show_bench.py < gpurun_out/r05_bench_default.json
  python3 bench.py --gpus 2 --size 256 --steps 3 --warmup 1 --pencil-extra off > gpurun_out/r05_bench_2ranks.json 2> gpurun_out/r05_bench_2ranks.err
  python3 -c "
import json
d=json.load(open('gpurun_out/r05_bench_2ranks.json'))
print('2 ranks:', d['value'], d['config']['exchange_transport'], d['cpu_baseline'])"
fi
