#!/bin/bash
# round 6, closing run on the final build (every command under its own timeout): 1. the perf gate -- a size sweep of THIS binary
# against the previous round's committed sweep; its output becomes profiles/r06_size_sweep.txt, the only source of the per-size
# numbers README / DESIGN quote -- 2. rocprofv3 passes of the headline and of the fused solver, 3. the whole -m gpu suite, the slow
# parametrisations, smoke, 4. the default bench line and a 2-rank line (RCCL's report through the stand-in).
STEP=${1:-all}
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out/r06
if [ "$STEP" = all ] || [ "$STEP" = gate ]; then
  timeout 1700 python3 scripts/perf_gate.py --out gpurun_out/r06/size_sweep_final.txt > gpurun_out/r06/perf_gate_final.log 2>&1
  echo "perf gate rc $?" >> gpurun_out/r06/perf_gate_final.log
  tail -48 gpurun_out/r06/perf_gate_final.log
  timeout 300 python3 scripts/pitchprof.py 1024 double none auto > gpurun_out/r06/pitched_final.txt 2>&1
  cat gpurun_out/r06/pitched_final.txt
fi
if [ "$STEP" = all ] || [ "$STEP" = prof ]; then
  bash scripts/profile_r06.sh > gpurun_out/r06/profile.log 2>&1
  python3 scripts/summarize_profiles.py r06_final gpurun_out/prof_r06/trace gpurun_out/prof_r06/fetch gpurun_out/prof_r06/write "bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off: 1024^3 fp64 slab R2C forward+inverse on one MI355X" > /dev/null
  python3 scripts/summarize_profiles.py sq r06_final gpurun_out/prof_r06/sq1 gpurun_out/prof_r06/sq2 > /dev/null
  python3 scripts/summarize_profiles.py r06_dns gpurun_out/prof_r06/dns_trace gpurun_out/prof_r06/dns_fetch gpurun_out/prof_r06/dns_write "examples/spectral_dns_device.py --M 9 --steps 2: fused Taylor-Green RK4 loop, 512^3 fp64 3/2-rule, pitched spectra, one MI355X" > /dev/null
  mkdir -p gpurun_out/r06/profiles_out; cp profiles/r06_final_* profiles/r06_dns_* gpurun_out/r06/profiles_out/ 2>/dev/null
  rm -rf gpurun_out/prof_r06
  ls gpurun_out/r06/profiles_out
fi
if [ "$STEP" = all ] || [ "$STEP" = tests ]; then
  ( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r06/gputests_final.log 2>&1
  echo "pytest rc $?" >> gpurun_out/r06/gputests_final.log
  tail -22 gpurun_out/r06/gputests_final.log
  MFFT_TEST_SLOW=1 timeout 900 python3 -m pytest tests -m "gpu and slow" -x -q > gpurun_out/r06/gputests_slow.log 2>&1
  echo "pytest (slow) rc $?" >> gpurun_out/r06/gputests_slow.log
  tail -3 gpurun_out/r06/gputests_slow.log
  timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()"
fi
if [ "$STEP" = all ] || [ "$STEP" = bench ]; then
  timeout 900 python3 bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
  python3 scripts/show_bench.py < gpurun_out/r06/bench_default.json
  timeout 600 python3 bench.py --gpus 2 --size 256 --steps 3 --warmup 1 --pencil-extra off > gpurun_out/r06/bench_2ranks.json 2> gpurun_out/r06/bench_2ranks.err
  python3 -c "
import json
d=json.load(open('gpurun_out/r06/bench_2ranks.json'))
c=d['config']
print('2 ranks:', d['value'], c['exchange_transport'], {k: c.get(k) for k in ('devices','rccl_nranks','rccl_version','rccl_user_ranks','rccl_devices')})"
  # the same over the RCCL entry points (the shared-memory stand-in: librccl refuses two ranks on one device)
  /opt/rocm/bin/hipcc -O2 -fPIC -shared -std=c++17 tests/mock_rccl/mock_rccl.cpp -o tests/mock_rccl/libmockrccl.so -lrt 2>/dev/null
  if [ -f tests/mock_rccl/libmockrccl.so ]; then
    MFFT_RCCL_LIB=$PWD/tests/mock_rccl/libmockrccl.so MFFT_TRANSPORT=rccl timeout 600 python3 bench.py --gpus 2 --size 256 --steps 3 --warmup 1 --pencil-extra off --transport rccl > gpurun_out/r06/bench_2ranks_mock.json 2> gpurun_out/r06/bench_2ranks_mock.err
    python3 -c "
import json
d=json.load(open('gpurun_out/r06/bench_2ranks_mock.json'))
c=d['config']
print('2 ranks over the RCCL entry points (stand-in):', d['value'], c['exchange_transport'], {k: c.get(k) for k in ('devices','rccl_nranks','rccl_version','rccl_user_ranks','rccl_devices')})"
  fi
fi
