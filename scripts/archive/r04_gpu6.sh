#!/bin/bash
# round 4: full suite on the build with the padded pipelines, then the rocprofv3 passes of the headline and of 720^3
python -m pytest tests -m gpu -q > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gputests.log
tail -4 gpurun_out/r04_gputests.log
bash scripts/profile_r04.sh > gpurun_out/r04_profile.log 2>&1
python3 scripts/summarize_profiles.py r04_final gpurun_out/prof_r04/trace gpurun_out/prof_r04/fetch gpurun_out/prof_r04/write "bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off: 1024^3 fp64 slab R2C forward+inverse on one MI355X" >> gpurun_out/r04_profile.log 2>&1
python3 scripts/summarize_profiles.py sq r04_final gpurun_out/prof_r04/sq1 gpurun_out/prof_r04/sq2 >> gpurun_out/r04_profile.log 2>&1
bash scripts/profile_cmd.sh b720 bench.py --size 720 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off >> gpurun_out/r04_profile.log 2>&1
python3 scripts/summarize_profiles.py r04_720 gpurun_out/prof_b720/trace gpurun_out/prof_b720/fetch gpurun_out/prof_b720/write "bench.py --size 720: 720^3 fp64 slab R2C forward+inverse on one MI355X" >> gpurun_out/r04_profile.log 2>&1
mkdir -p gpurun_out/r04_profiles_out; cp profiles/r04_final_* profiles/r04_720_* gpurun_out/r04_profiles_out/ 2>/dev/null
tail -20 gpurun_out/r04_profile.log
python3 bench.py --steps 10 --warmup 3 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python3 scripts/show_bench.py < gpurun_out/r04_bench_default.json
