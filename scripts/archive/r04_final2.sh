#!/bin/bash
# round 4, final build: profiles of the headline (kernel trace + PMC passes), smoke, the default bench line, a 2-rank line
bash scripts/profile_r04.sh bench > gpurun_out/r04_profile2.log 2>&1
python3 scripts/summarize_profiles.py r04_final gpurun_out/prof_r04/trace gpurun_out/prof_r04/fetch gpurun_out/prof_r04/write "bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off: 1024^3 fp64 slab R2C forward+inverse on one MI355X" > /dev/null
python3 scripts/summarize_profiles.py sq r04_final gpurun_out/prof_r04/sq1 gpurun_out/prof_r04/sq2 > /dev/null
mkdir -p gpurun_out/r04_profiles_out; cp profiles/r04_final_* gpurun_out/r04_profiles_out/
python3 -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python3 scripts/show_bench.py < gpurun_out/r04_bench_default.json
python3 bench.py --gpus 2 --size 256 --steps 3 --warmup 1 --pencil-extra off > gpurun_out/r04_bench_2ranks.json 2> gpurun_out/r04_bench_2ranks.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_bench_2ranks.json'))
print('2 ranks:', d['value'], d['config']['exchange_transport'], d['cpu_baseline'])"
rm -rf gpurun_out/prof_r04
