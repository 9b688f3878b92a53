#!/bin/bash
# round 4, closing run: the whole -m gpu suite, smoke, the default bench line
python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gputests.log
tail -4 gpurun_out/r04_gputests.log
python3 -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python3 scripts/show_bench.py < gpurun_out/r04_bench_default.json
python3 bench.py --gpus 2 --size 256 --steps 3 --warmup 1 --pencil-extra off > gpurun_out/r04_bench_2ranks.json 2> gpurun_out/r04_bench_2ranks.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_bench_2ranks.json'))
print('2 ranks:', d['value'], d['config']['exchange_transport'], d['cpu_baseline'])"
