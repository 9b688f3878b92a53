#!/bin/bash
# run-to-run spread of the stage times of the 1024^3 pair over fresh processes
out=gpurun_out/bimodal_r02.txt
: > $out
for i in $(seq 1 ${1:-16}); do
  python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); s=d['config']['stage_ms']
        print('%.2f ms/pair  ' % d['ms_per_step'] + '  '.join('%s %.3f' % (k, v) for k, v in sorted(s.items())))
" >> $out
done
cat $out
