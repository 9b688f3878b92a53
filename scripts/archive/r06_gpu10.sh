#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 300 python scripts/pitchprof.py 1024 double none auto >> gpurun_out/r06/pitched_ab.txt 2>&1
tail -4 gpurun_out/r06/pitched_ab.txt
timeout 900 python bench.py > gpurun_out/r06/bench2.json 2> gpurun_out/r06/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['extras'].get('pitched_spectrum')), json.dumps(d['extras'].get('dealias')))
print(json.dumps({k:(v.get('rk4_step_ms') if isinstance(v,dict) else v) for k,v in d['extras'].get('taylor_green_rk4',{}).items()}))
PY
