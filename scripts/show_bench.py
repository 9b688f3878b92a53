import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(d["config"]["workload"][:40], "| ms/pair", round(d["ms_per_step"], 2), "| pairs/s", round(d["value"], 1),
          "| frac8TB", round(d["config"]["whole_path_frac_of_8TBs"], 3),
          {k: round(v, 2) for k, v in d["config"]["stage_ms"].items()}, "rt", "%.1e" % d["config"]["roundtrip_rel_l2"])
