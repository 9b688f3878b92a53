import pytest

pytestmark = pytest.mark.gpu


def test_graft_smoke():
    import __graft_entry__ as g
    g.smoke()


def test_bench_one_gpu_line():
    """`python bench.py` on one GPU (a small cube, no CPU leg): one JSON line with the contract's keys, the roofline
    object measured live, the pencil and dealias extras, and a round trip inside the tolerance."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", "256", "--steps", "5", "--warmup", "2",
                        "--cpu-baseline", "off"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64" and not d.get("degraded")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["config"]["roundtrip_rel_l2"] < 1e-10
    ex = d["extras"]
    assert ex["pencil_R2CX"]["roundtrip_rel_l2"] < 1e-10
    dl = ex["dealias"]
    assert 0 < dl["ifftn_two_thirds_rule_ms"] and 0 < dl["ifftn_ms"] and 0 < dl["three_halves_rule_ifftn_fftn_pair_ms"]
