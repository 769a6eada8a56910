"""Run bench.py as the driver does (defaults, one GPU) and check the JSON line against the contract: the required keys,
the roofline and cpu_baseline blocks.  python tools/check_bench_line.py [extra bench args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def main():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        print(r.stderr[-2000:]); sys.exit(r.returncode)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"{len(lines)} JSON lines"
    d = json.loads(lines[0])
    missing = [k for k in REQUIRED if k not in d]
    assert not missing, missing
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]
    print({k: d[k] for k in REQUIRED if k not in ("config", "roofline")})
    print("roofline:", ro["bound"], ro.get("kernel"), round(ro["achieved"], 1), ro["unit"], "frac", round(ro["frac"], 3), "traffic", ro["traffic"],
          "launch ms", ro.get("avg_launch_ms"))
    if "cpu_baseline" in d:
        cb = d["cpu_baseline"]
        print("cpu_baseline:", round(cb["value"], 3), cb["unit"], "cores", cb["cores"], cb["kind"], "|", cb["sample"][:100])
    print("phases_ms:", d.get("phases_ms"))
    print("roofline_solve:", {k: d["roofline_solve"][k] for k in ("single_build_ms", "frac_single", "frac_batched")} if "roofline_solve" in d else None)
    print("workload:", d["config"]["workload"][:160], "...")


if __name__ == "__main__":
    main()
