#!/bin/bash
set -u
mkdir -p gpurun_out/r04
for c in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --alt-eval-cus $c > gpurun_out/r04/bench_alt_$c.json 2> gpurun_out/r04/bench_alt_$c.err || { tail -5 gpurun_out/r04/bench_alt_$c.err; exit 1; }
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
d=json.loads(open(f"gpurun_out/r04/bench_alt_{c}.json").read().strip().splitlines()[-1])
print(c, "value", round(d["value"]), "alternative", d["alternative"] and {k: (round(v) if k=="value" else v) for k, v in d["alternative"].items() if k != "build"})
PY
done
