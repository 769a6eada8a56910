#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" -- <program args...>   (counters in their own run, no tracing domains beyond kernel-trace)
tag=$1; ctrs=$2; shift 3
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$tag
timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o $tag -- "$@" > gpurun_out/pmc_$tag/run.log 2>&1
echo "pmc $tag exit $?"
python3 - "$tag" <<'PY'
import csv, sys, glob, collections
tag = sys.argv[1]
for f in glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "deform" not in k and "lu_" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
