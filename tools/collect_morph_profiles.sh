#!/bin/bash
# Morph-space passes (N = 1M, S = 50): kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in their
# own runs (--kernel-trace only).  Output under gpurun_out/profiles_morph/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/profiles_morph
rm -rf $OUT; mkdir -p $OUT
python tools/morph_bench.py 1000000 16,50,100 > $OUT/morph_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o morph -- python tools/morph_bench.py 1000000 50 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python tools/morph_bench.py 1000000 50 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python tools/morph_bench.py 1000000 50 > $OUT/write.log 2>&1
grep -v amdgpu.ids $OUT/morph_bench.txt
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/profiles_morph"
for name in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "morph" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:42s} {c:12s} mean {sum(v)/len(v):14.1f} KiB  n={len(v)}")
PY
