# bench.py over a few scheduling choices (same work per step): bash tools/bench_sweep.sh  [on a GPU box]
set -o pipefail
for args in "$@"; do
  echo "== $args"
  timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*32*1e3), 'us/group', d['phases_ms']['build_batch'], d['roofline']['avg_launch_ms'])" || exit 1
done
