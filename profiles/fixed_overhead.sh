# fm_criteo at several --steps values, one-step graphs and ten-step graphs: shows the fixed cost of a timed region.
for cfg in "RX_BENCH_NO_CYCLE=1 20" "RX_BENCH_NO_CYCLE=1 100" "RX_X=1 10" "RX_X=1 20" "RX_X=1 40" "RX_X=1 100"; do
  set -- $cfg
  env $1 timeout -k 10 200 python -u bench.py --extra none --no-cpu-baseline --steps $2 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],4), d['timed_regions']['ms_per_step'])" || exit 1
done
