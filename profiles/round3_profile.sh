# Profiles of the final tree (run on the GPU box from the repository root; results under gpurun_out/r3/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command -> kernel stats + the dominant op's
#      back-to-back launch durations
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only) of the eager fm_criteo step -> pmc_traffic.json
# Usage: bash profiles/round3_profile.sh [stats|pmc]
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r3 && mkdir -p $O
what=${1:-stats}
if [ "$what" = stats ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --extra none --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1 || exit 1
  cp $(ls $O/ks/*/*kernel_stats.csv | head -n 1) $O/fm_criteo_default_bench_kernel_stats.csv
  python3 $R/profiles/dominant_from_trace.py $(ls $O/ks/*/*kernel_trace.csv | head -n 1) "fg_reduce_kernel<4, 8, 5, false>" "fg_combine_kernel<4, 8, 5>" > $O/fm_criteo_dominant_kernel_durations.txt || exit 1
  rm -rf $O/ks
  tail -n 1 $O/bench_under_rocprof.log | cut -c1-400
  cat $O/fm_criteo_dominant_kernel_durations.txt
else
  for c in FETCH_SIZE WRITE_SIZE; do
    # few resident batches: under --pmc every dispatch is serialised and takes about a millisecond
    RX_BENCH_BATCHES=4 timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --gpus 1 --steps 4 --warmup 1 --repeats 1 --no-graph --extra none --no-cpu-baseline > $O/pmc_$c.log 2>&1 || exit 1
    echo pass $c done
  done
  python3 $R/profiles/pmc_traffic.py $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -n 1) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -n 1) > $O/fm_criteo_eager_pmc_traffic.txt || exit 1
  cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
  cat $O/fm_criteo_eager_pmc_traffic.txt
fi
