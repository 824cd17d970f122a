# Kernel timeline of replayed fm_criteo steps (graph segments of 10): rocprofv3 --kernel-trace of a short bench
# run, then profiles/timeline.py over a slice in the middle of the timed region.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r3 && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --repeats 2 --extra none --no-cpu-baseline > $O/bench_under_trace.log 2>&1 || exit 1
python3 $R/profiles/timeline_compact.py $(ls $O/kt/*/*kernel_trace.csv | head -n 1) ${1:-130} ${2:-260} > $O/fm_criteo_graph_timeline.txt
rm -rf $O/kt
echo done
