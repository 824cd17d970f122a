# Kernel stats + a serialized timeline slice of the one-rank sharded FM step (graph replay)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r3 && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sk -- python3 $R/bench.py --workload fm_criteo_sharded --child-entry --extra none --no-cpu-baseline > $O/fm_criteo_sharded_under_trace.log 2>&1 || exit 1
cp $(ls $O/sk/*/*kernel_stats.csv | head -n 1) $O/fm_criteo_sharded_kernel_stats.csv
python3 $R/profiles/timeline.py $(ls $O/sk/*/*kernel_trace.csv | head -n 1) 90 gather_multi_kernel:60 > $O/fm_criteo_sharded_graph_timeline.txt
rm -rf $O/sk
