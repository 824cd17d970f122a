# Kernel stats + a timeline slice of the graph-replayed fm_wide_d128 step (and the one-rank sharded FM with WL=fm_criteo_sharded)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r3 && mkdir -p $O
WL=${1:-fm_wide_d128}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wk -- python3 $R/bench.py --workload $WL --child-entry --extra none --no-cpu-baseline > $O/${WL}_under_trace.log 2>&1 || exit 1
cp $(ls $O/wk/*/*kernel_stats.csv | head -n 1) $O/${WL}_kernel_stats.csv
python3 $R/profiles/timeline.py $(ls $O/wk/*/*kernel_trace.csv | head -n 1) 60 fm2_fwd:150 > $O/${WL}_graph_timeline.txt
rm -rf $O/wk
