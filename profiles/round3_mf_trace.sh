# Kernel stats + a timeline slice of the graph-replayed mf_ml1m step
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r3 && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mk -- python3 $R/bench.py --workload mf_ml1m --child-entry --extra none --no-cpu-baseline > $O/mf_under_trace.log 2>&1 || exit 1
cp $(ls $O/mk/*/*kernel_stats.csv | head -n 1) $O/mf_ml1m_kernel_stats.csv
python3 $R/profiles/timeline.py $(ls $O/mk/*/*kernel_trace.csv | head -n 1) 60 mf_fwd_bce_kernel:150 > $O/mf_ml1m_graph_timeline.txt
rm -rf $O/mk
