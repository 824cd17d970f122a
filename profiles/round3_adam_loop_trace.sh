# Kernel stats of the NeuMF train loop with the reference pipeline's Adam (one graph launch per step)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r3 && mkdir -p $O
WL=${1:-neumf_ml1m_train_loop_adam}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ak -- python3 $R/bench.py --workload $WL --child-entry --extra none --no-cpu-baseline > $O/${WL}_under_trace.log 2>&1 || exit 1
cp $(ls $O/ak/*/*kernel_stats.csv | head -n 1) $O/${WL}_kernel_stats.csv
rm -rf $O/ak
