cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
for w in ${WORKLOADS:-neumf_ml1m mf_ml1m deepfm_criteo deepfm_criteo_bf16 dcnv2_criteo fm_wide_d128 fm_criteo_sharded deepfm_criteo_sharded}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$w -- python3 $R/bench.py --workload $w --steps 40 --warmup 16 --repeats 1 --no-cpu-baseline --extra none > $R/gpurun_out/ks_$w.log 2>&1 || exit 1
  cp $(ls $R/gpurun_out/ks_$w/*/*kernel_stats.csv | head -n 1) $R/gpurun_out/stats_$w.csv
  tail -n 1 $R/gpurun_out/ks_$w.log | cut -c1-200 > $R/gpurun_out/line_$w.txt
  rm -rf $R/gpurun_out/ks_$w
  echo done $w
done
