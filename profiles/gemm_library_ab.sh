for wl in dcnv2_criteo deepfm_criteo_bf16 deepfm_wide_d128_sharded; do
  for env in "RX_X=0" "RX_NO_LIBRARY_GEMM=1" "RX_LT_NO_GATED=1" "RX_LT_NO_BWD_W=1"; do
    echo -n "$wl $env: "
    env $env timeout -k 10 300 python bench.py --workload $wl --child-entry --extra none --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))" || exit 1
  done
done
