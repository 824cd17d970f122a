# MFMA counters of the dense-layer kernels: one rocprofv3 --pmc pass per workload (counters only,
# no other trace domain), folded by profiles/pmc_mfma.py.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
for w in dcnv2_criteo deepfm_criteo deepfm_criteo_bf16; do
  timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/mfma_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --extra none > $R/gpurun_out/mfma_$w.log 2>&1 || exit 1
  python3 $R/profiles/pmc_mfma.py $(ls $R/gpurun_out/mfma_$w/*/*counter_collection.csv | head -n 1) > $R/gpurun_out/mfma_$w.txt || exit 1
  rm -rf $R/gpurun_out/mfma_$w
  echo done $w
done
