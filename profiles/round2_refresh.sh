# Refresh of every committed profile on the final tree (run on the GPU box from the repository root):
# kernel stats + dominant-op durations of the default bench, the two PMC traffic passes, kernel stats of
# the other workloads, MFMA counters of the tower workloads.  Results land under gpurun_out/.
set -e
bash profiles/round2_profile.sh stats
bash profiles/round2_profile.sh pmc
WORKLOADS="neumf_ml1m mf_ml1m deepfm_criteo deepfm_criteo_bf16 dcnv2_criteo fm_wide_d128 fm_criteo_sharded" bash profiles/profile_workloads.sh
bash profiles/pmc_mfma.sh
