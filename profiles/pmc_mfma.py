#!/usr/bin/env python3
"""Folds a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_*,
GRBM_GUI_ACTIVE) into one line per GEMM kernel, averaged per launch:

  elapsed us            GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz (the counter is summed over the XCDs: this
                        reproduces the kernel durations of the kernel-trace runs)
  MfmaUtil %            SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs): share of SIMD cycles
                        with the matrix pipe executing
  GFLOP, TFLOP/s        hardware-counted matrix flops = MOPS x 512 (one count = 512 flops; padded
                        tiles included, so above the algorithmic 2 M N K), and their rate over
                        `elapsed`; of the 2.5 PFLOP/s (bf16) / 157 TFLOP/s (fp32) dense peaks

    python profiles/pmc_mfma.py counter_collection.csv"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm" not in k:
        continue
    acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':50s} {'calls':>5s} {'elapsed us':>10s} {'MfmaUtil %':>10s} {'GFLOP bf16':>10s} {'GFLOP f32':>10s} {'TFLOP/s':>8s} {'% of peak':>9s}")
for k, d in sorted(acc.items()):
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    us = cyc / 2400.0
    mb = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    fb = avg.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512
    ff = avg.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512
    rate = (fb + ff) / (us * 1e-6) / 1e12 if us else 0.0
    peak = 2500.0 if fb >= ff else 157.0
    print(f"{k[:50]:50s} {len(next(iter(d.values()))):5d} {us:10.1f} {100 * mb / (cyc * 1024) if cyc else 0:10.1f} "
          f"{fb / 1e9:10.2f} {ff / 1e9:10.2f} {rate:8.1f} {100 * rate / peak:9.1f}")
