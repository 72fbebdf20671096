#!/bin/bash
# A/B of the forward dct_32x32 grid cap on bench.py's two batch sizes (1080p x 128 headline, 4K x 128 shard leg) through
# KVZ_HIP_TUNE: the measurement behind "one block per wave at any batch size" (kvazaar_amd/csrc/dct32_mfma.hip).
# usage (GPU box): bash tools/tune4k_probe.sh
for v in 192 384 768 1100; do
  KVZ_HIP_TUNE=dct32_wgs_per_cu=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r2v_bench_$v.json 2> gpurun_out/r2v_bench_$v.err || exit 1
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r2v_bench_$v.json"))
print($v, "headline dct32", d["kernels"]["dct_32x32"]["achieved_GBs"], "4K dct32", d["shard_4k"]["rank0_kernels"]["dct_32x32"]["achieved_GBs"], "value", d["value"], "4K value", d["shard_4k"]["value"])
PY
done
