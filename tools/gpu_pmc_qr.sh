#!/bin/bash
# rocprofv3 PMC passes over the fused quantize_residual kernels (tools/bench_all.py --only quantize_residual):
# one pass per counter group (8 SQ slots; FETCH_SIZE and WRITE_SIZE cannot share a pass), program directly after `--`.
# usage: tools/gpu_pmc_qr.sh TAG   -> gpurun_out/TAG_pmc/{sq1,sq2,sq3,fetch,write}, summary gpurun_out/TAG_qr_pmc.txt
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_pmc
mkdir -p $OUT
CMD="python3 tools/bench_all.py --only quantize_residual --rounds 1 --mb 512"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/sq1 -o p --output-format csv -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $OUT/sq2 -o p --output-format csv -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 -d $OUT/sq3 -o p --output-format csv -- $CMD > $OUT/sq3.log 2>&1 || echo "sq3 pass failed (counter name?)" >> $OUT/sq3.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- $CMD > $OUT/write.log 2>&1
for p in sq1 sq2 sq3 fetch write; do python3 tools/pmc_filter.py $OUT/$p quantize_residual; done
python3 tools/pmc_kernels.py gpurun_out/${TAG}_qr_pmc.txt quantize_residual $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/fetch $OUT/write > /dev/null
tail -n 60 gpurun_out/${TAG}_qr_pmc.txt
