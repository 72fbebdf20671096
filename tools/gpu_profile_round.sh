#!/bin/bash
# End-of-work profile set of one build, written under gpurun_out/TAG_* (copy what is to be judged into profiles/):
#   0. bench.py with the driver's exact command, `python3 bench.py --gpus 1 --steps 20 --warmup 5` (no profiler): TAG_bench.json
#   1. the same command with the legs that launch the same kernels on other batches switched off (--no-shard-leg
#      --no-reference-workload --no-encoder-leg) under rocprofv3 --kernel-trace --stats: TAG_bench_profiled.json + TAG_kernel_stats.csv,
#      so that a kernel's average in the CSV is the average of the launches the bench line's roofline describes; with every leg on:
#      TAG_kernel_stats_all_legs.csv; the kernel trace of the headline leg (start / end of every launch): TAG_kernel_trace.csv
#   2. bench.py headline leg under --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -> TAG_pmc_traffic.json
#   3. the fused quantize_residual kernels under the SQ / TCC counter groups -> TAG_qr_pmc.txt
#   4. the frame-level sampling / SATD kernels under the same groups -> TAG_frame_kernels_pmc.txt
#   5. every entry of the ABI: TAG_bench_all_kernels.txt
#   6. (third argument `all`) front replay, frame pipeline, served encodes: TAG_front_replay.json, TAG_front_replay_kernel_stats.csv,
#      TAG_frame_pipeline.txt, TAG_gpu_served_encode.txt
# The program always follows `--` directly (no env / bash -c hop).  usage: tools/gpu_profile_round.sh TAG COMMIT [all | headline | service]
set -e
TAG=${1:-r02}
COMMIT=${2:-unknown}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O/${TAG}_prof
if [ "${3:-}" != "service" ]; then       # `service`: step 7 only (the search service and the served encodes)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_prof/bench.log
ONLY="--no-shard-leg --no-reference-workload --no-encoder-leg"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof/stats -o p --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $ONLY > $O/${TAG}_bench_profiled.json 2> $O/${TAG}_prof/stats.log
cp $O/${TAG}_prof/stats/p_kernel_stats.csv $O/${TAG}_kernel_stats.csv
python3 tools/trace_launches.py $O/${TAG}_prof/stats/p_kernel_trace.csv > $O/${TAG}_kernel_trace_summary.txt 2>&1 || true
rm -rf $O/${TAG}_prof/stats
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof/stats2 -o p --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-encoder-leg > $O/${TAG}_bench_profiled_all_legs.json 2> $O/${TAG}_prof/stats2.log
cp $O/${TAG}_prof/stats2/p_kernel_stats.csv $O/${TAG}_kernel_stats_all_legs.csv
rm -rf $O/${TAG}_prof/stats2
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $ONLY"
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_prof/fetch -o p --output-format csv -- $B > $O/${TAG}_prof/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_prof/write -o p --output-format csv -- $B > $O/${TAG}_prof/write.log 2>&1
for p in fetch write; do python3 tools/pmc_filter.py $O/${TAG}_prof/$p sad_nxn_kernel,satd8_kernel,dct32_mfma_kernel; done
python3 tools/pmc_traffic.py $O/${TAG}_prof/fetch $O/${TAG}_prof/write $O/${TAG}_pmc_traffic.json $COMMIT > /dev/null
if [ "${3:-}" = "headline" ]; then echo "profile set $TAG done (headline part)"; exit 0; fi    # steps 0-2 only: after a change that leaves the other kernels alone
tools/gpu_pmc_qr.sh ${TAG}
# frame-level kernels: sampling and descriptor SATD
F="python3 tools/bench_all.py --only sample_luma,image_satd --rounds 1"
KS=sample_small_kernel,sample_big_kernel,pair_satd
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/${TAG}_prof/f1 -o p --output-format csv -- $F > $O/${TAG}_prof/f1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $O/${TAG}_prof/f2 -o p --output-format csv -- $F > $O/${TAG}_prof/f2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_prof/f3 -o p --output-format csv -- $F > $O/${TAG}_prof/f3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_prof/f4 -o p --output-format csv -- $F > $O/${TAG}_prof/f4.log 2>&1
for p in f1 f2 f3 f4; do python3 tools/pmc_filter.py $O/${TAG}_prof/$p $KS; done
python3 tools/pmc_kernels.py $O/${TAG}_frame_kernels_pmc.txt $KS $O/${TAG}_prof/f1 $O/${TAG}_prof/f2 $O/${TAG}_prof/f3 $O/${TAG}_prof/f4 > /dev/null
python3 tools/bench_all.py > $O/${TAG}_bench_all_kernels.txt 2>&1
#   6. searches recorded from a real encode, front by front (+ S host threads), with the kernel statistics of that loop;
#      one 1080p frame of every stage; the GPU-served encodes of the reference encoder (their printed summaries)
if [ "${3:-}" = "all" ]; then
  R=$PWD
  python3 tools/front_replay.py --repeats 3 --sessions 2,4,8,16 --merged 2,8,32,128,512 --keep-case /tmp/kvz_case0.bin > $O/${TAG}_front_replay.json 2> $O/${TAG}_prof/front_replay.err
  python3 tools/frame_pipeline.py > $O/${TAG}_frame_pipeline.txt 2>&1
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${TAG}_prof/fr -o p -- $R/tools/front_replay /tmp/kvz_case0.bin 2 1 > $R/$O/${TAG}_prof/front_profiled.json 2> $R/$O/${TAG}_prof/front_prof.err)
  cp $O/${TAG}_prof/fr/p_kernel_stats.csv $O/${TAG}_front_replay_kernel_stats.csv
  rm -rf $O/${TAG}_prof/fr /tmp/kvz_case0.bin
  python3 -m pytest tests/test_gpu_dropin.py -m gpu -q -s -k "served or deblocked_by_one" 2>&1 | grep -E "frames: |frames, untouched|passed|failed" > $O/${TAG}_gpu_served_encode.txt
fi
fi
#   7. the search service: C pthread hosts hammering it, and the reference encoder with its own thread pool served by it /
#      with its SADs answered from tables, each next to the untouched encoder at the same thread count
(for t in 1 4 16 48; do tests/c_host/service_stress $t 1500; done) > $O/${TAG}_service_stress.txt 2>&1 || true
# 16 closed-loop C threads at 1080p, four pictures: resident workers against a launch per batch, without / with 60 us of work between a thread's requests
(export KVZ_HIP_SERVICE_DEBUG=1; for wk in 64 0; do for think in 0 60; do echo "== service_workers=$wk think_us=$think"; KVZ_HIP_TUNE=service_workers=$wk tests/c_host/service_stress 16 8000 1920 1080 4 $think | grep -v "^single"; done; done) > $O/${TAG}_service_stress_1080p.txt 2>&1 || true
python3 tools/service_latency.py --algos hexbs,full8,full16,full32 > $O/${TAG}_service_latency_by_size.jsonl 2> $O/${TAG}_prof/lat.err || true
python3 tools/served_encode.py --size 1920x1080 --frames 16 --threads 16 --min-size 8,16,32,64 --tables 16 --probe --upload-only > $O/${TAG}_served_encode_1080p_medium.jsonl 2> $O/${TAG}_prof/served1.err || true
KVZ_HIP_TUNE=service_workers=0 python3 tools/served_encode.py --size 1920x1080 --frames 16 --threads 16 --min-size 8,32 > $O/${TAG}_served_encode_1080p_medium_launches.jsonl 2> $O/${TAG}_prof/served1b.err || true
KVZ_HIP_SERVICE_DEBUG=1 python3 tools/served_encode.py --size 1920x1080 --frames 16 --opts preset=medium,qp=32,me=full16 --threads 16 --min-size 8,16 --tables 16 --probe > $O/${TAG}_served_encode_1080p_full16.jsonl 2> $O/${TAG}_served_encode_1080p_full16_worker_stats.txt || true
KVZ_HIP_TUNE=service_workers=0 python3 tools/served_encode.py --size 1920x1080 --frames 16 --opts preset=medium,qp=32,me=full16 --threads 16 --min-size 8 > $O/${TAG}_served_encode_1080p_full16_launches.jsonl 2> $O/${TAG}_prof/served2b.err || true
python3 tools/served_encode.py --size 1920x1080 --frames 16 --opts preset=medium,qp=32,me=full16 --threads 32 --min-size 8 > $O/${TAG}_served_encode_1080p_full16_32threads.jsonl 2> $O/${TAG}_prof/served2d.err || true
python3 tools/served_encode.py --size 1920x1080 --frames 32 --threads 32 --min-size 8,16,32 > $O/${TAG}_served_encode_1080p_medium_32threads.jsonl 2> $O/${TAG}_prof/served1c.err || true
python3 tools/served_encode.py --size 1920x1080 --frames 4 --opts preset=medium,qp=32,me=full32 --threads 16 --min-size 8 > $O/${TAG}_served_encode_1080p_full32.jsonl 2> $O/${TAG}_prof/served2c.err || true
python3 tools/served_encode.py --size 3840x2160 --frames 8 --threads 16 --min-size 8,32,64 --probe > $O/${TAG}_served_encode_4k_medium.jsonl 2> $O/${TAG}_prof/served3.err || true
python3 tools/served_encode.py --size 3840x2160 --frames 8 --opts preset=medium,qp=32,me=full16 --threads 16 --min-size 8 > $O/${TAG}_served_encode_4k_full16.jsonl 2> $O/${TAG}_prof/served4.err || true
echo "profile set $TAG done"
