# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun); results land in gpurun_out/prof_*.
#   1. kernel trace + stats of the default bench command (1 step = 32 micro-batches) -> kernel_stats.csv
#   2. separate PMC passes (FETCH_SIZE, WRITE_SIZE) restricted to the two roofline kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out; mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"   # default workload: global batch 1024 = 32 micro-batches per step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $CMD > $OUT/prof_kt.log 2>&1
cp $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $OUT/prof_kernel_stats.csv
cp $(ls /tmp/prof_kt/*/*agent_info.csv | head -1) $OUT/prof_agent_info.csv
tail -2 $OUT/prof_kt.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "bnact_bwd_k<true>|gemm_kernel<128, 128, 64, 2, 2, 0, 0, false, true>" --output-format csv -d /tmp/prof_$ctr -- $CMD > $OUT/prof_$ctr.log 2>&1
  cp $(ls /tmp/prof_$ctr/*/*counter_collection.csv | head -1) $OUT/prof_pmc_$ctr.csv
done
head -12 $OUT/prof_kernel_stats.csv
