# Round-6 rocprofv3 evidence (run on the GPU box through gpurun); everything lands in gpurun_out/prof_r6/.
#   usage: bash scripts/profile_round6.sh [stats|pmc|cfg4]   (default: stats = kernel trace + stats of one cfg3 step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/prof_r6; mkdir -p $OUT
MODE=${1:-stats}
C4="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-n8-load"
C3="python $R/bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-parity"
if [ "$MODE" = "stats" ] || [ "$MODE" = "all" ]; then
  # the default run (encoder chains on three streams: kernel durations are SHARED with concurrent launches) ...
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- $C3 --roofline-in-timed-region > $OUT/c3.log 2>&1
  cp $(ls /tmp/p_c3/*/*kernel_stats.csv | head -1) $OUT/cfg3_kernel_stats_streams3.csv
  cp $(ls /tmp/p_c3/*/*agent_info.csv | head -1) $OUT/agent_info.csv
  python $R/scripts/overlap_stats.py $(ls /tmp/p_c3/*/*kernel_trace.csv | head -1) > $OUT/cfg3_overlap_streams3.txt 2>&1
  # ... and the same command on ONE stream: every kernel alone on the GPU -- the durations bench.py's roofline objects quote
  MC_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3s0 -- $C3 > $OUT/c3_s0.log 2>&1
  cp $(ls /tmp/p_c3s0/*/*kernel_stats.csv | head -1) $OUT/cfg3_kernel_stats.csv
  python $R/scripts/overlap_stats.py $(ls /tmp/p_c3s0/*/*kernel_trace.csv | head -1) > $OUT/cfg3_overlap_one_stream.txt 2>&1
fi
if [ "$MODE" = "cfg4" ] || [ "$MODE" = "all" ]; then
  # one stream: the launch durations bench.py's roofline objects quote (its default run measures them in one-stream steps)
  MC_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- $C4 > $OUT/c4.log 2>&1
  cp $(ls /tmp/p_c4/*/*kernel_stats.csv | head -1) $OUT/cfg4_kernel_stats.csv
fi
if [ "$MODE" = "pmc" ] || [ "$MODE" = "all" ]; then
  export MC_STREAMS=0      # counters are attributed per dispatch: one kernel at a time on the GPU
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/p_pmc$i -- $C3 > $OUT/pmc$i.log 2>&1
    f=$(ls /tmp/p_pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "pass $i produced no counters"; tail -3 $OUT/pmc$i.log; continue; }
    python - "$f" "$OUT/cfg3_pmc_$(echo $grp | cut -d' ' -f1).csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f); w.writerow(['kernel', 'counter', 'launches', 'sum', 'avg'])
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
  done
fi
if [ "$MODE" = "pmc1" ]; then
  # round 6: the same one-stream cfg3 step in MBConv recompute mode 1 -- every narrow-input block runs the fused expand +
  # depthwise forward launch (mc_mbconv_xdw_fwd: the expanded tensor is not written / read in the forward, rebuilt once in the
  # backward): the forward arithmetic of every micro-batch of the headline run
  export MC_STREAMS=0
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3rc1 -- $C3 --recompute 1 > $OUT/c3rc1.log 2>&1
  cp $(ls /tmp/p_c3rc1/*/*kernel_stats.csv | head -1) $OUT/cfg3rc1_kernel_stats.csv
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/p1_pmc$i -- $C3 --recompute 1 > $OUT/pmc1_$i.log 2>&1
    f=$(ls /tmp/p1_pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "pass $i produced no counters"; tail -3 $OUT/pmc1_$i.log; continue; }
    python - "$f" "$OUT/cfg3rc1_pmc_$grp.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f); w.writerow(['kernel', 'counter', 'launches', 'sum', 'avg'])
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
  done
fi
if [ "$MODE" = "fwd" ]; then
  # round 6: HBM bytes and time of ONE graph-less train-mode forward of a 32-pair micro-batch (scripts/fwd_only_profile.py: the
  # first pass of the micro-batched headline step), with and without the fused expand + depthwise forward launch
  for xdw in 0 1; do
    MC_XDW=$xdw python $R/scripts/fwd_only_profile.py 4 > $OUT/fwd_xdw${xdw}_streams3.txt 2>&1
    MC_XDW=$xdw MC_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fwd$xdw -- python $R/scripts/fwd_only_profile.py 1 > $OUT/fwd_xdw${xdw}.log 2>&1
    cp $(ls /tmp/p_fwd$xdw/*/*kernel_stats.csv | head -1) $OUT/fwd${xdw}_kernel_stats.csv
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      MC_XDW=$xdw MC_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pf${xdw}_pmc$i -- python $R/scripts/fwd_only_profile.py 1 > $OUT/pmcf${xdw}_$i.log 2>&1
      f=$(ls /tmp/pf${xdw}_pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
      [ -z "$f" ] && { echo "pass $i produced no counters"; tail -3 $OUT/pmcf${xdw}_$i.log; continue; }
      python - "$f" "$OUT/fwd${xdw}_pmc_$grp.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f); w.writerow(['kernel', 'counter', 'launches', 'sum', 'avg'])
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
    done
  done
fi
if [ "$MODE" = "pmc4s" ]; then
  # !! DO NOT RUN without a short outer timeout.  Tried in round 5: the kernel-trace pass of this command completes, but BOTH
  # counter passes abort after ~1 minute with "HSA_STATUS_ERROR_INVALID_PACKET_FORMAT: The AQL packet is malformed" inside
  # rocprofv3's counter service and then hang until their timeout (2 x 20 GPU-minutes lost) -- the micro-batched step under
  # --pmc is not profilable on this stack (round 4 saw the cfg4 passes "not finish" for the same reason).  bench.py therefore
  # keeps labelling the cfg3-mix traffic (`roofline.traffic_source`).
  # HBM traffic on the DEFAULT run's launch mix, sampled: 128 pairs as 4 micro-batches with ONE kept graph = 4
  # graph-less forwards' worth of launches, 3 re-forwards with replayed statistics, 4 backwards -- the per-micro-batch launch mix
  # of the N = 1 run (25 of 32 micro-batches re-forwarded) at 1 / 8 of its launches, so a counter pass takes minutes
  C4S="python $R/bench.py --batch 128 --micro-batches 4 --keep-graphs 1 --steps 1 --warmup 1 --no-cpu-baseline --no-n8-load"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4s -- $C4S > $OUT/c4s.log 2>&1
  cp $(ls /tmp/p_c4s/*/*kernel_stats.csv | head -1) $OUT/cfg4_kernel_stats.csv
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/p4s_pmc$i -- $C4S > $OUT/pmc4s_$i.log 2>&1
    f=$(ls /tmp/p4s_pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "pass $i produced no counters"; tail -3 $OUT/pmc4s_$i.log; continue; }
    python - "$f" "$OUT/cfg4_pmc_$grp.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f); w.writerow(['kernel', 'counter', 'launches', 'sum', 'avg'])
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
  done
fi
if [ "$MODE" = "pmc4" ]; then
  # HBM traffic of the DEFAULT run's launch mix (cfg4: 32 micro-batches, re-forwards): FETCH_SIZE / WRITE_SIZE passes.
  # NOT part of `all`: with ~300 000 dispatches per run a counter pass does not finish in 20 minutes (tried in round 4: both
  # passes hit their timeouts) -- bench.py labels the cfg3-mix traffic instead (`roofline.traffic_source`)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- $C4 > $OUT/c4.log 2>&1
  cp $(ls /tmp/p_c4/*/*kernel_stats.csv | head -1) $OUT/cfg4_kernel_stats.csv
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 1200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/p4_pmc$i -- $C4 > $OUT/pmc4_$i.log 2>&1
    f=$(ls /tmp/p4_pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "pass $i produced no counters"; tail -3 $OUT/pmc4_$i.log; continue; }
    python - "$f" "$OUT/cfg4_pmc_$grp.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r['Counter_Value'])
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f); w.writerow(['kernel', 'counter', 'launches', 'sum', 'avg'])
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
  done
fi
if [ "$MODE" = "dw" ] || [ "$MODE" = "all" ]; then
  # SQ counters of the depthwise kernels (one launch per shape and kernel, scripts/pmc_dw.py): VALU activity, resident waves,
  # wait classes -- the evidence behind "VALU-issue bound" (marching 5x5) vs "four waves per SIMD" (lane = column form)
  bash $R/scripts/pmc_run.sh "python $R/scripts/pmc_dw.py" "dwconv" \
     "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
     "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
     "FETCH_SIZE" "WRITE_SIZE" > $OUT/dw_pmc.txt 2>&1
fi
ls -la $OUT
