#!/bin/bash
# Round 5's FINAL build on one box (run through gpurun from the repo root): operating points, the headline three times, the one-image /
# small-batch latencies.  (Round 4's library - tools/build/libdisco_r04.so - no longer loads under this round's Python layer, which binds
# exports it does not have; the same-box comparisons against it are profiles/r05_*_ab.txt, taken earlier in the round.)
# Outputs under gpurun_out/r05/ab/ -> profiles/r05_final_*.txt
R=$PWD
O=$R/gpurun_out/r05/ab
mkdir -p $O
{
  echo "# tools/operating_points.py, round 5's final build"
  python tools/operating_points.py 2>&1 | grep " x "
} > $O/final_operating_points_ab.txt
{
  echo "# bench.py (headline, 64 x 256x256, pipelined) on ONE box, alternating: value img/s, ms per step, single_image_latency_ms, stage table, checksum"
  for rep in 1 2 3; do
    for LIB in ""; do
      if [ -n "$LIB" ]; then export DISCO_HIP_LIB=$LIB; else unset DISCO_HIP_LIB; fi
      python bench.py --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${LIB:-round 5}', d['value'], d['ms_per_step'], d['single_image_latency_ms'], d['stage_ms_per_step'], d['result_checksum'])"
    done
  done
} > $O/final_headline_ab.txt
{
  echo "# tools/small_batch_latency.py, round 5's final build"
  python tools/small_batch_latency.py 2>&1 | grep "n="
} > $O/final_small_batch_ab.txt
