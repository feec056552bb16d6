#!/bin/bash
# Round 5's same-box A/B records (run through gpurun from the repo root; outputs under gpurun_out/r05/ab/, copied to profiles/r05_*.txt).
# Needs tools/build/libdisco_tl.so (conv_mx_ar2/ar3 built with -DMX_TIMELINE=1: tools/conv_timeline.py) for the two timeline records.
R=$PWD
O=$R/gpurun_out/r05/ab
mkdir -p $O
OFF="DISCO_CONV_LAT=0 DISCO_KMEANS_V1=1 DISCO_ENCODER_TAIL_ROWS=0"
{
  echo "# tools/small_batch_latency.py on ONE box: wall time of a synchronised forward / of back-to-back forwards, and a hash of the six outputs."
  echo "# 'r5 switches off' = the round-5 kernels disabled by their A/B switches (DISCO_CONV_LAT=0: round 4's tiles and loop; DISCO_KMEANS_V1=1:"
  echo "# the general k-means kernel; DISCO_ENCODER_TAIL_ROWS=0: 64-row encoder tiles + a q/k/v launch per layer); the remaining difference to"
  echo "# that line is what has no switch (kernel-argument prefetch, branch-free DMA offsets, upfeat / conv_c1 / attention small-grid forms)."
  for rep in 1 2; do
    echo "== all round-5 switches off (pass $rep)"; env $OFF python tools/small_batch_latency.py 2>&1 | grep "n="
    echo "== only the conv latency loop on (pass $rep)"; env DISCO_KMEANS_V1=1 DISCO_ENCODER_TAIL_ROWS=0 python tools/small_batch_latency.py 2>&1 | grep "n="
    echo "== latency loop + k-means kernels (pass $rep)"; env DISCO_ENCODER_TAIL_ROWS=0 python tools/small_batch_latency.py 2>&1 | grep "n="
    echo "== default (everything on) (pass $rep)"; python tools/small_batch_latency.py 2>&1 | grep "n="
  done
} > $O/latency_loop_ab.txt
{
  echo "# tools/operating_points.py on ONE box: round 4's library (tools/build/libdisco_r04.so, built from commit 78da8e6) and this round's"
  echo "== round 4"; DISCO_HIP_LIB=tools/build/libdisco_r04.so python tools/operating_points.py 2>&1 | grep " x "
  echo "== round 5"; python tools/operating_points.py 2>&1 | grep " x "
} > $O/operating_points_ab.txt
{
  echo "# bench.py (headline, 64 x 256x256, pipelined) on ONE box, alternating: value img/s, ms per step, single_image_latency_ms, stage table"
  for rep in 1 2; do
    for LIB in tools/build/libdisco_r04.so ""; do
      if [ -n "$LIB" ]; then export DISCO_HIP_LIB=$LIB; else unset DISCO_HIP_LIB; fi
      python bench.py --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${LIB:-round 5}', d['value'], d['ms_per_step'], d['single_image_latency_ms'], d['stage_ms_per_step'], d['result_checksum'])"
    done
  done
} > $O/headline_ab.txt
{
  echo "# tools/kmeans_latency.py: the k-means + anchors launch on one image's tokens; general kernel (DISCO_KMEANS_V1=1) vs kmeans_small / kmeans_tiled"
  echo "== DISCO_KMEANS_V1=1"; DISCO_KMEANS_V1=1 python tools/kmeans_latency.py 2>&1 | grep "L="
  echo "== default"; python tools/kmeans_latency.py 2>&1 | grep "L="
} > $O/kmeans_latency_ab.txt
{
  echo "# forward time and stage split with the encoder layers' second half on 64-row tiles (DISCO_ENCODER_TAIL_ROWS=0) and on 16-row tiles (default)"
  echo "== DISCO_ENCODER_TAIL_ROWS=0"; DISCO_ENCODER_TAIL_ROWS=0 python tools/forward_points.py 2>&1 | grep " x "
  echo "== default"; python tools/forward_points.py 2>&1 | grep " x "
} > $O/encoder_tail_ab.txt
if [ -f tools/build/libdisco_tl.so ]; then
  {
    echo "# tools/conv_timeline.py --n 1 --raw: s_memtime stamps of workgroup (0,0), one image; first with round 4's loop (DISCO_CONV_LAT=0), then the latency loop"
    echo "# (the latency loop's second trace is a PRODUCER wave: p14/p15 = its first weight / pixel pieces issued)"
    echo "== DISCO_CONV_LAT=0"
    DISCO_CONV_LAT=0 DISCO_HIP_LIB=tools/build/libdisco_tl.so python tools/conv_timeline.py --x3 --n 1 --raw 512 2>&1 | grep -v amdgpu
    DISCO_CONV_LAT=0 DISCO_HIP_LIB=tools/build/libdisco_tl.so python tools/conv_timeline.py "256->256" --n 1 --raw 2>&1 | grep -v amdgpu
    echo "== latency loop"
    DISCO_HIP_LIB=tools/build/libdisco_tl.so python tools/conv_timeline.py --x3 --n 1 --raw 512 2>&1 | grep -v amdgpu
    DISCO_HIP_LIB=tools/build/libdisco_tl.so python tools/conv_timeline.py "256->256" --n 1 --raw 2>&1 | grep -v amdgpu
  } > $O/conv_timeline_n1.txt
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg1 -- python $R/tools/trace_gaps.py --run --batch 1 > /tmp/tg_run.txt 2>&1
(cd $R; { echo "# rocprofv3 --kernel-trace -- python tools/trace_gaps.py --run --batch 1 ; python tools/trace_gaps.py --report DIR   (round 5 build, one 256x256 image per forward)"; python tools/trace_gaps.py --report /tmp/tg1; } > $O/trace_gaps_n1.txt 2>&1)
ls -la $O
