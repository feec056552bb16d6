#!/bin/bash
# What would ONE image win if the deep conv layers' accumulation chain were halved (two workgroups per tile, each summing half of the chunks, on the
# CUs a one-image launch leaves idle - DESIGN.md section 7 "Next" (i))?  Bound: an ablation build whose latency loop walks half the chunks of every
# chain of >= 16 chunks (tools/build_ablations.sh 64; results wrong by design, timing only): the launch time of one half, before the exchange of the
# partial sums.  One image per launch (bench_conv --n 1) and the per-layer table of a one-image forward.
for rep in 1 2; do
  for lib in "" tools/build/libdisco_abl64.so; do
    echo "== ${lib:-as built} (pass $rep): one image per launch"
    for shape in "512->512 @32" "256->256 @64"; do
      DISCO_HIP_LIB=$lib python tools/bench_conv.py --n 1 --only "$shape" --iters 200 2>&1 | grep "@" | grep -v "^up\|^cat\|^s2" | sed 's/$/   [f16x3]/'
      DISCO_HIP_LIB=$lib python tools/bench_conv.py --n 1 --mx 6 --only "$shape" --iters 200 2>&1 | grep "@" | grep -v "^up\|^cat\|^s2"
    done
  done
done
for lib in "" tools/build/libdisco_abl64.so; do
  echo "== ${lib:-as built}: one-image forward, conv launches"
  DISCO_HIP_LIB=$lib python tools/profile_layers.py --batch 1 2>&1 | grep "stage totals\|conv launches"
done
