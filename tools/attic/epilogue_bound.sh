#!/bin/bash
# What could ANY scheme that hides the conv epilogue under the next tile's taps (verdict r4 item 6: two half-workgroups out of phase; or a second
# accumulator set with the epilogue sliced between the taps) win at best?  Hiding cannot beat REMOVING: an ablation build without the epilogue
# (tools/build_ablations.sh 32: no math, no stores; results wrong by design, timing only) against the kernel as built, N = 64, both arithmetics.
mkdir -p gpurun_out/r05
for rep in 1 2; do
  for lib in "" tools/build/libdisco_abl32.so; do
    echo "== ${lib:-as built} (pass $rep)"
    for shape in "64->64 @256" "128->128 @128" "256->256 @64" "512->512 @32"; do
      DISCO_HIP_LIB=$lib python tools/bench_conv.py --only "$shape" --iters 30 2>&1 | grep "@" | grep -v "^up\|^cat\|^s2" | sed 's/$/   [f16x3]/'
      DISCO_HIP_LIB=$lib python tools/bench_conv.py --mx 6 --only "$shape" --iters 30 2>&1 | grep "@" | grep -v "^up\|^cat\|^s2"
    done
  done
done
