#!/bin/bash
# Same-box A/B of the deferred conv epilogue (tools/build_conv_variants.sh): per-layer-shape times of both arithmetics, the parity tests of the
# conv kernel under the variant, and the headline with its result checksum (must equal the default build's: bit-identical by construction).
for rep in 1 2; do
  for v in "" defer defer_s0 defer_s2; do
    lib=${v:+tools/build/libdisco_conv_$v.so}
    [ -n "$v" ] && [ ! -f "$lib" ] && continue
    echo "== ${v:-as built} (pass $rep)"
    for shape in "64->64 @256" "128->128 @128" "256->256 @64" "512->512 @32"; do
      DISCO_HIP_LIB=$lib python tools/bench_conv.py --only "$shape" --iters 30 2>&1 | grep "@" | grep -v "^up\|^cat\|^s2" | sed 's/$/   [f16x3]/'
      DISCO_HIP_LIB=$lib python tools/bench_conv.py --mx 6 --only "$shape" --iters 30 2>&1 | grep "@" | grep -v "^up\|^cat\|^s2"
    done
  done
done
for v in "" defer; do
  lib=${v:+tools/build/libdisco_conv_$v.so}
  echo "== headline, ${v:-as built}"
  DISCO_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'checksum', d['result_checksum'], 'conv TF', d['roofline']['achieved'], d['stage_ms_per_step'])"
done
echo "== conv parity tests under the deferred build"
DISCO_HIP_LIB=tools/build/libdisco_conv_defer.so python -m pytest tests/test_gpu_mx.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -4
