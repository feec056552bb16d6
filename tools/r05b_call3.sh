set -x
O=gpurun_out/r05b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention or encoder" > $O/attn_tests2.log 2>&1; tail -5 $O/attn_tests2.log
{
echo "== attention_kernel (DISCO_ATTN_MFMA=0)"; DISCO_ATTN_MFMA=0 python tools/attn_ab.py 2>&1 | grep tokens
echo "== attention_mfma_kernel v2 (software-pipelined) as built"; python tools/attn_ab.py 2>&1 | grep tokens
for v in novf abl1 abl2 abl3; do
  echo "== variant $v"; DISCO_HIP_LIB=tools/build/libdisco_attn_$v.so python tools/attn_ab.py 2>&1 | grep tokens
done
} > $O/attn_variants2.txt 2>&1
grep -v "^+" $O/attn_variants2.txt
