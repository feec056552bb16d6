set -x
O=gpurun_out/r05b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention or encoder" > $O/attn_tests3.log 2>&1; tail -5 $O/attn_tests3.log
{
echo "== attention_kernel (DISCO_ATTN_MFMA=0)"; DISCO_ATTN_MFMA=0 python tools/attn_ab.py 2>&1 | grep tokens
echo "== attention_mfma_kernel v3 (key split on small grids) as built"; python tools/attn_ab.py 2>&1 | grep tokens
} > $O/attn_variants3.txt 2>&1
grep -v "^+" $O/attn_variants3.txt
timeout 1500 bash tools/defer_ab.sh > $O/defer_ab.txt 2>&1
grep -v "amdgpu.ids" $O/defer_ab.txt | tail -50
