set -x
O=gpurun_out/r05b
mkdir -p $O
{
for f in 1 2 3; do echo "== DISCO_ATTN_FORM=$f"; DISCO_ATTN_FORM=$f python tools/attn_ab.py 2>&1 | grep tokens; done
for v in kpt1 kpt4; do echo "== variant $v (auto form)"; DISCO_HIP_LIB=tools/build/libdisco_attn_$v.so python tools/attn_ab.py 2>&1 | grep tokens; done
} > $O/attn_forms.txt 2>&1
grep -v "^+" $O/attn_forms.txt
