set -x
O=gpurun_out/r05b
mkdir -p $O
{
echo "== attention_kernel (DISCO_ATTN_MFMA=0)"; DISCO_ATTN_MFMA=0 python tools/attn_ab.py 2>&1 | grep tokens
echo "== attention_mfma_kernel as built"; python tools/attn_ab.py 2>&1 | grep tokens
for v in vf abl1 abl2 abl4 abl3 abl6 vfabl2; do
  echo "== variant $v"; DISCO_HIP_LIB=tools/build/libdisco_attn_$v.so python tools/attn_ab.py 2>&1 | grep tokens
done
} > $O/attn_variants.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/attn_prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_ab.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/attn_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/attn_kernel_stats.csv
find $O/attn_prof -name "*kernel_trace.csv" -delete; find $O/attn_prof -name "*agent_info.csv" -delete
head -12 $O/attn_kernel_stats.csv
