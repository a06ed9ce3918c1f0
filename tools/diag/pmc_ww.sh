set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_ww_pmc; rm -rf $out; mkdir -p $out; cd $R
for v in 0 1; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
    tag=ww${v}_$(echo $set | cut -d' ' -f1)
    PTK_WAVE_WINDOWS=$v rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o pmc -- python3 tools/diag/render_loop.py cornell.txt 0 1 24 > /dev/null 2> $out/$tag.err || { tail -3 $out/$tag.err; continue; }
    python3 tools/pmc_sum.py $(find $out/$tag -name "*counter_collection.csv" | head -1) > $out/$tag.txt; rm -rf $out/$tag
  done
done
grep -h -A8 "k_shade_trace" $out/*.txt | grep -v "^--" 
