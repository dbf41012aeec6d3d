# Developer tool (GPU box): PMC passes over the bf16 batch-512 forward for the up3 kernels.  Usage: bash tools/prof_up3.sh <outdir>
set -e
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/prof_up3}
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL" \
         "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -o pmc -- python3 tools/prof_fwd.py > $O/p$i.log 2>&1 || echo "pass $i failed"
  echo pass $i done
done
python3 tools/pmc_dump.py "${PAT:-cand8}" $(ls -d $O/p*/ | sed 's#/$##' | while read d; do f=$(find $d -name pmc_counter_collection.csv | head -1); [ -n "$f" ] && dirname $f; done) > $O/dump.txt 2>&1 || true
tail -80 $O/dump.txt
