set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r01
rm -rf $O; mkdir -p $O
timeout -k 10 500 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1
echo bench done
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 10 --warmup 3 --cpu-faces 0 --bf16-batch 0 > $O/stats.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o s -- python3 tools/prof_bf16.py > $O/stats_bf16.log 2>&1
echo stats bf16 done
for p in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  d=${p%%:*}; c=${p#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$d -o pmc -- python3 bench.py --steps 3 --warmup 1 --cpu-faces 0 --bf16-batch 0 > $O/$d.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/bf16/$d -o pmc -- python3 tools/prof_bf16.py > $O/bf16_$d.log 2>&1
  echo pmc $d done
done
ls $O $O/stats | head -30
