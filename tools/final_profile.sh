# Round profile (GPU box): un-profiled bench line, rocprofv3 kernel-trace stats and three PMC passes (FETCH_SIZE;
# WRITE_SIZE; SQ + GRBM set) for the fp32 headline, the bf16 configuration and the HBM-bound kernels, the bare MFMA
# loops.  Usage: bash tools/final_profile.sh <round tag, e.g. r03>   (results under gpurun_out/final_<tag>/)
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
O=gpurun_out/final_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $O/bench.log 2>&1
echo bench done
F32="bench.py --steps 10 --warmup 3 --cpu-faces 0 --bf16-batch 0 --f32-big-batch 0 --no-hbm-kernels --settle-ms 0"
BF16="bench.py --config 3 --steps 10 --warmup 3 --cpu-faces 0 --no-hbm-kernels --settle-ms 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $F32 > $O/stats.log 2>&1
echo stats f32 done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o s -- python3 $BF16 > $O/stats_bf16.log 2>&1
echo stats bf16 done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_hbm -o s -- python3 tools/bench_hbm.py > $O/stats_hbm.log 2>&1
echo stats hbm done
for p in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  d=${p%%:*}; c=${p#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/f32/$d -o pmc -- python3 bench.py --steps 3 --warmup 1 --cpu-faces 0 --bf16-batch 0 --f32-big-batch 0 --no-hbm-kernels --settle-ms 0 > $O/f32_$d.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/bf16/$d -o pmc -- python3 bench.py --config 3 --steps 3 --warmup 1 --cpu-faces 0 --no-hbm-kernels --settle-ms 0 > $O/bf16_$d.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/hbm/$d -o pmc -- python3 tools/bench_hbm.py > $O/hbm_$d.log 2>&1
  echo pmc $d done
done
hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak > /dev/null 2>&1 && timeout -k 10 200 /tmp/mfma_peak > $O/mfma_peak.txt 2>&1
echo peak done
# the raw counter CSVs are large: keep what tools/pmc_report.py reads
find $O -name "*.csv" -size +20M -delete
ls $O
