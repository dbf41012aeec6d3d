set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/dec_traffic; rm -rf $O; mkdir -p $O
for b in 64 512; do for c in FETCH_SIZE WRITE_SIZE; do
  ONLY=decode_top4_b$b timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/b$b/$c -o pmc -- python3 tools/bench_hbm.py > $O/b${b}_$c.log 2>&1
done; done
find $O -name "*.csv" -size +20M -delete; ls $O
