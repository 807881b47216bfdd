set -u
TAG=${1:-r03_v1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MCR_LANES=1
export MCR_FORK=0
SHORT="$REPO/bench.py --steps 20 --warmup 2 --windows 1 --no-cpu-baseline --no-moments --no-validate --no-probe --no-extras"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/sq_a -o a -- python3 $SHORT > /dev/null 2> $OUT/a.err
find /tmp/sq_a -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_sq_a_c1.csv \;
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/sq_b -o b -- python3 $SHORT > /dev/null 2> $OUT/b.err
find /tmp/sq_b -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_sq_b_c1.csv \;
cd $REPO && python tools/trim_pmc.py $OUT/${TAG}_pmc_sq_a_c1.csv 4 && python tools/trim_pmc.py $OUT/${TAG}_pmc_sq_b_c1.csv 4
cd $REPO && python tools/sq_summary.py $OUT/${TAG}_pmc_sq_a_c1.csv $OUT/${TAG}_pmc_sq_b_c1.csv > $OUT/${TAG}_sq_summary.json
tail -3 $OUT/a.err $OUT/b.err; ls -la $OUT
