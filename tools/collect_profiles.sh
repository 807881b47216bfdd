#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/prof/ (copy what is to be judged into
# profiles/).  usage (through gpurun):  bash tools/collect_profiles.sh <tag>
#   1. bench.py default run (the line the driver will reproduce)
#   2. rocprofv3 --kernel-trace --stats of the same bench with MCR_LANES=1 (kernels alone, as the HIP-event pass times them)
#   3. rocprofv3 --kernel-trace --stats of the long-chain cases (tools/fft_prof.py: FFT tier, tier-3 kernels)
#   4. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, counters only) -> per-kernel HBM traffic per launch
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd "$REPO" || exit 1
python bench.py > "$OUT/${TAG}_c1_bench.json" 2> "$OUT/${TAG}_c1_bench.err" || echo "bench failed"
python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_c1_bench_driver_flags.json" 2>> "$OUT/${TAG}_c1_bench.err" || echo "bench (driver flags) failed"
python bench.py --workload corpus --steps 200 --no-moments > "$OUT/${TAG}_corpus_bench.json" 2>> "$OUT/${TAG}_c1_bench.err" || echo "corpus bench failed"
python tools/ingest_bench.py 9 > "$OUT/${TAG}_ingest_parquet.json" 2> "$OUT/${TAG}_ingest.err" || echo "ingest bench failed"
python tools/ingest_bench.py 9 tests/golden/corpus/draws > "$OUT/${TAG}_ingest_real_corpus.json" 2>> "$OUT/${TAG}_ingest.err" || echo "ingest bench (real corpus) failed"
python tools/cov_bench.py --out "$OUT/${TAG}_cov_mfma.json" > /dev/null 2>> "$OUT/${TAG}_ingest.err" || echo "cov bench failed"
python bench.py --workload stress --steps 3 --warmup 2 --windows 1 > "$OUT/${TAG}_stress_16GB_bench.json" 2>> "$OUT/${TAG}_c1_bench.err" || echo "stress bench failed"
python tests/manual/stress.py --pipeline-params 10000 --check-params 16 > "$OUT/${TAG}_stress_16GB_pipeline.json" 2>> "$OUT/${TAG}_c1_bench.err" || echo "stress profile failed"
cd /tmp && export TMPDIR=/tmp
export MCR_LANES=1
export MCR_FORK=0     # per-kernel numbers: no lone-call fork (it splits k_acov_seg / k_diag_combine into two half launches)
BENCH="$REPO/bench.py --steps 200 --warmup 20 --windows 1 --no-cpu-baseline --no-moments --no-probe --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python3 $BENCH > "$OUT/${TAG}_rocprof_bench_lanes1.json" 2> /tmp/prof_kt.err
find /tmp/prof_kt -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_c1_kernel_stats_lanes1.csv" \;
SHORT="$REPO/bench.py --steps 20 --warmup 2 --windows 1 --no-cpu-baseline --no-moments --no-validate --no-probe --no-extras"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python3 $SHORT > /dev/null 2> /tmp/prof_f.err
find /tmp/prof_f -name "*counter_collection.csv" -exec cp {} "$OUT/${TAG}_pmc_fetch_c1.csv" \;
python3 $REPO/tools/trim_pmc.py "$OUT/${TAG}_pmc_fetch_c1.csv" 20
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python3 $SHORT > /dev/null 2> /tmp/prof_w.err
find /tmp/prof_w -name "*counter_collection.csv" -exec cp {} "$OUT/${TAG}_pmc_write_c1.csv" \;
python3 $REPO/tools/trim_pmc.py "$OUT/${TAG}_pmc_write_c1.csv" 20
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fft -o fft -- python3 $REPO/tools/fft_prof.py 20 > "$OUT/${TAG}_fft_prof.txt" 2> /tmp/prof_fft.err
find /tmp/prof_fft -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_fft_kernel_stats.csv" \;
cd "$REPO" && python tools/pmc_summary.py "$OUT/${TAG}_pmc_fetch_c1.csv" "$OUT/${TAG}_pmc_write_c1.csv" > "$OUT/pmc_traffic.json"
ls -la "$OUT"
