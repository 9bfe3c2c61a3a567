#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block is checked against.
# Runs ON THE GPU BOX (via gpurun), from the repo root:
#     bash tools/profile.sh r1 ["program.py args"]
# One kernel-trace pass (timings) and separate --pmc passes (counters): the
# TCC block holds FETCH_SIZE (3 slots) or WRITE_SIZE (2 slots), never both.
# Raw output lands under gpurun_out/prof_<tag>/; tools/summarize_profile.py
# condenses it into profiles/<tag>_*.{csv,json} (done back in the container).
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$R"
# the fingerprint of the sources the profiled library was built from, taken HERE, where the counters are collected (the
# summariser runs later, in the container, possibly after the sources have moved on)
python3 -c "from pointcloudregistration_amd.stamp import kernel_sources_stamp; print(kernel_sources_stamp())" > "$OUT/kernel_sources_stamp.txt"
# second argument: the program to profile and its arguments (default: the bench)
PROG=${2:-bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-knn --no-extra --no-live-pmc}
ARGS="$PROG"
PARGS="$PROG"

timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace done"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 $PARGS > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.err"
echo "sq done"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 $PARGS > "$OUT/bench_pmc_sq2.json" 2> "$OUT/pmc_sq2.err"
echo "sq2 done"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $PARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "fetch done"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $PARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
echo "write done"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcc" -- python3 $PARGS > "$OUT/bench_pmc_tcc.json" 2> "$OUT/pmc_tcc.err"
echo "tcc done"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc_tcp" -- python3 $PARGS > "$OUT/bench_pmc_tcp.json" 2> "$OUT/pmc_tcp.err"
echo "tcp done"
# keep only what the summariser reads (the merge back is capped at 64 MiB)
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*.csv" | sort
