#!/bin/bash
# One measurement pass on the GPU box: bench line, rocprofv3 kernel trace and the two HBM PMC passes of bench.py.
#   usage (inside gpurun): bash tools/gpu_profile.sh OUTNAME [bench.py arguments, e.g. --config tiled]
# Writes gpurun_out/OUTNAME/{bench.json, kernel_summary.txt, kernel_stats.csv, hbm_traffic.json}; copy what is to be
# judged into profiles/.  Every GPU step runs under its own time limit and the script stops at the first failure.
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
NAME="$1"; shift
O="$R/gpurun_out/$NAME"
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
echo "[gpu_profile] bench"
timeout -k 10 300 python3 "$R/bench.py" --steps 5 --warmup 2 "$@" > "$O/bench.json" 2> "$O/bench.err"
echo "[gpu_profile] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > "$O/trace.log" 2>&1
python3 "$R/tools/prof_summary.py" "$O/trace" > "$O/kernel_summary.txt"
S=$(find "$O/trace" -name '*kernel_stats.csv' | head -n 1)
[ -n "$S" ] && cp "$S" "$O/kernel_stats.csv"
echo "[gpu_profile] pmc FETCH_SIZE"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > "$O/pmc_fetch.log" 2>&1
echo "[gpu_profile] pmc WRITE_SIZE"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > "$O/pmc_write.log" 2>&1
python3 "$R/tools/hbm_traffic.py" "$O/pmc_fetch" "$O/pmc_write" "$O/hbm_traffic.json" > /dev/null
# the raw traces are large: keep the summaries only
rm -rf "$O/trace" "$O/pmc_fetch" "$O/pmc_write"
echo "[gpu_profile] done: $(cat "$O/bench.json" | head -c 200)"
