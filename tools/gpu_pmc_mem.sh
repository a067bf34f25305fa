#!/bin/bash
# memory-path counter passes (TLB, L1 -> L2 read latency, L2 -> fabric read queue) of bench.py (one step of BASELINE configs[1]) -> per-kernel sums per dispatch.
#   usage (inside gpurun): bash tools/gpu_pmc_mem.sh OUTNAME
# Each pass holds at most 4 TCP / 4 TCC counters and runs under its own hard time limit (rocprofv3 hangs when a pass asks for more
# than the hardware collects).  Writes gpurun_out/OUTNAME/pmc_mem_pass{1,2,3}.txt (tools/pmc_summary.py).
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
NAME="$1"; shift
O="$R/gpurun_out/$NAME"
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
P1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
P2="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
P3="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  echo "[gpu_pmc_mem] pass $i: $P"
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d "$O/pmc_mem_$i" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > "$O/pmc_mem_$i.log" 2>&1
  python3 "$R/tools/pmc_summary.py" "$O/pmc_mem_$i" > "$O/pmc_mem_pass$i.txt"
  rm -rf "$O/pmc_mem_$i"
  i=$((i+1))
done
echo "[gpu_pmc_mem] done"
