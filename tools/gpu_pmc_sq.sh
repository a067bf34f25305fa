#!/bin/bash
# SQ counter passes of bench.py (one step of BASELINE configs[1]) -> per-kernel sums per dispatch.
#   usage (inside gpurun): bash tools/gpu_pmc_sq.sh OUTNAME
# Each pass holds at most 8 SQ counters and runs under its own hard time limit (rocprofv3 hangs when a pass asks for more
# than the hardware collects).  Writes gpurun_out/OUTNAME/pmc_sq_pass{1,2}.txt (tools/pmc_summary.py).
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
NAME="$1"; shift
O="$R/gpurun_out/$NAME"
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
P1="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2"; do
  echo "[gpu_pmc_sq] pass $i: $P"
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d "$O/pmc_sq_$i" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > "$O/pmc_sq_$i.log" 2>&1
  python3 "$R/tools/pmc_summary.py" "$O/pmc_sq_$i" > "$O/pmc_sq_pass$i.txt"
  rm -rf "$O/pmc_sq_$i"
  i=$((i+1))
done
echo "[gpu_pmc_sq] done"
