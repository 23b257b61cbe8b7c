#!/bin/bash
# dev tool: SQ counter passes of the fused kernel at the bench shape (500k samples x 1M rows)
#   tools/pmc1.sh [out-tag]     -> gpurun_out/pmc_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
: > "$O/summary.txt"
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$O/$n" -o "$n" --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --mode fused --no-cpu-baseline --no-extras $NPS_BENCH_EXTRA > "$O/$n.log" 2>&1
  f=$(ls "$O/$n"/*/*counter_collection.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls "$O/$n"/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY' | tee -a "$O/summary.txt"
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'fused_cw' in r['Kernel_Name'] or 'fused_mx' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU
run c SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL
