#!/bin/bash
# dev tool: SQ counter passes of the multi-score product kernel (8 scores x 1M rows x 500k samples)
#   tools/pmc_multi.sh [out-tag [qb_multi.py options]]     -> gpurun_out/pmc_multi_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
shift
QB=("$@")
O=$R/gpurun_out/pmc_multi_$TAG
mkdir -p "$O"
: > "$O/summary.txt"
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$O/$n" -o "$n" --output-format csv -- python3 "$R/tools/qb_multi.py" --steps 3 "${QB[@]}" > "$O/$n.log" 2>&1
  f=$(ls "$O/$n"/*/*counter_collection.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls "$O/$n"/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY' | tee -a "$O/summary.txt"
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'multi_mfma' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v[1:])/len(v[1:]), len(v[1:]))
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run c SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES SQ_ACTIVE_INST_VMEM
