#!/usr/bin/env bash
# PMC passes on one GEMM shape (counters in their own runs, kernel-trace only alongside).
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
SHAPE="${SHAPE:-8192 10240 1280}"
VAR="${VAR:-2}"
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- \
     python "$GRAFT_REPO_ROOT/tools/one_gemm.py" $SHAPE $VAR 3 > "$out/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
  f=$(find "$out/pmc_$name" -name "*counter_collection.csv" | head -1)
  if [[ -n "$f" ]]; then
     python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" not in k: continue
    print(k)
    for c, v in d.items():
        print(f"   {c:32s} {v / max(cnt[(k, c)], 1):16.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
  fi
  find "$out/pmc_$name" -name "*.csv" -size +5M -delete
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
