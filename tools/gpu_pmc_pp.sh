#!/usr/bin/env bash
# PMC passes (counters in their own runs, kernel-trace only alongside) on the bench's dominant GEMM:
# the FF up-projection at num_samples 16 (M=32768, N=10240 packed GEGLU, K=1280) -> gemm_pp_kernel<0>.
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
SHAPE="${SHAPE:-65536 10240 1280}"
EPI="${EPI:-geglu_ln}"    # geglu_ln: the fused-LayerNorm consumer gemm_pp_kernel<half,0,9> the benchmark's GEGLU projection runs (round 4); geglu: <half,0,0>
sum="$out/pmc_pp_summary.txt"
: > "$sum"
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmcpp_$name" -o p -- \
     python "$GRAFT_REPO_ROOT/tools/one_gemm.py" $SHAPE 0 3 $EPI > "$out/pmcpp_$name.log" 2>&1
  echo "pass $name ($*) rc=$?  $(grep -h 'TF/s' "$out/pmcpp_$name.log" | tail -1)" | tee -a "$sum"
  f=$(find "$out/pmcpp_$name" -name "*counter_collection.csv" | head -1)
  if [[ -n "$f" ]]; then
     python - "$f" <<'PY' | tee -a "$sum"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm_pp" not in k: continue
    print("  ", k)
    for c, v in d.items():
        print(f"      {c:32s} {v / max(cnt[(k, c)], 1):18.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
  fi
  rm -rf "$out/pmcpp_$name"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
