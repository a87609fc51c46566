#!/usr/bin/env bash
# PMC passes (counters in their own runs, kernel-trace only alongside) on the bench's second and third kernels:
# conv_halo256_kernel and self_attn_kernel<2> / <1> (tools/one_op.py).
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
sum="$out/r02_pmc_conv_attn_summary.txt"
: > "$sum"
run() { # op, name, counters...
  op=$1; name=$2; shift 2
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmcop_$name" -o p -- \
     python "$GRAFT_REPO_ROOT/tools/one_op.py" $op 3 > "$out/pmcop_$name.log" 2>&1
  echo "pass $op/$name ($*) rc=$?  $(grep -h 'TF/s' "$out/pmcop_$name.log" | tail -1)" | tee -a "$sum"
  f=$(find "$out/pmcop_$name" -name "*counter_collection.csv" | head -1)
  if [[ -n "$f" ]]; then
     python - "$f" <<'PY' | tee -a "$sum"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "conv_halo" not in k and "self_attn" not in k: continue
    print("  ", k)
    for c, v in d.items():
        print(f"      {c:32s} {v / max(cnt[(k, c)], 1):18.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
  fi
  rm -rf "$out/pmcop_$name"
}
for op in conv attn attn1k; do
  run $op ${op}_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run $op ${op}_fetch FETCH_SIZE          # FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE in ONE pass hung until the timeout (round 2):
  run $op ${op}_write WRITE_SIZE          # one derived-size counter per pass, like tools/gpu_pmc_pp.sh
  run $op ${op}_grbm GRBM_GUI_ACTIVE
  run $op ${op}_valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVES
done
