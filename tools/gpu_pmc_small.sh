#!/usr/bin/env bash
# PMC passes (counters in their own runs, kernel-trace only alongside) on the two one-block-per-CU kernels of a batch-1 request:
# conv_halo_deep_kernel (B=2, 32x32, 1280->1280), gemm_t160_kernel (2048 x 1280 x 1280; its 128-row tiles at 2048 x 2560 x 1280) and
# gemm_g320_kernel (2048 x 10240 x 1280 GEGLU).   OPS="g320 t160tall" tools/gpu_pmc_small.sh selects; SUM names the summary file.
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
sum="$out/${SUM:-r06_pmc_small_grid_summary.txt}"; : > "$sum"
run() { # op, name, counters...
  op=$1; name=$2; shift 2
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmcs_$name" -o p -- \
     python "$GRAFT_REPO_ROOT/tools/one_op.py" $op 20 > "$out/pmcs_$name.log" 2>&1
  echo "pass $op/$name ($*) rc=$?  $(grep -h 'TF/s' "$out/pmcs_$name.log" | tail -1)" | tee -a "$sum"
  f=$(find "$out/pmcs_$name" -name "*counter_collection.csv" | head -1)
  if [[ -n "$f" ]]; then
     python - "$f" <<'PY' | tee -a "$sum"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "conv_halo" not in k and "t160" not in k and "g320" not in k: continue
    print("  ", k)
    for c, v in d.items():
        print(f"      {c:32s} {v / max(cnt[(k, c)], 1):18.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
  fi
  rm -rf "$out/pmcs_$name"
}
for op in ${OPS:-conv_b2 t160}; do
  run $op ${op}_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run $op ${op}_grbm GRBM_GUI_ACTIVE
  run $op ${op}_tcc TCC_HIT_sum TCC_MISS_sum
  run $op ${op}_fetch FETCH_SIZE
  run $op ${op}_write WRITE_SIZE
done
