#!/usr/bin/env bash
# PMC passes (one counter group per run, kernel-trace only alongside) on ip_attn_kernel after its round-6 changes (tools/one_ipattn.py 64 20 32 32:
# the level-2 launch of the batch-64 forward)
set -u
R="$GRAFT_REPO_ROOT"
out="$R/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
sum="$out/r06_pmc_ip_attn_summary.txt"; : > "$sum"
run() { # tag, cmd..., -- counters
  tag=$1; shift; cmd=(); while [[ $1 != "--" ]]; do cmd+=("$1"); shift; done; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmcx" -o p -- python "${cmd[@]}" > "$out/pmcx.log" 2>&1
  echo "== $tag ($*) rc=$? $(grep -h 'TF/s\|TB/s' "$out/pmcx.log" | tail -1)" >> "$sum"
  f=$(find "$out/pmcx" -name "*counter_collection.csv" | head -1)
  [[ -n "$f" ]] && python - "$f" >> "$sum" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "conv_halo" not in k and "ip_attn" not in k: continue
    print("  ", k)
    for c, v in d.items(): print(f"      {c:32s} {v / max(cnt[(k, c)], 1):18.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
  rm -rf "$out/pmcx"
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
for t in ip; do
  if [[ $t == conv ]]; then c=("$R/tools/one_op.py" conv 3); else c=("$R/tools/one_ipattn.py" 64 20 32 32 3); fi
  run "$t sq" "${c[@]}" -- $SQ
  run "$t fetch" "${c[@]}" -- FETCH_SIZE
  run "$t write" "${c[@]}" -- WRITE_SIZE
  run "$t tcc" "${c[@]}" -- TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
done
cat "$sum"
