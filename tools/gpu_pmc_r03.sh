#!/usr/bin/env bash
# Round 3 counter passes (each counter set in its OWN rocprofv3 run, kernel-trace only alongside): FETCH_SIZE and WRITE_SIZE as
# single-counter passes on conv_halo256_kernel, self_attn_sp_kernel and ip_attn_kernel (the 3-block pass of round 2 timed
# out), the SQ set on the new attention kernel, and the conv's LDS bank-conflict counter after the column-swizzle fix.
# Sizes are reported per launch with the guide's gfx950 correction (FETCH_SIZE x 2, both in KiB -> bytes).
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
sum="$out/r03_pmc_summary.txt"
: > "$sum"
python -c "import torch" > /dev/null 2>&1      # page the image in once, outside the timed passes
run() { # label, filter, "cmd", counters...
  label=$1; filt=$2; cmd=$3; shift 3
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc3_$label" -o p -- $cmd > "$out/pmc3_$label.log" 2>&1
  echo "pass $label ($*) rc=$?  $(grep -h 'us' "$out/pmc3_$label.log" | tail -1)" | tee -a "$sum"
  f=$(find "$out/pmc3_$label" -name "*counter_collection.csv" | head -1)
  if [[ -n "$f" ]]; then
     python - "$f" "$filt" <<'PY' | tee -a "$sum"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:80]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if sys.argv[2] not in k: continue
    print("  ", k)
    for c, v in d.items():
        avg = v / max(cnt[(k, c)], 1)
        extra = ""
        if c == "FETCH_SIZE": extra = f"  -> {avg * 1024 * 2 / 1e6:10.1f} MB per launch (x2 gfx950 correction)"
        if c == "WRITE_SIZE": extra = f"  -> {avg * 1024 / 1e6:10.1f} MB per launch"
        print(f"      {c:32s} {avg:18.1f}  (avg over {cnt[(k, c)]} dispatches){extra}")
PY
  fi
  rm -rf "$out/pmc3_$label"
}
R="$GRAFT_REPO_ROOT/tools"
for spec in "conv|conv_halo|python $R/one_op.py conv 3" "attnsp|self_attn_sp|python $R/one_op.py attn 3" \
            "attnsp1k|self_attn_sp|python $R/one_op.py attn1k 3" "ipattn|ip_attn|python $R/one_ipattn.py 32 20 32 32 3"; do
  IFS='|' read -r name filt cmd <<< "$spec"
  run ${name}_fetch "$filt" "$cmd" FETCH_SIZE
  run ${name}_write "$filt" "$cmd" WRITE_SIZE
  run ${name}_sq "$filt" "$cmd" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run ${name}_valu "$filt" "$cmd" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES
done
cp "$sum" "$out/r03_pmc_conv_attn_ip_summary.txt"
