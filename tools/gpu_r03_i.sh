#!/usr/bin/env bash
# Round 3, GPU call I: sp attention with all query blocks of a head on one XCD (variant 3) vs the plain block order (4): tests,
# timing, fabric fetch per launch.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "self_attention_s or sp_" 2>&1 | tail -3
VARS=3,4 ROUNDS=7 timeout 300 python tools/attn_bench.py 2>&1 | grep "^B=" | tee "$out/r03_i_attn_xcd_ab.txt"
cd /tmp
for v in 3 4; do
  DS_OPTIONS=attn_variant=$v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_i_$v" -o p -- python "$GRAFT_REPO_ROOT/tools/one_op.py" attn 3 > /dev/null 2>&1
  f=$(find "$GRAFT_REPO_ROOT/$out/pmc_i_$v" -name "*counter_collection.csv" | head -1)
  python - "$f" $v <<'PY' | tee -a "$GRAFT_REPO_ROOT/$out/r03_i_attn_xcd_ab.txt"
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "self_attn_sp" in r.get("Kernel_Name", "") and r["Counter_Name"] == "FETCH_SIZE"]
v = sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1)
print(f"attn_variant {sys.argv[2]}: FETCH_SIZE {v:.0f} KB -> {v * 2048 / 1e6:.0f} MB per launch (x2), {len(rows)} dispatches")
PY
  rm -rf "$GRAFT_REPO_ROOT/$out/pmc_i_$v"
done
