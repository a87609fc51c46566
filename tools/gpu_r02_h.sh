#!/usr/bin/env bash
# Round 2, GPU call H: interleaved dispatch A/B (auto vs gemm_pp vs one-buffer 128x128) at UNet batches 6..16, and PMC passes on
# ip_attn_kernel at the benchmark's level-2 shape.
set -u
out="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 400 python tools/pp_dispatch_ab.py 2>&1 | grep -v amdgpu.ids | tee "$out/r02_pp_dispatch_ab.txt"
cd /tmp
sum="$out/r02_pmc_ip_attn_summary.txt"
: > "$sum"
run() { name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmcip_$name" -o p -- \
     python "$GRAFT_REPO_ROOT/tools/one_ipattn.py" 32 20 32 32 3 > "$out/pmcip_$name.log" 2>&1
  echo "pass $name ($*) rc=$?  $(grep -h 'ip_attn B' "$out/pmcip_$name.log" | tail -1)" | tee -a "$sum"
  f=$(find "$out/pmcip_$name" -name "*counter_collection.csv" | head -1)
  if [[ -n "$f" ]]; then
     python - "$f" <<'PY' | tee -a "$sum"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "ip_attn" not in k: continue
    print("  ", k)
    for c, v in d.items():
        print(f"      {c:32s} {v / max(cnt[(k, c)], 1):18.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
  fi
  rm -rf "$out/pmcip_$name" "$out/pmcip_$name.log"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
