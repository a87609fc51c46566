#!/usr/bin/env bash
# round 5, final part 2: the bench line, the rocprofv3 kernel trace of the same workload, counter passes on the dominant GEMM and on
# self_attn_sp_kernel (round-4 library vs this tree), static-priority A/B
set -u
R="$GRAFT_REPO_ROOT"
out="$R/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd "$R"
python - <<'PY' || exit 3
from diffsensei_amd import build
import os
assert open(os.path.join(build.LIBDIR, "build.stamp")).read().strip() == build._digest(), "sources changed after the library was built"
PY
# ---- the small-batch conv kernel with the tail split (tools/patches/conv_halo_small_tail_only.patch, applied): tests, then
# split vs full tail tiles in this build and the same shapes under the round-4 library (the spill question)
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vae.py -q -m gpu -p no:cacheprovider -k "conv" > "$out/f2_pytest_conv.log" 2>&1
echo "pytest conv rc=$?"; tail -2 "$out/f2_pytest_conv.log"
{ for b in 2 8; do python tools/conv_small_ab.py $b; DIFFSENSEI_LIB=$R/diffsensei_amd/lib/libdiffsensei_hip_base.so python tools/conv_small_ab.py $b; done; } 2>&1 | grep -v amdgpu.ids | tee "$out/r05_conv_small_tail_ab.txt"
timeout 900 python bench.py --steps 3 --warmup 1 > "$out/r05_bench_default_ns32_final.json" 2> "$out/r05_bench_default_ns32_final.err"
echo "bench rc=$?"; tail -c 1500 "$out/r05_bench_default_ns32_final.json"; echo
python - "$out/r05_bench_default_ns32_final.err" "$out/r05_bench_default_ns32_final_per_kernel.json" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
i = txt.find('{\n "unet_forward_ms_event_sum"')
if i >= 0:
    dec = json.JSONDecoder()
    obj, _ = dec.raw_decode(txt[i:])
    json.dump(obj, open(sys.argv[2], "w"), indent=1)
    print("forward", obj["unet_forward_ms_event_sum"], {k: v["ms"] for k, v in list(obj["per_kernel"].items())[:6]})
PY
# ---- kernel trace of the same workload, graph replay off (one call = 50 forwards + encoders + VAE)
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_r05" -o bench_ns32_eager -- \
    python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > "$out/r05_prof_bench.json" 2> "$out/r05_prof_bench.err"
echo "rocprof rc=$?"
cd "$R"
f=$(find "$out/prof_r05" -name "*kernel_stats.csv" | head -1)
mkdir -p "$out/r05_rocprof_kernel_stats"
[[ -n "$f" ]] && cp "$f" "$out/r05_rocprof_kernel_stats/bench_ns32_eager_kernel_stats_final.csv" && head -12 "$f"
rm -rf "$out/prof_r05"
# ---- counter passes on the dominant launch (GEGLU projection, fused-LayerNorm consumer, M = 65536)
SHAPE="65536 10240 1280" EPI=geglu_ln bash tools/gpu_pmc_pp.sh > /dev/null 2>&1
cp "$out/pmc_pp_summary.txt" "$out/r05_pmc_gemm_pp_summary.txt"; cat "$out/r05_pmc_gemm_pp_summary.txt"
python tools/pmc_pp_json.py "$out/pmc_pp_summary.txt" 65536 10240 1280 geglu_ln > "$out/r05_pmc_gemm_pp.json"
# ---- self_attn_sp_kernel: round-4 library vs this tree, SQ counters
cd /tmp
sum="$out/r05_pmc_self_attn_sp_summary.txt"; : > "$sum"
for tag in base new; do
  for op in attn attn1k; do
    for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES GRBM_GUI_ACTIVE"; do
      lib=""; [[ $tag == base ]] && lib="$R/diffsensei_amd/lib/libdiffsensei_hip_base.so"
      DIFFSENSEI_LIB=$lib DS_OPTIONS=attn_variant=3 timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmcsp" -o p -- \
          python "$R/tools/one_op.py" $op 3 > "$out/pmcsp.log" 2>&1
      echo "== $tag $op ($grp) rc=$? $(grep -h 'TF/s' "$out/pmcsp.log" | tail -1)" >> "$sum"
      f=$(find "$out/pmcsp" -name "*counter_collection.csv" | head -1)
      [[ -n "$f" ]] && python - "$f" >> "$sum" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:50]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "self_attn" not in k: continue
    print("  ", k)
    for c, v in d.items(): print(f"      {c:32s} {v / max(cnt[(k, c)], 1):18.1f}  (avg over {cnt[(k, c)]} dispatches)")
PY
      rm -rf "$out/pmcsp"
    done
  done
done
cat "$sum"
cd "$R"
# ---- static priority (waves 4..7 at prio 1, no per-cluster flips) vs the per-cluster s_setprio, batch-64 forward
for r in 1 2; do
  AB_TAG=new timeout 300 python tools/forward_lib_ab.py 64 "$out/f2_new_$r.json" 2>&1 | tail -1
  DIFFSENSEI_LIB=$R/diffsensei_amd/lib/libdiffsensei_hip_sprio.so AB_TAG=sprio timeout 300 python tools/forward_lib_ab.py 64 "$out/f2_sprio_$r.json" 2>&1 | tail -1
done
python tools/forward_lib_ab.py --compare "$out"/f2_new_*.json "$out"/f2_sprio_*.json > "$out/r05_pp_static_prio_ab.txt"; head -14 "$out/r05_pp_static_prio_ab.txt"
