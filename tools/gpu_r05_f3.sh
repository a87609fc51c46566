#!/usr/bin/env bash
# round 5, final part 3: whole-trajectory parity on the final tree (bench.py --cpu-full) and the other BASELINE configs on one GPU
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
perk() { python - "$1" "$2" <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
i = txt.find('{\n "unet_forward_ms_event_sum"')
if i >= 0:
    obj, _ = json.JSONDecoder().raw_decode(txt[i:])
    json.dump(obj, open(sys.argv[2], "w"), indent=1)
    print("  forward", obj["unet_forward_ms_event_sum"], "ms")
PY
}
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ln_fusion.py -q -m gpu -p no:cacheprovider -k "gemm or ln or fused or conv" > "$out/f3_pytest_gemm_conv.log" 2>&1
echo "pytest gemm+conv rc=$?"; tail -2 "$out/f3_pytest_gemm_conv.log"
# ---- the bench line of the final tree and the rocprofv3 kernel trace of the same workload on the same box (graph replay off)
timeout 900 python bench.py --steps 3 --warmup 1 > "$out/r05_bench_default_ns32_final.json" 2> "$out/r05_bench_default_ns32_final.err"
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/r05_bench_default_ns32_final.json'));print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])")"; perk "$out/r05_bench_default_ns32_final.err" "$out/r05_bench_default_ns32_final_per_kernel.json"
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r05" -o bench_ns32_eager -- \
    python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > "$out/r05_prof_bench.json" 2> "$out/r05_prof_bench.err"
echo "rocprof rc=$?"
cd "$R"
f=$(find "$out/prof_r05" -name "*kernel_stats.csv" | head -1)
mkdir -p "$out/r05_rocprof_kernel_stats"
[[ -n "$f" ]] && cp "$f" "$out/r05_rocprof_kernel_stats/bench_ns32_eager_kernel_stats_final.csv" && head -14 "$f" | cut -c1-220
rm -rf "$out/prof_r05"
timeout 600 python bench.py --steps 3 --warmup 1 --num-samples 1 --refs 1 --no-dialog --no-cpu-baseline > "$out/r05_bench_c2_ns1_1ref_final.json" 2> "$out/c2.err"
echo "c2 rc=$? $(python -c "import json;print(json.load(open('$out/r05_bench_c2_ns1_1ref_final.json'))['value'])")"; perk "$out/c2.err" "$out/r05_bench_c2_ns1_1ref_final_per_kernel.json"
timeout 900 python bench.py --steps 3 --warmup 1 --mllm --num-samples 4 --no-cpu-baseline > "$out/r05_bench_c3_mllm_ns4_final.json" 2> "$out/c3.err"
echo "c3 rc=$? $(python -c "import json;print(json.load(open('$out/r05_bench_c3_mllm_ns4_final.json'))['value'])")"; perk "$out/c3.err" "$out/r05_bench_c3_mllm_ns4_final_per_kernel.json"
timeout 900 python bench.py --steps 2 --warmup 1 --size 2048 --refs 4 --num-samples 1 --no-cpu-baseline > "$out/r05_bench_c5_2048_ns1_fp16_final.json" 2> "$out/c5.err"
echo "c5 fp16 rc=$? $(python -c "import json;print(json.load(open('$out/r05_bench_c5_2048_ns1_fp16_final.json'))['value'])")"; perk "$out/c5.err" "$out/r05_bench_c5_2048_ns1_fp16_final_per_kernel.json"
timeout 900 python bench.py --steps 2 --warmup 1 --size 2048 --refs 4 --num-samples 1 --attn fp8 --no-cpu-baseline > "$out/r05_bench_c5_2048_ns1_fp8_final.json" 2> "$out/c5f.err"
echo "c5 fp8 rc=$? $(python -c "import json;print(json.load(open('$out/r05_bench_c5_2048_ns1_fp8_final.json'))['value'])")"
timeout 900 python tools/mixed_bench.py > "$out/r05_mixed_bucket_serving.json" 2> "$out/mixed.err"
echo "mixed rc=$?"; tail -c 600 "$out/r05_mixed_bucket_serving.json"; echo
timeout 1500 python bench.py --steps 2 --warmup 1 --cpu-full > "$out/r05_bench_cpu_full.json" 2> "$out/cpufull.err"
echo "cpu-full rc=$?"; python - "$out/r05_bench_cpu_full.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], json.dumps(d["parity"])[:900]); print(json.dumps(d["cpu_baseline"])[:600])
PY
