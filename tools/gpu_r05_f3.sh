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
