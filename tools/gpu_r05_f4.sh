#!/usr/bin/env bash
# round 5, final tree (GEGLU epilogue without LDS transposition, static priority): the whole GPU suite, smoke, the bench line and the
# rocprofv3 kernel trace of the same workload on the same box
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
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -s --durations=8 > "$out/r05_pytest_gpu_final_tree.log" 2>&1
echo "pytest rc=$?"; grep -v "amdgpu.ids" "$out/r05_pytest_gpu_final_tree.log" | tail -14
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r05_smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$out/r05_smoke.log"
timeout 900 python bench.py --steps 3 --warmup 1 > "$out/r05_bench_default_ns32_final.json" 2> "$out/r05_bench_default_ns32_final.err"
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/r05_bench_default_ns32_final.json'));print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['unet_forward'])")"
python - "$out/r05_bench_default_ns32_final.err" "$out/r05_bench_default_ns32_final_per_kernel.json" <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
i = txt.find('{\n "unet_forward_ms_event_sum"')
if i >= 0:
    obj, _ = json.JSONDecoder().raw_decode(txt[i:])
    json.dump(obj, open(sys.argv[2], "w"), indent=1)
    print({k: v["ms"] for k, v in list(obj["per_kernel"].items())[:6]})
PY
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r05" -o bench_ns32_eager -- \
    python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > "$out/r05_prof_bench.json" 2> "$out/r05_prof_bench.err"
echo "rocprof rc=$?"
cd "$R"
f=$(find "$out/prof_r05" -name "*kernel_stats.csv" | head -1)
mkdir -p "$out/r05_rocprof_kernel_stats"
[[ -n "$f" ]] && cp "$f" "$out/r05_rocprof_kernel_stats/bench_ns32_eager_kernel_stats_final.csv" && head -8 "$f" | cut -c1-200
rm -rf "$out/prof_r05"
