#!/usr/bin/env bash
# Round 2, GPU call O: gemm_pp_kernel tile hand-over (C stores drain under the next tile's first k-tiles, bias through LDS,
# residual rows requested a piece ahead): parity of every GEMM test, bit-equality + timing A/B against the drained variant and
# hipBLASLt, then the default bench line.
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "gemm or geglu or pp" 2>&1 | tail -4
timeout 600 python tools/pp_epilogue_ab.py 2> "$out/r02_pp_epilogue_ab.err" | tee "$out/r02_pp_epilogue_ab.txt"
tail -3 "$out/r02_pp_epilogue_ab.err"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_o.err" | tail -1 > "$out/r02_bench_o.json"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_o.json"))
print("bench: %.4f panels/s, %.1f ms per call, forward %.1f ms, roofline frac %.4f (avg %.1f us), parity %s" % (
    d["value"], d["ms_per_step"], d["unet_forward"]["unet_forward_ms_event_sum"], d["roofline"]["frac"],
    d["roofline"]["avg_launch_us"], d.get("parity")))
PY
tail -2 "$out/r02_bench_o.err"
