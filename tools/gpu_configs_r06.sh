#!/usr/bin/env bash
# The non-headline BASELINE configs on one MI355X (literal command lines; the headline line is `python bench.py`):
#   configs[1]  1024^2, 50 steps, 1 ref, no dialog boxes, batch 1        configs[2]  + MLLM pre-pass, 2 refs + dialog, batch 4
#   configs[3]  mixed bucket {512, 768, 1024, 1536}, 32 requests (ONE GPU)  configs[4]  2048^2, 4 refs, batch 1 (fp16 attention)
set -u
out=gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/r06_configs_build.log" 2>&1
run() { # name, args...
  n=$1; shift
  timeout 900 python bench.py "$@" --no-cpu-baseline --no-parity > "$out/r06_bench_$n.json" 2> "$out/r06_bench_$n.err"
  echo "$n rc=$? $(python - "$out/r06_bench_$n.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "forward", d.get("unet_forward", {}).get("unet_forward_ms_event_sum"), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("unparsed", e)
PY
)"
}
run c2_ns1_1ref --num-samples 1 --refs 1 --no-dialog --steps 3 --warmup 1
run c3_mllm_ns4 --mllm --num-samples 4 --steps 3 --warmup 1
run c5_2048_ns1 --size 2048 --refs 4 --num-samples 1 --steps 2 --warmup 1
timeout 900 python tools/mixed_bench.py > "$out/r06_mixed_bucket_serving.json" 2> "$out/r06_mixed_bucket_serving.err"
echo "mixed rc=$? $(tail -1 $out/r06_mixed_bucket_serving.json | cut -c1-300)"
