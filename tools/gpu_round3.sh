#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/run3.log" 2>&1
timeout 900 python tools/gemm_bench.py --variants 2,4 --reps 10 > "$out/gemm_bench3.log" 2>&1
echo "gemm_bench rc=$?" | tee -a "$out/run3.log"
tail -16 "$out/gemm_bench3.log"
DS_GEMM_VARIANT=4 timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x > "$out/pytest_gpu3.log" 2>&1
echo "pytest(v4) rc=$?" | tee -a "$out/run3.log"
tail -8 "$out/pytest_gpu3.log"
DS_GEMM_VARIANT=4 timeout 900 python bench.py --steps 2 --warmup 1 --num-samples 4 --no-cpu-baseline > "$out/bench3_ns4.json" 2> "$out/bench3_ns4.err"
echo "bench rc=$?" | tee -a "$out/run3.log"
cut -c1-200 "$out/bench3_ns4.json"
grep -A4 '"gemm\|"self_attn\|"groupnorm\|"ip_attn\|"layernorm' "$out/bench3_ns4.err" | head -70
