#!/usr/bin/env bash
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/run6.log" 2>&1
timeout 900 python tools/gemm_bench.py --variants 2,5,6 --reps 10 > "$out/gemm_bench6.log" 2>&1
echo "gemm_bench rc=$?" | tee -a "$out/run6.log"
tail -16 "$out/gemm_bench6.log"
for v in 5 6; do
DS_GEMM_VARIANT=$v timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x > "$out/pytest_gpu6_v$v.log" 2>&1
echo "pytest(v$v) rc=$?" | tee -a "$out/run6.log"
tail -6 "$out/pytest_gpu6_v$v.log"
done
DS_GEMM_VARIANT=5 timeout 900 python bench.py --steps 2 --warmup 1 --num-samples 4 --no-cpu-baseline > "$out/bench6_ns4.json" 2> "$out/bench6_ns4.err"
echo "bench rc=$?" | tee -a "$out/run6.log"
cut -c1-200 "$out/bench6_ns4.json"
grep -A4 '"gemm' "$out/bench6_ns4.err" | head -70
