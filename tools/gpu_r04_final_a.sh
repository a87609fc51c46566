#!/usr/bin/env bash
# Round 4 evidence, part A (final tree): the default bench line with the live roofline figures, then the rocprofv3 kernel-trace
# summary of the same workload on the SAME box (graph replay off under the profiler so that every launch is traced).
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
if [[ -n "${RETEST:-}" ]]; then
  timeout 600 python -m pytest $RETEST -q -m gpu -p no:cacheprovider > "$out/r04_pytest_gpu_retest.log" 2>&1
  echo "retest rc=$?"; tail -3 "$out/r04_pytest_gpu_retest.log"
fi
timeout 1500 python bench.py --steps 3 --warmup 1 > "$out/r04_bench_default_ns32_final.json" 2> "$out/r04_bench_default_ns32_final.err"
echo "bench rc=$?"; tail -1 "$out/r04_bench_default_ns32_final.json" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value',d['value'],'frac',r['frac'],'avg_launch_us',r['avg_launch_us'],'traffic',r['traffic'])
print(json.dumps(r.get('instantiations'),indent=0))
print('forward',d['unet_forward']); print('parity',{k:d['parity'][k] for k in ('rel_l2','image_rel_l2','u8_frac_gt_1lsb','u8_max_diff')})"
cd /tmp
DIFFSENSEI_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/r04_final_prof" -o bench -- \
   python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/r04_final_prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/r04_final_prof_bench.err"
echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/r04_final_prof" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$out/r04_final_kernel_stats.csv" && head -16 "$f" | cut -c1-170
rm -rf "$out/r04_final_prof"
