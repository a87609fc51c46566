#!/usr/bin/env bash
# Round 4, last evidence: a second final-tree bench line on whatever box comes (box-to-box spread of the headline), and the kernels
# of ONE replayed step graph of BASELINE configs[1] (UNet batch 2) as rocprofv3 sees them - what does a small launch cost in there?
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/r04_bench_default_ns32_final_box2.json" 2> "$out/r04_bench_default_ns32_final_box2.err"
echo "bench rc=$?"; tail -1 "$out/r04_bench_default_ns32_final_box2.json" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value',d['value'],'frac',r['frac'],'avg_launch_us',r['avg_launch_us'],'forward',d['unet_forward'])"
cd /tmp
for g in 1 0; do
DIFFSENSEI_GRAPH=$g timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/r04_c2_prof_g$g" -o c2 -- \
   python "$GRAFT_REPO_ROOT/bench.py" --num-samples 1 --refs 1 --no-dialog --steps 1 --warmup 1 --no-cpu-baseline --no-roofline \
   > "$GRAFT_REPO_ROOT/$out/r04_c2_prof_g$g.json" 2> "$GRAFT_REPO_ROOT/$out/r04_c2_prof_g$g.err"
echo "rocprof graph=$g rc=$?"
f=$(find "$GRAFT_REPO_ROOT/$out/r04_c2_prof_g$g" -name "*kernel_stats.csv" | head -1)
[[ -n "$f" ]] && cp "$f" "$GRAFT_REPO_ROOT/$out/r04_c2_kernel_stats_graph$g.csv" && head -14 "$f" | cut -c1-150
rm -rf "$GRAFT_REPO_ROOT/$out/r04_c2_prof_g$g"
tail -1 "$GRAFT_REPO_ROOT/$out/r04_c2_prof_g$g.json" | cut -c1-200
done
