#!/usr/bin/env bash
# End of round 6: same-box A/B of the round-5 library (with its own launch planner, _ab/r05) against the final tree, one UNet forward
# at the four shapes of tools/gpu_lib_generations_ab.sh, two interleaved rounds each.  Results: gpurun_out/r06_final_gen_ab_*.txt
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/gen_fin"
run() { ( cd "$2" && AB_TAG=$1 AB_SIZE=$4 timeout 600 python tools/forward_lib_ab.py $3 "$out/gen_fin/${1}_b$3_s$4_$5.json" 2>&1 | grep -v amdgpu.ids ); }
for shape in "2 128" "8 128" "64 128" "2 256"; do
  set -- $shape
  for rnd in 1 2; do
    run r05 "$root/_ab/r05" $1 $2 $rnd
    run r06 "$root" $1 $2 $rnd
  done
  { echo "=== UNet batch $1, $(( $2 * 8 )) x $(( $2 * 8 )): round-5 library + planner -> final tree of round 6"; python "$root/tools/forward_lib_ab.py" --compare "$out"/gen_fin/r05_b$1_s$2_*.json "$out"/gen_fin/r06_b$1_s$2_*.json; } > "$out/r06_final_gen_ab_b$1_s$2.txt" 2>&1
  head -16 "$out/r06_final_gen_ab_b$1_s$2.txt"
done
