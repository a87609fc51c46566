#!/usr/bin/env bash
# Same-box A/B of the round-4, round-5 and current kernel libraries (each with its own Python launch planner: _ab/r04, _ab/r05 are
# `git archive` exports of the round-end commits, built in place) on one UNet forward at the shapes of BASELINE configs[4]
# (2048 x 2048, batch 2), configs[1] (1024 x 1024, batch 2), batch 8 and the benched batch 64 - two interleaved rounds each
# (VERDICT r5 items 7 / 8).  Results: gpurun_out/gen_ab_<shape>.txt
set -u
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out"; mkdir -p "$out/gen_ab"
run() { # tag tree batch latent
  ( cd "$2" && AB_TAG=$1 AB_SIZE=$4 timeout 600 python tools/forward_lib_ab.py $3 "$out/gen_ab/${1}_b$3_s$4_$5.json" 2>&1 | grep -v amdgpu.ids )
}
for shape in "2 256" "2 128" "8 128" "64 128"; do
  set -- $shape
  for rnd in 1 2; do
    run r04 "$root/_ab/r04" $1 $2 $rnd
    run r05 "$root/_ab/r05" $1 $2 $rnd
    run r06 "$root" $1 $2 $rnd
  done
  for pair in "r04 r05" "r05 r06"; do
    set -- $shape $pair
    { echo "=== UNet batch $1, $(( $2 * 8 )) x $(( $2 * 8 )): $3 -> $4"; python "$root/tools/forward_lib_ab.py" --compare "$out"/gen_ab/${3}_b$1_s$2_*.json "$out"/gen_ab/${4}_b$1_s$2_*.json; } > "$out/gen_ab_b$1_s$2_$3_$4.txt" 2>&1
  done
done
head -30 "$out"/gen_ab_b2_s256_r04_r05.txt
