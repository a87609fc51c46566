#!/usr/bin/env bash
# Device ISA of one source file with the production flags: tools/isa.sh gemm_t160.hip [extra -D flags] > /tmp/x.s
src="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -fno-math-errno -fno-trapping-math -fno-signed-zeros \
  -freciprocal-math -fapprox-func -ffp-contract=fast -Wno-unused-result -S --cuda-device-only "$@" -o - "$(dirname "$0")/../diffsensei_amd/csrc/$src"
