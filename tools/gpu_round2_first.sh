#!/usr/bin/env bash
# First GPU call of the next round: the runs round 1 wired but could not take before its GPU budget ended.
#   1. whole GPU suite (incl. tests/test_gpu_preprocess.py and the agent-dimension resampler tests)
#   2. bench.py with the MLLM pre-pass in the timed region at BASELINE config 3 (num_samples 4)
#   3. pipeline with device_preprocess switched on vs the host processors (character tokens must agree)
set -u
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee "$out/r02_pytest_gpu_first.log"
timeout 300 python bench.py --mllm --num-samples 4 --steps 2 --warmup 1 --no-cpu-baseline 2> "$out/r02_bench_c3_mllm.err" \
    | tail -1 | tee "$out/r02_bench_c3_mllm.json" | cut -c1-600
timeout 200 python - <<'PY' 2>&1 | tail -5 | tee "$out/r02_device_preprocess_e2e.log"
import numpy as np, torch, bench
from PIL import Image
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev, 1, 0, with_vae=False)
imgs = [Image.fromarray(np.random.RandomState(s).randint(0, 256, (300 + 40 * s, 200 + 90 * s, 3), dtype=np.uint8)) for s in (1, 2)]
host = pipe.encode_ip_tokens(list(imgs)).float()
pipe.device_preprocess = True
devt = pipe.encode_ip_tokens(list(imgs)).float()
rel = ((devt - host).norm() / host.norm()).item()
print("device_preprocess vs host processors: rel L2 of the 80 character tokens =", rel)
assert rel < 2e-3
PY
