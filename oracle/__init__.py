"""CPU oracle for the DiffSensei sampling hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in `diffsensei_amd/` may import from this package.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` call it, and only
as the checker.  It is a plain-PyTorch fp32 restatement (CPU) of:

  * reference src/models/attention_processor.py  (self-attn + region-masked IP cross-attn)
  * reference src/models/unet.py                 (dialog-bbox add + the diffusers SDXL UNet it subclasses)
  * reference src/models/resampler.py            (perceiver resampler)
  * reference src/pipelines/pipeline_diffsensei.py:204-367 (CFG batch assembly, denoise loop)
  * the diffusers Euler-discrete / DDIM scheduler arithmetic the pipeline calls [3P, not vendored]

PARITY PINNING.  The reference ships no tests and no golden vectors.  The two
reference files that import only torch (attention_processor.py, resampler.py) were executed in the
build container and their outputs are committed under tests/golden/ (generator:
oracle/make_golden.py); the oracle is asserted equal to those.  Everything that lives in
`diffusers` (UNet blocks, schedulers) is NOT importable here (diffusers is not installed, no
network), is restated from its published semantics, and is therefore "parity unpinned" against the
third-party code: see DESIGN.md §Oracle.
"""
