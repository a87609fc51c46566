"""Build the gfx950 kernel library in-tree: diffsensei_amd/lib/libdiffsensei_hip.so.

    python -m diffsensei_amd.build [--force] [--ablation]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting .so travels
with the tree to the GPU box.  One hipcc invocation per source (parallel), then one link.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdiffsensei_hip.so")
SOURCES = ["gemm.hip", "gemm_pp.hip", "gemm_t160.hip", "gemm_g320.hip", "conv_halo.hip", "vae.hip", "norm.hip", "attention.hip", "attention_sp.hip", "elementwise.hip", "llm.hip", "preprocess.hip",
           "capi.hip"]
HEADERS = ["ds_common.h", "ds_kernels.h", os.path.join("..", "..", "include", "diffsensei_hip.h")]
# -ffast-math spelled out WITHOUT -fassociative-math and -ffinite-math-only (round 3): with reassociation
# allowed, two kernel variants that share a source expression could be rounded differently at the optimizer's whim, and the
# suite's "bit-identical between variants" guarantees (ping-pong vs register-staged GEMM, 8x16 vs 16x16 conv blocks, flash
# attention <1> vs <2>, hipGraph replay vs eager) would rest on luck.  What is left - no errno, no traps, no signed zeros,
# reciprocal and approximate library functions, fma contraction - is decided per expression, not per schedule.  A/B on one
# box, UNet forward event sum at batch 32, two interleaved rounds: 239.5 / 240.0 ms with -ffast-math, 240.7 / 241.1 ms with
# this set (+0.5 %), all 139 kernel / UNet tests incl. every torch.equal check green (profiles/r03_fastmath_ab.txt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-fno-math-errno", "-fno-trapping-math",
         "-fno-signed-zeros", "-freciprocal-math", "-fapprox-func", "-ffp-contract=fast", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(extra=()) -> str:
    h = hashlib.sha256()
    for f in SOURCES + list(extra) + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, ablation: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    # the ablation build (extra template instantiations behind -DDS_ABLATION) is a SEPARATE library, selected per process with
    # DIFFSENSEI_LIB=.../libdiffsensei_hip_ablation.so: the production library is never replaced by it
    lib = LIB.replace(".so", "_ablation.so") if ablation else LIB
    stamp = os.path.join(LIBDIR, "build_ablation.stamp" if ablation else "build.stamp")
    flags = FLAGS + (["-DDS_ABLATION"] if ablation else [])
    sources = SOURCES
    dig = _digest() + ("+ablation" if ablation else "")
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.basename(src).replace(".hip", "_abl.o" if ablation else ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(compile_one, sources))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {lib}")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, ablation="--ablation" in sys.argv)
