#!/usr/bin/env python
"""Build-time A/B of ONE source: libdiffsensei_hip_<name>.so = the production objects with <source> recompiled under extra -D
switches.  Load with DIFFSENSEI_LIB=<path> (never a fallback; tests and tools/forward_lib_ab.py take it from the environment).
    python tools/build_variants.py gemm_pp.hip ex16:-DPP_EX=16 pf:-DPP_RES_PF=1 ex16pf:-DPP_EX=16,-DPP_RES_PF=1"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsensei_amd import build

src = sys.argv[1]
build.build(verbose=False)
for spec in sys.argv[2:]:
    name, _, defs = spec.partition(":")
    obj = os.path.join(build.LIBDIR, src.replace(".hip", f"_{name}.o"))
    subprocess.run([build._hipcc(), *build.FLAGS, *defs.split(","), "-c", os.path.join(build.CSRC, src), "-o", obj], check=True)
    objs = [os.path.join(build.LIBDIR, s.replace(".hip", ".o")) for s in build.SOURCES if s != src] + [obj]
    lib = build.LIB.replace(".so", f"_{name}.so")
    subprocess.run([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print("built", lib)
