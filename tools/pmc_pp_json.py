#!/usr/bin/env python
"""Turn the summary of tools/gpu_pmc_pp.sh (gpurun_out/pmc_pp_summary.txt) into the JSON bench.py attaches as `roofline.traffic`.

    python tools/pmc_pp_json.py gpurun_out/pmc_pp_summary.txt M N K [epilogue] > profiles/r03_pmc_gemm_pp.json
"""
import json, re, sys

path, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
epi = sys.argv[5] if len(sys.argv) > 5 else "geglu"
fused = epi.endswith("_ln")
vals = {}
for line in open(path):
    m = re.match(r"\s+(\w+)\s+([0-9.]+)\s+\(avg over", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
n_out = N // 2 if epi.startswith("geglu") else N
alg = 2 * (M * K + N * K + M * n_out)                       # A + W read once, C written once (f16)
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
traffic = int(fetch_kb * 1024 * 2 + write_kb * 1024)
hit = vals.get("TCC_HIT_sum", 0.0) / max(vals.get("TCC_HIT_sum", 0.0) + vals.get("TCC_MISS_sum", 0.0), 1.0)
# GRBM_GUI_ACTIVE is summed over the 8 XCDs, each with 32 CUs x 4 SIMDs (the round-1/2 definition of the figure)
busy = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(128.0 * vals.get("GRBM_GUI_ACTIVE", 0.0), 1.0)
print(json.dumps({
    "kernel": "gemm_pp_kernel" + ("<half,0,9> (fused-LayerNorm consumer, GEGLU epilogue)" if fused else "<half,0,0>"),
    "command": f"bash tools/gpu_pmc_pp.sh (SHAPE='{M} {N} {K}'; rocprofv3 --pmc <counters> --kernel-trace, one pass per counter group; "
               f"target: tools/one_gemm.py {M} {N} {K} 0 3 {epi}); JSON by tools/pmc_pp_json.py",
    "shape": {"M": M, "N": N, "K": K, "epilogue": epi},
    "FETCH_SIZE_KB": fetch_kb,
    "fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section: 16 B/lane streaming reads are tallied at half their bytes on gfx950)",
    "WRITE_SIZE_KB": write_kb,
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round(traffic / alg, 3),
    "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"), "l2_hit_rate": round(hit, 3),
    "GRBM_GUI_ACTIVE_sum_over_8_xcd": vals.get("GRBM_GUI_ACTIVE"),
    "SQ_VALU_MFMA_BUSY_CYCLES": vals.get("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_BUSY_CYCLES": vals.get("SQ_BUSY_CYCLES"),
    "mfma_busy_fraction_of_simd_cycles": round(busy, 3),
    "SQ_LDS_BANK_CONFLICT": vals.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": vals.get("SQ_LDS_IDX_ACTIVE"),
    "note": "FETCH_SIZE counts L2 fabric-side requests, Infinity-Cache hits included; WRITE_SIZE equals the C bytes.",
}, indent=1))
