#!/usr/bin/env python
"""MLLM pre-pass decode rate at LLaMA-2-13B dimensions (random weights): tokens/s of the captured one-token plan and
the HBM roofline fraction (algorithmic bytes = every layer matrix + lm_head once per token).
    python tools/mllm_bench.py [--layers 40] [--prompt 96] [--new 192]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd.mllm import LlamaConfig, LlamaDecodeEngine, random_llama_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=40)
ap.add_argument("--prompt", type=int, default=96)
ap.add_argument("--new", type=int, default=192)
ap.add_argument("--graph", choices=["both", "on", "off"], default="both")
ap.add_argument("--gemv-variant", type=int, default=0, help="0 pipelined, 1 one column per wavefront, 2 streaming")
a = ap.parse_args()
dev = torch.device("cuda", 0)
from diffsensei_amd import _lib
assert _lib.load().ds_set_option(b"llm_gemv_variant", a.gemv_variant) == 0
cfg = LlamaConfig(num_hidden_layers=a.layers)
t0 = time.perf_counter()
sd = random_llama_state_dict(cfg, dev, 0)
eng = LlamaDecodeEngine(cfg, sd, dev, max_positions=a.prompt + a.new + 8, max_new_tokens=a.new, poll_every=16)
del sd
torch.cuda.synchronize()
init_s = time.perf_counter() - t0
emb = (torch.randn(a.prompt, cfg.hidden_size, device=dev) * 0.5).half()
rows = []
for graph in {"both": (True, False), "on": (True,), "off": (False,)}[a.graph]:
    eng.use_graph = graph
    for rep in range(2):                                   # rep 0 warms up (and captures)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.generate(emb, 1, -1, a.new)              # eos -1: never stops early
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = int(out["ids"].numel())
    assert n == a.new and torch.isfinite(out["hidden"].float()).all()
    rows.append({"graph": graph, "seconds": round(dt, 4), "new_tokens": n})
prompt_ms = {}
for path in ("mfma", "chunks"):                            # prompt pass alone (max_new_tokens = 1)
    eng.prompt_path = path
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(emb, 1, -1, 1)
        torch.cuda.synchronize()
        prompt_ms[path] = round((time.perf_counter() - t0) * 1e3, 2)
eng.prompt_path = "mfma"
# token loop alone = whole call - prompt pass; one weight pass per token after the first
for r in rows:
    r["ms_per_token"] = round((r["seconds"] * 1e3 - prompt_ms["mfma"]) / (r["new_tokens"] - 1), 4)
best = min(r["ms_per_token"] for r in rows)
gbs = eng.weight_bytes_per_token() / (best * 1e-3) / 1e9
print(json.dumps({"workload": f"LLaMA-2-13B dims x {a.layers} layers, batch 1 greedy, prompt {a.prompt} + {a.new} new tokens",
                  "init_s": round(init_s, 1), "runs": rows, "prompt_ms": prompt_ms,
                  "weight_bytes_per_token": eng.weight_bytes_per_token(), "decode_tokens_per_s": round(1e3 / best, 2),
                  "roofline": {"bound": "hbm", "kernel": "llm_gemv_pipe_kernel", "achieved": round(gbs, 1), "peak": 8000.0,
                               "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                               "note": "whole token step (204 launches) priced against the weight bytes"},
                  "ops_per_token": eng.last_run_info["ops_per_token"], "gemv_variant": a.gemv_variant}))
