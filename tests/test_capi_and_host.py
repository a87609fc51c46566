"""CPU: the C-ABI library loads and exports every symbol include/diffsensei_hip.h declares; host-side logic
(config tables, weight packing, plan construction, scheduler tables, input checks) matches the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "diffsensei_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_lib):
    from diffsensei_amd import _lib
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in include/diffsensei_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes signature table and header disagree"
    assert hip_lib.ds_version() >= 100
    assert ctypes.sizeof(_lib.DsOp) == 4 + 16 * 4 + 4 * 4 + 4 + 12 * 8 + 10 * 8  # matches struct ds_op (4 B pad)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from diffsensei_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DiffSenseiHipError):
        _lib.load()


def test_option_knobs_are_thread_local(hip_lib):
    """"no global mutable state except the last-error TLS slot".  The A/B knobs of `ds_set_option` are
    thread-local since round 5: a kernel family forced on one thread does not change what another
    thread - another serving handle of the same process - dispatches.  `ds_gemm_ln_fusable` is a pure host query of the
    dispatch rule (no GPU needed): it answers 0 while the calling thread forces the register-staged GEMM family."""
    import threading
    q = lambda: hip_lib.ds_gemm_ln_fusable(65536, 1280, 1280, 0, 1)
    auto = q()
    assert auto == 1
    other = []
    try:
        assert hip_lib.ds_set_option(b"gemm_variant", 1) == 0
        assert q() == 0
        t = threading.Thread(target=lambda: other.append(q()))
        t.start()
        t.join()
    finally:
        hip_lib.ds_set_option(b"gemm_variant", 0)
    assert other == [auto] and q() == auto


def test_env_options_reach_every_thread(hip_lib, monkeypatch):
    """DS_OPTIONS / DS_GEMM_VARIANT are a process-wide request but the knobs are thread-local, so `_lib.load()`
    applies the environment once per thread (a plan built on one thread and launched from another must see one dispatch)."""
    import threading
    from diffsensei_amd import _lib
    q = lambda: _lib.load().ds_gemm_ln_fusable(65536, 1280, 1280, 0, 1)
    assert q() == 1
    monkeypatch.setenv("DS_OPTIONS", "gemm_variant=1")
    seen = []
    t = threading.Thread(target=lambda: seen.append(q()))     # a NEW thread: its first load() applies the environment
    t.start()
    t.join()
    assert seen == [0]
    assert q() == 1            # this thread had its (empty) environment applied long ago and is not touched
    monkeypatch.delenv("DS_OPTIONS")
    t = threading.Thread(target=lambda: seen.append(q()))
    t.start()
    t.join()
    assert seen == [0, 1]


def test_ops_refuse_cpu_tensors(hip_lib):
    from diffsensei_amd import _lib, ops
    with pytest.raises(_lib.DiffSenseiHipError):
        ops.gemm(torch.zeros(8, 64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16))


def test_sdxl_config_inventory():
    from diffsensei_amd.unet_config import attn_processor_names, build_topology, param_shapes, sdxl_config
    cfg = sdxl_config()
    shapes = param_shapes(cfg)
    n = sum(int(np.prod(s)) for s in shapes.values())
    # SDXL-base UNet 2.567 B + 140 IP projection matrices (70 x 2 x C x 2048) + dialog embedding
    ip = sum(int(np.prod(s)) for k, s in shapes.items() if "_ip.weight" in k)
    assert ip == 2 * 2048 * (10 * 640 + 60 * 1280)
    assert abs((n - ip) - 2_567_463_684) < 2_000, n - ip
    names = attn_processor_names(cfg)
    assert len(names) == 140 and sum(x.endswith("attn2.processor") for x in names) == 70
    topo = build_topology(cfg)
    assert [r.cin for r in topo.up[0]["resnets"]] == [2560, 2560, 1920]
    assert [r.cin for r in topo.up[1]["resnets"]] == [1920, 1280, 960]
    assert [r.cin for r in topo.up[2]["resnets"]] == [960, 640, 640]


def test_pack_geglu_layout():
    from diffsensei_amd.engine import pack_geglu
    c = 128
    w = torch.arange(8 * c, dtype=torch.float32)[:, None].repeat(1, 4)
    b = torch.arange(8 * c, dtype=torch.float32)
    wp, bp = pack_geglu(w, b)
    for t in range(4 * c // 64):
        assert (bp[t * 128: t * 128 + 64] == torch.arange(t * 64, t * 64 + 64)).all()
        assert (bp[t * 128 + 64: t * 128 + 128] == 4 * c + torch.arange(t * 64, t * 64 + 64)).all()
    assert (wp[:, 0] == bp).all()


def test_mask_grid_size_matches_oracle():
    from diffsensei_amd.attention_processor import mask_grid_size
    from oracle.attention_ref import mask_grid_size as ref
    for h, w in [(128, 128), (64, 64), (32, 32), (24, 40), (48, 32), (28, 48), (16, 16), (8, 12), (192, 128)]:
        for lvl in (1, 2):
            hh, ww = h >> lvl, w >> lvl
            if hh * ww == 0:
                continue
            assert mask_grid_size(hh * ww, h / w) == ref(hh * ww, h / w)


def test_dialog_pixel_boxes_match_reference_loop():
    from diffsensei_amd.unet import dialog_pixel_boxes
    torch.manual_seed(0)
    db = torch.rand(3, 8, 4).to(torch.float16)
    db[0, 0] = torch.tensor([0.65, 0.02, 0.95, 0.15])
    db[1, 1] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    db[2, 2] = 0
    for (h, w) in [(128, 128), (64, 96), (17, 23)]:
        got = dialog_pixel_boxes(db, h, w)
        for i in range(3):
            for j in range(8):
                x1, y1 = int(db[i, j, 0] * w), int(db[i, j, 1] * h)   # reference src/models/unet.py:102-105
                x2, y2 = int(db[i, j, 2] * w), int(db[i, j, 3] * h)
                exp = [max(0, x1), max(0, y1), min(w, x2), min(h, y2)]
                assert got[i, j].tolist() == exp


def test_scheduler_tables_match_oracle():
    from diffsensei_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    from oracle.scheduler_ref import DDIMOracle, EulerDiscreteOracle
    for n in (20, 30, 50):
        e, eo = EulerDiscreteScheduler(), EulerDiscreteOracle().set_timesteps(n)
        e.set_timesteps(n)
        tab = e.coef_table(7.5)
        assert np.array_equal(tab[:, 0], eo.timesteps)
        assert np.array_equal(tab[:, 2], eo.sigmas[:-1]) and np.array_equal(tab[:, 3], eo.sigmas[1:])
        assert abs(e.init_noise_sigma - eo.init_noise_sigma) < 1e-6
        assert np.allclose(tab[:, 1], np.sqrt(eo.sigmas[:-1].astype(np.float64) ** 2 + 1), rtol=1e-6)
        assert (tab[:, 7] == 7.5).all() and tab[-1, 3] == 0 and tab[-1, 6] == 1
        d, do = DDIMScheduler(), DDIMOracle().set_timesteps(n)
        d.set_timesteps(n)
        tab = d.coef_table(5.0)
        assert np.array_equal(tab[:, 0].astype(np.int64), do.timesteps)
        a = do.alphas_cumprod.numpy()
        assert np.allclose(tab[:, 2], np.sqrt(a[do.timesteps]), rtol=1e-6)


def test_scheduler_config_keys_that_change_the_schedule_are_refused():
    """scheduler_config.json keys the local Euler / DDIM classes do not implement must not be swallowed - a
    checkpoint with karras sigmas, zero-SNR rescaling, trained betas, ... would silently sample on another schedule than
    the reference's diffusers scheduler (reference src/pipelines/pipeline_diffsensei.py:248-249 uses whatever the
    checkpoint's scheduler_config.json builds)."""
    from diffsensei_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    ok = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
              timestep_spacing="leading", prediction_type="epsilon", interpolation_type="linear", use_karras_sigmas=False,
              trained_betas=None, clip_sample=False, set_alpha_to_one=False, skip_prk_steps=True, sample_max_value=1.0,
              rescale_betas_zero_snr=False, final_sigmas_type="zero", timestep_type="discrete")
    EulerDiscreteScheduler(**ok).set_timesteps(20)
    DDIMScheduler(**ok).set_timesteps(20)
    for key, bad in (("use_karras_sigmas", True), ("rescale_betas_zero_snr", True), ("trained_betas", [0.1, 0.2]),
                     ("interpolation_type", "log_linear"), ("final_sigmas_type", "sigma_min"), ("clip_sample", True),
                     ("set_alpha_to_one", True), ("timestep_type", "continuous"), ("use_exponential_sigmas", True),
                     ("thresholding", True)):
        for cls in (EulerDiscreteScheduler, DDIMScheduler):
            with pytest.raises(NotImplementedError):
                cls(**dict(ok, **{key: bad}))
    with pytest.raises(NotImplementedError):
        EulerDiscreteScheduler(timestep_spacing="trailing")


def test_pipeline_check_inputs_errors():
    from diffsensei_amd.pipeline import DiffSenseiPipeline
    p = DiffSenseiPipeline.__new__(DiffSenseiPipeline)
    with pytest.raises(ValueError):
        p.check_inputs(None, None, [], None, [])
    with pytest.raises(ValueError):
        p.check_inputs(["a"], None, [], None, [])
    with pytest.raises(ValueError):
        p.check_inputs("a", 3, [], None, [])
    with pytest.raises(ValueError):
        p.check_inputs("a", None, [object()], torch.zeros(1, 16, 8), [[0, 0, 1, 1]])
    with pytest.raises(ValueError):
        p.check_inputs("a", None, [object()], None, [])
    p.check_inputs("a", None, [object()], None, [[0, 0, 1, 1]])


def test_plan_builds_on_host_for_tiny_config(hip_lib):
    """The launch plan is pure host logic + pointers: build it over CPU buffers (no launch) and count the ops."""
    from diffsensei_amd.engine import PackedUNet, UNetEngine
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    cfg = tiny_config()
    sd = random_state_dict(cfg, 0)
    pk = PackedUNet(cfg, sd, torch.device("cpu"))
    eng = UNetEngine(pk, 2, 16, 16)
    n_blocks = sum(a.depth for a in pk.attns)
    n_res = len(pk.resnets)
    n_tr = len(pk.attns)
    n_short = sum(r.has_shortcut for r in pk.resnets)
    expected = 4 + 1 + n_res * 8 + n_short + n_tr * (3 + 2 + 1) + n_blocks * 12 + 2 + 3 + 1
    # time(4) conv_in(1) resnet: 2 GN(x3 kernels inside one op)=2 ops + 2 conv = 4 ops ... recount generically below
    assert eng.forward_plan.n == len(eng.forward_ops) > 100
    eng.build_sampler(1, 0, True)
    assert eng.step_plan.n == eng.forward_plan.n + 2
    assert pk.kv_total == sum(a.channels * a.depth for a in pk.attns)
    assert pk.temb_total == sum(r.cout for r in pk.resnets)
    # any latent size builds, like the reference (image sides that are multiples of 8): levels halve with ceil, the
    # upsamplers resize to the skip's size, V^T rows are padded to 8 keys
    odd = UNetEngine(pk, 2, 18, 13)
    assert odd.hw == [(18, 13), (9, 7), (5, 4)]
    ups = [op for op in odd.forward_ops if op.code == 2 and op.i[6] == 1]          # CONV3X3 with the upsample flag
    assert [(op.i[1], op.i[2], op.i[8], op.i[9]) for op in ups] == [(5, 4, 9, 7), (9, 7, 18, 13)]
    att = [op for op in odd.forward_ops if op.code == 5]                            # SELF_ATTN: ldv = N rounded up to 8
    assert {(op.i[2], op.l[2]) for op in att} == {(63, 64), (20, 24)}
    with pytest.raises(ValueError):
        UNetEngine(pk, 2, 0, 16)


def test_plan_folds_every_layernorm_into_the_gemms_around_it(hip_lib, monkeypatch):
    """Host logic of the fused LayerNorm (engine._transformer, csrc/gemm.hip ds_gemm_ln_kind): on the tiny config every GEMM of a
    transformer block runs the 128-wide kernels, so the plan carries no LAYERNORM and no LN_FINALIZE op - each block's two
    out-projections and its FF down-projection (or proj_in) emit statistics, q|k, the transposed to_v, attn2.to_q and the GEGLU
    projection consume the partial sums; DIFFSENSEI_LN_FUSION=0 brings the three launches per block back.  The dispatch query
    itself is pinned for the SDXL shapes of both operating points."""
    from collections import Counter
    from diffsensei_amd.engine import PackedUNet, UNetEngine
    from diffsensei_amd.unet_config import random_state_dict, tiny_config
    cfg = tiny_config()
    pk = PackedUNet(cfg, random_state_dict(cfg, 0), torch.device("cpu"))
    blocks = sum(a.depth for a in pk.attns)
    eng = UNetEngine(pk, 2, 16, 16)
    codes = Counter(op.code for op in eng.forward_ops)
    gemms = [op for op in eng.forward_ops if op.code == 1]
    assert codes[4] == 0 and codes[27] == 0 and eng.ln_fused_launches == 3 * blocks and eng.ln_finalize_launches == 0
    assert sum(1 for op in gemms if op.p[9]) == 3 * blocks                       # producers: one per fused norm
    assert sum(1 for op in gemms if op.p[7] and op.i[9]) == 4 * blocks           # consumers of partial sums: q|k, V^T, to_q, GEGLU
    assert sum(1 for op in gemms if op.i[8]) == blocks                           # ... of which the operand-swapped V^T
    assert all(op.l[11] == op.l[10] * op.i[5] for op in gemms if op.i[8])        # ln_rows = tokens per image x images
    with monkeypatch.context() as mp:
        mp.setenv("DIFFSENSEI_LN_FUSION", "0")
        off = UNetEngine(pk, 2, 16, 16)
    assert Counter(op.code for op in off.forward_ops)[4] == 3 * blocks and len(off.forward_ops) == len(eng.forward_ops) + 3 * blocks
    assert not any(op.p[7] or op.p[9] for op in off.forward_ops if op.code == 1)
    kind = lambda *a: hip_lib.ds_gemm_ln_fusable(*a)
    # SDXL at UNet batch 2 (BASELINE configs[1]): everything on the 128-wide kernels except the 64 x 64-token GEGLU projections
    assert [kind(2048, 1280, 1280, 0, 1), kind(2048, 2560, 1280, 0, 1), kind(2048, 1280, 5120, 0, 1), kind(2048, 10240, 1280, 1, 1),
            kind(1280, 1024, 1280, 0, 2), kind(8192, 640, 640, 0, 1), kind(8192, 5120, 640, 1, 1)] == [2, 2, 2, 2, 2, 2, 1]
    # ... and at batch 64 (the metric line): the 1280-channel level on gemm_pp_kernel, and since round 6 the 640-channel
    # out-projections too (whole 64-column strips of their ragged third tile column); gemm_pp_narrow 1: 128-wide, as until round 5
    assert [kind(65536, 1280, 1280, 0, 1), kind(65536, 10240, 1280, 1, 1), kind(1280, 1024, 1280, 0, 64), kind(262144, 640, 640, 0, 1),
            kind(262144, 5120, 640, 1, 1)] == [1, 1, 1, 1, 1]
    assert kind(262144, 640, 2560, 0, 1) == 1 and kind(640, 4096, 640, 0, 64) == 1     # (the operand-swapped V^T: 2.5 tile ROWS, whole 32-row pieces)
    assert kind(648, 4096, 640, 0, 64) != 1          # (no whole 32-row pieces: the 128-wide kernels)
    try:
        assert hip_lib.ds_set_option(b"gemm_pp_narrow", 1) == 0
        assert kind(262144, 640, 640, 0, 1) == 2
    finally:
        hip_lib.ds_set_option(b"gemm_pp_narrow", 0)
    assert kind(2048, 1280, 1288, 0, 1) == 0 and kind(0, 1280, 1280, 0, 1) == 0 and kind(2048, 1280, 1280, 2, 1) == 0


def test_committed_bench_line_follows_the_contract():
    """The round-end bench line kept under profiles/ carries every key the driver's contract names."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench_default_ns32_final.json")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in line, k
    par = line["parity"]            # the SAME whole call (prompt strings -> uint8 image) on the GPU and on the fp32 CPU oracle
    assert par["path"] == "__call__" and par["rel_l2"] <= par["tolerance"] <= 3e-2 and par["steps"] >= 2 and "512x512" in par["config"]
    assert par["image_rel_l2"] <= par["image_tolerance"] <= 5e-2 and 0 <= par["u8_frac_gt_1lsb"] <= 0.05 and par["u8_max_diff"] <= 8
    assert line["config"]["output"] == "pil"
    assert line["unit"] == "panels/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"]
    inst = r["instantiations"]          # round 4: the dominant kernel has five instantiations; their launches add up to the kernel's
    assert sum(v["launches"] for v in inst.values()) == r["launches_per_forward"] and all(k.startswith(r["kernel"]) for k in inst)
    vp = line["config"]["vae_precision"]
    assert vp["engine"] == "fp16-scaled" and vp["reference"].startswith("fp32") and \
        0 < vp["fp32_decode_exposure"]["value_lower_bound_if_fp32_decode"] < line["value"]
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_bench_mllm_request_path_contract():
    """Host side of `bench.py --mllm` (BASELINE config 3; its only full-size run crashed in round 1 on a kernel shape
    limit): the synthetic instruction is consistent with what `ContinuousLVLM.generate` and the engine capacities expect,
    and every matrix width of the agent at LLaMA-2-13B / resampler dimensions is one the GPU suite exercises."""
    import bench
    from diffsensei_amd import mllm as M
    inp = bench.mllm_synthetic_inputs()
    ids, mask, chain = inp["input_ids"], inp["ids_cmp_mask"], inp["chain"]
    assert ids.shape == mask.shape == (111,) and int(mask.sum()) == 64
    assert ids[0] == 1 and ids[-1] == chain[0] and ids[41] == chain[0] and ids[106] == chain[-1]
    assert ids[42:106].tolist() == chain[1:-1] and bool(mask[42:106].all())
    assert len(chain) == 66 and inp["max_new"] == 66
    cfg = M.LlamaConfig()
    assert len(ids) + inp["max_new"] <= 256 and inp["max_new"] <= 128            # build_mllm_agent's cache / token capacity
    assert M.image_token_ids(None, 64, chain) == (chain, chain[-1], chain[1:-1])
    # widths that reach ds_layernorm_f16 / the GEMV kernels in the agent: hidden 5120, resampler dims 5120 / 2048 -
    # tests/test_gpu_ops.py::test_layernorm covers C up to 8192, test_gpu_mllm.py runs both resamplers at these dims
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads) == (5120, 13824, 40)
    assert max(cfg.hidden_size, 2048) <= 8192


def test_one_block_per_cu_tile_rules_are_pure_host_logic(hip_lib):
    """Round 6: the planner's host queries for the tile shapes built for the reference's own call shape (one request, batch 1;
    reference scripts/demo/gradio_wo_mllm.py:45-62) answer without a GPU, are thread-local A/B switches, and pick exactly the
    shapes whose grid is one block per CU on >= 5/8 of a 256-CU part:
      ds_gemm_t160_fits   64 x 160 tiles (M = 2048, N = 1280), else 128 x 160 tiles (q|k: N = 2560; M = 4096, N = 1280)
      ds_gemm_g320_fits   256 x 320 tiles (the GEGLU projection M = 2048, N = 10240 packed; plain q|k at M = 8192, N = 2560)."""
    t160 = lambda m, n, k, b=1: int(hip_lib.ds_gemm_t160_fits(m, n, k, b))
    g320 = lambda m, n, k, b=1: int(hip_lib.ds_gemm_g320_fits(m, n, k, b))
    assert t160(2048, 1280, 1280) == 1 and t160(2048, 1280, 5120) == 1          # 32 x 8 = 256 blocks of 64 x 160
    assert t160(2048, 2560, 1280) == 1 and t160(4096, 1280, 1280) == 1          # 16 x 16 / 32 x 8 = 256 blocks of 128 x 160
    assert t160(8192, 2560, 1280) == 0 and t160(65536, 1280, 1280) == 0 and t160(1024, 1280, 1280) == 0
    assert t160(2048, 1280, 1280, 2) == 0 and t160(2048, 1288, 1280) == 0 and t160(2048, 1280, 192) == 0
    assert g320(2048, 10240, 1280) == 1 and g320(8192, 2560, 1280) == 1         # 8 x 32 / 32 x 8 = 256 blocks of 256 x 320
    assert g320(1152, 10240, 1280) == 0                                          # 768 x 768, batch 1: its 5 x 40 tiles of 256 x 256 already fit one round
    assert g320(4096, 10240, 1280) == 0 and g320(65536, 10240, 1280) == 0 and g320(6144, 2560, 1280) == 0   # > 256 blocks / one round of 256 x 256 tiles
    assert g320(2048, 10240, 1280, 2) == 0 and g320(2048, 10240 + 64, 1280) == 0
    for key, q, args in ((b"gemm_t160", t160, (2048, 1280, 1280)), (b"gemm_g320", g320, (2048, 10240, 1280))):
        try:
            assert hip_lib.ds_set_option(key, 1) == 0 and q(*args) == 0
        finally:
            hip_lib.ds_set_option(key, 0)
        assert q(*args) == 1
    try:
        assert hip_lib.ds_set_option(b"gemm_t160", 2) == 0                      # 64-row tiles only
        assert t160(2048, 1280, 1280) == 1 and t160(2048, 2560, 1280) == 0
    finally:
        hip_lib.ds_set_option(b"gemm_t160", 0)
    # the fused-LayerNorm query follows: a consumer of partial sums on every one of these kernels
    assert hip_lib.ds_gemm_ln_fusable(2048, 10240, 1280, 4, 1) == 2 and hip_lib.ds_gemm_ln_fusable(8192, 2560, 1280, 0, 1) == 2
    assert hip_lib.ds_gemm_ln_fusable(2048, 2560, 1280, 0, 1) == 2 and hip_lib.ds_gemm_ln_fusable(2048, 10240, 1280, 4, 2) == 0
