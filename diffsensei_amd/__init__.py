"""diffsensei_amd — MI355X (gfx950) native sampling engine for DiffSensei-style manga panel generation.

Drop-in for ONE path of jianzongwu/DiffSensei: `DiffSenseiPipeline.__call__` -> `UNetMangaModel.forward` ->
attention processors -> scheduler step (reference src/pipelines/pipeline_diffsensei.py, src/models/unet.py,
src/models/attention_processor.py, src/models/resampler.py).  All arithmetic on that path runs in the hand-written
HIP kernels of `diffsensei_amd/csrc` behind the C ABI in `include/diffsensei_hip.h`; importing the compute
classes without the built library raises (no CPU / PyTorch fallback exists).
"""
from .unet_config import UNetMangaConfig, sdxl_config, tiny_config  # noqa: F401

__all__ = ["UNetMangaConfig", "sdxl_config", "tiny_config", "UNetMangaModel", "DiffSenseiPipeline", "Resampler",
           "AttnProcessor2_0", "MaskedIPAttnProcessor2_0", "EulerDiscreteScheduler", "DDIMScheduler"]


def __getattr__(name):
    # lazy: keeps `import diffsensei_amd.unet_config` (pure tables) usable by the oracle without touching torch ops
    if name == "UNetMangaModel":
        from .unet import UNetMangaModel
        return UNetMangaModel
    if name == "DiffSenseiPipeline":
        from .pipeline import DiffSenseiPipeline
        return DiffSenseiPipeline
    if name == "Resampler":
        from .resampler import Resampler
        return Resampler
    if name in ("AttnProcessor2_0", "MaskedIPAttnProcessor2_0"):
        from . import attention_processor
        return getattr(attention_processor, name)
    if name in ("EulerDiscreteScheduler", "DDIMScheduler"):
        from . import schedulers
        return getattr(schedulers, name)
    raise AttributeError(name)
