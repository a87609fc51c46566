"""GPU: device pre-processing of the character references (csrc/preprocess.hip, diffsensei_amd/preprocess.py) against the
oracle that is pinned bit-exact to Pillow (oracle/image_preprocess_ref.py).  Bar: the resized + cropped BYTES are identical
(integer work); the normalised fp32 pixels agree to 1e-6 (one division, compiled with fast-math)."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("hw", [(300, 200), (224, 386), (97, 333), (640, 512), (224, 224)])
def test_clip_and_vit_preprocess_match_pillow_semantics(hip_lib, hw):
    from diffsensei_amd.preprocess import DevicePreprocessor
    from oracle import image_preprocess_ref as P
    img = np.random.RandomState(hw[0] * 7 + hw[1]).randint(0, 256, hw + (3,), dtype=np.uint8)
    img[: hw[0] // 3] = 255                                              # saturated area: the bicubic lobes clamp
    pil = Image.fromarray(img)
    pre = DevicePreprocessor(DEV)
    pre.keep_bytes = True
    got = pre.clip([pil, pil.convert("L")])                              # second image: grey -> RGB conversion path
    nh, nw = P.shortest_edge_size(hw[0], hw[1], 224)
    ref_bytes = P.pil_resize_u8(img, nh, nw, "bicubic")[(nh - 224) // 2:(nh - 224) // 2 + 224,
                                                         (nw - 224) // 2:(nw - 224) // 2 + 224]
    assert np.array_equal(pre.last_bytes[0].cpu().numpy(), ref_bytes), "CLIP bytes differ from Pillow's"
    assert got.shape == (2, 3, 224, 224) and got.dtype == torch.float32
    assert np.abs(got[0].cpu().numpy() - P.clip_preprocess(img)).max() <= 1e-6
    grey = np.asarray(pil.convert("L").convert("RGB"))
    assert np.abs(got[1].cpu().numpy() - P.clip_preprocess(grey)).max() <= 1e-6
    got_v = pre.vit([pil])
    assert np.array_equal(pre.last_bytes[0].cpu().numpy(), P.pil_resize_u8(img, 224, 224, "bilinear"))
    assert np.abs(got_v[0].cpu().numpy() - P.vit_preprocess(img)).max() <= 1e-6
