"""Super-resolution stage of the *_sr models (reference: modules/radnerfs/radnerf_sr.py:14-43 on top of
modules/eg3ds/models/superresolution.py:159-258 and networks_stylegan2.py:37-94, 286-478).

SURVEY.md section 8f-1 ranks this stage "next" after the NeRF hot path; until it is built the class below only
carries the interface (``input_resolution``) and refuses to run, so that the NeRF part of the *_sr models (256x256 rays,
landmark-conditioned head-aware torso) can be rendered and verified on its own.
"""
import torch.nn as nn


class Superresolution(nn.Module):
    ready = False

    def __init__(self, channels=3, img_resolution=512, sr_antialias=True):
        super().__init__()
        assert img_resolution == 512
        self.input_resolution = 256
        self.w_dim = 16

    def forward(self, rgb, **block_kwargs):
        raise NotImplementedError("the StyleGAN2 super-resolution stage is not built yet (SURVEY.md 8f-1); "
                                  "use result['rgb_map'] (256x256) of the *_sr models")
