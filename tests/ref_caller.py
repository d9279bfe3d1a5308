#!/usr/bin/env python
"""Test helper (child interpreter; not shipped): run the REFERENCE's OWN CALLER -- ``GeneFace2Infer.load_secc2video`` and
``GeneFace2Infer.forward_secc2video`` of /root/reference/inference/genefacepp_infer.py, unmodified, imported from the mount -- on a
synthetic on-disk checkpoint (genefaceplusplus_amd.synthetic.write_checkpoint), on the CPU with the native kernels served by the oracle:

    --mode reference   modules.radnerfs.* are the reference's own classes over oracle/ref_backends.py (its four extension names)
    --mode product     modules.radnerfs.* are this package's drop-in classes (compat.install()), their C-ABI calls redirected to the oracle
                       (tests/product_on_oracle.py) -- the one place where the product is allowed to run on CPU tensors

Everything between ``set_hparams(config.yaml)`` and the uint8 frames handed to the video writer is the reference's code in both modes:
``load_ckpt(model, dir, model_name='model', strict=True)``, ``torch.compile(model)``, the dataset construction (served by
genefaceplusplus_amd.dataset.RADNeRFDataset on a synthetic trainval_dataset.npy in both modes: the reference's reader needs the 3DMM
assets), ``render(..., **hparams)`` under ``torch.cuda.amp.autocast``, ``model_out['rgb_map'][0].reshape([512,512,3])`` /
``model_out['sr_rgb_map'][0]``, the uint8 conversion.  What is substituted: third-party imports the path never calls (librosa, cv2, kornia,
mediapipe, tensorboardX, trimesh, mcubes, lpips), ``imageio`` (a writer that keeps the frames), ``os.system`` (ffmpeg), ``Tensor.cuda()``.

    python tests/ref_caller.py --mode product --variant may_torso --frames 2 --out /tmp/x.npz
"""
import argparse
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("GFPP_REFERENCE", "/root/reference")
STUB_ROOTS = ("librosa", "cv2", "kornia", "mediapipe", "tensorboardX", "trimesh", "mcubes", "lpips", "pytorch3d", "face_alignment")


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


class _StubModule(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


class CapturedVideo:
    """imageio.get_writer(...) stand-in: keeps what the caller appends."""
    frames = []

    def __init__(self, *a, **k):
        pass

    def append_data(self, img):
        assert img.dtype == np.uint8
        CapturedVideo.frames.append(np.array(img, copy=True))

    def close(self):
        pass


def prepare(mode, data_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    sys.path.insert(0, REF)
    sys.meta_path.insert(0, _StubFinder())
    imageio = types.ModuleType("imageio")
    imageio.get_writer = CapturedVideo
    sys.modules["imageio"] = imageio
    torch.Tensor.cuda = lambda self, *a, **k: self
    if mode == "reference":
        from oracle import ref_backends
        ref_backends.install()
    else:
        from genefaceplusplus_amd import compat
        compat.install()
        import ctypes
        import product_on_oracle
        oracle_dispatch = product_on_oracle.dispatch

        def dispatch(name, *a):
            """gfpp_get_rays: the stand-in for the ray kernel evaluates utils.py:352-363 with the SAME torch expression the reference's own
            get_rays uses in --mode reference (the numpy oracle differs from it in the last bit of ~0.01 % of the direction components, and a
            frame of 786 432 truncated uint8 values shows one such bit); everything else goes to the C oracle."""
            if name != "gfpp_get_rays":
                return oracle_dispatch(name, *a)
            fp = ctypes.POINTER(ctypes.c_float)
            pose = torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(a[0], fp), shape=(16,)).reshape(1, 4, 4).copy())
            fx, fy, cx, cy, H, W = a[1], a[2], a[3], a[4], int(a[5]), int(a[6])
            jj, ii = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
            i = ii.reshape([1, H * W]) + 0.5
            j = jj.reshape([1, H * W]) + 0.5
            zs = torch.ones_like(i)
            directions = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
            directions = directions / torch.norm(directions, dim=-1, keepdim=True)
            rays_d = directions @ pose[:, :3, :3].transpose(-1, -2)
            np.ctypeslib.as_array(ctypes.cast(a[8], fp), shape=(H * W, 3))[:] = rays_d[0].numpy()
            np.ctypeslib.as_array(ctypes.cast(a[7], fp), shape=(H * W, 3))[:] = pose[0, :3, 3].numpy()[None]
            return 0
        product_on_oracle.dispatch = dispatch
        product_on_oracle.patch()
        # the SR stage has no CPU kernels to redirect: the harness runs the module's own torch-op formulation (the training path of
        # genefaceplusplus_amd/radnerfs/superres.py, same parameters), drawing the noise like the reference does (one randn per layer, in order)
        from genefaceplusplus_amd.radnerfs import superres
        superres.Superresolution.forward = lambda self, rgb, noise_mode="random", **kw: self._forward_autograd(rgb.float(), noise_mode)
    # the dataset: the reference's reader needs the BFM assets (Face3DHelper); both modes get this package's reader of the same file schema
    from genefaceplusplus_amd import dataset as gds

    class _Dataset(gds.RADNeRFDataset):
        def __init__(self, prefix, data_dir_=None, training=True):
            from utils.commons.hparams import hparams as ref_hparams
            super().__init__(prefix, ref_hparams, data_dir=data_dir_ or data_dir, training=training, device="cpu", allow_bfm68_fallback=True)

    mod = types.ModuleType("tasks.radnerfs.dataset_utils")
    mod.RADNeRFDataset = _Dataset
    mod.get_boundary_mask = mod.dilate_boundary_mask = mod.get_lf_boundary_mask = _Any()
    sys.modules["tasks.radnerfs.dataset_utils"] = mod
    os.chdir(REF)


def run(mode, variant, n_frames, work, thresh=0.01):
    from dataset_fixture import write_synthetic_dataset
    sys.path.insert(0, REPO)
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.configs import may_hparams
    hp = may_hparams(variant)
    side = 512                                                   # the frame the caller writes; the *_sr models render 256^2 rays
    data_dir = os.path.join(work, "binary", hp["video_id"])
    os.makedirs(data_dir, exist_ok=True)
    write_synthetic_dataset(os.path.join(data_dir, "trainval_dataset.npy"), T=11, H=side, W=side)
    ckpt_dir = os.path.join(work, "ckpt_" + variant)
    syn.write_checkpoint(ckpt_dir, variant, extra_hparams={"binary_data_dir": os.path.join(work, "binary"), "infer_bg_img_fname": "",
                                                           "infer_smooth_camera_path": False, "polygon_face_mask": False, "n_rays": 65536,
                                                           "load_imgs_to_memory": False})
    prepare(mode, data_dir)
    import warnings
    warnings.filterwarnings("ignore")
    import inference.genefacepp_infer as gi
    from utils.commons.hparams import hparams as ref_hparams

    shell = object.__new__(gi.GeneFace2Infer)                    # no __init__: that loads audio2motion, the 3DMM, HuBERT ...
    model = shell.load_secc2video("", ckpt_dir)                  # genefacepp_infer.py:163-191, the reference's code
    shell.secc2video_model = model
    shell.secc2video_model.to("cpu").eval()                      # GeneFace2Infer.__init__ (genefacepp_infer.py:128), on the torch.compile wrapper
    assert not getattr(model, "_orig_mod", model).training
    if mode == "product":
        # the fused frame pipeline has no extension-level calls the oracle could serve (it is GPU-tested frame by frame against the oracle);
        # on the CPU the drop-in runs its reference-shaped executor: the same Python layer, one C-ABI call per reference extension call
        getattr(model, "_orig_mod", model).executor = "staged"
    info = {"model_class": type(getattr(model, "_orig_mod", model)).__module__ + "." + type(getattr(model, "_orig_mod", model)).__name__,
            "compiled_wrapper": type(model).__name__, "with_sr": bool(ref_hparams.get("with_sr", False)),
            "closed_eye": shell.closed_eye_area_percent, "opened_eye": shell.opened_eye_area_percent}
    # the render batch, the way prepare_batch_from_inp / get_pose_from_ds lay it out (genefacepp_infer.py:246-275, 411-431)
    ds = shell.dataset
    batch = {"rays_o": [], "rays_d": [], "poses": [], "cond_wins": [], "lm68": [], "eye_area_percent": []}
    for i in range(n_frames):
        ngp_pose = torch.from_numpy(syn.synthetic_pose(i))[None]
        rays = gi.get_rays(ngp_pose.cuda(), ds.intrinsics, ds.H, ds.W, N=-1)
        batch["rays_o"].append(rays["rays_o"].cuda())
        batch["rays_d"].append(rays["rays_d"].cuda())
        batch["poses"].append(gi.convert_poses(ngp_pose).cuda())
        fi = syn.synthetic_frame_inputs(hp, i)
        batch["cond_wins"].append(torch.from_numpy(fi["cond"]))
        batch["lm68"].append(torch.from_numpy(fi["lm68"]))
        batch["eye_area_percent"].append(torch.from_numpy(fi["eye_area_percent"]))
    batch["lm68"] = torch.stack(batch["lm68"])
    batch["eye_area_percent"] = torch.stack(batch["eye_area_percent"])
    batch["bg_img"] = ds.bg_img.reshape([1, -1, 3]).cuda()
    batch["bg_coords"] = ds.bg_coords.cuda()
    shell.wav16k_name = os.path.join(work, "none_16k.wav")
    inp = {"low_memory_usage": False, "raymarching_end_threshold": thresh,    # 0.05 = the CLI's --fast (genefacepp_infer.py:591-592)
            "debug": False, "out_name": os.path.join(work, f"{variant}_{mode}.mp4")}
    os.system = lambda cmd: 0                                     # ffmpeg mux + rm of the temporaries
    import random
    random.seed(0)
    torch.manual_seed(1234)                                       # the SR layers draw their noise from torch's generator ('random' mode)
    CapturedVideo.frames = []
    shell.forward_secc2video(batch, inp)                          # genefacepp_infer.py:433-519, the reference's code
    frames = np.stack(CapturedVideo.frames)
    assert frames.shape == (n_frames, 512, 512, 3), frames.shape
    return frames, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["reference", "product"])
    ap.add_argument("--variant", required=True, choices=["may_torso", "may_torso_sr", "may_head"])
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--thresh", type=float, default=0.01, help="raymarching_end_threshold; --fast sets 0.05 (genefacepp_infer.py:566,591-592)")
    ap.add_argument("--work", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    frames, info = run(a.mode, a.variant, a.frames, a.work, a.thresh)
    np.savez_compressed(a.out, frames=frames, info=np.array([repr(info)]))
    print("REFCALLER", a.mode, a.variant, frames.shape, info)


if __name__ == "__main__":
    main()
