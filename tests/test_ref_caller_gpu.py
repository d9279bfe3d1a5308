"""The reference caller's sequence (tests/caller_sequence.py = genefacepp_infer.py:163-191 + :433-486 restated; the reference file itself runs in
tests/test_ref_caller_cpu.py) on the REAL kernels: checkpoint directory on disk -> yaml hparams -> class -> strict load -> torch.compile(model) ->
.to('cuda').eval() -> render(..., **hparams) per frame under autocast -> uint8 frames, against tests/golden/caller_golden.npz (the frames the
reference's own caller produced with the reference's own classes on the CPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _psnr(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def _setup(tmp_path, variant):
    from dataset_fixture import write_synthetic_dataset
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.configs import may_hparams
    from genefaceplusplus_amd.dataset import RADNeRFDataset
    import caller_sequence as cs
    hp = may_hparams(variant)
    data_dir = tmp_path / "binary" / hp["video_id"]
    data_dir.mkdir(parents=True)
    write_synthetic_dataset(str(data_dir / "trainval_dataset.npy"), T=11, H=512, W=512)       # same fixture as tests/ref_caller.py
    ckpt_dir = str(tmp_path / ("ckpt_" + variant))
    syn.write_checkpoint(ckpt_dir, variant, extra_hparams={"binary_data_dir": str(tmp_path / "binary"), "infer_bg_img_fname": "", "infer_smooth_camera_path": False,
                                                           "polygon_face_mask": False, "n_rays": 65536, "load_imgs_to_memory": False})
    dev = torch.device("cuda:0")
    model, hparams = cs.load_secc2video(ckpt_dir, dev)
    ds = RADNeRFDataset("trainval", hparams, training=False, device=dev, allow_bfm68_fallback=True)      # load_secc2video's dataset (:181-182)
    return cs, model, hparams, ds, dev


def test_caller_sequence_torso_512_under_torch_compile(tmp_path):
    from genefaceplusplus_amd.radnerfs import camera
    g = np.load(os.path.join(HERE, "golden", "caller_golden.npz"))
    cs, model, hparams, ds, dev = _setup(tmp_path, "may_torso")
    inner = getattr(model, "_orig_mod", model)
    assert type(model).__name__ == "OptimizedModule" and not inner.training and inner.density_bitfield.is_cuda
    assert inner.executor == "fused"
    batch = cs.make_batch(ds, hparams, 2, dev, camera.get_rays, camera.convert_poses)
    want = g["may_torso.sub"]
    # exact-fp32 arithmetic (no autocast): the reference's bytes up to the kernels' fp32 summation order and 1-ulp ray directions
    f32 = cs.forward_secc2video(model, hparams, batch, 0.01, autocast=False)
    assert f32.shape == (2, 512, 512, 3) and f32.dtype == np.uint8
    d = np.abs(f32[:, ::4, ::4].astype(np.int32) - want.astype(np.int32))
    stats = {"differ": float((d != 0).mean()), "over_1": float((d > 1).mean()), "max": int(d.max()), "psnr": _psnr(f32[:, ::4, ::4], want)}
    print("fp32 vs the reference caller's frames", stats)
    assert stats["over_1"] <= 5e-4 and stats["differ"] <= 2e-2 and stats["psnr"] >= 60, stats
    # the caller's own setting: torch.cuda.amp.autocast(enabled=True) -> 16-bit MFMA operands (precision 'auto' follows the context)
    f16 = cs.forward_secc2video(model, hparams, batch, 0.01, autocast=True)
    s16 = {"psnr": _psnr(f16[:, ::4, ::4], want), "max": int(np.abs(f16[:, ::4, ::4].astype(np.int32) - want.astype(np.int32)).max())}
    print("autocast vs the reference caller's frames", s16)
    assert s16["psnr"] >= 45, s16
    assert not np.array_equal(f16, f32)                       # the autocast context really switched the arithmetic
    # the same frames through the module without the torch.compile wrapper and the on-disk round trip: identical bytes
    from genefaceplusplus_amd import synthetic as syn, radnerfs
    plain = radnerfs.RADNeRFTorso(hparams)
    plain.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in syn.synthetic_state_dict(hparams, "may_torso").items()}, strict=True)
    plain = plain.to(dev).eval()
    np.testing.assert_array_equal(cs.forward_secc2video(plain, hparams, batch, 0.01, autocast=False), f32)


def test_caller_sequence_fast_threshold_eight_frames(tmp_path):
    """`--fast` (raymarching_end_threshold 0.05, genefacepp_infer.py:566,591-592), 8 frames (eight poses, conditioning windows and individual-code rows) against the
    frames the reference's own caller produced with the reference's own classes (tests/golden/make_golden_caller.py --only fast)."""
    from genefaceplusplus_amd.radnerfs import camera
    g = np.load(os.path.join(HERE, "golden", "caller_golden.npz"))
    want = g["may_torso.fast.sub"]
    assert want.shape[0] == 8
    cs, model, hparams, ds, dev = _setup(tmp_path, "may_torso")
    batch = cs.make_batch(ds, hparams, 8, dev, camera.get_rays, camera.convert_poses)
    f32 = cs.forward_secc2video(model, hparams, batch, 0.05, autocast=False)
    d = np.abs(f32[:, ::4, ::4].astype(np.int32) - want.astype(np.int32))
    per_frame = [float((x > 1).mean()) for x in d]
    stats = {"differ": float((d != 0).mean()), "over_1": float((d > 1).mean()), "max": int(d.max()), "psnr": _psnr(f32[:, ::4, ::4], want), "over_1_per_frame": per_frame}
    print("--fast, fp32 vs the reference caller's 8 frames", stats)
    assert stats["over_1"] <= 5e-4 and max(per_frame) <= 1e-3 and stats["differ"] <= 2e-2 and stats["psnr"] >= 60, stats
    f16 = cs.forward_secc2video(model, hparams, batch, 0.05, autocast=True)
    p16 = [_psnr(a, b) for a, b in zip(f16[:, ::4, ::4], want)]
    print("--fast, autocast per-frame PSNR", p16)
    assert min(p16) >= 45, p16
    # and the default threshold renders another picture with the same graph key family (T_thresh is part of the captured frame's key)
    f16_default = cs.forward_secc2video(model, hparams, batch, 0.01, autocast=True)
    assert not np.array_equal(f16_default, f16)
    np.testing.assert_array_equal(cs.forward_secc2video(model, hparams, batch, 0.05, autocast=True), f16)


def test_caller_sequence_torso_sr_under_torch_compile(tmp_path):
    from genefaceplusplus_amd.radnerfs import camera
    g = np.load(os.path.join(HERE, "golden", "caller_golden.npz"))
    cs, model, hparams, ds, dev = _setup(tmp_path, "may_torso_sr")
    assert (ds.H, ds.W) == (256, 256)
    batch = cs.make_batch(ds, hparams, 2, dev, camera.get_rays, camera.convert_poses)
    torch.manual_seed(7)
    out = cs.forward_secc2video(model, hparams, batch, 0.01, autocast=True)
    assert out.shape == (2, 512, 512, 3)
    # the reference draws fresh unit-normal noise per SR layer and frame (networks_stylegan2.py:329-331) from ITS generator: frames agree up to that noise
    want = g["may_torso_sr.sub"]
    stats = {"psnr": _psnr(out[:, ::4, ::4], want), "mean_abs": float(np.abs(out[:, ::4, ::4].astype(np.int32) - want.astype(np.int32)).mean()),
             "mean": float(out.mean()), "want_mean": float(want.mean())}
    print("sr frames vs the reference caller's (different noise draws)", stats)
    # (the synthetic SR net's noise_strength 0.05-0.1 on unit normals through two more conv layers: ~18 dB between two draws of the same frame)
    assert stats["psnr"] >= 15 and abs(stats["mean"] - stats["want_mean"]) <= 3.0, stats
    # a second pass over the same batch (graph replays now, other noise draws): the same picture up to the SR noise
    again = cs.forward_secc2video(model, hparams, batch, 0.01, autocast=True)
    s2 = {"psnr": _psnr(again, out)}
    print("second pass vs first", s2)
    assert s2["psnr"] >= 15, s2
