"""RADNeRFDataset.__getitem__ on the device, and a clip rendered straight from the dataset's driving signals."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dataset_fixture import write_synthetic_dataset
from genefaceplusplus_amd.dataset import RADNeRFDataset
from helpers import frame_case, build_model
from test_dataset_cpu import _hp


@pytest.fixture(scope="module")
def npy(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("ds") / "trainval_dataset.npy")
    return p, write_synthetic_dataset(p, T=22, H=64, W=64)


def test_getitem_inference_and_training(npy, oracle_mod):
    path, d = npy
    dev = torch.device("cuda:0")
    ds = RADNeRFDataset("val", _hp(), data_dir=path, training=False, device=dev, allow_bfm68_fallback=True)
    s = ds[1]
    assert s["rays_o"].shape == s["rays_d"].shape == (1, 64 * 64, 3) and s["rays_o"].is_cuda
    ref = oracle_mod.get_rays(ds.poses[1][None].numpy(), ds.intrinsics, 64, 64)
    np.testing.assert_array_equal(s["rays_o"].cpu().numpy(), ref["rays_o"])
    np.testing.assert_allclose(s["rays_d"].cpu().numpy(), ref["rays_d"], atol=2e-7)
    assert s["cond_wins"].shape == (5, 1, 204) and s["pose"].shape == (1, 6) and s["bg_img"].shape == (1, 4096, 3)
    np.testing.assert_allclose(s["bg_img"].cpu().numpy().reshape(64, 64, 3), d["bg_img"].astype(np.float32) / 255, atol=1e-6)
    # rectangle face mask (polygon_face_mask False): rows [16,48) x cols [16,48) of pixel centres
    fm = s["face_mask"].reshape(64, 64).cpu().numpy()
    assert fm[16:48, 16:48].all() and fm.sum() == 32 * 32
    assert torch.equal(s["cond_mask"], s["face_mask"].reshape(-1))
    tr = RADNeRFDataset("train", _hp(n_rays=500), data_dir=path, training=True, device=dev, allow_bfm68_fallback=True)
    t = tr[3]
    assert t["rays_o"].shape == (1, 500, 3) and t["bg_img"].shape == (1, 500, 3) and t["face_mask"].shape == (1, 500)
    tr.finetune_lip_flag = True
    t = tr[3]
    assert t["rays_o"].shape == (1, 8 * 8, 3)                                  # the lip rectangle of the fixture


def test_clip_from_dataset_equals_per_frame_render(npy):
    """dataset.clip_batch -> ClipRenderer == model.render() per frame with the dataset's own rays / windows (the loop of genefacepp_infer.py)."""
    from genefaceplusplus_amd.clip import ClipRenderer
    from genefaceplusplus_amd import frames
    path, _ = npy
    dev = torch.device("cuda:0")
    case = frame_case("may_torso", 64)
    model = build_model(case, dev, "fused")
    model.precision = "fp16"
    hp = dict(case["hp"], **{k: v for k, v in _hp().items() if k not in case["hp"]})
    ds = RADNeRFDataset("trainval", hp, data_dir=path, training=False, device=dev, allow_bfm68_fallback=True)
    cr = ClipRenderer(model, ds.H, ds.W, ds.intrinsics, bg_img=ds.bg_img.reshape(1, -1, 3), T_thresh=0.01, use_graph=True, lanes=1)
    clip = cr.prepare(ds.clip_batch(), dev)
    got = cr.render_to_device(clip, [0, 5, 21]).cpu().numpy()
    for k, i in enumerate((0, 5, 21)):
        s = ds[i]
        with torch.no_grad():
            # (the 6-vector pose on the DEVICE, like the clip renderer derives it: the CPU's atan2 / asin differ from the GPU's in the last bit,
            # which the pose-conditioned torso turns into single-LSB differences of the uint8 frame)
            from genefaceplusplus_amd.radnerfs import camera
            r = model.render(s["rays_o"], s["rays_d"], s["cond_wins"].to(dev), ds.bg_coords.to(dev), camera.convert_poses(ds.poses[i][None].float().to(dev)), index=0, bg_color=ds.bg_img.reshape(1, -1, 3),
                             perturb=False, T_thresh=0.01, max_steps=16, dt_gamma=hp["dt_gamma"])
        want = frames.to_uint8_hwc(r["rgb_map"].reshape(64, 64, 3)).cpu().numpy()
        np.testing.assert_array_equal(got[k], want)


def test_sample_key_set_matches_the_reference(tmp_path):
    """Every key the reference's __getitem__ returns (dataset_utils.py:318-434; the task steps read sample['bg_coords'], tasks/radnerfs/radnerf.py:115)
    -- 'camera' excepted, which needs the reference's Face3DHelper route."""
    from PIL import Image
    from test_dataset_cpu import REF_SAMPLE_KEYS
    p = str(tmp_path / "trainval_dataset.npy")
    d = write_synthetic_dataset(p, T=11, H=64, W=64)
    rng = np.random.default_rng(3)
    for s in d["train_samples"] + d["val_samples"]:
        s["torso_img_fname"] = str(tmp_path / f"torso_{s['idx']}.png")
        s["gt_img_fname"] = str(tmp_path / f"gt_{s['idx']}.png")
        Image.fromarray(rng.integers(0, 256, (64, 64, 4)).astype(np.uint8), "RGBA").save(s["torso_img_fname"])
        Image.fromarray(rng.integers(0, 256, (64, 64, 3)).astype(np.uint8), "RGB").save(s["gt_img_fname"])
    np.save(p, d, allow_pickle=True)
    dev = torch.device("cuda:0")
    for variant in ("may_torso", "may_torso_sr"):
        for training in (True, False):
            ds = RADNeRFDataset("train", _hp(variant, n_rays=300), data_dir=p, training=training, device=dev, allow_bfm68_fallback=True)
            s = ds[2]
            assert set(s.keys()) == REF_SAMPLE_KEYS - {"camera"}, (variant, training, set(s.keys()) ^ REF_SAMPLE_KEYS)
            n = s["rays_o"].shape[1]
            assert s["bg_coords"].shape == (1, n, 2) and s["bg_img"].shape == (1, n, 3) and s["gt_img"].shape == (1, n, 3) and s["bg_torso_img"].shape == (1, n, 3)
            assert s["torso_img"].shape == (64, 64, 4) and s["gt_img_512"].shape == (1, 3, 64, 64)
            if training and variant == "may_torso":
                assert n == 300                                                # gathered by rays['inds'] like the reference (dataset_utils.py:428-430)
                full = ds.bg_coords.to(dev)
                # the sampled coordinates are rows of the full table
                assert bool((s["bg_coords"][0, :, None, :] == full[0, None, :, :]).all(-1).any(-1).all())
            else:
                assert torch.equal(s["bg_coords"].cpu(), ds.bg_coords.cpu())


@pytest.mark.parametrize("variant", ["may_torso", "may_torso_sr"])
def test_bench_takes_a_checkpoint_directory_and_a_dataset_in_the_reference_layout(tmp_path, variant):
    """SURVEY 8d, last bullet: files dropped at checkpoints/motion2video_nerf/<name> + data/binary/videos/<id>/trainval_dataset.npy are benched by the same harness.
    Here: a synthetic checkpoint written the way the reference's trainer leaves it (synthetic.write_checkpoint) and a synthetic dataset file in the binarizer's
    schema; `bench.py --ckpt-dir --data-dir` must load both through the reference layout, say so in the line, and agree with the CPU oracle on the same weights."""
    import json
    import subprocess
    import sys
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.configs import may_hparams
    side = 512                  # the file holds the 512 x 512 video frames; the *_sr models' reader halves the ray grid (dataset_utils.py:216-230)
    hp = may_hparams(variant)
    data_dir = tmp_path / "binary" / hp["video_id"]
    data_dir.mkdir(parents=True)
    write_synthetic_dataset(str(data_dir / "trainval_dataset.npy"), T=11, H=side, W=side)
    ckpt_dir = str(tmp_path / "ckpt")
    syn.write_checkpoint(ckpt_dir, variant, extra_hparams={"binary_data_dir": str(tmp_path / "binary"), "infer_bg_img_fname": "", "infer_smooth_camera_path": False,
                                                           "polygon_face_mask": False, "n_rays": 65536, "load_imgs_to_memory": False})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ckpt-dir", ckpt_dir, "--data-dir", str(data_dir / "trainval_dataset.npy"), "--steps", "12", "--warmup", "4",
                        "--precision", "fp16", "--no-modes", "--no-configs", "--no-grid-stage", "--no-cpu-baseline", "--ckpt-parity", "1"], cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    assert d["data"] == "real checkpoint + real driving signals" and "model_ckpt_steps_250000.ckpt" in d["config"]["workload"] and d["value"] > 0
    chk = d["config"]["ckpt_parity_fp32_vs_oracle"]
    assert isinstance(chk, list) and len(chk) == 1, chk
    assert chk[0]["frac_over_2e-4"] <= 5e-4, chk                                    # SURVEY 8c's tolerance, on the checkpoint's own weights and the dataset's own pose
