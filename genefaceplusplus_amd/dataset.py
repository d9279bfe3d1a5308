"""Reader of the reference's binarised person dataset ``trainval_dataset.npy`` -- the frame-I/O row next to the render path.

Restates ``tasks/radnerfs/dataset_utils.py:160-420`` (``RADNeRFDataset``) on top of the schema ``data_gen/runs/binarizer_nerf.py:197-320`` writes:

    {'H','W','focal','cx','cy', 'bg_img' u8 [H,W,3], 'id' [T,80], 'exp' [T,64], 'euler' [T,3], 'trans' [T,3], 'eye_area_percent' [T,1],
     'idexp_lm3d' [T,204], 'idexp_lm3d_mean' [204], 'idexp_lm3d_std' [204], 'hubert', 'mel', 'f0', ('esperanto' [T,16,44]),
     'train_samples' / 'val_samples': [{'idx', 'head_img_fname', 'torso_img_fname', 'gt_img_fname', 'face_rect', 'lip_rect', 'c2w' [4,4]}]}

What it yields is exactly what the renderer and ``clip.ClipRenderer.prepare`` eat: ngp poses (``nerf_matrix_to_ngp`` of c2w with ``camera_scale`` /
``camera_offset``), intrinsics (dataset_utils.py:216-230: centre H/2, focal rescaled by (H/2)/cx), ``bg_img``, ``bg_coords``, conditioning windows
(``get_audio_features(conds, att_mode=2, index)``), eye-area values and -- when they can be had -- the 68 2-D landmarks of the torso model.

Two things of the reference need assets that are not part of this path and are therefore OPTIONAL inputs here:
  * the 3DMM (``deep_3drecon`` BFM files behind ``Face3DHelper``): the reference re-derives the landmark conditioning and the 2-D landmarks from
    'id'/'exp'/'euler'/'trans' with it.  Pass ``face3d_helper=<the reference's object>`` to get the same arrays.  Without it the reader REFUSES
    the landmark conditioning unless ``allow_bfm68_fallback=True``: the file's own 'idexp_lm3d' is the binarizer's Face3DHelper(keypoint_mode='lm68')
    reconstruction = the BFM's 68 keypoint vertices (binarizer_nerf.py:241,335), whereas the reference dataset rebuilds in keypoint_mode='mediapipe'
    and takes index_lm68_from_lm478 (dataset_utils.py:247-273) -- different mesh vertices, i.e. a different conditioning signal from the one
    reference checkpoints were trained on.  ``lm68s`` then comes from an 'lm68' / 'lm2d' array in the file if present, else None (the SR torso
    model then needs ``lm68=`` from the caller, as at inference where it comes from audio2motion, genefacepp_infer.py:420-431);
  * image decoding for ``gt_img`` / ``torso_img`` (training targets): PIL if importable; inference never reads them.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .radnerfs import camera


def _t(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to(dtype)


def smooth_camera_path(poses, kernel_size=7):
    """dataset_utils.py:138-157: box filter over the translations, rotation = mean rotation of the window (scipy Rotation.mean)."""
    from scipy.spatial.transform import Rotation
    poses = np.array(poses, dtype=np.float64, copy=True)
    N, K = poses.shape[0], kernel_size // 2
    trans, rots = poses[:, :3, 3].copy(), poses[:, :3, :3].copy()
    for i in range(N):
        a, b = max(0, i - K), min(N, i + K + 1)
        poses[i, :3, 3] = trans[a:b].mean(0)
        try:
            poses[i, :3, :3] = Rotation.from_matrix(rots[a:b]).mean().as_matrix()
        except Exception:
            poses[i, :3, :3] = rots[i] if i == 0 else poses[i - 1, :3, :3]
    return poses


def _load_image_u8(path):
    try:
        from PIL import Image
    except Exception as exc:                                                   # pragma: no cover
        raise RuntimeError(f"reading {path} needs PIL (only training targets are images; inference does not read them)") from exc
    return torch.from_numpy(np.array(Image.open(path)))


class RADNeRFDataset(torch.utils.data.Dataset):
    """``RADNeRFDataset(prefix, hparams, data_dir=None | npy path, training=True, device=None, face3d_helper=None, allow_bfm68_fallback=False)``.

    Attributes the reference's callers use (genefacepp_infer.py:246-275, tasks/radnerfs/*): ``H, W, focal, cx, cy, near, far, intrinsics, poses
    [F,4,4], bg_img [H,W,3], bg_img_512, bg_coords [1,HW,2], conds, eye_area_percents, lm68s, lips_rect, ds_dict, samples``."""

    def __init__(self, prefix, hparams, data_dir=None, training=True, device=None, face3d_helper=None, allow_bfm68_fallback=False):
        super().__init__()
        self.hparams = hp = hparams
        self.allow_bfm68_fallback = bool(allow_bfm68_fallback)
        if data_dir is None:
            data_dir = os.path.join(hp["binary_data_dir"], hp["video_id"])
        path = data_dir if data_dir.endswith(".npy") else os.path.join(data_dir, "trainval_dataset.npy")
        self.ds_dict = ds = np.load(path, allow_pickle=True).tolist()
        if prefix == "train":
            raw = list(ds["train_samples"])
        elif prefix == "val":
            raw = list(ds["val_samples"])
        elif prefix == "trainval":
            raw = list(ds["train_samples"]) + list(ds["val_samples"])
        else:
            raise ValueError("prefix should in train/val !")
        n_lim = hp.get("num_train_samples", 0)
        if n_lim and len(raw) >= n_lim:
            raw = raw[:n_lim]
        self.samples = [dict(s, c2w=_t(s["c2w"])) for s in raw]
        self.prefix, self.training = prefix, training
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.cond_type = hp["cond_type"]
        self.H, self.W = int(ds["H"]), int(ds["W"])
        if hp.get("with_sr"):
            self.H, self.W = self.H // 2, self.W // 2                         # the NeRF renders half resolution, the SR net restores it
        self.focal, self.cx, self.cy = float(ds["focal"]), float(ds["cx"]), float(ds["cy"])
        self.near, self.far = hp["near"], hp["far"]
        self._init_background(ds, hp)
        cx, cy = self.H / 2, self.W / 2
        self.intrinsics = np.array([self.focal * (cx / self.cx), self.focal * (cy / self.cy), cx, cy])
        if not training and hp.get("infer_smooth_camera_path", False):
            smo = smooth_camera_path(torch.stack([s["c2w"] for s in self.samples]).numpy(), hp.get("infer_smooth_camera_path_kernel_size", 7))
            for s, m in zip(self.samples, smo):
                s["c2w"] = torch.from_numpy(m).float()
        key = "mp_c2w" if hp.get("use_mp_pose", False) else "c2w"
        self.poses = torch.from_numpy(np.stack([camera.nerf_matrix_to_ngp(np.asarray(s[key]), scale=hp["camera_scale"], offset=hp["camera_offset"])
                                                for s in self.samples]))
        if torch.any(torch.isnan(self.poses)):
            raise ValueError("Found NaN in transform_matrix, please check the face_tracker process!")
        self.bg_coords = camera.get_bg_coords(self.H, self.W, "cpu")
        self._init_conditioning(ds, hp, face3d_helper)
        self.finetune_lip_flag = False
        self.lips_rect = [s["lip_rect"] for s in self.samples]
        if hp.get("with_sr"):
            self.lips_rect = (np.array(self.lips_rect) / 2).astype(int).tolist()
        self.global_step = 0

    # -- pieces of __init__ ----------------------------------------------------------------------------------------------------
    def _init_background(self, ds, hp):
        name = hp.get("infer_bg_img_fname", "")
        if name == "":
            bg = torch.from_numpy(np.asarray(ds["bg_img"])).float() / 255.0
            self.bg_img_512 = bg.to(self.device)
            bg = F.interpolate(bg.unsqueeze(0).permute(0, 3, 1, 2), mode="bilinear", size=(self.H, self.W), antialias=True).permute(0, 2, 3, 1)
            bg = bg.reshape(self.H, self.W, 3)
        elif name in ("white", "black"):
            bg = torch.full((self.H, self.W, 3), 1.0 if name == "white" else 0.0)
            self.bg_img_512 = torch.full((int(ds["H"]), int(ds["W"]), 3), 1.0 if name == "white" else 0.0, device=self.device)
        else:
            img = _load_image_u8(name)[..., :3].float() / 255.0               # RGB (the reference converts cv2's BGR to RGB)
            self.bg_img_512 = F.interpolate(img.unsqueeze(0).permute(0, 3, 1, 2), size=(int(ds["H"]), int(ds["W"])), mode="area").permute(0, 2, 3, 1)[0].to(self.device)
            bg = F.interpolate(img.unsqueeze(0).permute(0, 3, 1, 2), size=(self.H, self.W), mode="area").permute(0, 2, 3, 1)[0]
        self.bg_img = bg.to(self.device)

    def _init_conditioning(self, ds, hp, helper):
        n = len(self.samples)
        take = (lambda a: a[:n]) if self.prefix == "train" else (lambda a: a[-n:] if self.prefix == "val" else a)
        self.lm2ds = self.lm68s = self.eg3d_cameras = None
        self.eye_area_percents = take(_t(ds["eye_area_percent"])) if "eye_area_percent" in ds else None
        self.idexp_lm3d_mean = _t(ds["idexp_lm3d_mean"]) if "idexp_lm3d_mean" in ds else None
        self.idexp_lm3d_std = _t(ds["idexp_lm3d_std"]) if "idexp_lm3d_std" in ds else None
        if self.cond_type == "deepspeech":
            raise NotImplementedError("We no longer support DeepSpeech")        # the reference's own message
        if self.cond_type == "esperanto":
            self.conds = torch.tensor(np.asarray(ds["esperanto"]))             # [T, 16, 44]
            return
        if self.cond_type != "idexp_lm3d_normalized":
            raise NotImplementedError
        mode = hp.get("nerf_keypoint_mode", "lm68")
        if helper is not None:                                                  # the reference's route (needs its BFM assets)
            from data_gen.utils.mp_feature_extractors.face_landmarker import index_lm68_from_lm478, index_lm131_from_lm478
            id_, exp = _t(ds["id"]), _t(ds["exp"])
            arr = helper.reconstruct_idexp_lm3d(id_, exp)
            normed = (arr - arr.mean(dim=0, keepdim=True)) / arr.std(dim=0, keepdim=True)
            self.lm2ds = take(helper.reconstruct_lm2d_nerf(id_, exp, _t(ds["euler"]), _t(ds["trans"])))
            self.lm68s = torch.as_tensor(self.lm2ds[:, index_lm68_from_lm478, :])
            try:                                                                # sample['camera'] (dataset_utils.py:268-269, 335); only the eg3d-style tasks read it
                from data_gen.eg3d.convert_to_eg3d_convention import get_eg3d_convention_camera_pose_intrinsic
                cam = get_eg3d_convention_camera_pose_intrinsic({"euler": _t(ds["euler"]), "trans": _t(ds["trans"])})
                self.eg3d_cameras = _t(np.concatenate([cam["c2w"].reshape([-1, 16]), cam["intrinsics"].reshape([-1, 9])], axis=-1))
            except ImportError:
                self.eg3d_cameras = None
            sel = {"lm68": index_lm68_from_lm478, "lm131": index_lm131_from_lm478, "lm468": slice(None)}.get(mode)
            if sel is None:
                raise NotImplementedError()
            normed = normed[:, sel]
            self.keypoint_num = normed.shape[1]
        else:
            # NOT what reference checkpoints were trained on: the binarizer stores ds['idexp_lm3d'] from Face3DHelper(keypoint_mode='lm68') = the BFM's own
            # 68 keypoint vertices (binarizer_nerf.py:241,335), while the reference dataset rebuilds the landmarks in keypoint_mode='mediapipe' and takes
            # index_lm68_from_lm478 of those (dataset_utils.py:247-273) -- different mesh vertices.  Served only on explicit request.
            if mode != "lm68":
                raise NotImplementedError("without the reference's Face3DHelper only nerf_keypoint_mode='lm68' can be served (the file stores the lm68 reconstruction)")
            if not self.allow_bfm68_fallback:
                raise ValueError("RADNeRFDataset: no face3d_helper given.  The file's own 'idexp_lm3d' holds the BFM-68 keypoints, not the mediapipe-indexed lm68 "
                                 "the reference conditions on (dataset_utils.py:247-273): pass the reference's Face3DHelper(keypoint_mode='mediapipe'), or "
                                 "allow_bfm68_fallback=True to condition on the stored array anyway (a different signal from a reference checkpoint's)")
            import warnings
            warnings.warn("RADNeRFDataset: conditioning on the file's BFM-68 'idexp_lm3d' (allow_bfm68_fallback=True); reference checkpoints were trained on "
                          "the mediapipe-indexed lm68 landmarks, so this is NOT the signal they expect", stacklevel=3)
            arr = _t(ds["idexp_lm3d"]).reshape(-1, 68, 3)
            normed = (arr - arr.mean(dim=0, keepdim=True)) / arr.std(dim=0, keepdim=True)
            self.keypoint_num = 68
            for key in ("lm68", "lm68s", "lm2d", "lm2ds"):                     # optional 2-D landmarks somebody put into the file
                if key in ds:
                    lm = _t(ds[key])
                    self.lm68s = take(lm.reshape(lm.shape[0], -1, 2)[:, :68])
                    break
        conds = normed.reshape(-1, 1, self.keypoint_num * 3)
        self.conds = conds[:n] if self.prefix == "train" else conds[-n:]

    # -- access ---------------------------------------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.samples)

    @property
    def num_rays(self):
        return self.hparams["n_rays"] if self.training else -1

    def cond_window(self, idx):
        return camera.get_audio_features(self.conds, 2, idx, self.hparams["smo_win_size"])

    def clip_batch(self, indices=None):
        """The driving signals of a clip in the form ``ClipRenderer.prepare`` takes (poses, conditioning windows, landmarks, eye values)."""
        idx = list(range(len(self))) if indices is None else list(indices)
        out = {"ngp_poses": self.poses[idx].float().numpy(), "cond_wins": torch.stack([self.cond_window(i) for i in idx]).float().numpy()}
        if self.lm68s is not None:
            out["lm68"] = self.lm68s[idx].reshape(len(idx), -1).float().numpy()
        if self.eye_area_percents is not None:
            out["eye_area_percent"] = self.eye_area_percents[idx].reshape(len(idx), 1, 1).float().numpy()
        return out

    def _images(self, idx):
        s = self.samples[idx]
        if self.hparams.get("load_imgs_to_memory", True):
            if "torso_img" not in s:
                s["torso_img"], s["gt_img"] = _load_image_u8(s["torso_img_fname"]), _load_image_u8(s["gt_img_fname"])
            return s["torso_img"], s["gt_img"]
        return _load_image_u8(s["torso_img_fname"]), _load_image_u8(s["gt_img_fname"])

    def __getitem__(self, idx):
        hp, raw, dev = self.hparams, self.samples[idx], self.device
        sample = {"H": self.H, "W": self.W, "focal": self.focal, "cx": self.cx, "cy": self.cy, "near": self.near, "far": self.far, "idx": raw["idx"],
                  "face_rect": raw["face_rect"], "lip_rect": self.lips_rect[idx], "bg_img": self.bg_img, "c2w": raw["c2w"]}
        if self.eg3d_cameras is not None:
            sample["camera"] = self.eg3d_cameras[idx].unsqueeze(0)
        last = len(self) - 1
        sample["cond_wins"] = self.cond_window(idx)
        sample["cond_wins_prev"] = self.cond_window(max(idx - 1, 0))
        sample["cond_wins_next"] = self.cond_window(min(idx + 1, last))
        ngp_pose = self.poses[idx].unsqueeze(0)
        sample["pose"], sample["pose_matrix"] = camera.convert_poses(ngp_pose), ngp_pose
        pose_dev = ngp_pose.float().to(dev)
        if hp.get("with_sr"):
            rays = camera.get_rays(pose_dev, self.intrinsics, self.H, self.W, N=-1, rect=None)
        elif self.training:
            rect = sample["lip_rect"] if self.finetune_lip_flag else None
            rays = camera.get_rays(pose_dev, self.intrinsics, self.H, self.W, N=-1 if self.finetune_lip_flag else self.num_rays, rect=rect)
        else:
            rays = camera.get_rays(pose_dev, self.intrinsics, self.H, self.W, N=-1)
        sample["rays_o"], sample["rays_d"] = rays["rays_o"], rays["rays_d"]
        if self.eye_area_percents is not None:
            sample["eye_area_percent"] = self.eye_area_percents[idx]
        if self.lm68s is not None:
            sample["lm68"] = self.lm68s[idx].reshape(-1)
        # face mask: the polygon variant needs the 2-D landmarks (reference default); the rectangle variant (RAD-NeRF) only the stored face_rect
        if hp.get("polygon_face_mask", True) and self.lm2ds is not None:
            from tasks.radnerfs.dataset_utils import dilate_boundary_mask, get_boundary_mask       # the reference's helpers (need cv2)
            f_mask = dilate_boundary_mask(get_boundary_mask(self.lm2ds[idx], index_mode="lm68", h=self.H, w=self.W).unsqueeze(0).to(dev), ksize=3)
            face_mask = f_mask.reshape(-1).bool()[rays["inds"]]
        else:
            xmin, xmax, ymin, ymax = (float(v) / (2 if hp.get("with_sr") else 1) for v in raw["face_rect"])
            face_mask = (rays["j"] >= xmin) & (rays["j"] < xmax) & (rays["i"] >= ymin) & (rays["i"] < ymax)
        sample["face_mask"] = face_mask
        sample["cond_mask"] = face_mask.reshape(-1)
        inds3 = torch.stack(3 * [rays["inds"]], -1)
        sample["bg_img"] = torch.gather(self.bg_img.view(1, -1, 3).to(dev), 1, inds3)
        have_images = all(os.path.exists(raw.get(k, "")) for k in ("torso_img_fname", "gt_img_fname"))
        if have_images:                                                         # training targets (dataset_utils.py:304-420)
            torso_u8, gt_u8 = self._images(idx)
            torso, gt = torso_u8.to(dev).float() / 255.0, gt_u8.to(dev).float() / 255.0
            sample["gt_img_512"] = gt_u8.to(dev).unsqueeze(0).permute(0, 3, 1, 2) / 255.0
            sample["torso_img"] = torso                                        # dataset_utils.py:352-355 (the full-size RGBA frame, before the SR halving)
            full = torso.shape[0]
            bt512 = torso[..., :3] * torso[..., 3:] + self.bg_img_512 * (1 - torso[..., 3:])
            if hp.get("with_sr"):
                torso = F.interpolate(torso.view(1, full, full, -1).permute(0, 3, 1, 2), size=(self.H, self.W), mode="bilinear", antialias=True).permute(0, 2, 3, 1)
                gt = F.interpolate(gt.view(1, full, full, 3).permute(0, 3, 1, 2), size=(self.H, self.W), mode="bilinear", antialias=True).permute(0, 2, 3, 1)
            bt = (torso[..., :3] * torso[..., 3:] + self.bg_img * (1 - torso[..., 3:])).reshape(1, -1, 3)
            C = gt.shape[-1]
            sample["bg_torso_img"] = torch.gather(bt, 1, inds3)
            sample["bg_torso_img_512"] = bt512.reshape(1, -1, 3)
            sample["gt_img"] = torch.gather(gt.reshape(1, -1, C), 1, torch.stack(C * [rays["inds"]], -1))
        # dataset_utils.py:428-432: every task step reads sample['bg_coords'] (tasks/radnerfs/radnerf.py:115, radnerf_torso.py:83, ...)
        if self.training:
            sample["bg_coords"] = torch.gather(self.bg_coords.to(dev), 1, torch.stack(2 * [rays["inds"]], -1))
        else:
            sample["bg_coords"] = self.bg_coords
        return sample

    def collater(self, samples):
        assert len(samples) == 1                                               # the reference trains with batch size 1 (one frame of rays)
        return samples[0]
