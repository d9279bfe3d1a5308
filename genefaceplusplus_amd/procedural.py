"""A procedural talking-head clip: the training target of tools/make_trained_checkpoint.py (no May video ships with the reference mount).

What a person-specific dataset gives the reference's trainer (tasks/radnerfs/dataset_utils.py:160-420) is, per frame: a camera pose, a
conditioning vector (normalised 3-D landmarks, 68 x 3), an eye-area value, 68 2-D landmarks, and three images -- the full frame
(`gt_img`), the torso composited over the background (`bg_torso_img`) and the static background (`bg_img`).  This module produces the
same set analytically, at any resolution and for any subset of pixels, from three driving signals of the frame (mouth opening, smile,
eye opening) and its pose:

  head   an opaque ellipsoid in NeRF space (semi-axes 0.21 / 0.17 / 0.26 around (0, 0, 0.10), inside the May AABB) with a procedural
         texture in its own coordinates -- skin with a low-frequency pattern, hair cap, two eyes whose lids follow the eye signal, a mouth
         whose height follows `open` and width `smile` --, Lambert-shaded by a head-fixed light;
  torso  a neck + shoulders silhouette in IMAGE space (the reference's torso is a 2-D field over the pixel grid, radnerf_torso.py:51-84)
         with striped cloth, shifted by the frame's yaw / pitch like a body under a turning head;
  bg     a static smooth image.

The conditioning vector of a frame is a fixed random linear map of the three signals into the 204 landmark coordinates plus a little
noise, normalised per coordinate over the clip like the dataset does (dataset_utils.py:255-262) -- the networks have to find the
signals in it, as they have to find the mouth in real landmarks.  Everything is torch, device-agnostic and deterministic in (seed, T).
"""
import math

import numpy as np
import torch

HEAD_AXES = (0.21, 0.17, 0.26)
HEAD_CENTRE = (0.0, 0.0, 0.10)
_CHIN = [5, 6, 7, 8, 9, 10, 11]


def _sig(x):
    return torch.sigmoid(x)


class ProceduralClip:
    """T frames of driving signals + analytic target images.  Attributes mirror what the reference's RADNeRFDataset exposes to the trainer:
    `ngp_poses` [T,4,4] (numpy f32, ngp convention), `conds` [T,1,204] (torch f32, normalised), `eye_area_percents` [T,1], `lm68s` [T,68,2]."""

    def __init__(self, T=256, seed=0, distance=4.0):
        self.T = T
        t = np.arange(T, dtype=np.float64)
        self.open = 0.5 + 0.5 * np.sin(2 * np.pi * t / 17.0 + 0.3) * np.cos(2 * np.pi * t / 71.0)
        self.smile = 0.5 + 0.5 * np.sin(2 * np.pi * t / 41.0 + 1.0)
        self.eye = 1.0 - 0.85 * np.exp(-(((t % 53.0) - 26.0) ** 2) / 8.0)
        self.yaw = 0.10 * np.sin(2 * np.pi * t / 67.0)
        self.pitch = 0.05 * np.sin(2 * np.pi * t / 45.0 + 1.0)
        self.shift = np.stack([0.03 * np.sin(2 * np.pi * t / 90.0), np.zeros(T), 0.02 * np.cos(2 * np.pi * t / 70.0)], 1)
        base = np.array([[-1, 0, 0, 0], [0, 0, -1, distance], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
        poses = []
        for k in range(T):
            c, s = math.cos(self.yaw[k]), math.sin(self.yaw[k])
            rz = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
            c, s = math.cos(self.pitch[k]), math.sin(self.pitch[k])
            rx = np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float64)
            p = rz @ rx @ base
            p[:3, 3] += self.shift[k]
            poses.append(p)
        self.ngp_poses = np.stack(poses).astype(np.float32)
        rng = np.random.default_rng(seed)
        U = rng.standard_normal((3, 204))
        raw = np.stack([self.open, self.smile, self.eye], 1) @ U + 0.02 * rng.standard_normal((T, 204))
        normed = (raw - raw.mean(0, keepdims=True)) / raw.std(0, keepdims=True)
        self.conds = torch.from_numpy(normed.astype(np.float32)).reshape(T, 1, 204)
        self.eye_area_percents = torch.from_numpy((0.1 + 0.3 * self.eye).astype(np.float32)).reshape(T, 1)
        # 68 2-D landmarks in [0, 1]^2 image units: a fixed face layout that follows the torso shift (the SR torso conditions on the chin points)
        layout = rng.uniform(0.35, 0.65, (68, 2))
        layout[_CHIN] = np.stack([np.linspace(0.40, 0.60, 7), 0.70 - 0.04 * np.cos(np.linspace(-1.2, 1.2, 7))], 1)
        sx, sy = self.torso_shift()
        self.lm68s = torch.from_numpy((layout[None] + 0.5 * np.stack([sx, sy], 1)[:, None, :]).astype(np.float32))

    # the torso's image-space shift under the head's rotation (u to the right, v downwards, both in [-1, 1] units)
    def torso_shift(self):
        return 1.2 * self.yaw, 0.8 * self.pitch

    def cond_window(self, idx, smo_win_size):
        from .radnerfs.camera import get_audio_features
        return get_audio_features(self.conds, 2, idx, smo_win_size)

    # -- images -----------------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def background(u, v):
        """Static background colour at image coordinates u (column), v (row) in [-1, 1]."""
        wave = 0.04 * torch.sin(5.0 * u) * torch.sin(4.0 * v)
        return torch.stack([0.80 - 0.10 * v + wave, 0.84 - 0.08 * v - wave, 0.90 - 0.05 * v + 0.5 * wave], -1).clamp(0, 1)

    def torso(self, k, u, v):
        """-> (rgb [n,3], alpha [n]) of frame k's torso layer."""
        sx, sy = (float(a[k]) for a in self.torso_shift())
        uu, vv = u - sx, v - sy
        shoulders = _sig((vv - (0.55 + 0.9 * uu * uu)) * 60.0)
        neck = _sig((0.15 - uu.abs()) * 60.0) * _sig((vv - 0.25) * 60.0)
        alpha = torch.maximum(shoulders, neck)
        stripes = torch.sin(18.0 * uu) * torch.sin(14.0 * vv)
        cloth = torch.stack([0.18 + 0.10 * stripes, 0.28 + 0.06 * stripes, 0.58 - 0.12 * stripes], -1)
        collar = _sig((0.10 - (vv - (0.55 + 0.9 * uu * uu))) * 40.0) * shoulders
        cloth = cloth * (1 - collar[:, None]) + torch.tensor([0.92, 0.92, 0.88], device=u.device) * collar[:, None]
        skin = torch.tensor([0.78, 0.60, 0.50], device=u.device).expand(u.shape[0], 3)
        w = (shoulders / (alpha + 1e-6)).unsqueeze(-1)
        return (cloth * w + skin * (1 - w)).clamp(0, 1), alpha

    def head(self, k, rays_o, rays_d):
        """-> (rgb [n,3], alpha [n], front [n]) of frame k's head layer for rays in NeRF space; `front` marks the face side (the trainer's face mask)."""
        dev = rays_o.device
        ax = torch.tensor(HEAD_AXES, device=dev)
        o = (rays_o - torch.tensor(HEAD_CENTRE, device=dev)) / ax
        d = rays_d / ax
        A = (d * d).sum(-1)
        B = (o * d).sum(-1)
        C = (o * o).sum(-1) - 1.0
        closest = (C + 1.0 - B * B / A).clamp(min=0).sqrt()             # distance of the ray from the centre, unit-sphere space
        alpha = ((1.0 - closest) / 0.012).clamp(0, 1)
        tt = (-B - (B * B - A * C).clamp(min=0).sqrt()) / A
        q = o + tt.unsqueeze(-1) * d                                     # point on the unit sphere (meaningless where alpha == 0)
        qx, qy, qz = q[:, 0], q[:, 1], q[:, 2]
        n = q / ax
        n = n / n.norm(dim=-1, keepdim=True).clamp(min=1e-6)
        light = torch.tensor([0.30, 0.80, 0.52], device=dev)
        shade = 0.62 + 0.38 * (n * (light / light.norm())).sum(-1).clamp(min=0)
        pat = torch.sin(6.0 * math.pi * qx) * torch.sin(5.0 * math.pi * qz)
        rgb = torch.stack([0.80 + 0.05 * pat, 0.62 - 0.04 * pat, 0.52 + 0.03 * pat], -1)
        front = _sig(qy * 12.0)
        opn, smi, eye = float(self.open[k]), float(self.smile[k]), float(self.eye[k])
        # hair cap
        hair = _sig((qz - (0.50 + 0.10 * torch.sin(8.0 * qx)) + 0.5 * (1 - front)) * 30.0)
        hair_rgb = torch.tensor([0.20, 0.13, 0.10], device=dev) + 0.05 * torch.sin(40.0 * qx).unsqueeze(-1)
        rgb = rgb * (1 - hair[:, None]) + hair_rgb * hair[:, None]
        # eyes: white with a dark pupil, lid height follows the eye signal
        for cx in (-0.36, 0.36):
            e2 = ((qx - cx) / 0.15) ** 2 + ((qz - 0.18) / (0.025 + 0.065 * eye)) ** 2
            white = _sig((1.0 - e2) * 6.0) * front
            p2 = ((qx - cx) / 0.06) ** 2 + ((qz - 0.18) / 0.06) ** 2
            pupil = _sig((1.0 - p2) * 6.0) * white
            rgb = rgb * (1 - white[:, None]) + torch.tensor([0.95, 0.95, 0.95], device=dev) * white[:, None]
            rgb = rgb * (1 - pupil[:, None]) + torch.tensor([0.10, 0.20, 0.35], device=dev) * pupil[:, None]
        # mouth: lips ring + dark opening
        mw, mh = 0.22 + 0.08 * smi, 0.035 + 0.11 * opn
        m2 = (qx / mw) ** 2 + ((qz + 0.45 - 0.05 * smi * (qx / mw) ** 2) / mh) ** 2
        lips = _sig((1.0 - m2 / 1.6) * 6.0) * front
        inner = _sig((1.0 - m2) * 8.0) * front
        rgb = rgb * (1 - lips[:, None]) + torch.tensor([0.72, 0.25, 0.28], device=dev) * lips[:, None]
        rgb = rgb * (1 - inner[:, None]) + torch.tensor([0.25, 0.04, 0.06], device=dev) * inner[:, None]
        return (rgb * shade.unsqueeze(-1)).clamp(0, 1), alpha, (front > 0.5) & (alpha > 0)

    def target(self, k, rays_o, rays_d, bg_coords):
        """The training images of frame k at the given rays / pixels: bg_coords [n,2] = (row, col) coordinates in [-1,1] (camera.get_bg_coords).
        -> {'gt' [n,3], 'bg_torso' [n,3], 'bg' [n,3], 'head_alpha' [n], 'torso_alpha' [n], 'face_mask' [n] bool}."""
        v, u = bg_coords[:, 0], bg_coords[:, 1]
        bg = self.background(u, v)
        trgb, talpha = self.torso(k, u, v)
        bg_torso = trgb * talpha[:, None] + bg * (1 - talpha[:, None])
        hrgb, halpha, face = self.head(k, rays_o, rays_d)
        gt = hrgb * halpha[:, None] + bg_torso * (1 - halpha[:, None])
        return {"gt": gt, "bg_torso": bg_torso, "bg": bg, "head_alpha": halpha, "torso_alpha": talpha, "face_mask": face}

    def frame(self, k, HW, intr, device):
        """All pixels of frame k at HW x HW: rays through the library's own ray generator, -> target() + 'rays_o' / 'rays_d'."""
        from .radnerfs import camera
        pose = torch.from_numpy(self.ngp_poses[k:k + 1]).to(device)
        rays = camera.get_rays(pose, intr, HW, HW)
        out = self.target(k, rays["rays_o"][0], rays["rays_d"][0], camera.get_bg_coords(HW, HW, device)[0])
        out.update(rays_o=rays["rays_o"], rays_d=rays["rays_d"])
        return out

    def clip_batch(self, smo_win_size, frames=None):
        """The driving signals of (a subset of) the clip in the form clip.ClipRenderer.prepare takes (RADNeRFDataset.clip_batch's keys)."""
        idx = list(range(self.T)) if frames is None else [int(k) % self.T for k in frames]
        return {"ngp_poses": self.ngp_poses[idx].astype(np.float32),
                "cond_wins": torch.stack([self.cond_window(k, smo_win_size) for k in idx]).numpy(),
                "lm68": self.lm68s[idx].reshape(len(idx), -1).numpy(),
                "eye_area_percent": self.eye_area_percents[idx].reshape(len(idx), 1, 1).numpy()}

    def background_image(self, HW, device="cpu"):
        """[1, HW*HW, 3]: the static background at every pixel (the `bg_img` a torso model composites over)."""
        rows = torch.arange(HW, device=device) / (HW - 1) * 2 - 1
        vv, uu = torch.meshgrid(rows, rows, indexing="ij")
        return self.background(uu.reshape(-1), vv.reshape(-1)).reshape(1, HW * HW, 3)


#: keys of a torso model's state that a head-only model of the same family does not have
_TORSO_ONLY = ("torso_", "density_grid_torso", "head_color_weights_encoder", "lm68_embedder")


def head_only_state(state, keep_sr=True):
    """The head model's part of a torso checkpoint's state_dict (the reverse of radnerf_torso.py:31-37's strict=False load)."""
    return {k: v for k, v in state.items() if not k.startswith(_TORSO_ONLY) and (keep_sr or not k.startswith("sr_net."))}
