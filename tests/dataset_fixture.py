"""A synthetic ``trainval_dataset.npy`` with the schema data_gen/runs/binarizer_nerf.py:197-320 writes (test fixture writer)."""
import numpy as np


def write_synthetic_dataset(path, T=22, H=64, W=64, seed=0, with_esperanto=True, with_lm68=True):
    rng = np.random.default_rng(seed)
    d = {"H": H, "W": W, "focal": 1015.0, "cx": 112.0, "cy": 112.0,                      # face_model.focal / center (bfm.py:35-36)
         "bg_img": rng.integers(0, 256, (H, W, 3)).astype(np.uint8),
         "id": rng.standard_normal((T, 80)).astype(np.float32), "exp": rng.standard_normal((T, 64)).astype(np.float32),
         "euler": (rng.standard_normal((T, 3)) * 0.1).astype(np.float32), "trans": (rng.standard_normal((T, 3)) * 0.1).astype(np.float32),
         "eye_area_percent": rng.uniform(0.1, 0.5, (T, 1)).astype(np.float32),
         "hubert": rng.standard_normal((2 * T, 1024)).astype(np.float32), "mel": rng.standard_normal((4 * T, 80)).astype(np.float32),
         "f0": rng.uniform(80, 300, (4 * T,)).astype(np.float32)}
    lm3d = rng.standard_normal((T, 204)).astype(np.float32) * 0.05
    d["idexp_lm3d"], d["idexp_lm3d_mean"], d["idexp_lm3d_std"] = lm3d, lm3d.mean(axis=0), lm3d.std(axis=0)
    if with_esperanto:
        d["esperanto"] = rng.standard_normal((T, 16, 44)).astype(np.float32)
    if with_lm68:
        d["lm68"] = rng.uniform(0.3, 0.7, (T, 68, 2)).astype(np.float32)
    n_train = T // 11 * 10
    samples = []
    for i in range(T):
        th = 0.05 * np.sin(i / 3.0)
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
        c2w[:3, 3] = [0.01 * i, -0.02, 1.0 + 0.002 * i]
        samples.append({"idx": i, "head_img_fname": f"/nonexistent/head_imgs/{i:08d}.png", "torso_img_fname": f"/nonexistent/inpaint_torso_imgs/{i:08d}.png",
                        "gt_img_fname": f"/nonexistent/com_imgs/{i:08d}.jpg", "face_rect": np.array([H // 4, 3 * H // 4, W // 4, 3 * W // 4]),
                        "lip_rect": [H // 2, H // 2 + 8, W // 2 - 4, W // 2 + 4], "c2w": c2w})
    d["train_samples"], d["val_samples"] = samples[:n_train], samples[n_train:]
    np.save(path, d, allow_pickle=True)
    return d
