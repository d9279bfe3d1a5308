"""Host logic of the clip renderer that needs no GPU: the packed per-frame input row (genefaceplusplus_amd/clip.py::ClipRenderer.prepare / _views)."""
import numpy as np
import torch

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.clip import ClipRenderer
from genefaceplusplus_amd.configs import may_hparams
from genefaceplusplus_amd.radnerfs import camera


def test_packed_rows_round_trip_and_alignment():
    hp = may_hparams("may_torso_sr")
    F = 5
    fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
    batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
             "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
    clip = ClipRenderer.prepare(batch, torch.device("cpu"))
    assert clip["frames"] == F and clip["packed"].shape[0] == F and clip["packed"].is_contiguous()
    assert clip["packed"].shape[1] % 4 == 0                                  # every row (and so every row start) is 16-byte aligned
    pose6 = camera.convert_poses(torch.from_numpy(batch["ngp_poses"]))
    for i in range(F):
        v = ClipRenderer._views(clip["packed"][i], clip["layout"])
        assert set(v) == {"pose", "pose6", "cond", "lm68", "eye"}
        for name, t in v.items():
            assert (t.data_ptr() - clip["packed"][i].data_ptr()) % 16 == 0, name   # fields start on 16-byte boundaries
        np.testing.assert_array_equal(v["pose"].numpy(), batch["ngp_poses"][i])
        np.testing.assert_array_equal(v["cond"].numpy(), batch["cond_wins"][i])
        np.testing.assert_array_equal(v["lm68"].numpy(), batch["lm68"][i])
        np.testing.assert_array_equal(v["eye"].numpy(), batch["eye_area_percent"][i].reshape(1, 1))
        np.testing.assert_array_equal(v["pose6"].numpy(), pose6[i:i + 1].numpy())
    # optional signals default to zeros
    clip2 = ClipRenderer.prepare({"ngp_poses": batch["ngp_poses"], "cond_wins": batch["cond_wins"]}, torch.device("cpu"))
    v = ClipRenderer._views(clip2["packed"][0], clip2["layout"])
    assert float(v["lm68"].abs().sum()) == 0.0 and float(v["eye"].abs().sum()) == 0.0


def test_rows_grow_by_the_per_frame_constants_of_a_job():
    """ClipRenderer._with_cond_features: the model's per-job constants become one more 16-byte-aligned field of every row; a model that cannot
    provide them (or the switch off) leaves the clip untouched.  (Host bookkeeping only: the model here is a stand-in that records what it is asked.)"""
    hp = may_hparams("may_torso")
    F = 4
    fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
    batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
             "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
    clip = ClipRenderer.prepare(batch, torch.device("cpu"))
    asked = {}

    class Model:
        def frame_consts_rows(self, rows, cond_at, eye_at, count):
            asked.update(rows=rows, cols={"cond": cond_at, "eye": eye_at}, count=count)
            return torch.arange(count * 258, dtype=torch.float32).reshape(count, 258)       # a width that needs 2 floats of padding

    cr = object.__new__(ClipRenderer)
    cr.model, cr.precompute_cond = Model(), True
    ext = cr._with_cond_features(clip)
    assert asked["count"] == F and asked["rows"] is clip["packed"]
    # the columns the model is told are where _views finds the fields
    v0 = ClipRenderer._views(clip["packed"][0], clip["layout"])
    for name in ("cond", "eye"):
        assert asked["cols"][name] * 4 == v0[name].data_ptr() - clip["packed"][0].data_ptr(), name
    assert ext["frames"] == F and ext["packed"].shape == (F, clip["packed"].shape[1] + 260) and ext["packed"].is_contiguous()
    assert ext["layout"][:-1] == clip["layout"] and ext["layout"][-1] == ("cond_feat", (258,)) and ext["strides"][-1] == 260
    for i in range(F):
        v = ClipRenderer._views(ext["packed"][i], ext["layout"])
        np.testing.assert_array_equal(v["cond"].numpy(), batch["cond_wins"][i])
        np.testing.assert_array_equal(v["cond_feat"].numpy(), np.arange(i * 258, (i + 1) * 258, dtype=np.float32))
        assert (v["cond_feat"].data_ptr() - ext["packed"][i].data_ptr()) % 16 == 0
    # the same clip under the same weights again (a caller that starts one job per chunk): the extended rows come from the one-entry cache
    asked.clear()
    assert cr._with_cond_features(clip) is ext and not asked
    # no constants from the model, or the switch off: the clip itself
    cr._cond_cache = None
    Model.frame_consts_rows = lambda self, rows, cond_at, eye_at, count: None
    assert cr._with_cond_features(clip) is clip
    cr.precompute_cond = False
    assert cr._with_cond_features(clip) is clip
