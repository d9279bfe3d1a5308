"""The trained procedural field (tests/golden/trained/*.npz; made by tools/make_trained_checkpoint.py on an MI355X with the package's own training path, fit curves in
fit_log.json beside the files) without a GPU: the files carry the reference's state_dict layout, their 16-bit-stored arrays are exact, and the CPU ORACLE --
which shares no code with the training path -- renders the held-out frame of the procedural clip from these weights at > 38 dB against the analytic target.
That last check ties three independent pieces together: the HIP training kernels that produced the weights, the oracle's restatement of the reference's
render path, and genefaceplusplus_amd.procedural's images."""
import json
import os

import numpy as np
import pytest
import torch

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams
from helpers import trained_case, oracle_render, TRAINED_DIR


@pytest.mark.parametrize("variant", ["may_torso", "may_torso_sr"])
def test_layout_is_the_reference_state_dict(variant):
    sd = syn.load_compact_state(os.path.join(TRAINED_DIR, variant + ".npz"))
    hp = may_hparams(variant)
    want = dict(syn.synthetic_state_dict(hp, variant))
    if hp.get("with_sr"):
        want.update(syn.synthetic_sr_state())
    assert set(sd) == set(want)
    for k, v in want.items():
        assert sd[k].shape == np.asarray(v).shape and sd[k].dtype == np.asarray(v).dtype, k
    # exact float16 storage of the big arrays
    for k, v in sd.items():
        if syn.compact_f16_key(k) and v.dtype == np.float32:
            np.testing.assert_array_equal(v, v.astype(np.float16).astype(np.float32), err_msg=k)
    # a trained occupancy: a thin shell of the 128^3 cells, consistent with the grid it was packed from at SOME threshold <= density_thresh
    occ = np.unpackbits(sd["density_bitfield"], bitorder="little").astype(bool)
    grid = sd["density_grid"].reshape(-1)
    assert 0.01 < occ.mean() < 0.06
    assert grid[occ].min() > grid[~occ].max() - 1e-3 and grid[occ].min() <= hp["density_thresh"]
    # the torso occupancy after training is everything (sigmoid alpha > 0 everywhere, threshold min(0.01, mean_density_torso = 0 at inference) = 0; radnerf_torso.py:22,201-244)
    assert float((sd["density_grid_torso"] > 0).mean()) == 1.0


def test_fit_log_meets_the_30_db_bar():
    log = json.load(open(os.path.join(TRAINED_DIR, "fit_log.json")))
    seen = set()
    for run in log["runs"]:
        for variant, per_prec in run["final_psnr_vs_target"].items():
            seen.add(variant)
            for prec in ("fp32", "fp16", "bf16"):
                assert per_prec[prec]["val"]["min"] >= 30.0 and per_prec[prec]["val"]["mean"] >= 35.0, (variant, prec, per_prec[prec])
        head = run["stages"]["head"]["curve"]
        assert head[0]["train_psnr"] < 25.0 < head[-1]["train_psnr"]                          # it was a fit, not a lucky start
        assert head[-1]["occupied_cells"] < 0.2 * head[0]["occupied_cells"]                   # update_extra_state carved the occupancy down to the head
    assert seen == {"may_head", "may_torso", "may_head_sr", "may_torso_sr"}


@pytest.mark.parametrize("variant,HW", [("may_torso", 96), ("may_torso_sr", 96), ("may_head", 64), ("may_head_sr", 64)])
def test_oracle_renders_the_target(oracle_mod, variant, HW):
    case = trained_case(variant, HW)
    trace = []
    ref = oracle_render(oracle_mod, case, trace=trace)
    r = oracle_mod.get_rays(case["pose"], case["intr"], HW, HW)
    tgt = case["clip"].target(case["frame_idx"], torch.from_numpy(r["rays_o"][0]), torch.from_numpy(r["rays_d"][0]), torch.from_numpy(oracle_mod.get_bg_coords(HW, HW))[0])
    mse = float(((ref["rgb_map"].reshape(-1, 3) - tgt["gt"].numpy()) ** 2).mean())
    psnr = -10 * np.log10(mse)
    print(variant, HW, "oracle vs procedural target", psnr, "dB; trips", trace)
    assert psnr >= 38.0
    # the schedule of a trained field: most rays leave after the first trip (they miss the head's occupied cells), the rest march on in a few long trips
    assert trace[0][0] == HW * HW and trace[1][0] < 0.4 * HW * HW
