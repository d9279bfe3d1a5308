"""The outermost drop-in boundary: the reference's OWN caller (inference/genefacepp_infer.py: load_secc2video :163-191 -- set_hparams(config.yaml),
load_ckpt(strict=True), torch.compile -- and forward_secc2video :433-519), imported unmodified from /root/reference, run on this package's
classes (compat.install()) with the kernels served by the oracle, on a synthetic ON-DISK checkpoint in the reference's layout.  Compared with
tests/golden/caller_golden.npz = the same caller over the reference's own classes (tests/golden/make_golden_caller.py)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GFPP_REFERENCE", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "inference")), reason="needs the reference checkout (build container only)")


def _golden():
    return np.load(os.path.join(HERE, "golden", "caller_golden.npz"))


def _run(mode, variant, frames, work, thresh=0.01):
    out = os.path.join(work, f"{variant}_{mode}.npz")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_caller.py"), "--mode", mode, "--variant", variant, "--frames", str(frames), "--thresh", str(thresh),
                        "--work", str(work), "--out", out], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = np.load(out)
    return d["frames"], str(d["info"][0])


@needs_reference
def test_reference_caller_on_the_dropin_delivers_the_reference_bytes_torso_512(tmp_path):
    """RADNeRFTorso, 512x512 rays, 2 frames (index 0 and 1: two individual-code rows): the bytes the reference's caller hands to its video writer
    are the same whether modules.radnerfs.* is the reference or this package."""
    g = _golden()
    frames, info = _run("product", "may_torso", 2, tmp_path)
    assert "genefaceplusplus_amd.radnerfs.torso.RADNeRFTorso" in info and "OptimizedModule" in info       # strict load + torch.compile wrapper went through
    assert hashlib.sha256(np.ascontiguousarray(frames).tobytes()).hexdigest() == str(g["may_torso.sha256"][0])
    np.testing.assert_array_equal(frames[:, ::4, ::4], g["may_torso.sub"])
    assert frames.std() > 20                                                                                # a picture, not a constant


@needs_reference
def test_reference_caller_on_the_dropin_torso_sr(tmp_path):
    """RADNeRFTorsowithSR (the released May checkpoint's class): 256x256 rays + super-resolution, `sr_rgb_map`.  The SR stage runs as fp32 torch
    ops in two formulations on the CPU (the product's HIP kernels are GPU-tested against oracle/sr_oracle.py): <= 1 LSB on <= 0.1 % of the values."""
    g = _golden()
    frames, info = _run("product", "may_torso_sr", 1, tmp_path)
    assert "RADNeRFTorsowithSR" in info and "'with_sr': True" in info
    diff = np.abs(frames[:, ::4, ::4].astype(np.int32) - g["may_torso_sr.sub"][:1].astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3, (int(diff.max()), float((diff != 0).mean()))


@needs_reference
def test_reference_caller_fast_threshold_eight_frames(tmp_path):
    """The CLI's `--fast` (raymarching_end_threshold 0.05, genefacepp_infer.py:566,591-592): the fixture holds 8 frames of the reference's caller over the
    reference's classes (the generator asserted byte equality of the drop-in on all 8; tests/test_ref_caller_gpu.py renders all 8 on the real kernels);
    here all 8 through the drop-in again -- every frame has its own pose, conditioning window and individual-code row -- byte for byte."""
    g = _golden()
    assert g["may_torso.fast.sub"].shape[0] == 8 and list(g["may_torso.fast.product_differs"]) == [0, 0]
    assert not np.array_equal(g["may_torso.fast.sub"][:2], g["may_torso.sub"])                              # the threshold changed the picture
    frames, info = _run("product", "may_torso", 8, tmp_path, thresh=0.05)
    np.testing.assert_array_equal(frames[:, ::4, ::4], g["may_torso.fast.sub"])
    assert hashlib.sha256(np.ascontiguousarray(frames).tobytes()).hexdigest() == str(g["may_torso.fast.sha256"][0])


def test_checkpoint_writer_layout(tmp_path):
    """synthetic.write_checkpoint leaves what Trainer._atomic_save / dump_checkpoint leave (utils/commons/trainer.py:542-567): a legacy-pickle file
    model_ckpt_steps_N.ckpt with state_dict['model'], and a flat config.yaml; the newest step wins (ckpt_utils.py:19-26)."""
    import yaml
    import zipfile
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.configs import may_hparams
    d = str(tmp_path / "ck")
    syn.write_checkpoint(d, "may_torso_sr", steps=1000, seed=1)
    path, cfg = syn.write_checkpoint(d, "may_torso_sr", steps=250000)
    assert os.path.basename(path) == "model_ckpt_steps_250000.ckpt" and not zipfile.is_zipfile(path)       # _use_new_zipfile_serialization=False
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "global_step", "checkpoint_callback_best", "optimizer_states", "state_dict"} and set(ck["state_dict"]) == {"model"}
    assert ck["global_step"] == 250000
    hp = yaml.safe_load(open(cfg))
    for k, v in may_hparams("may_torso_sr").items():
        assert hp[k] == v, k
    sd, used = syn.read_checkpoint(d)
    assert used == path                                                                                     # newest of the two
    want = dict(syn.synthetic_state_dict(may_hparams("may_torso_sr"), "may_torso_sr"))
    want.update(syn.synthetic_sr_state())
    assert set(sd) == set(want)
    for k in ("sigma_net.net.0.weight", "sr_net.block1.conv0.weight", "density_bitfield", "torso_embedder.embeddings"):
        np.testing.assert_array_equal(sd[k].numpy(), want[k])
    # the flat layout ("model.<key>") is read too (ckpt_utils.py:39-41)
    flat = {"state_dict": {"model." + k: v for k, v in sd.items()}}
    torch.save(flat, str(tmp_path / "flat.ckpt"))
    sd2, _ = syn.read_checkpoint(str(tmp_path / "flat.ckpt"))
    assert set(sd2) == set(sd)
    # and the drop-in classes take it with strict=True
    from genefaceplusplus_amd import radnerfs
    m = radnerfs.RADNeRFTorsowithSR(hp)
    m.load_state_dict(sd, strict=True)
