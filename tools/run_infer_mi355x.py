#!/usr/bin/env python
"""One command for a user who has the reference checkout and the May files: run the reference's own inference script with its motion2video
NeRF renderer replaced by this package (everything else -- HuBERT, audio2motion, post-net, video writer -- is the reference's).

    python tools/run_infer_mi355x.py --reference /path/to/GeneFacePlusPlus -- \\
        --a2m_ckpt=checkpoints/audio2motion_vae --head_ckpt= --torso_ckpt=checkpoints/motion2video_nerf/may_torso \\
        --drv_aud=data/raw/val_wavs/MacronSpeech.wav --out_name=may_demo.mp4

What happens: the reference root goes on sys.path and becomes the working directory (its scripts use relative paths); ``compat.install()``
registers ``modules.radnerfs.*`` (RADNeRF, RADNeRFTorso, RADNeRFTorsowithSR, RADNeRFwithSR, get_rays, ...) from genefaceplusplus_amd, so
``inference/genefacepp_infer.py:39-43`` imports the MI355X renderer; the checkpoint loads with strict=True because the state-dict layout is the
reference's; ``GeneFace2Infer.forward_secc2video`` then calls ``model.render()`` per frame under autocast -> the fused 16-bit hipGraph path.
``--dataset-reader`` additionally swaps ``tasks.radnerfs.dataset_utils.RADNeRFDataset`` for genefaceplusplus_amd.dataset.RADNeRFDataset (no 3DMM
assets needed for the lm68 / esperanto conditioning; see that module's docstring for what it cannot derive without them)."""
import argparse
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="root of a yerfor/GeneFacePlusPlus checkout (with checkpoints/ and data/binary/)")
    ap.add_argument("--dataset-reader", action="store_true", help="also replace the reference's RADNeRFDataset by this package's reader")
    ap.add_argument("--script", default="inference/genefacepp_infer.py", help="reference script to run (relative to --reference)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="arguments after `--` go to the reference script unchanged")
    args = ap.parse_args()
    ref = os.path.abspath(args.reference)
    if not os.path.exists(os.path.join(ref, args.script)):
        sys.exit(f"{args.script} not found under {ref}")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, ref)
    os.chdir(ref)
    import torch
    if not torch.cuda.is_available():
        sys.exit("no MI355X visible: the renderer has no CPU path")
    from genefaceplusplus_amd import _lib, compat
    _lib.lib()                                                     # fail now, loudly, if the HIP library is not built
    shimmed = compat.install()
    print(f"[gfpp] modules.radnerfs.* -> genefaceplusplus_amd ({len(shimmed)} modules), device {torch.cuda.get_device_name(0)}")
    if args.dataset_reader:
        import types
        from genefaceplusplus_amd import dataset as gds
        from utils.commons.hparams import hparams as ref_hparams       # the reference's global dict, filled by its set_hparams()

        class _Dataset(gds.RADNeRFDataset):                             # the reference constructs RADNeRFDataset(prefix, data_dir=None, training=True)
            def __init__(self, prefix, data_dir=None, training=True):
                super().__init__(prefix, ref_hparams, data_dir=data_dir, training=training)

        mod = types.ModuleType("tasks.radnerfs.dataset_utils")
        mod.RADNeRFDataset = _Dataset
        mod.smooth_camera_path = gds.smooth_camera_path
        sys.modules["tasks.radnerfs.dataset_utils"] = mod
    rest = args.rest[1:] if args.rest and args.rest[0] == "--" else args.rest
    sys.argv = [args.script] + rest
    runpy.run_path(os.path.join(ref, args.script), run_name="__main__")


if __name__ == "__main__":
    main()
