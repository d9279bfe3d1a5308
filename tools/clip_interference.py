"""Do a clip's frames depend on WHAT ELSE is in flight?  One clip rendered again and again by the same ClipRenderer (lanes > 1: frames of different lanes overlap on the
device); after every render the frames and the last frame's intermediate buffers of every lane (head workspace, torso outputs, SR input and first activation) are
compared bit for bit with the render before.  Prints how often each buffer changed.

Round 6: with the polyphase up-sampling launch (gfpp_tuning.sr_up_poly = 1) the torso / pre-march kernels of the OTHER lane return different bits in 2-40 % of
the renders -- 16 consecutive pixels of one 32-pixel torso pass off by 1e-3 .. 5e-2 --, with the composed launch never (0 of 1 800 renders).  The bisect of the
launch (docs/LAB_NOTEBOOK.md): it needs the launch's MFMA phase WITH its LDS operand reads (either alone: nothing), on the same CU as the victim (the launch padded to
the whole LDS of a CU: nothing); not its stores, not its LDS DMA, not its scratch, not its data (zero weights and zero activations: same rate).

    python tools/clip_interference.py [renders] [variant] [HW] [precision]        env: GROUP (frames per head launch, 1), LANES (2), GRAPH (1), GFPP_SR_UP_POLY
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_model, frame_case                                    # noqa: E402
from test_clip_gpu import _clip_batch                                          # noqa: E402
from genefaceplusplus_amd.clip import ClipRenderer                             # noqa: E402
from genefaceplusplus_amd.radnerfs.frame_pipeline import FramePipeline        # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution            # noqa: E402


def main():
    renders = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    variant = sys.argv[2] if len(sys.argv) > 2 else "may_torso_sr"
    HW = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    precision = sys.argv[4] if len(sys.argv) > 4 else "fp16"
    F = 10
    dev = torch.device("cuda:0")
    case = frame_case(variant, HW)
    model = build_model(case, dev, "fused")
    model.precision = precision
    kw = dict(case["hp"], use_head_for_torso=True)
    if variant.endswith("_sr"):
        kw["sr_noise_mode"] = "const"               # 'random' draws per launch: not comparable between two renders
    batch = _clip_batch(case["hp"], F)

    # the graphs' static memory: what the last captured call handed to / got from these two stays valid and is rewritten by every replay
    sr_in, torso_out = {}, {}
    sr_forward, head_torso = Superresolution.forward, FramePipeline.render_head_torso

    def spy_sr(self, rgb, *a, **k):
        sr_in[self.lane] = rgb
        return sr_forward(self, rgb, *a, **k)

    def spy_torso(self, *a, **k):
        o = head_torso(self, *a, **k)
        torso_out[self.lane] = o
        return o
    Superresolution.forward, FramePipeline.render_head_torso = spy_sr, spy_torso

    cr = ClipRenderer(model, HW, HW, case["intr"], bg_img=torch.from_numpy(case["bg_color"]), T_thresh=case["T_thresh"], render_kwargs=kw,
                      group=int(os.environ.get("GROUP", "1")), lanes=int(os.environ.get("LANES", "2")), use_graph=os.environ.get("GRAPH", "1") != "0")
    clip = cr.prepare(batch, dev)

    def snapshot():
        torch.cuda.synchronize()
        out = {}
        if hasattr(model, "sr_net") and model.sr_net._packed is not None:
            for lane, (_ws, bufs) in model.sr_net._packed["ws"].items():
                out[(lane, "sr.img256")] = bufs["img256"].clone()
        for lane, v in sr_in.items():
            out[(lane, "sr input")] = v.clone()
        for lane, o in torso_out.items():
            for k, v in o.items():
                if torch.is_tensor(v):
                    out[(lane, "torso." + k)] = v.clone()
        for wkey, ent in model.pipeline()._ws.items():
            for k, v in ent[-1].items():
                if torch.is_tensor(v):
                    out[(wkey[-1], "frame_ws." + k)] = v.clone()
        return out

    frames = cr.render_to_device(clip).cpu().numpy()
    before = snapshot()
    tally, spans = {}, []
    for rep in range(renders):
        again = cr.render_to_device(clip).cpu().numpy()
        now = snapshot()
        if (again != frames).any():
            tally["frames (uint8)"] = tally.get("frames (uint8)", 0) + 1
        for key in sorted(before):
            if key in now and not torch.equal(now[key], before[key]):
                tally[key[1]] = tally.get(key[1], 0) + 1
                if key[1] == "torso.torso_alpha" and len(spans) < 6:
                    d = (now[key].float() - before[key].float()).abs().reshape(-1)
                    w = torch.nonzero(d > 0).reshape(-1).tolist()
                    spans.append((rep, key[0], w[0], len(w), sorted(set((n // 256, (n % 256) // 64) for n in w)), float(d.max())))
        frames, before = again, now
    print(f"{variant} {HW} {precision} group {cr.group} lanes {cr.lanes}: buffers that changed between consecutive renders of the same clip, of {renders}: {tally or 'none'}")
    for rep, lane, first, n, wgs, worst in spans:
        print(f"   render {rep} lane {lane}: torso_alpha of {n} pixels from {first} on ((workgroup, wavefront) of k_torso_lp: {wgs}), worst {worst:.3g}")


main()
