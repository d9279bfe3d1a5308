#!/bin/bash
# round 5, GPU call 6: where a 20-frame job's start-up goes (host profile of ClipRenderer.start) + the driver-shaped line three times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/clip_start_profile.py may_torso 512 bf16 20 > gpurun_out/r5c6_start.log 2>&1
timeout 300 python - > gpurun_out/r5c6_cprofile.log 2>&1 <<'PY'
import cProfile, pstats, io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import frame_case, build_model
from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.clip import ClipRenderer
dev = torch.device("cuda:0"); case = frame_case("may_torso", 512); model = build_model(case, dev, "fused"); model.precision = "bf16"; hp = case["hp"]
F = 25
fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
         "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
cr = ClipRenderer(model, 512, 512, case["intr"], bg_img=torch.from_numpy(case["bg_color"]), T_thresh=0.01, render_kwargs=dict(hp, use_head_for_torso=True))
clip = cr.prepare(batch, dev); out = torch.empty(20, 512, 512, 3, dtype=torch.uint8, device=dev)
cr.render_to_device(clip, range(5), out=out[:5]); torch.cuda.synchronize()
for rep in range(3):
    cr._cond_cache = None; cr.start(clip, range(5, 25), out); cr.issue(); cr.join(); torch.cuda.synchronize()
pr = cProfile.Profile()
for rep in range(20):
    cr._cond_cache = None
    torch.cuda.synchronize()
    pr.enable(); cr.start(clip, range(5, 25), out); t=cr.issue(); pr.disable()
    cr.join(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
PY
for rep in 1 2 3; do
  ( timeout 300 python bench.py --steps 20 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_issue_ms_per_frame'], d['config']['timed_frames_check']['ok'], d['roofline']['frac'])" ) >> gpurun_out/r5c6_bench20.log 2>&1
done
echo done
