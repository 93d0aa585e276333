"""Developer check (GPU box): what the corrupted stage-3 outputs of the two-stream mode look like (tools/dev/two_stream_check.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rc_mvsnet_amd import _lib, synthetic
from rc_mvsnet_amd.casmvsnet import CascadeMVSNet_eval
from rc_mvsnet_amd.scene_pipeline import ScenePipeline
_lib.load()
dev = "cuda:0"
scenes = []
for seed in range(4):
    imgs, pm, dv = synthetic.cascade_inputs(1, 3, 512, 640, seed)
    scenes.append((imgs.to(dev), {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))


def make():
    m = CascadeMVSNet_eval(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    m.load_state_dict(synthetic.cascade_state_dict(0)); return m.to(dev).eval()


with torch.no_grad():
    ref = make()
    want = [ref(*s)["depth"].clone() for s in scenes]
    torch.cuda.synchronize()
    shown = 0
    for rnd in range(10):
        pipe = ScenePipeline(make, 2, dev)
        got = [pipe(*scenes[i % 4])[0]["depth"] for i in range(16)]
        pipe.synchronize()
        for i, o in enumerate(got):
            bad = (o != want[i % 4])[0]
            if bool(bad.any()) and shown < 3:
                shown += 1
                cells = bad.view(64, 8, 80, 8).any(dim=3).any(dim=1)
                print('\n'.join(''.join('#' if c else '.' for c in row) for row in cells.tolist()))
                rows = bad.any(dim=1).nonzero().flatten(); cols = bad.any(dim=0).nonzero().flatten()
                eq_other = [j for j in range(4) if j != i % 4 and bool(torch.equal(o, want[j]))]
                frac_other = {j: round(float((o == want[j])[0][bad].float().mean()), 3) for j in range(4) if j != i % 4}
                print(f"round {rnd} scene #{i} (stream {i % 2}, input {i % 4}): {float(bad.float().mean()):.4f} of the pixels differ; rows {int(rows.min())}..{int(rows.max())} "
                      f"({len(rows)} rows), cols {int(cols.min())}..{int(cols.max())} ({len(cols)} cols); equals another scene's output: {eq_other}; "
                      f"of the differing pixels, fraction equal to scene j's reference: {frac_other}; nan {int(torch.isnan(o).sum())}")
