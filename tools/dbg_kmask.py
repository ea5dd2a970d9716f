import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import localdiffusion_hallucination_amd as ldh
ldh.configure_runtime()
from localdiffusion_hallucination_amd import dist as ldist, rng
from test_hip_sampler import make
H, B, T = 32, 2, 50
cond = torch.from_numpy(rng.uniform((B, 1, H, H), 6, 1, 0.0, 2.0)).cuda()
mask = torch.zeros(B, 1, H, H); mask[:, :, :, :H // 4] = 1.0
masks2 = torch.cat([mask, 1.0 - (mask >= 1.0).float()], 1).cuda()
conf = dict(data="mri", branch_out=True, start_intermediate=True, start_timestep=2, mask_x=True)
gd = make(dict(mode="mri"), conf, H, T)
want = gd.sample(cond, None, batch_size=B, mask=masks2, min_max_val=(0.0, 2.0)).cpu()
gd.reset_call_state()
pay, where = gd.kmask_branch_units(cond, masks2, (0.0, 2.0), 0, 2 * B)
print("where", where, "payload x range", float(pay[:, 0].min()), float(pay[:, 0].max()), "x0 range", float(pay[:, 1].min()), float(pay[:, 1].max()))
got = gd.kmask_fuse_joint(cond, masks2, (0.0, 2.0), pay, where, 0, B).cpu()
print("halves vs sample():", float((got - want).abs().max()), "want range", float(want.min()), float(want.max()), "got range", float(got.min()), float(got.max()))
# stepwise: the unsharded loop's state at the fusion step, re-derived with the library calls
gd.reset_call_state()
for t_stop in (48, 40, 3):
    gd2 = make(dict(mode="mri"), dict(conf, start_timestep=t_stop), H, T)
    p2, w2 = gd2.kmask_branch_units(cond, masks2, (0.0, 2.0), 0, 2 * B)
    print("fusion at", t_stop, "where", w2, "x", float(p2[:, 0].min()), float(p2[:, 0].max()), "x0", float(p2[:, 1].min()), float(p2[:, 1].max()))
