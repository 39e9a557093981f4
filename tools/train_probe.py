import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lss2_multimodal_nu_amd as L
from lss2_multimodal_nu_amd import dp
from oracle import lss_oracle as lo
GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}
B = 4
dev = torch.device("cuda")
torch.manual_seed(0)
feats = torch.randn(B * 6, 512, 8, 22, device=dev)
calib = lo.synthetic_rig(B, train_aug=True, seed=0)
tgt = torch.randint(0, 4, (B, 200, 200), device=dev)
def run(tag, use_bucket, hooks, fused):
    m = L.compile_model_lss(B, GRID, AUG, 4).to(dev).train()
    if use_bucket:
        bucket = dp.GradBucket(m.parameters())
        if not hooks:
            for h in bucket._hooks: h.remove()
        params = bucket.params
    else:
        params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-8)
    sl = L.SimpleLoss().cuda()
    def one():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m.forward_loss(feats, *calib, tgt) if fused else sl(m(feats, *calib).float(), tgt)
        if use_bucket:
            bucket.zero(); loss.backward(); bucket.all_reduce_mean(); bucket.clip_grad_norm_(5.0)
            if hooks:
                un = bucket.hide_unused(); opt.step(); bucket.attach()
            else:
                opt.step()
        else:
            opt.zero_grad(set_to_none=True); loss.backward(); torch.nn.utils.clip_grad_norm_(params, 5.0); opt.step()
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): one()
    torch.cuda.synchronize(); print("%-40s %.2f ms/step" % (tag, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
run("plain params, two-step loss", False, False, False)
run("plain params, fused loss", False, False, True)
run("bucket no hooks, fused", True, False, True)
run("bucket + hooks, fused", True, True, True)
