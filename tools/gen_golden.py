#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs + the reference's outputs) and are committed.  The
reference has no tests / golden vectors of its own, so
these are the pins for oracle/ and, through it, for the HIP kernels.

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from _ref_loader import load_reference, make_lss_shell  # noqa: E402
from oracle.lss_oracle import synthetic_rig  # noqa: E402  (input generator only)

OUT = os.path.join(ROOT, "tests", "golden")

GRID_DEFAULT = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5],
                    zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
GRID_HIRES = dict(xbound=[-50.0, 50.0, 0.25], ybound=[-50.0, 50.0, 0.25],
                  zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 61.0, 1.0])
GRID_COARSE = dict(xbound=[-50.0, 50.0, 2.0], ybound=[-50.0, 50.0, 2.0],
                   zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
GRID_SMALL = dict(xbound=[-10.0, 10.0, 2.0], ybound=[-10.0, 10.0, 2.0],
                  zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 9.0, 1.0])
GRID_SMALL_Z2 = dict(xbound=[-10.0, 10.0, 2.0], ybound=[-10.0, 10.0, 2.0],
                     zbound=[-4.0, 4.0, 4.0], dbound=[4.0, 9.0, 1.0])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-32s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def ref_indices(shell, geom):
    """The reference's own quantise + filter lines (src/model_BEV_TXT.py:92,
    99-103) applied through its voxel_pooling would not return the indices, so
    re-evaluate exactly those two expressions on the reference's tensors."""
    gf = ((geom - (shell.bx - shell.dx / 2.)) / shell.dx).long()
    kept = (gf[..., 0] >= 0) & (gf[..., 0] < shell.nx[0]) \
        & (gf[..., 1] >= 0) & (gf[..., 1] < shell.nx[1]) \
        & (gf[..., 2] >= 0) & (gf[..., 2] < shell.nx[2])
    return gf, kept


def main():
    os.makedirs(OUT, exist_ok=True)
    rt, rm, rmod = load_reference()
    torch.set_num_threads(8)

    # ---- G1 gen_dx_bx ------------------------------------------------------
    g1 = {}
    for tag, gc in (("default", GRID_DEFAULT), ("hires", GRID_HIRES), ("coarse", GRID_COARSE),
                    ("small_z2", GRID_SMALL_Z2)):
        dx, bx, nx = rt.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])
        g1[tag + "_bounds"] = np.array([gc["xbound"], gc["ybound"], gc["zbound"]], dtype=np.float64)
        g1[tag + "_dx"], g1[tag + "_bx"], g1[tag + "_nx"] = dx.numpy(), bx.numpy(), nx.numpy()
    save("g1_gen_dx_bx", **g1)

    # ---- G2 frustum --------------------------------------------------------
    s_def = make_lss_shell(rt, rm, rmod, 1, GRID_DEFAULT, {"final_dim": (128, 352), "Ncams": 6})
    s_hi = make_lss_shell(rt, rm, rmod, 1, GRID_HIRES, {"final_dim": (256, 704), "Ncams": 6})
    fr_hi = s_hi.frustum.detach().numpy()
    save("g2_frustum",
         default=s_def.frustum.detach().numpy(),
         hires_sha256=np.array(sha(fr_hi)),
         hires_shape=np.array(fr_hi.shape),
         hires_xs=fr_hi[0, 0, :, 0].copy(), hires_ys=fr_hi[0, :, 0, 1].copy(),
         hires_ds=fr_hi[:, 0, 0, 2].copy())

    # ---- G3 geometry -> voxel index ----------------------------------------
    def g3_case(tag, shell, calib, store_idx=True):
        rots, trans, intr, prot, ptr = calib
        with torch.no_grad():
            geom = shell.get_geometry(rots, trans, intr, prot, ptr)
            gf, kept = ref_indices(shell, geom)
            inv_pr = torch.inverse(prot)
            comb = rots.matmul(torch.inverse(intr))
        B = geom.shape[0]
        X, Y = int(shell.nx[0]), int(shell.nx[1])
        b = torch.arange(B).view(B, 1, 1, 1, 1)
        cell = torch.where(kept, (b * X + gf[..., 0]) * Y + gf[..., 1],
                           torch.full_like(gf[..., 0], -1)).to(torch.int32).reshape(-1)
        iz = torch.where(kept, gf[..., 2], torch.zeros_like(gf[..., 2])).to(torch.int32).reshape(-1)
        d = dict(rots=rots.numpy(), trans=trans.numpy(), intrins=intr.numpy(),
                 post_rots=prot.numpy(), post_trans=ptr.numpy(),
                 inv_post_rots=inv_pr.numpy(), combine=comb.numpy(),
                 geom_sha256=np.array(sha(geom.numpy())),
                 cell_sha256=np.array(sha(cell.numpy())),
                 n_kept=np.array(int(kept.sum())),
                 geom_sample=geom.reshape(-1, 3)[::97].numpy().copy())
        if store_idx:
            d["idx_i16"] = gf.reshape(-1, 3).clamp(-32768, 32767).to(torch.int16).numpy()
            d["kept"] = np.packbits(kept.reshape(-1).numpy())
            d["cell"] = cell.numpy()
            d["iz"] = iz.numpy().astype(np.int8)
        save(tag, **d)
        return geom

    for seed in (0, 1, 2):
        g3_case("g3_val_b1_s%d" % seed, s_def, synthetic_rig(1, seed=seed), store_idx=(seed == 0))
        g3_case("g3_train_b1_s%d" % seed, s_def, synthetic_rig(1, train_aug=True, seed=seed))
    # randn calibrations as in the reference's own smoke test
    # (src/model_vovnet_transformer.py:729-734): exercises NaN/inf/huge paths.
    for seed in (0, 1):
        g = torch.Generator().manual_seed(100 + seed)
        calib = (torch.randn(1, 6, 3, 3, generator=g), torch.randn(1, 6, 3, generator=g),
                 torch.randn(1, 6, 3, 3, generator=g), torch.randn(1, 6, 3, 3, generator=g),
                 torch.randn(1, 6, 3, generator=g))
        g3_case("g3_randn_b1_s%d" % seed, s_def, calib)
    s_def4 = make_lss_shell(rt, rm, rmod, 4, GRID_DEFAULT, {"final_dim": (128, 352), "Ncams": 6})
    g3_case("g3_train_b4_s0", s_def4, synthetic_rig(4, train_aug=True, seed=0), store_idx=False)
    s_hi2 = make_lss_shell(rt, rm, rmod, 2, GRID_HIRES, {"final_dim": (256, 704), "Ncams": 6})
    g3_case("g3_hires_b2_s0", s_hi2, synthetic_rig(2, final_dim=(256, 704), seed=0), store_idx=False)

    # ---- G4 / G7 small splat forward + backward -----------------------------
    def small_case(tag, gc, B, N, fH, fW, C, seed):
        torch.manual_seed(seed)
        shell = make_lss_shell(rt, rm, rmod, B, gc, {"final_dim": (fH * 16, fW * 16), "Ncams": N}, camC=C)
        D = shell.D
        # a rig that lands most points inside the tiny 20 m grid
        rots, trans, intr, prot, ptr = synthetic_rig(B, N=N, final_dim=(fH * 16, fW * 16),
                                                     train_aug=True, seed=seed)
        intr = intr.clone(); intr[..., 0, 0] = 60.0; intr[..., 1, 1] = 60.0
        intr[..., 0, 2] = fW * 8.0; intr[..., 1, 2] = fH * 8.0
        prot = torch.eye(3).repeat(B, N, 1, 1) * 1.0
        prot[..., 0, 0] = 0.9; prot[..., 1, 1] = 1.1
        ptr = torch.zeros(B, N, 3); ptr[..., 0] = 1.5; ptr[..., 1] = -2.0
        feat_in = torch.randn(B * N, 512, fH, fW, requires_grad=True)
        Gup = torch.randn(B, C * int(shell.nx[2]), int(shell.nx[0]), int(shell.nx[1]))
        geom = shell.get_geometry(rots, trans, intr, prot, ptr)
        depth, lifted = shell.camencode.get_depth_feat(feat_in)
        lifted.retain_grad()
        x = lifted.view(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
        out = shell.voxel_pooling(geom, x)
        (out * Gup).sum().backward()
        gf, kept = ref_indices(shell, geom.detach())
        with torch.no_grad():
            inv_pr = torch.inverse(prot); comb = rots.matmul(torch.inverse(intr))
        X, Y = int(shell.nx[0]), int(shell.nx[1])
        b = torch.arange(B).view(B, 1, 1, 1, 1)
        cell = torch.where(kept, (b * X + gf[..., 0]) * Y + gf[..., 1],
                           torch.full_like(gf[..., 0], -1)).to(torch.int32).reshape(-1)
        iz = torch.where(kept, gf[..., 2], torch.zeros_like(gf[..., 2])).to(torch.int32).reshape(-1)
        save(tag,
             bounds=np.array([gc["xbound"], gc["ybound"], gc["zbound"], gc["dbound"]], dtype=np.float64),
             dims=np.array([B, N, D, fH, fW, C]),
             rots=rots.numpy(), trans=trans.numpy(), intrins=intr.numpy(),
             post_rots=prot.numpy(), post_trans=ptr.numpy(),
             inv_post_rots=inv_pr.numpy(), combine=comb.numpy(),
             frustum=shell.frustum.detach().numpy(),
             depthnet_weight=shell.camencode.depthnet.weight.detach().numpy(),
             depthnet_bias=shell.camencode.depthnet.bias.detach().numpy(),
             feat_in=feat_in.detach().numpy(), grad_out=Gup.numpy(),
             geom=geom.detach().numpy(), cell=cell.numpy(), iz=iz.numpy(),
             depth=depth.detach().numpy(), out=out.detach().numpy(),
             grad_lifted=lifted.grad.numpy(), grad_feat_in=feat_in.grad.numpy(),
             grad_weight=shell.camencode.depthnet.weight.grad.numpy(),
             grad_bias=shell.camencode.depthnet.bias.grad.numpy())

    small_case("g4_small_z1", GRID_SMALL, B=2, N=2, fH=3, fW=4, C=4, seed=7)
    small_case("g4_small_z2", GRID_SMALL_Z2, B=2, N=3, fH=2, fW=5, C=64, seed=8)
    small_case("g4_small_c64", GRID_SMALL, B=1, N=2, fH=2, fW=3, C=64, seed=9)

    # ---- G4b full-size splat stats -------------------------------------------
    def full_case(tag, shell, B, final_dim, seed, train_aug):
        torch.manual_seed(seed)
        fH, fW = final_dim[0] // 16, final_dim[1] // 16
        calib = synthetic_rig(B, final_dim=final_dim, train_aug=train_aug, seed=seed)
        feat_in = torch.randn(B * 6, 512, fH, fW)
        with torch.no_grad():
            geom = shell.get_geometry(*calib)
            x = shell.get_cam_feats(feat_in)
            out = shell.voxel_pooling(geom, x)
            inv_pr = torch.inverse(calib[3]); comb = calib[0].matmul(torch.inverse(calib[2]))
        o = out.numpy()
        Bc, C, X, Y = o.shape
        occ = (np.abs(o).sum(1) > 0)
        g = np.random.RandomState(seed)
        occ_idx = np.argwhere(occ)
        pick = occ_idx[g.choice(len(occ_idx), size=384, replace=False)]
        emp = np.argwhere(~occ); pick_e = emp[g.choice(len(emp), size=128, replace=False)]
        pick = np.concatenate([pick, pick_e])
        rows = o[pick[:, 0], :, pick[:, 1], pick[:, 2]]
        save(tag, dims=np.array([B, 6, shell.D, fH, fW, 64]),
             seed=np.array(seed), train_aug=np.array(train_aug),
             rots=calib[0].numpy(), trans=calib[1].numpy(), intrins=calib[2].numpy(),
             post_rots=calib[3].numpy(), post_trans=calib[4].numpy(),
             inv_post_rots=inv_pr.numpy(), combine=comb.numpy(),
             depthnet_weight=shell.camencode.depthnet.weight.detach().numpy(),
             depthnet_bias=shell.camencode.depthnet.bias.detach().numpy(),
             feat_seed_check=feat_in.reshape(-1)[:16].numpy().copy(),
             feat_sha256=np.array(sha(feat_in.numpy())),
             n_occupied=np.array(int(occ.sum())),
             chan_sum=o.astype(np.float64).sum((0, 2, 3)),
             chan_l2=np.sqrt((o.astype(np.float64) ** 2).sum((0, 2, 3))),
             pick=pick.astype(np.int32), rows=rows)

    full_case("g4_full_b1_val", s_def, 1, (128, 352), 0, False)
    full_case("g4_full_b4_train", s_def4, 4, (128, 352), 1, True)

    # ---- G5 QuickCumsum / cumsum_trick hand-sized ----------------------------
    torch.manual_seed(5)
    x = torch.randn(10, 2, requires_grad=True)
    ranks = torch.tensor([3, 3, 3, 7, 9, 9, 12, 12, 12, 40])
    gfe = torch.arange(40).view(10, 4)
    y, gk = rt.QuickCumsum.apply(x, gfe, ranks)
    gy = torch.randn_like(y)
    y.backward(gy)
    y2, gk2 = rt.cumsum_trick(x.detach(), gfe, ranks)
    save("g5_quickcumsum", x=x.detach().numpy(), ranks=ranks.numpy(), geom=gfe.numpy(),
         y=y.detach().numpy(), geom_kept=gk.numpy(), grad_y=gy.numpy(), grad_x=x.grad.numpy(),
         y_cumsum_trick=y2.numpy(), geom_kept_cumsum_trick=gk2.numpy())

    # ---- G6 CamEncode --------------------------------------------------------
    torch.manual_seed(6)
    ce = rm.CamEncode(41, 64, 16)
    xin = torch.randn(2, 512, 2, 3)
    with torch.no_grad():
        logits = ce.depthnet(xin)
        depth, lifted = ce.get_depth_feat(xin)
        fwd = ce(xin)
    assert torch.equal(fwd, lifted)
    save("g6_camencode", x=xin.numpy(), weight=ce.depthnet.weight.detach().numpy(),
         bias=ce.depthnet.bias.detach().numpy(), logits=logits.numpy(),
         depth=depth.numpy(), lifted=lifted.numpy())

    # ---- G9 Up (bilinear align_corners + cat + 2x conv-BN-ReLU) ---------------
    for tag, sf, train in (("g9_up_x2_eval", 2, False), ("g9_up_x4_train", 4, True)):
        torch.manual_seed(9)
        up = rm.Up(8 + 4, 6, scale_factor=sf)
        for m in up.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0)
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_()
        up.train(train)
        x1 = torch.randn(2, 8, 3, 5); x2 = torch.randn(2, 4, 3 * sf, 5 * sf)
        sd0 = {k: v.clone() for k, v in up.state_dict().items()}
        with torch.no_grad():
            yup = up.up(x1)
            y = up(x1, x2)
        d = {"sd_" + k: v.numpy() for k, v in sd0.items()}
        d.update({"sd_after_" + k: v.numpy() for k, v in up.state_dict().items() if "running" in k})
        save(tag, x1=x1.numpy(), x2=x2.numpy(), upsampled=yup.numpy(), y=y.numpy(),
             scale=np.array(sf), training=np.array(train), **d)


def gen_host_input_path(rtools):
    """G12: the reference's own img_transform (src/tools.py:117-142) and sample_augmentation
    (src/data.py:90-112) on seeded draws - the calibration half of the loader."""
    import types

    from PIL import Image
    import src.data as rdata
    conf = {"resize_lim": (0.193, 0.225), "final_dim": (128, 352), "rot_lim": (-5.4, 5.4), "H": 900, "W": 1600,
            "rand_flip": True, "bot_pct_lim": (0.0, 0.22)}
    out = {}
    for mode, is_train in (("train", True), ("val", False)):
        np.random.seed(123)
        fake = types.SimpleNamespace(data_aug_conf=conf, is_train=is_train)
        rows, prs, pts = [], [], []
        for _ in range(8):
            resize, resize_dims, crop, flip, rotate = rdata.NuscData.sample_augmentation(fake)
            img = Image.new("RGB", (1600, 900))
            _, pr, pt = rtools.img_transform(img, torch.eye(2), torch.zeros(2), resize=resize, resize_dims=resize_dims,
                                             crop=crop, flip=flip, rotate=rotate)
            rows.append([resize, resize_dims[0], resize_dims[1], *crop, float(flip), rotate])
            prs.append(pr.numpy())
            pts.append(pt.numpy())
        out[mode + "_params"] = np.array(rows, dtype=np.float64)
        out[mode + "_post_rot"] = np.stack(prs)
        out[mode + "_post_tran"] = np.stack(pts)
    save("g12_host_input_path", **out)


if __name__ == "__main__":
    main()
    gen_host_input_path(load_reference()[0])
