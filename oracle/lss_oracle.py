"""CPU restatement of the reference's lift-splat path (torch-CPU and numpy).

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Two flavours of every step:
  * `*_torch`: the same ATen op sequence the reference issues (used as the
    timed `cpu_baseline` "port" and as the autograd oracle for backward);
  * `*_np`: explicit scalar-order numpy arithmetic (what the HIP kernels must
    reproduce bit for bit on the integer side) and an fp64 direct segmented sum
    (the clean numeric reference for the feature side).

Reference lines are cited per function as `ref: file:lines`
(paths relative to /root/reference).
"""
import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


# ----------------------------------------------------------------------------
# grid / frustum
# ----------------------------------------------------------------------------
def gen_dx_bx(xbound, ybound, zbound):
    """ref: src/tools.py:172-178.  dx=step, bx=lo+step/2 (python float math then
    fp32), nx=trunc((hi-lo)/step) in python float math -> int64."""
    rows = [xbound, ybound, zbound]
    dx = torch.tensor([r[2] for r in rows], dtype=torch.float32)
    bx = torch.tensor([r[0] + r[2] / 2.0 for r in rows], dtype=torch.float32)
    nx = torch.tensor([int((r[1] - r[0]) / r[2]) for r in rows], dtype=torch.int64)
    return dx, bx, nx


def create_frustum(final_dim, downsample, dbound):
    """ref: src/model_BEV_TXT.py:37-48.  (D,fH,fW,3) fp32 of (x_pix, y_pix, depth)."""
    ogfH, ogfW = final_dim
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = torch.arange(*dbound, dtype=torch.float)
    D = ds.shape[0]
    xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float)
    ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float)
    fr = torch.empty(D, fH, fW, 3, dtype=torch.float32)
    fr[..., 0] = xs.view(1, 1, fW)
    fr[..., 1] = ys.view(1, fH, 1)
    fr[..., 2] = ds.view(D, 1, 1)
    return fr


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
def calib_matrices(rots, intrins, post_rots):
    """The two per-camera matrices of ref: src/model_BEV_TXT.py:60,66 -
    inv(post_rots) and rots @ inv(intrins) - through the same torch-CPU calls
    (MKL LAPACK / bmm) the reference makes."""
    return torch.inverse(post_rots), rots.matmul(torch.inverse(intrins))


def get_geometry_torch(frustum, rots, trans, intrins, post_rots, post_trans):
    """ref: src/model_BEV_TXT.py:50-70, op for op."""
    B, N, _ = trans.shape
    pts = frustum - post_trans.view(B, N, 1, 1, 1, 3)
    pts = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
    pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
    combine = rots.matmul(torch.inverse(intrins))
    pts = combine.view(B, N, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1)
    pts += trans.view(B, N, 1, 1, 1, 3)
    return pts


def _mat3_apply_np(m, p):
    """Row i of a 3x3 * 3-vector with every mul and add rounded to fp32 and
    the association ((m0*p0 + m1*p1) + m2*p2): what torch-CPU's batched 3x3
    matmul produces (SURVEY.md 8a-3 probe, re-checked by test_oracle_golden)."""
    out = np.empty(p.shape, dtype=f32)
    for i in range(3):
        t = (m[..., i, 0] * p[..., 0]).astype(f32)
        t = (t + (m[..., i, 1] * p[..., 1]).astype(f32)).astype(f32)
        out[..., i] = (t + (m[..., i, 2] * p[..., 2]).astype(f32)).astype(f32)
    return out


def geometry_points_np(frustum, inv_post_rots, post_trans, combine, trans):
    """Explicit-order fp32 restatement of ref: src/model_BEV_TXT.py:59-68 given
    the two precomputed matrices.  All arrays numpy fp32.
    frustum (D,fH,fW,3); inv_post_rots, combine (B,N,3,3); post_trans, trans (B,N,3)
    -> (B,N,D,fH,fW,3)."""
    fr = frustum[None, None].astype(f32)
    p = (fr - post_trans[:, :, None, None, None, :]).astype(f32)
    q = _mat3_apply_np(inv_post_rots[:, :, None, None, None], p)
    r = np.empty_like(q)
    r[..., 0] = (q[..., 0] * q[..., 2]).astype(f32)
    r[..., 1] = (q[..., 1] * q[..., 2]).astype(f32)
    r[..., 2] = q[..., 2]
    g = _mat3_apply_np(combine[:, :, None, None, None], r)
    return (g + trans[:, :, None, None, None, :]).astype(f32)


def voxel_indices_np(geom, dx, bx, nx):
    """ref: src/model_BEV_TXT.py:92,99-103.  idx = trunc((geom-(bx-dx/2))/dx)
    as int64 with x86 cvttss2si semantics (NaN/inf/|q|>=2^63 -> INT64_MIN), and
    the in-box mask.  geom (...,3) fp32 -> idx (...,3) int64, kept (...) bool."""
    dx = np.asarray(dx, dtype=f32)
    bx = np.asarray(bx, dtype=f32)
    lo = (bx - (dx / f32(2.0)).astype(f32)).astype(f32)
    q = ((geom.astype(f32) - lo).astype(f32) / dx).astype(f32)
    bad = ~np.isfinite(q) | (np.abs(q) >= f32(2.0 ** 63))
    idx = np.where(bad, 0, np.trunc(np.where(bad, 0, q))).astype(np.int64)
    idx[bad] = np.iinfo(np.int64).min
    nx = np.asarray(nx, dtype=np.int64)
    kept = np.ones(idx.shape[:-1], dtype=bool)
    for a in range(3):
        kept &= (idx[..., a] >= 0) & (idx[..., a] < nx[a])
    return idx, kept


def cell_ids_np(geom, dx, bx, nx):
    """The build's flat cell id per frustum point: (b*X + ix)*Y + iy with iz
    carried separately, -1 when the reference would drop the point.
    geom (B,N,D,fH,fW,3) -> cell (B*N*D*fH*fW,) int32, iz int32."""
    B = geom.shape[0]
    idx, kept = voxel_indices_np(geom, dx, bx, nx)
    X, Y = int(nx[0]), int(nx[1])
    b = np.arange(B, dtype=np.int64).reshape(B, 1, 1, 1, 1)
    cell = (b * X + idx[..., 0]) * Y + idx[..., 1]
    cell = np.where(kept, cell, -1).astype(np.int32).reshape(-1)
    iz = np.where(kept, idx[..., 2], 0).astype(np.int32).reshape(-1)
    return cell, iz


# ----------------------------------------------------------------------------
# CamEncode
# ----------------------------------------------------------------------------
def cam_encode_torch(x, weight, bias, D, C):
    """ref: src/modules.py:79-91.  x (BN,512,fH,fW); weight (D+C,512,1,1).
    Returns depth (BN,D,fH,fW) and the lifted tensor (BN,C,D,fH,fW)."""
    y = F.conv2d(x, weight, bias)
    depth = y[:, :D].softmax(dim=1)
    lifted = depth.unsqueeze(1) * y[:, D:D + C].unsqueeze(2)
    return depth, lifted


def get_cam_feats_torch(x, weight, bias, bsize, D, C):
    """ref: src/model_BEV_TXT.py:72-82 -> (B,N,D,fH,fW,C) permuted view."""
    BN, _, imH, imW = x.shape
    _, lifted = cam_encode_torch(x, weight, bias, D, C)
    lifted = lifted.view(bsize, BN // bsize, C, D, imH, imW)
    return lifted.permute(0, 1, 3, 4, 5, 2)


# ----------------------------------------------------------------------------
# cumsum trick
# ----------------------------------------------------------------------------
def cumsum_trick(x, geom_feats, ranks):
    """ref: src/tools.py:181-189."""
    x = x.cumsum(0)
    kept = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
    kept[:-1] = ranks[1:] != ranks[:-1]
    x, geom_feats = x[kept], geom_feats[kept]
    x = torch.cat((x[:1], x[1:] - x[:-1]))
    return x, geom_feats


class QuickCumsum(torch.autograd.Function):
    """ref: src/tools.py:192-218."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        x = x.cumsum(0)
        kept = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
        kept[:-1] = ranks[1:] != ranks[:-1]
        x, geom_feats = x[kept], geom_feats[kept]
        x = torch.cat((x[:1], x[1:] - x[:-1]))
        ctx.save_for_backward(kept)
        ctx.mark_non_differentiable(geom_feats)
        return x, geom_feats

    @staticmethod
    def backward(ctx, gradx, gradgeom):
        kept, = ctx.saved_tensors
        back = torch.cumsum(kept, 0)
        back[kept] -= 1
        return gradx[back], None, None


# ----------------------------------------------------------------------------
# voxel pooling
# ----------------------------------------------------------------------------
def voxel_pooling_torch(geom, x, dx, bx, nx, use_quickcumsum=True):
    """ref: src/model_BEV_TXT.py:84-126, op for op.
    geom (B,N,D,H,W,3) f32; x (B,N,D,H,W,C) f32 -> (B, C*nz, nx, ny) f32."""
    B, N, D, H, W, C = x.shape
    Nprime = B * N * D * H * W
    x = x.reshape(Nprime, C)
    gf = ((geom - (bx - dx / 2.)) / dx).long().view(Nprime, 3)
    batch_ix = torch.cat([torch.full([Nprime // B, 1], ix, dtype=torch.long)
                          for ix in range(B)])
    gf = torch.cat((gf, batch_ix), 1)
    kept = (gf[:, 0] >= 0) & (gf[:, 0] < nx[0]) \
        & (gf[:, 1] >= 0) & (gf[:, 1] < nx[1]) \
        & (gf[:, 2] >= 0) & (gf[:, 2] < nx[2])
    x = x[kept]
    gf = gf[kept]
    ranks = gf[:, 0] * (nx[1] * nx[2] * B) + gf[:, 1] * (nx[2] * B) \
        + gf[:, 2] * B + gf[:, 3]
    order = ranks.argsort()
    x, gf, ranks = x[order], gf[order], ranks[order]
    if use_quickcumsum:
        x, gf = QuickCumsum.apply(x, gf, ranks)
    else:
        x, gf = cumsum_trick(x, gf, ranks)
    final = torch.zeros((B, C, int(nx[2]), int(nx[0]), int(nx[1])))
    final[gf[:, 3], :, gf[:, 2], gf[:, 0], gf[:, 1]] = x
    return torch.cat(final.unbind(dim=2), 1)


def splat_direct_np(cell, iz, depth, feat, B, N, D, fH, fW, C, X, Y, Z,
                    dtype=np.float64):
    """Clean numeric reference: per-voxel direct sum of depth[p]*feat[pix(p),:]
    accumulated in `dtype` (fp64 by default).  Never materialises more than
    the (P,C) products.
    cell (P,) int32 (-1 = dropped), iz (P,) int32, depth (B*N,D,fH,fW),
    feat (B*N,C,fH,fW) -> (B, C*Z, X, Y) in `dtype`."""
    P = B * N * D * fH * fW
    dep = np.asarray(depth, dtype=dtype).reshape(B * N, D, fH * fW)
    ft = np.asarray(feat, dtype=dtype).reshape(B * N, C, fH * fW)
    # lifted[bn, d, pix, c]
    lifted = dep[:, :, :, None] * ft.transpose(0, 2, 1)[:, None, :, :]
    lifted = lifted.reshape(P, C)
    out = np.zeros((B * X * Y, Z, C), dtype=dtype)
    keep = cell >= 0
    np.add.at(out, (cell[keep].astype(np.int64), iz[keep].astype(np.int64)),
              lifted[keep])
    out = out.reshape(B, X, Y, Z, C).transpose(0, 3, 4, 1, 2)  # B,Z,C,X,Y
    return np.ascontiguousarray(out).reshape(B, Z * C, X, Y)


def lift_splat_torch(feat_in, weight, bias, frustum, rots, trans, intrins,
                     post_rots, post_trans, dx, bx, nx, bsize, D, C):
    """ref: src/model_BEV_TXT.py:128-133 `get_voxels` (geometry || CamEncode ->
    voxel_pooling) as one callable: the L1 CPU baseline."""
    geom = get_geometry_torch(frustum, rots, trans, intrins, post_rots, post_trans)
    x = get_cam_feats_torch(feat_in, weight, bias, bsize, D, C)
    return voxel_pooling_torch(geom, x, dx, bx, nx)


# ----------------------------------------------------------------------------
# synthetic calibration rig - shared by tests, bench, goldens
# ----------------------------------------------------------------------------
def synthetic_rig(B, N=6, final_dim=(128, 352), train_aug=False, seed=0,
                  H=900, W=1600):
    """nuScenes-like 6-camera rig (values from ref: src/data.py:93-133 and
    src/tools.py:118-142).  Returns rots, trans, intrins, post_rots, post_trans
    as fp32 torch CPU tensors of shapes (B,N,3,3)/(B,N,3)."""
    g = np.random.RandomState(seed)
    yaws = np.deg2rad([55.0, 0.0, -55.0, 110.0, 180.0, -110.0])
    base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
    fH, fW = final_dim
    rots = np.zeros((B, N, 3, 3)); trans = np.zeros((B, N, 3))
    intr = np.zeros((B, N, 3, 3)); prot = np.zeros((B, N, 3, 3)); ptr = np.zeros((B, N, 3))
    for b in range(B):
        for n in range(N):
            y = yaws[n % 6]
            Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
            rots[b, n] = Rz @ base
            trans[b, n] = [1.5 * np.cos(y), 0.5 * np.sin(y), 1.5]
            intr[b, n] = [[1266.0, 0, 816.0], [0, 1266.0, 491.0], [0, 0, 1]]
            if not train_aug:
                resize = max(fH / H, fW / W)
                newW, newH = int(W * resize), int(H * resize)
                crop_h = int((1 - 0.22 / 2) * newH) - fH
                crop_w = int(max(0, newW - fW) / 2)
                flip, rot = False, 0.0
            else:
                lo, hi = (0.193, 0.225) if fH == 128 else (0.193 * fH / 128, 0.225 * fH / 128)
                resize = g.uniform(lo, hi)
                newW, newH = int(W * resize), int(H * resize)
                crop_h = int((1 - g.uniform(0.0, 0.22)) * newH) - fH
                crop_w = int(g.uniform(0, max(0, newW - fW)))
                flip = bool(g.choice([0, 1]))
                rot = g.uniform(-5.4, 5.4)
            # img_transform (ref: src/tools.py:118-142)
            pr = np.eye(2) * resize
            pt = -np.array([crop_w, crop_h], dtype=np.float64)
            if flip:
                A = np.array([[-1.0, 0], [0, 1]]); bvec = np.array([fW, 0.0])
                pr = A @ pr; pt = A @ pt + bvec
            h = rot / 180 * np.pi
            A = np.array([[np.cos(h), np.sin(h)], [-np.sin(h), np.cos(h)]])
            bvec = np.array([fW, fH]) / 2
            bvec = A @ (-bvec) + bvec
            pr = A @ pr; pt = A @ pt + bvec
            prot[b, n] = np.eye(3); prot[b, n, :2, :2] = pr
            ptr[b, n, :2] = pt
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    return t(rots), t(trans), t(intr), t(prot), t(ptr)
