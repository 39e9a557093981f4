"""CPU restatement of the vovnet-model pieces of the hot path and of the BEV
transformer that follows it (ref: src/model_vovnet_transformer.py:22-173,
483-554; src/transformer_modules.py:12-263).

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Functional style: every function takes a flat `state_dict`-like mapping whose
keys are the reference modules' own (`depth_head.0.weight`, `depth_c3.1.running_var`,
`feat_proj.weight`, `transformer.encoder.self_attn.sampling_offsets.weight`, ...),
so the product modules' state_dicts can be fed in directly.  Eval-mode
semantics (BatchNorm running statistics, dropout off).

PINNED by tests/golden/g10_*.npz and g11_*.npz, which tools/gen_golden_vovnet.py
produced by running the reference's own classes on CPU (the VoVNet trunk itself
is `timm`, absent offline and outside the hot path: the fixtures start at its
C3/C4 outputs).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import lss_oracle


# --------------------------------------------------------------------------
# deterministic, platform-independent parameters for fixtures and tests
# --------------------------------------------------------------------------
def seeded_state(shapes, seed):
    """`shapes`: ordered (key, shape) pairs of a state_dict.  Values come from
    numpy's RandomState (NOT torch's RNG) so the generator (which loads them into
    the reference's modules) and the tests (which load them into the product's)
    see identical numbers on any host."""
    g = np.random.RandomState(seed)
    out = {}
    for key, shape in shapes:
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.long)
            continue
        if leaf == "running_var":
            v = g.uniform(0.5, 1.5, size=shape)
        elif leaf == "running_mean":
            v = g.normal(0.0, 0.1, size=shape)
        elif len(shape) == 1 and leaf == "weight":  # BatchNorm / LayerNorm gain
            v = g.uniform(0.5, 1.5, size=shape)
        elif len(shape) == 1:  # biases
            v = g.normal(0.0, 0.1, size=shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = g.normal(0.0, 1.0 / math.sqrt(fan_in), size=shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return out


def _bn_shapes(pre, c):
    return [(pre + ".weight", (c,)), (pre + ".bias", (c,)), (pre + ".running_mean", (c,)),
            (pre + ".running_var", (c,)), (pre + ".num_batches_tracked", ())]


def _head_shapes(pre, cin, D):
    """Conv3x3(cin,256)+bias, BN(256), ReLU, Conv1x1(256,D)+bias (ref :27-32, :78-83)."""
    return ([(pre + ".0.weight", (256, cin, 3, 3)), (pre + ".0.bias", (256,))] + _bn_shapes(pre + ".1", 256)
            + [(pre + ".3.weight", (D, 256, 1, 1)), (pre + ".3.bias", (D,))])


def standard_depthnet_shapes(c3_channels=768, D=41):
    return _head_shapes("depth_head", c3_channels, D)


def multiscale_depthnet_shapes(c3_channels=768, c4_channels=1024, D=41):
    return (_head_shapes("depth_c3", c3_channels, D) + _head_shapes("depth_c4", c4_channels, D)
            + [("fusion.0.weight", (D, 2 * D, 1, 1)), ("fusion.0.bias", (D,))] + _bn_shapes("fusion.1", D))


def camencode_v2_shapes(C_in=768, C_out=128):
    return [("feat_proj.weight", (C_out, C_in, 1, 1)), ("feat_proj.bias", (C_out,))]


def transformer_shapes(pre="", d=256, heads=8, points=8, ff=1024):
    """LightweightBEVTransformer's state_dict (ref: src/transformer_modules.py:62-85,
    164-183, 210-224; PositionEmbeddingSine has no parameters)."""
    e = pre + "encoder."
    a = e + "self_attn."
    return [(a + "sampling_offsets.weight", (heads * points * 2, d)), (a + "sampling_offsets.bias", (heads * points * 2,)),
            (a + "attention_weights.weight", (heads * points, d)), (a + "attention_weights.bias", (heads * points,)),
            (a + "value_proj.weight", (d, d)), (a + "value_proj.bias", (d,)),
            (a + "output_proj.weight", (d, d)), (a + "output_proj.bias", (d,)),
            (e + "linear1.weight", (ff, d)), (e + "linear1.bias", (ff,)),
            (e + "linear2.weight", (d, ff)), (e + "linear2.bias", (d,)),
            (e + "norm1.weight", (d,)), (e + "norm1.bias", (d,)),
            (e + "norm2.weight", (d,)), (e + "norm2.bias", (d,))]


def bev_encoder_transformer_shapes(in_channels=128, outC=4):
    """ref: src/model_vovnet_transformer.py:127-154."""
    return ([("compress.0.weight", (256, in_channels, 1, 1)), ("compress.0.bias", (256,))] + _bn_shapes("compress.1", 256)
            + transformer_shapes("transformer.")
            + [("seg_head.0.weight", (128, 256, 3, 3)), ("seg_head.0.bias", (128,))] + _bn_shapes("seg_head.1", 128)
            + [("seg_head.3.weight", (64, 128, 3, 3)), ("seg_head.3.bias", (64,))] + _bn_shapes("seg_head.4", 64)
            + [("seg_head.6.weight", (outC, 64, 1, 1)), ("seg_head.6.bias", (outC,))])


# --------------------------------------------------------------------------
# depth heads + CamEncodeV2
# --------------------------------------------------------------------------
def _bn_eval(x, sd, pre, eps=1e-5):
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], sd[pre + ".weight"],
                        sd[pre + ".bias"], training=False, eps=eps)


def _depth_head(x, sd, pre):
    """Conv3x3 - BN - ReLU - Conv1x1 -> raw depth logits."""
    h = F.conv2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], padding=1)
    h = F.relu(_bn_eval(h, sd, pre + ".1"))
    return F.conv2d(h, sd[pre + ".3.weight"], sd[pre + ".3.bias"])


def standard_depthnet(c3, sd):
    """ref: src/model_vovnet_transformer.py:85-87."""
    return F.softmax(_depth_head(c3, sd, "depth_head"), dim=1)


def upsample_bilinear_half_pixel(x, size):
    """F.interpolate(mode='bilinear', align_corners=False) written out (ref :62):
    src = (dst + 0.5) * in/out - 0.5, clamped at 0; neighbours clamped at the edge."""
    B, C, H, W = x.shape
    oh, ow = size

    def axis(n_in, n_out):
        s = torch.arange(n_out, dtype=torch.float32)
        src = (np.float32(n_in) / np.float32(n_out)) * (s + 0.5) - 0.5
        src = torch.clamp(src, min=0.0)
        i0 = src.floor().long()
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        l1 = src - i0.float()
        return i0, i1, 1.0 - l1, l1

    h0, h1, lh0, lh1 = axis(H, oh)
    w0, w1, lw0, lw1 = axis(W, ow)
    top = x[:, :, h0][:, :, :, w0] * lw0 + x[:, :, h0][:, :, :, w1] * lw1
    bot = x[:, :, h1][:, :, :, w0] * lw0 + x[:, :, h1][:, :, :, w1] * lw1
    return top * lh0[:, None] + bot * lh1[:, None]


def multiscale_depthnet(c3, c4, sd):
    """ref: src/model_vovnet_transformer.py:50-70."""
    d3 = _depth_head(c3, sd, "depth_c3")
    d4 = _depth_head(c4, sd, "depth_c4")
    d4u = upsample_bilinear_half_pixel(d4, d3.shape[2:])
    f = F.conv2d(torch.cat([d3, d4u], dim=1), sd["fusion.0.weight"], sd["fusion.0.bias"])
    f = F.relu(_bn_eval(f, sd, "fusion.1"))
    return F.softmax(f, dim=1)


def camencode_v2(features, depth, sd):
    """ref: src/model_vovnet_transformer.py:100-122 -> (BN, C_out, D, H, W)."""
    feat = F.conv2d(features, sd["feat_proj.weight"], sd["feat_proj.bias"])
    return feat.unsqueeze(2) * depth.unsqueeze(1)


def vovnet_lift_splat(c3, c4, depth_sd, cam_sd, lss_version, frustum, rots, trans, intrins, post_rots,
                      post_trans, dx, bx, nx, bsize):
    """BEV branch of VoVNetBEVTransformer.forward up to voxel_pooling
    (ref: src/model_vovnet_transformer.py:586-602; get_geometry :496-511 and
    voxel_pooling :513-554 are the same arithmetic as src/model_BEV_TXT.py:50-126)."""
    depth = multiscale_depthnet(c3, c4, depth_sd) if lss_version == "v2" else standard_depthnet(c3, depth_sd)
    cam = camencode_v2(c3, depth, cam_sd)
    BN, C, D, fH, fW = cam.shape
    cam = cam.view(bsize, BN // bsize, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
    geom = lss_oracle.get_geometry_torch(frustum, rots, trans, intrins, post_rots, post_trans)
    return lss_oracle.voxel_pooling_torch(geom, cam, dx, bx, nx), depth


# --------------------------------------------------------------------------
# BEV transformer (ref: src/transformer_modules.py)
# --------------------------------------------------------------------------
def position_embedding_sine(H, W, num_pos_feats=128, temperature=10000, scale=2 * math.pi):
    """ref: src/transformer_modules.py:25-59 (normalize=True) -> (2*npf, H, W):
    channels [0,npf) encode the row (y), [npf, 2npf) the column (x);
    even channel = sin, odd = cos of coord / T^(2*(i//2)/npf)."""
    y = torch.arange(H, dtype=torch.float32) / (H - 1) * scale
    x = torch.arange(W, dtype=torch.float32) / (W - 1) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, None] / dim_t
    py = y[:, None] / dim_t
    even = (torch.arange(num_pos_feats) % 2 == 0)
    px = torch.where(even, px.sin(), px.cos())
    py = torch.where(even, py.sin(), py.cos())
    pos = torch.cat([py[:, None, :].expand(H, W, -1), px[None, :, :].expand(H, W, -1)], dim=2)
    return pos.permute(2, 0, 1).contiguous()


def reference_points(H, W):
    """ref: src/transformer_modules.py:243-247 -> (H*W, 2) of (x, y) in [0, 1]."""
    gy, gx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    return torch.stack([gx, gy], dim=-1).view(-1, 2)


def bilinear_zero_pad(v, px, py):
    """grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False) written
    out.  v (B, H, W, ch); px, py (B, M) pixel coordinates (x = loc*W - 0.5).
    Returns (B, M, ch)."""
    B, H, W, ch = v.shape
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    out = 0
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wgt = (1 - (px - xi).abs()) * (1 - (py - yi).abs())
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()
            g = torch.gather(v.reshape(B, H * W, ch), 1, idx[..., None].expand(-1, -1, ch))
            out = out + g * (wgt * ok)[..., None]
    return out


def deformable_attention(query, value, ref_pts, sd, pre, n_heads=8, n_points=8):
    """ref: src/transformer_modules.py:104-161.  query/value (B, N, C), ref_pts (N, 2)."""
    B, N, C = query.shape
    H = W = int(math.sqrt(N))
    ch = C // n_heads
    off = F.linear(query, sd[pre + "sampling_offsets.weight"], sd[pre + "sampling_offsets.bias"])
    off = off.view(B, N, n_heads, n_points, 2)
    aw = F.linear(query, sd[pre + "attention_weights.weight"], sd[pre + "attention_weights.bias"])
    aw = F.softmax(aw.view(B, N, n_heads, n_points), dim=-1)
    loc = (ref_pts[None, :, None, None, :] + off / H).clamp(0, 1)  # BOTH axes divided by H (ref :124)
    v = F.linear(value, sd[pre + "value_proj.weight"], sd[pre + "value_proj.bias"]).view(B, H, W, n_heads, ch)
    out = torch.zeros(B, N, n_heads, ch)
    for h in range(n_heads):
        # grid = 2*loc - 1; align_corners=False pixel coordinate = ((grid + 1) * size - 1) / 2
        g = loc[:, :, h] * 2.0 - 1.0  # (B, N, P, 2)
        px = ((g[..., 0] + 1) * W - 1) / 2
        py = ((g[..., 1] + 1) * H - 1) / 2
        s = bilinear_zero_pad(v[:, :, :, h], px.reshape(B, -1), py.reshape(B, -1)).view(B, N, n_points, ch)
        out[:, :, h] = (s * aw[:, :, h, :, None]).sum(dim=2)
    return F.linear(out.view(B, N, C), sd[pre + "output_proj.weight"], sd[pre + "output_proj.bias"])


def transformer_encoder_layer(src, pos, ref_pts, sd, pre, n_heads=8):
    """ref: src/transformer_modules.py:185-207 (eval: dropout = identity; GELU = erf form)."""
    C = src.shape[-1]
    q = src + pos.flatten(1).t()[None]
    src2 = deformable_attention(q, src, ref_pts, sd, pre + "self_attn.", n_heads)
    src = F.layer_norm(src + src2, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    ff = F.linear(F.gelu(F.linear(src, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])),
                  sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return F.layer_norm(src + ff, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])


def lightweight_bev_transformer(x, sd, pre="", n_heads=8):
    """ref: src/transformer_modules.py:226-258.  x (B, C, H, W) -> (B, C, H, W)."""
    B, C, H, W = x.shape
    pos = position_embedding_sine(H, W, C // 2)
    y = transformer_encoder_layer(x.flatten(2).permute(0, 2, 1), pos, reference_points(H, W), sd,
                                  pre + "encoder.", n_heads)
    return y.permute(0, 2, 1).reshape(B, C, H, W)


def bev_encoder_transformer(x, sd):
    """ref: src/model_vovnet_transformer.py:156-173 -> (seg, refined)."""
    h = F.conv2d(x, sd["compress.0.weight"], sd["compress.0.bias"])
    h = F.relu(_bn_eval(h, sd, "compress.1"))
    r = lightweight_bev_transformer(h, sd, "transformer.")
    s = F.conv2d(r, sd["seg_head.0.weight"], sd["seg_head.0.bias"], padding=1)
    s = F.relu(_bn_eval(s, sd, "seg_head.1"))
    s = F.conv2d(s, sd["seg_head.3.weight"], sd["seg_head.3.bias"], padding=1)
    s = F.relu(_bn_eval(s, sd, "seg_head.4"))
    return F.conv2d(s, sd["seg_head.6.weight"], sd["seg_head.6.bias"]), r
