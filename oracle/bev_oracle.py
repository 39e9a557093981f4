"""CPU restatement of `BevEncode` / `Up` (ref: src/modules.py:9-27, 94-130).

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Functional: every function takes a flat `state_dict`-style mapping
(name -> torch.Tensor) whose keys are the reference module's own
(`conv1.weight`, `bn1.*`, `layer1.0.conv1.weight`, ..., `up1.conv.0.weight`,
`up2.1.weight`, `up2.4.bias`) so the product module's state_dict can be fed in
directly.

`Up` is pinned by tests/golden/g9_*.npz (the reference's own class, run on
CPU).  `layer1..3` are torchvision==0.13.1 resnet18 BasicBlocks
(environment.yaml:62; call sites src/modules.py:98,104-106) which are absent
offline: restated from the published definition - PARITY UNPINNED:
    out = relu(bn1(conv3x3(x, stride)))
    out = bn2(conv3x3(out))
    out = relu(out + (downsample(x) if stride != 1 or cin != cout else x))
    downsample = conv1x1(stride) -> BN;  all convs bias-free.
"""
import torch
import torch.nn.functional as F


def _bn(x, sd, pre, training, momentum=0.1, eps=1e-5, stats_out=None):
    rm, rv = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if training:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, sd[pre + ".weight"], sd[pre + ".bias"],
                     training=training, momentum=momentum, eps=eps)
    if stats_out is not None and training:
        stats_out[pre + ".running_mean"] = rm
        stats_out[pre + ".running_var"] = rv
    return y


def upsample_bilinear_ac(x, scale):
    """nn.Upsample(scale_factor, 'bilinear', align_corners=True), ref: src/modules.py:13-14."""
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)


def up_block(x1, x2, sd, pre, scale, training=False, stats_out=None):
    """ref: src/modules.py:9-27 `Up.forward`: upsample x1, cat([x2, x1_up]) (skip
    first), then 2 x [3x3 conv - BN - ReLU]."""
    x1 = upsample_bilinear_ac(x1, scale)
    x = torch.cat([x2, x1], dim=1)
    x = F.conv2d(x, sd[pre + ".conv.0.weight"], None, padding=1)
    x = F.relu(_bn(x, sd, pre + ".conv.1", training, stats_out=stats_out))
    x = F.conv2d(x, sd[pre + ".conv.3.weight"], None, padding=1)
    x = F.relu(_bn(x, sd, pre + ".conv.4", training, stats_out=stats_out))
    return x


def basic_block(x, sd, pre, stride, training=False, stats_out=None):
    idt = x
    out = F.conv2d(x, sd[pre + ".conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(_bn(out, sd, pre + ".bn1", training, stats_out=stats_out))
    out = F.conv2d(out, sd[pre + ".conv2.weight"], None, padding=1)
    out = _bn(out, sd, pre + ".bn2", training, stats_out=stats_out)
    if (pre + ".downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[pre + ".downsample.0.weight"], None, stride=stride)
        idt = _bn(idt, sd, pre + ".downsample.1", training, stats_out=stats_out)
    return F.relu(out + idt)


def bev_encode(x, sd, training=False, stats_out=None, return_intermediates=False):
    """ref: src/modules.py:118-130 `BevEncode.forward`.  x (B,inC,X,Y) -> (B,outC,X,Y)."""
    inter = {}
    x = F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, sd, "bn1", training, stats_out=stats_out))
    inter["stem"] = x
    x = basic_block(x, sd, "layer1.0", 1, training, stats_out)
    x1 = basic_block(x, sd, "layer1.1", 1, training, stats_out)
    inter["layer1"] = x1
    x = basic_block(x1, sd, "layer2.0", 2, training, stats_out)
    x = basic_block(x, sd, "layer2.1", 1, training, stats_out)
    inter["layer2"] = x
    x = basic_block(x, sd, "layer3.0", 2, training, stats_out)
    x = basic_block(x, sd, "layer3.1", 1, training, stats_out)
    inter["layer3"] = x
    x = up_block(x, x1, sd, "up1", 4, training, stats_out)
    inter["up1"] = x
    x = upsample_bilinear_ac(x, 2)
    x = F.conv2d(x, sd["up2.1.weight"], None, padding=1)
    x = F.relu(_bn(x, sd, "up2.2", training, stats_out=stats_out))
    inter["up2"] = x
    x = F.conv2d(x, sd["up2.4.weight"], sd["up2.4.bias"])
    if return_intermediates:
        return x, inter
    return x


def bev_encode_state_shapes(inC=64, outC=4):
    """Ordered (key, shape) list of BevEncode's state_dict:
    used by tests to check the product module exposes identical keys."""
    out = [("conv1.weight", (64, inC, 7, 7))]

    def bn(pre, c):
        return [(pre + ".weight", (c,)), (pre + ".bias", (c,)), (pre + ".running_mean", (c,)),
                (pre + ".running_var", (c,)), (pre + ".num_batches_tracked", ())]

    out += bn("bn1", 64)
    cin = 64
    for li, c in ((1, 64), (2, 128), (3, 256)):
        for bi in (0, 1):
            pre = "layer%d.%d" % (li, bi)
            out.append((pre + ".conv1.weight", (c, cin if bi == 0 else c, 3, 3)))
            out += bn(pre + ".bn1", c)
            out.append((pre + ".conv2.weight", (c, c, 3, 3)))
            out += bn(pre + ".bn2", c)
            if bi == 0 and cin != c:
                out.append((pre + ".downsample.0.weight", (c, cin, 1, 1)))
                out += bn(pre + ".downsample.1", c)
        cin = c
    out.append(("up1.conv.0.weight", (256, 320, 3, 3)))
    out += bn("up1.conv.1", 256)
    out.append(("up1.conv.3.weight", (256, 256, 3, 3)))
    out += bn("up1.conv.4", 256)
    out.append(("up2.1.weight", (128, 256, 3, 3)))
    out += bn("up2.2", 128)
    out.append(("up2.4.weight", (outC, 128, 1, 1)))
    out.append(("up2.4.bias", (outC,)))
    return out
