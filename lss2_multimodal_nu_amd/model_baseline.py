"""Drop-in for the reference's `src/model_baseline.py` (used by `pre_train.py:7,36`).

* `LSS` (:11-140) is byte-identical to `src/model_BEV_TXT.py`'s, so the same class serves
  `compile_model_lss` (:293-294).
* `BEV_TXT` (:143-290) is the "only-BEV" multi-task variant: the BEV half is the same camera->BEV
  hot path (HIP kernels K2..K8 through `_LiftSplatMixin`), but the action / description logits are
  predicted from the BEV map ALONE - `bev[:, :, 60:140, 56:144]` (NOT detached, :283) -> `BevPost`
  -> `Embedder_f2(8)` -> two `Predictor`s (:283-288) - and no camera feature reaches the heads.
  `sceneunder` is constructed (it is a `state_dict` entry of the reference, :170) but never called.
  Factory: `compile_model_onlybev` (:295-296).

The heads (8 x 8 x 22 values per sample) are stock PyTorch like the other TXT heads (SURVEY.md
section 2: out of the hot path); under autograd their gradient flows back into the BEV map, as in
the reference.
"""
import torch
from torch import nn

from .heads import BevPost, Embedder_f2, Predictor, SceneUnder
from .model_BEV_TXT import LSS, _LiftSplatMixin, _bev_class_weights, compile_model_lss  # noqa: F401
from .tools import head_weighted_cross_entropy


class BEV_TXT(_LiftSplatMixin, nn.Module):
    """Only-BEV variant: (bev, act_f, desc) with both heads fed by the BEV crop around the ego vehicle."""

    def __init__(self, bsize, grid_conf, data_aug_conf, outC, encoder=None, precision=None):
        nn.Module.__init__(self)

        def heads():
            self.sceneunder = SceneUnder()  # parameter container only (ref: constructed :170, unused in forward)
            self.embeder_bev = Embedder_f2(out_channels=8)
            self.predictor_bev1 = Predictor(num_in=8, classes=4)
            self.predictor_bev2 = Predictor(num_in=8, classes=8)

        self._init_lift_splat(bsize, grid_conf, data_aug_conf, outC, encoder, precision, heads)
        self.bevpost = BevPost()

    def forward(self, x, rots, trans, intrins, post_rots, post_trans):
        x = self.encoder(x)
        bev = self._bev(x, rots, trans, intrins, post_rots, post_trans)
        bev_post = self.embeder_bev(self.bevpost(bev[:, :, 60:140, 56:144]))
        return bev, self.predictor_bev1(bev_post), self.predictor_bev2(bev_post)


    def forward_loss(self, x, rots, trans, intrins, post_rots, post_trans, binimgs, act_gt, desc_gt):
        """`MultiLoss(*self(x, ...), binimgs, act_gt, desc_gt)` with the BEV head + weighted cross-entropy fused
        (see `model_BEV_TXT.BEV_TXT.forward_loss`); the heads' crop logits stay differentiable (ref :283)."""
        x = self.encoder(x)
        y = self.bevencode.features(self._train_voxels(x, rots, trans, intrins, post_rots, post_trans))
        head = self.bevencode.up2[4]
        loss_bev = head_weighted_cross_entropy(y, head, binimgs, _bev_class_weights(y.device))
        bev_post = self.embeder_bev(self.bevpost(head(y[:, :, 60:140, 56:144].float())))
        F = torch.nn.functional
        w1 = torch.tensor([1.0, 5.0, 5.0, 5.0], device=y.device)
        w2 = torch.tensor([1.0, 5.0, 5.0, 5.0, 1.0, 1.0, 1.0, 1.0], device=y.device)
        return (loss_bev + F.binary_cross_entropy_with_logits(self.predictor_bev1(bev_post), act_gt, weight=w1)
                + F.binary_cross_entropy_with_logits(self.predictor_bev2(bev_post), desc_gt, weight=w2))


def compile_model_onlybev(bsize, grid_conf, data_aug_conf, outC, **kw):
    return BEV_TXT(bsize, grid_conf, data_aug_conf, outC, **kw)
