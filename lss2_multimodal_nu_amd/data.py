"""Host-side input path of the camera->BEV model: the calibration half of the
reference's data augmentation (`src/tools.py:110-142` `get_rot`, `img_transform`;
`src/data.py:90-112` `sample_augmentation`; `:114-159` `get_image_data`) and a loader-side
`CalibrationPack` that moves the per-step host linear algebra of the model - the two 3x3
inverses per camera that the exact-index contract keeps on the host - out of the training
step and into the DataLoader workers, already packed in ONE pinned buffer for a single H2D copy.

    pack = prepare_calibration(rots, trans, intrins, post_rots, post_trans)   # in the worker / collate_fn
    bev = model(x, pack, None, None, None, None)                               # same forward(), 5 tensors in one

No dataset code here (nuScenes access stays in the reference's `src/data.py`).
"""
import numpy as np
import torch
import torch.utils.data


def get_rot(h):
    """2x2 rotation used by `img_transform` (ref: src/tools.py:110-114)."""
    return torch.Tensor([[np.cos(h), np.sin(h)], [-np.sin(h), np.cos(h)]])


def sample_augmentation(data_aug_conf, is_train, rng=np.random):
    """(resize, resize_dims, crop, flip, rotate) as the reference draws them (ref: src/data.py:90-112):
    random resize / crop / flip / rotation in training, the fixed centre crop in validation."""
    H, W = data_aug_conf["H"], data_aug_conf["W"]
    fH, fW = data_aug_conf["final_dim"]
    if is_train:
        resize = rng.uniform(*data_aug_conf["resize_lim"])
        resize_dims = (int(W * resize), int(H * resize))
        newW, newH = resize_dims
        crop_h = int((1 - rng.uniform(*data_aug_conf["bot_pct_lim"])) * newH) - fH
        crop_w = int(rng.uniform(0, max(0, newW - fW)))
        flip = bool(data_aug_conf["rand_flip"] and rng.choice([0, 1]))
        rotate = rng.uniform(*data_aug_conf["rot_lim"])
    else:
        resize = max(fH / H, fW / W)
        resize_dims = (int(W * resize), int(H * resize))
        newW, newH = resize_dims
        crop_h = int((1 - np.mean(data_aug_conf["bot_pct_lim"])) * newH) - fH
        crop_w = int(max(0, newW - fW) / 2)
        flip, rotate = False, 0
    crop = (crop_w, crop_h, crop_w + fW, crop_h + fH)
    return resize, resize_dims, crop, flip, rotate


def img_transform(img, post_rot, post_tran, resize, resize_dims, crop, flip, rotate):
    """Apply the augmentation to a PIL image (or pass `img=None`) and compose its 2-D homography
    into (post_rot 2x2, post_tran 2) - same argument order and result as ref src/tools.py:117-142."""
    if img is not None:
        from PIL import Image
        img = img.resize(resize_dims).crop(crop)
        if flip:
            img = img.transpose(method=Image.FLIP_LEFT_RIGHT)
        img = img.rotate(rotate)
    post_rot = post_rot * resize
    post_tran = post_tran - torch.Tensor(crop[:2])
    if flip:
        A = torch.Tensor([[-1, 0], [0, 1]])
        b = torch.Tensor([crop[2] - crop[0], 0])
        post_rot = A.matmul(post_rot)
        post_tran = A.matmul(post_tran) + b
    A = get_rot(rotate / 180 * np.pi)
    b = torch.Tensor([crop[2] - crop[0], crop[3] - crop[1]]) / 2
    b = A.matmul(-b) + b
    return img, A.matmul(post_rot), A.matmul(post_tran) + b


def augmentation_matrices(post_rot2, post_tran2):
    """2x2 / 2-vector homography -> the 3x3 `post_rot` and 3-vector `post_tran` the model takes
    (ref: src/data.py:143-147)."""
    post_rot, post_tran = torch.eye(3), torch.zeros(3)
    post_rot[:2, :2] = post_rot2
    post_tran[:2] = post_tran2
    return post_rot, post_tran


def calib_matrices(rots, intrins, post_rots):
    """inv(post_rots) and rots @ inv(intrins) on the HOST (exact-index contract):
    one flattened (2*B*N, 3, 3) LAPACK call, bitwise the matrices of the reference's two 4-D calls."""
    r, i, p = (t.detach().float().cpu().reshape(-1, 3, 3) for t in (rots, intrins, post_rots))
    n = p.shape[0]
    inv = torch.inverse(torch.cat([p, i]))
    shape = tuple(rots.shape)
    return inv[:n].view(shape), torch.bmm(r, inv[n:]).view(shape)


class CalibrationPack:
    """The four per-camera quantities K3 consumes - inv(post_rots), rots @ inv(intrins), post_trans,
    trans - in one contiguous (optionally pinned) fp32 buffer, plus the batch shape."""

    __slots__ = ("buffer", "shape")

    def __init__(self, buffer, shape):
        self.buffer, self.shape = buffer, tuple(shape)  # shape = (B, N)

    def views(self, buf=None):
        """(inv_post_rots (B,N,3,3), combine (B,N,3,3), post_trans (B,N,3), trans (B,N,3)) as views."""
        buf = self.buffer if buf is None else buf
        B, N = self.shape
        n = B * N
        return (buf[:9 * n].view(B, N, 3, 3), buf[9 * n:18 * n].view(B, N, 3, 3), buf[18 * n:21 * n].view(B, N, 3),
                buf[21 * n:24 * n].view(B, N, 3))


def prepare_calibration(rots, trans, intrins, post_rots, post_trans, pin=False):
    """Loader-side half of `LSS.forward`'s calibration handling: do the host inverses once per sample
    batch (in a DataLoader worker or collate_fn) and pack everything for one H2D copy.
    Returns a `CalibrationPack` to pass as the model's `rots` argument.

    `pin` defaults to False: page-locking needs a HIP context, which a forked DataLoader worker must not
    create ("Cannot re-initialize CUDA in forked subprocess"), and pinning does not survive the worker ->
    main-process queue anyway.  Let `DataLoader(pin_memory=True)` pin the pack's buffer in the main process, or
    call this with pin=True there.  (With <= 36 cameras the buffer rides in the kernel arguments and is never
    copied at all: pinning is irrelevant on that path.)"""
    inv_pr, comb = calib_matrices(rots, intrins, post_rots)
    B, N = trans.shape[:2]
    buf = torch.cat([inv_pr.reshape(-1), comb.reshape(-1), post_trans.detach().float().cpu().reshape(-1),
                     trans.detach().float().cpu().reshape(-1)])
    if pin:
        if torch.utils.data.get_worker_info() is not None:
            raise RuntimeError("prepare_calibration(pin=True) inside a DataLoader worker: pin in the main process")
        buf = buf.pin_memory()
    return CalibrationPack(buf, (B, N))
