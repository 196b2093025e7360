"""The step after the renderer in the reference's training loop (run_mvtn.py:186-187, 244-245):
`regualarize_rendered_views` (ops.py:168-176) — view dropout + batch-wise flip / replicate-pad / random crop
(`applied_transforms`, ops.py:138-146; util.py:168-242) — as one fused HIP pass per direction
(`mvr_regularize_views_forward/backward`, include/mvr.h).

Random draws are made here with torch's generators, in the reference's order and on the reference's devices
(the keep mask on the GPU generator; flips and crop offsets on the CPU generator), so a seeded run picks the
same mask, flips and windows as the reference code would; the kernels only apply them."""
import torch

from . import _lib
from .functional import _f32c, _require_cuda, _stream


def crop_padding(image_size, crop_ratio):
    """ops.py:143: ReplicationPad2d(int((1 + crop_ratio) * H) - H)."""
    return int((1 + crop_ratio) * image_size) - image_size


def draw_regularization(batch, nb_views, image_size, device, dropout_p=0.0, augment_training=False, crop_ratio=0.3):
    """The random choices of one `regualarize_rendered_views` call, per image in MEMORY order n = b*V + v:
    (scale (N,) f32 or None, flip (N,) u8 or None, crop_i, crop_j (N,) i32 or None, pad)."""
    B, V, N = batch, nb_views, batch * nb_views
    scale = None
    if dropout_p > 0:
        # dropout2d on the 5-D tensor = feature dropout with a (B,V,1,1,1) mask (ops.py:170-171); drawing it on a ones
        # tensor of that shape consumes the device generator exactly as the reference's call does
        scale = torch._VF.feature_dropout(torch.ones(B, V, 1, 1, 1, dtype=torch.float32, device=device), float(dropout_p), True).reshape(N)
    flip = ci = cj = None
    pad = 0
    if augment_training:
        pad = crop_padding(image_size, crop_ratio)
        # super_batched_op(1, ...) (ops.py:149-153) flattens (B,V) view-major: image m = v*B + b gets the m-th draw
        flipped = torch.rand(N) < 0.5                               # util.py:193 (CPU generator)
        if pad > 0:                                                 # util.py:224-231
            i = torch.randint(0, 2 * pad + 1, (N,))
            j = torch.randint(0, 2 * pad + 1, (N,))
        else:
            i = j = torch.zeros(N, dtype=torch.int64)
        host = torch.stack([flipped.to(torch.int32), i.to(torch.int32), j.to(torch.int32)])          # (3, V*B)
        dev = host.view(3, V, B).transpose(1, 2).contiguous().view(3, N).to(device, non_blocking=True)    # -> n = b*V + v
        flip, ci, cj = dev[0].to(torch.uint8), dev[1], dev[2]
    return scale, flip, ci, cj, pad


def _cfloats(vals, n):
    import ctypes
    if vals is None:
        return None
    vals = [float(v) for v in vals]
    if len(vals) == 1:
        vals = vals * n
    if len(vals) != n:
        raise ValueError("mean / std need one value per channel (%d)" % n)
    return (ctypes.c_float * n)(*vals)


class _RegularizeViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, scale, flip, ci, cj, pad, mean=None, std=None):
        x = _f32c(images)
        _require_cuda(x, scale, flip, ci, cj)
        n, (c, h, w) = x.numel() // (x.shape[-3] * x.shape[-2] * x.shape[-1]), x.shape[-3:]
        out = torch.empty_like(x)
        cm, cs = _cfloats(mean, c), _cfloats(std, c)
        if cm is None:
            _lib.check(_lib.load().mvr_regularize_views_forward(x.data_ptr(), n, c, h, w, _p(scale), _p(flip), _p(ci), _p(cj), int(pad),
                                                                out.data_ptr(), _stream()), "mvr_regularize_views_forward")
        else:
            _lib.check(_lib.load().mvr_regularize_normalize_views_forward(x.data_ptr(), n, c, h, w, _p(scale), _p(flip), _p(ci), _p(cj),
                                                                          int(pad), cm, cs, out.data_ptr(), _stream()),
                       "mvr_regularize_normalize_views_forward")
        ctx.draws = (scale, flip, ci, cj, int(pad), n, c, h, w, cm, cs)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        scale, flip, ci, cj, pad, n, c, h, w, cm, cs = ctx.draws
        g = _f32c(grad_out)
        gin = torch.empty_like(g)
        if cm is None:
            _lib.check(_lib.load().mvr_regularize_views_backward(g.data_ptr(), n, c, h, w, _p(scale), _p(flip), _p(ci), _p(cj), pad,
                                                                 gin.data_ptr(), _stream()), "mvr_regularize_views_backward")
        else:
            _lib.check(_lib.load().mvr_regularize_normalize_views_backward(g.data_ptr(), n, c, h, w, _p(scale), _p(flip), _p(ci), _p(cj),
                                                                           pad, cm, cs, gin.data_ptr(), _stream()),
                       "mvr_regularize_normalize_views_backward")
        return gin, None, None, None, None, None, None, None


def _p(t):
    return None if t is None else t.data_ptr()


def apply_regularization(rendered_images, scale=None, flip=None, crop_i=None, crop_j=None, pad=0, mean=None, std=None):
    """Apply given draws (see `draw_regularization`) to (..., C, H, W) images; differentiable w.r.t. the images.
    `mean` / `std` (per channel, or one value for all): the fused `(x - mean) / std` of `normalize_views`."""
    if scale is None and flip is None and crop_i is None and mean is None:
        return rendered_images
    return _RegularizeViews.apply(rendered_images, scale, flip, crop_i, crop_j, pad, mean, std)


IMAGENET_MEAN, IMAGENET_STD = (0.456, 0.456, 0.456), (0.225, 0.225, 0.225)   # viewGCN/tools/Trainer_mvt.py:45,47


def mvctosvc(x):
    """ops.py:283 rearrange(x, 'b m c h w -> (b m) c h w'): a view of the contiguous block the renderer returns."""
    return x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))


def normalize_views(x, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """viewGCN/tools/Trainer_mvt.py:41-49 `Normalize`: (x - mean) / std per channel on (N, C, H, W) (or (B, V, C, H, W))
    images, one fused pass; differentiable."""
    return apply_regularization(x, mean=mean, std=std)


def regularize_normalize_views(rendered_images, dropout_p=0, augment_training=False, crop_ratio=0.3,
                               mean=IMAGENET_MEAN, std=IMAGENET_STD, flatten=True):
    """The three steps between the renderer and the classifier in the reference's training loop as ONE pass over the
    frames: regualarize_rendered_views (run_mvtn.py:186-187; ops.py:168-176) -> mvctosvc (run_mvtn.py:249; ops.py:283) ->
    Normalize (viewGCN/tools/Trainer_mvt.py:41-49).  Returns (B V, C, H, W) (`flatten`) or (B, V, C, H, W)."""
    if rendered_images.dim() != 5:
        raise ValueError("expected (B, V, C, H, W) rendered images")
    B, V, _, H, W = rendered_images.shape
    if augment_training and H != W:
        raise ValueError("applied_transforms crops to (H, H): square images only")
    draws = draw_regularization(B, V, H, rendered_images.device, dropout_p, augment_training, crop_ratio)
    out = apply_regularization(rendered_images, *draws, mean=mean, std=std)
    return mvctosvc(out) if flatten else out


def regualarize_rendered_views(rendered_images, dropout_p=0, augment_training=False, crop_ratio=0.3):
    """Drop-in for ops.py:168-176 (the reference's spelling): (B,V,C,H,W) -> (B,V,C,H,W)."""
    if rendered_images.dim() != 5:
        raise ValueError("expected (B, V, C, H, W) rendered images")
    B, V, _, H, W = rendered_images.shape
    if augment_training and H != W:
        raise ValueError("applied_transforms crops to (H, H): square images only")
    draws = draw_regularization(B, V, H, rendered_images.device, dropout_p, augment_training, crop_ratio)
    return apply_regularization(rendered_images, *draws)


regularize_rendered_views = regualarize_rendered_views
