"""Host-side helpers on the renderer path, restated from the reference's util.py / ops.py
(only the functions models/renderer.py calls)."""
import torch


def torch_color(color_type, custom_color=(1.0, 0, 0), max_lightness=False, epsilon=0.00001):
    """util.py:314-334.  (3,) float32 HOST tensor.  "white" with max_lightness is 1/(1+1e-5) =
    0.99999 per channel.  The reference's "custom" branch is unreachable (it tests an unbound
    name and raises UnboundLocalError, util.py:330); here it returns `custom_color`."""
    named = {"white": (1.0, 1.0, 1.0), "red": (1.0, 0.0, 0.0), "green": (0.0, 1.0, 0.0),
             "blue": (0.0, 0.0, 1.0), "black": (0.0, 0.0, 0.0)}
    if color_type in named:
        color = torch.tensor(named[color_type])
    elif color_type == "random":
        color = torch.rand(3)
    elif color_type == "custom":
        color = torch.as_tensor(custom_color, dtype=torch.float32)
    else:
        raise ValueError("unknown color_type %r" % (color_type,))
    if max_lightness and color_type != "black":
        color = color / (torch.max(color) + epsilon)
    return color


def batch_tensor(tensor, dim=1, squeeze=False):
    """util.py:509-521: move `dim` into the batch dimension 0."""
    batch_size, dim_size = tensor.shape[0], tensor.shape[dim]
    returned_size = list(tensor.shape)
    returned_size[0] = batch_size * dim_size
    returned_size[dim] = 1
    out = tensor.transpose(0, dim).reshape(returned_size)
    return out.squeeze(dim) if squeeze else out


def unbatch_tensor(tensor, batch_size, dim=1, unsqueeze=False):
    """util.py:524-534: inverse of batch_tensor."""
    nb_chunks = int(tensor.shape[0] / batch_size)
    if unsqueeze:
        tensor = tensor.unsqueeze(dim)
    return torch.cat(torch.chunk(tensor, nb_chunks, dim=0), dim=dim).contiguous()


def flatten_views(x):
    """(B, V) -> (N,) with n = b*V + v: what batch_tensor(x.T, dim=1, squeeze=True) yields at
    models/renderer.py:79-80 (also for stride-(0,1) expanded selector outputs)."""
    return x.reshape(-1)


def check_valid_rotation_matrix(R, tol=1e-6):
    """util.py:403-420 (torch version, host-syncing; the renderer uses the fused device check
    of mvr_look_at instead)."""
    N = R.shape[0]
    eye = torch.eye(3, dtype=R.dtype, device=R.device).view(1, 3, 3).expand(N, -1, -1)
    orthogonal = torch.allclose(R.bmm(R.transpose(1, 2)), eye, atol=tol)
    det_R = torch.det(R)
    no_distortion = torch.allclose(det_R, torch.ones_like(det_R))
    return bool(orthogonal and no_distortion)
