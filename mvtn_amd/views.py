"""Non-learned view selectors — the step immediately BEFORE the renderer, restated
from models/mvtn.py:13-111 so a harness can run `selector -> MVRenderer` without the reference.  Same buffers,
same (B, V) stride-(0,1) expanded outputs, same training-time noise / distance jitter.  The learned selectors
(models/mvtn.py:114-407) need the point encoders and stay out of scope."""
import torch
from torch import nn

from .synthetic import unit_spherical_grid


class _FixedGridSelector(nn.Module):
    def __init__(self, views_azim, views_elev, nb_views, canonical_distance, transform_distance, input_view_noise):
        super().__init__()
        self.nb_views = nb_views
        self.transform_distance = transform_distance
        self.canonical_distance = canonical_distance
        self.input_view_noise = input_view_noise
        self.register_buffer("views_azim", views_azim.to(torch.float))
        self.register_buffer("views_elev", views_elev.to(torch.float))
        self.register_buffer("views_dist", torch.ones(nb_views, dtype=torch.float) * canonical_distance)

    def forward(self, shape_features=None, c_batch_size=1):
        azim = self.views_azim.expand(c_batch_size, self.nb_views)
        elev = self.views_elev.expand(c_batch_size, self.nb_views)
        dist = self.views_dist.expand(c_batch_size, self.nb_views)
        dist = dist + float(self.transform_distance) * 1.0 * dist * (
            torch.rand((c_batch_size, self.nb_views), device=dist.device) - 0.5)
        if self.input_view_noise > 0.0 and self.training:
            azim = azim + torch.normal(0.0, 180.0 * self.input_view_noise, azim.size(), device=azim.device)
            elev = elev + torch.normal(0.0, 90.0 * self.input_view_noise, elev.size(), device=elev.device)
            dist = dist + torch.normal(0.0, self.canonical_distance * self.input_view_noise, dist.size(), device=dist.device)
        return azim, elev, dist


class CircularViewSelector(_FixedGridSelector):
    """models/mvtn.py:13-45: V azimuths evenly spaced on a circle at one elevation."""

    def __init__(self, nb_views=12, canonical_elevation=35.0, canonical_distance=2.2, transform_distance=False,
                 input_view_noise=0.0):
        azim = torch.linspace(-180, 180, nb_views + 1)[:-1] - 90.0
        elev = torch.ones_like(azim) * canonical_elevation
        super().__init__(azim, elev, nb_views, canonical_distance, transform_distance, input_view_noise)


class SphericalViewSelector(_FixedGridSelector):
    """models/mvtn.py:48-80: V views on util.unit_spherical_grid."""

    def __init__(self, nb_views=12, canonical_distance=2.2, transform_distance=False, input_view_noise=0.0):
        azim, elev = unit_spherical_grid(nb_views)
        super().__init__(torch.from_numpy(azim), torch.from_numpy(elev), nb_views, canonical_distance,
                         transform_distance, input_view_noise)


class RandomViewSelector(nn.Module):
    """models/mvtn.py:83-111: azim ~ U[-180,180), elev ~ U[-90,90) per call."""

    def __init__(self, nb_views=12, canonical_distance=2.2, transform_distance=False):
        super().__init__()
        self.nb_views = nb_views
        self.transform_distance = transform_distance
        self.canonical_distance = canonical_distance
        self.register_buffer("views_dist", torch.ones(nb_views, dtype=torch.float) * canonical_distance)
        self.register_buffer("views_elev", torch.zeros(nb_views, dtype=torch.float))
        self.register_buffer("views_azim", torch.zeros(nb_views, dtype=torch.float))

    def forward(self, shape_features=None, c_batch_size=1):
        shape = (c_batch_size, self.nb_views)
        dev = self.views_azim.device
        azim = self.views_azim.expand(*shape) + torch.rand(shape, device=dev) * 360.0 - 180.0
        elev = self.views_elev.expand(*shape) + torch.rand(shape, device=dev) * 180.0 - 90.0
        dist = self.views_dist.expand(*shape)
        dist = dist + float(self.transform_distance) * 1.0 * dist * (torch.rand(shape, device=dev) - 0.499)
        return azim, elev, dist


def view_selector(views_config, nb_views, canonical_elevation=30.0, canonical_distance=2.2, transform_distance=False,
                  input_view_noise=0.0):
    """The non-learned branches of models/mvtn.py:298-324 ViewSelector ("custom" and spherical with exactly 4 views
    fall back to circular, as the reference does at :303)."""
    if views_config in ("circular", "custom") or (views_config == "spherical" and nb_views == 4):
        return CircularViewSelector(nb_views, canonical_elevation, canonical_distance, transform_distance, input_view_noise)
    if views_config == "spherical":
        return SphericalViewSelector(nb_views, canonical_distance, transform_distance, input_view_noise)
    if views_config == "random":
        return RandomViewSelector(nb_views, canonical_distance, transform_distance)
    raise NotImplementedError("learned view selectors are outside the renderer path (models/mvtn.py:114-407)")
