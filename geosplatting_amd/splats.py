"""Container for raw (pre-activation) Gaussian parameters: field layout of rfstudio.graphics.Splats
(rfstudio/graphics/_splats.py:18-32) -- log-scales, logit-opacities, wxyz quaternions."""
from __future__ import annotations

from dataclasses import dataclass

from torch import Tensor


@dataclass
class SplatSet:
    means: Tensor       # [N,3]
    scales: Tensor      # [N,3] log
    quats: Tensor       # [N,4] wxyz, not normalised
    opacities: Tensor   # [N,1] logit
    colors: Tensor      # [N,3]

    def to(self, device):
        return SplatSet(*(t.to(device) for t in (self.means, self.scales, self.quats, self.opacities, self.colors)))

    @property
    def num(self) -> int:
        return self.means.shape[0]
