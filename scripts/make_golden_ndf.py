#!/usr/bin/env python3
"""Golden vectors for the lobe cut-off cosines of the split-sum prefilter, produced by the reference's OWN
`__ndfBounds` (rfstudio/graphics/_mesh/_splitsum/_wrap.py:120-135): the module is loaded from its file (the package
`__init__` would JIT-compile CUDA), `_get_plugin` is replaced by a stub whose `specular_bounds` only records its
arguments -- everything `__ndfBounds` computes itself (the float64 NumPy CDF over 10^6 angles, the arg-max) is the
reference's code running unchanged.

    cd /tmp && python /root/repo/scripts/make_golden_ndf.py        (build container only)

Writes tests/golden/ref_ndf_cutoff.npz: (res, roughness, cutoff) -> cos(theta_cutoff) exactly as passed to the plugin.
"""
import importlib.util
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SRC = "/root/reference/rfstudio/graphics/_mesh/_splitsum/_wrap.py"

spec = importlib.util.spec_from_file_location("ref_splitsum_wrap", SRC)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

seen = []


class _Plugin:
    @staticmethod
    def specular_bounds(res, costheta_cutoff, index):
        seen.append((res, float(costheta_cutoff), index))
        return None


mod._get_plugin = lambda: _Plugin
ndf_bounds = getattr(mod, "__ndfBounds")

rows = []
# the six levels of TextureCubeMap.as_splitsum for a 512^2 map (rfstudio/graphics/_mesh/_texture.py:530-557) ...
n = 6
cases = [(512 >> idx, (idx / (n - 2)) * (0.5 - 0.08) + 0.08, 0.99) for idx in range(n - 1)] + [(16, 1.0, 0.99)]
# ... the levels of the 64^2 / 128^2 test maps, and a few other (roughness, cutoff) pairs
for R0 in (64, 128):
    m = 1
    while (R0 >> (m - 1)) > 16:
        m += 1
    cases += [(R0 >> idx, (idx / (m - 2)) * (0.5 - 0.08) + 0.08, 0.99) for idx in range(m - 1)]
cases += [(32, 0.3, 0.95), (32, 0.05, 0.99), (32, 0.7, 0.999), (32, 1.0, 0.5)]
for res, rough, cutoff in cases:
    ct, _ = ndf_bounds(res, rough, cutoff, 0)
    assert seen[-1][0] == res and seen[-1][1] == float(ct)
    rows.append((res, rough, cutoff, float(ct)))
a = np.array(rows, dtype=np.float64)
np.savez_compressed(os.path.join(OUT, "ref_ndf_cutoff.npz"), res=a[:, 0].astype(np.int32), roughness=a[:, 1], cutoff=a[:, 2],
                    costheta=a[:, 3])
for r in rows:
    print(r)
