"""CPU restatement of the FlexiCubes dual-marching-cubes extraction -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; geosplatting_amd never does.

Restates `FlexiCubes.dual_marching_cubes` with `grad_func=None` (rfstudio/graphics/_mesh/_flexicubes.py:559-713), its
helpers `_get_case_id` (:459-506), `_identify_surf_edges` (:508-538), `_linear_interp` (:540-557), `_compute_reg_loss`
(:727-741), `_triangulate` (:743-802), `compute_entropy` (:715-725) and the grid of `from_resolution` (:397-457), in
torch of any dtype (float64 for gradient checks; autograd supplies the reference gradients).

The 256-case tables (`_flexicubes.py:36-366`) are NOT taken from the reference: `dmc_patches` / `check_entry` below
derive them from the cube's corner/edge incidence by rule, and scripts/make_golden_flexicubes.py asserts (in the build
container, where the reference is importable) that the derived tables equal the reference's entry for entry.
PINNED by tests/golden/ref_flexicubes.npz, which that script produced by executing the reference extraction itself.

Index formulation (differs from the reference's torch.unique / sort pipeline, produces the same numbering):
  * grid vertex  (i0,i1,i2) -> id  i0 + (R0+1) * (i1 + (R1+1) * i2)   (corner bit 0 moves i0; `from_resolution` :443)
  * a grid edge is keyed (first vertex f, kind): kind 0 = (f, f - s1) [cube edges 8..11, stored high-to-low], kind 1 =
    (f, f + 1) [edges 0,2,4,6], kind 2 = (f, f + s2) [edges 1,3,5,7]; the reference's `unique(dim=0)` order of edge rows
    is exactly ascending slot = 3 f + kind, so the rank of a surface edge is a prefix count over the slots;
  * dual vertices are numbered by (number of patches of the cube, cube id, patch) (:648-664);
  * quads by (flip class, edge rank) (:760-772), the four cubes around an edge in ascending cube id.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

CUBE_EDGES = ((0, 1), (1, 5), (4, 5), (0, 4), (2, 3), (3, 7), (6, 7), (2, 6), (2, 0), (3, 1), (7, 5), (6, 4))   # :18-33


# ------------------------------------------------------------------------------------------------ derived tables
def _components(corners: set) -> List[List[int]]:
    left, out = set(corners), []
    while left:
        seed = min(left)
        comp, frontier = {seed}, [seed]
        while frontier:
            c = frontier.pop()
            for a, b in CUBE_EDGES:
                for p, q in ((a, b), (b, a)):
                    if p == c and q in left and q not in comp:
                        comp.add(q); frontier.append(q)
        left -= comp
        out.append(sorted(comp))
    return out


def dmc_patches(case: int) -> List[List[int]]:
    """Edge groups (one dual vertex each) of a corner-occupancy case.  One patch per connected component of the
    occupied corners, components by their lowest corner, edges ascending -- except the four 'tunnel' cases whose two
    empty corners are opposite ends of a body diagonal, which take the components of the EMPTY corners instead."""
    occ = {c for c in range(8) if case >> c & 1}
    emp = set(range(8)) - occ
    side = occ
    if len(emp) == 2 and sum(emp) == 7 and (min(emp) ^ max(emp)) == 7:
        side = emp
    patches = []
    for comp in _components(side):
        edges = [i for i, (a, b) in enumerate(CUBE_EDGES) if (a in comp) != (b in comp)]
        if edges:
            patches.append(edges)
    return patches


_FACES = [((axis, val), [c for c in range(8) if (c >> axis & 1) == val]) for axis in range(3) for val in (0, 1)]


def check_entry(case: int) -> Optional[Tuple[int, int, int, int]]:
    """(d0, d1, d2, inverted case) for the cases whose 2 or 3 EMPTY corners put a checkerboard on exactly one face
    (`_flexicubes.py:36-78`): the neighbour across that face decides whether both cubes flip to the complement case."""
    emp = [c for c in range(8) if not case >> c & 1]
    if len(emp) not in (2, 3):
        return None
    amb = []
    for (axis, val), fc in _FACES:
        e = [c for c in fc if c in emp]
        if len(e) == 2 and (e[0] ^ e[1]) == (7 ^ (1 << axis)):
            amb.append((axis, val))
    if len(amb) != 1:
        return None
    axis, val = amb[0]
    d = [0, 0, 0]
    d[axis] = 1 if val else -1
    return d[0], d[1], d[2], 255 - case


def tables():
    """(patch edge lists [256][<=4][<=7], num patches [256], check [256,5] as the reference lays it out)"""
    dmc = [dmc_patches(c) for c in range(256)]
    chk = np.zeros((256, 5), dtype=np.int64)
    for c in range(256):
        e = check_entry(c)
        if e is not None:
            chk[c] = (1,) + e
    return dmc, np.array([len(p) for p in dmc], dtype=np.int64), chk


# ------------------------------------------------------------------------------------------------ grid
def grid(res: Tuple[int, int, int], scale: float = 1.0, dtype=torch.float32):
    """`from_resolution` (:397-457): vertices [(R0+1)(R1+1)(R2+1), 3] and cube corner indices [R0 R1 R2, 8]."""
    R0, R1, R2 = res
    a, b, c = np.meshgrid(np.arange(R0 + 1), np.arange(R1 + 1), np.arange(R2 + 1), indexing="ij")
    coords = torch.from_numpy(np.stack((a, b, c), -1).reshape(-1, 3)).float()
    verts = coords / torch.tensor([R0, R1, R2], dtype=torch.long)
    n = np.arange(R0 * R1 * R2)
    base = np.stack((n % R0, (n // R0) % R1, n // (R0 * R1)), -1)[:, None, :]
    corners = np.array([[k & 1, k >> 1 & 1, k >> 2 & 1] for k in range(8)])[None]
    cc = base + corners
    idx = (cc[..., 2] * (1 + R1) + cc[..., 1]) * (1 + R0) + cc[..., 0]
    return ((2 * verts - 1) * scale).to(dtype), torch.from_numpy(idx)


def _interp(sa, sb, xa, xb, sdf_eps):
    w = sa / (sa - sb)                                                   # :551-556
    if sdf_eps is not None:
        w = (1 - sdf_eps) * w + sdf_eps / 2
    return xb * w + xa * (1 - w)


def extract(vertices: torch.Tensor, sdf: torch.Tensor, res, alpha=None, beta=None, gamma=None, *, weight_scale=0.99,
            sdf_eps=None, return_aux: bool = False):
    """vertices [Vg,3], sdf [Vg], raw alpha [C,8] / beta [C,12] / gamma [C,1] (or None) ->
    (mesh vertices [Q+nq,3], faces [4 nq,3] int64, L_dev [K])"""
    R0, R1, R2 = (int(r) for r in res)
    s1, s2 = R0 + 1, (R0 + 1) * (R1 + 1)
    Vg, C = s2 * (R2 + 1), R0 * R1 * R2
    assert vertices.shape == (Vg, 3) and sdf.numel() == Vg
    sdf = sdf.reshape(-1)
    dmc, nvd_t, chk = tables()
    occ = (sdf.detach() < 0).numpy()
    n = np.arange(C)
    origin = (n % R0) + s1 * ((n // R0) % R1) + s2 * (n // (R0 * R1))
    corner_off = np.array([(k & 1) + s1 * (k >> 1 & 1) + s2 * (k >> 2 & 1) for k in range(8)])
    cube_v = origin[:, None] + corner_off[None]                          # [C,8]
    raw = (occ[cube_v] << np.arange(8)).sum(-1)
    surf = (raw > 0) & (raw < 255)
    assert surf.any()
    # ambiguity resolution (:459-506): positions are the C-order unravel of the cube id over `res`
    case = raw.copy()
    prob = surf & (chk[raw, 0] == 1)
    pid = n[prob]
    u = np.stack(np.unravel_index(pid, (R0, R1, R2)), -1)
    adj = u + chk[raw[pid], 1:4]
    ok = (adj >= 0).all(-1) & (adj < np.array([R0, R1, R2])).all(-1)
    adj_id = np.ravel_multi_index(tuple(adj[ok].T), (R0, R1, R2)) if ok.any() else np.zeros(0, dtype=np.int64)
    flip = prob[adj_id]
    case[pid[ok][flip]] = chk[raw[pid[ok][flip]], 4]

    sid = n[surf]                                                        # surface cubes, ascending
    scase = case[sid]
    nvd = nvd_t[scase]
    # grid-edge slots
    edge_first = np.array([max(a, b) if e >= 8 else min(a, b) for e, (a, b) in enumerate(CUBE_EDGES)])
    edge_kind = np.array([0 if e >= 8 else (1 if (a ^ b) == 1 else 2) for e, (a, b) in enumerate(CUBE_EDGES)])
    slot_of = lambda cubes: 3 * (origin[cubes][:, None] + corner_off[edge_first][None]) + edge_kind[None]   # [.,12]
    f_all = np.arange(Vg)
    second = np.stack((f_all - s1, f_all + 1, f_all + s2), -1)            # [Vg,3]
    i0, i1, i2 = f_all % s1, (f_all // s1) % (R1 + 1), f_all // s2
    exists = np.stack((i1 >= 1, i0 < R0, i2 < R2), -1)
    sec_c = np.where(exists, second, 0)
    cross = exists & (occ[f_all][:, None] != occ[sec_c])
    cross_f = cross.reshape(-1)
    rank = np.cumsum(cross_f) - cross_f                                   # rank of a surface edge among surface edges
    E = int(cross_f.sum())
    e_first = np.repeat(f_all, 3)[cross_f]; e_second = second.reshape(-1)[cross_f]

    # dual vertices in reference order (:648-664)
    xs_a, xs_b = vertices[e_first], vertices[e_second]
    s_a, s_b = sdf[e_first], sdf[e_second]
    zero_cross = _interp(s_a[:, None], s_b[:, None], xs_a, xs_b, sdf_eps)

    k_edge, k_vd, k_cube, k_slot, vd_cube, vd_cnt = [], [], [], [], [], []
    vd_base = np.zeros(sid.shape[0], dtype=np.int64)
    total = 0
    for num in range(1, 5):
        sel = np.nonzero(nvd == num)[0]
        if sel.size == 0:
            continue
        vd_base[sel] = total + num * np.arange(sel.size)
        for j, s in enumerate(sel):
            for p, edges in enumerate(dmc[scase[s]]):
                v = total + num * j + p
                vd_cube.append(s); vd_cnt.append(len(edges))
                for e in edges:
                    k_edge.append(e); k_vd.append(v); k_cube.append(s)
        total += num * sel.size
    Q = total
    k_edge, k_vd, k_cube = (np.asarray(a, dtype=np.int64) for a in (k_edge, k_vd, k_cube))
    vd_cube, vd_cnt = np.asarray(vd_cube, dtype=np.int64), np.asarray(vd_cnt, dtype=np.int64)
    k_slot = slot_of(sid[k_cube])[np.arange(k_edge.size), k_edge]
    assert cross_f[k_slot].all()
    k_rank = rank[k_slot]
    dt = vertices.dtype
    if alpha is not None:
        a_act = torch.tanh(alpha[sid]) * weight_scale + 1                 # [N,8]
        ce = np.array(CUBE_EDGES)
        al_a = a_act[k_cube, ce[k_edge, 0]]; al_b = a_act[k_cube, ce[k_edge, 1]]
        ue = _interp((s_a[k_rank] * al_a)[:, None], (s_b[k_rank] * al_b)[:, None], xs_a[k_rank], xs_b[k_rank], sdf_eps)
    else:
        ue = zero_cross[k_rank]
    if beta is not None:
        b_act = torch.tanh(beta[sid]) * weight_scale + 1
        bk = b_act[k_cube, k_edge][:, None]
    else:
        bk = torch.ones(k_edge.size, 1, dtype=dt)
    g_act = (torch.sigmoid(gamma[sid]) * weight_scale + (1 - weight_scale) / 2).reshape(-1) if gamma is not None \
        else torch.ones(sid.size, dtype=dt)
    k_vd_t = torch.from_numpy(k_vd)
    bsum = torch.zeros(Q, 1, dtype=dt).index_add_(0, k_vd_t, bk)
    vd = torch.zeros(Q, 3, dtype=dt).index_add_(0, k_vd_t, ue * bk) / bsum
    dist = (zero_cross[k_rank] - vd[k_vd_t]).norm(dim=-1)                 # :727-741
    mean = torch.zeros(Q, dtype=dt).index_add_(0, k_vd_t, dist) / torch.from_numpy(vd_cnt).to(dt)
    L_dev = (dist - mean[k_vd_t]).abs()
    vd_gamma = g_act[vd_cube]

    # which dual vertex owns (surface cube, cube edge)
    owner = np.zeros((sid.size, 12), dtype=np.int64)
    owner[k_cube, k_edge] = k_vd
    surf_pos = np.full(C, -1, dtype=np.int64); surf_pos[sid] = np.arange(sid.size)

    # quads (:752-772): interior surface edges; the 4 cubes around the edge in ascending cube id
    ef, ek = np.repeat(f_all, 3)[cross_f], np.tile(np.arange(3), Vg)[cross_f]
    j0, j1, j2 = ef % s1, (ef // s1) % (R1 + 1), ef // s2
    interior = np.where(ek == 0, (j0 >= 1) & (j0 < R0) & (j2 >= 1) & (j2 < R2),
                        np.where(ek == 1, (j1 >= 1) & (j1 < R1) & (j2 >= 1) & (j2 < R2),
                                 (j0 >= 1) & (j0 < R0) & (j1 >= 1) & (j1 < R1)))
    eid = np.nonzero(interior)[0]
    # lower vertex of the edge and the cubes touching it
    lo = np.where(ek == 0, ef - s1, ef)
    l0, l1, l2 = lo % s1, (lo // s1) % (R1 + 1), lo // s2
    cid = lambda x, y, z: x + R0 * (y + R1 * z)
    quad_vd = np.zeros((eid.size, 4), dtype=np.int64)
    for q, e in enumerate(eid):
        x, y, z, k = l0[e], l1[e], l2[e], ek[e]
        if k == 1:      # along i0: cubes (x, y-1..y, z-1..z)
            cubes = [cid(x, y - 1, z - 1), cid(x, y, z - 1), cid(x, y - 1, z), cid(x, y, z)]
        elif k == 0:    # along i1: cubes (x-1..x, y, z-1..z)
            cubes = [cid(x - 1, y, z - 1), cid(x, y, z - 1), cid(x - 1, y, z), cid(x, y, z)]
        else:           # along i2: cubes (x-1..x, y-1..y, z)
            cubes = [cid(x - 1, y - 1, z), cid(x, y - 1, z), cid(x - 1, y, z), cid(x, y, z)]
        slot = 3 * ef[e] + k
        for t, c in enumerate(cubes):
            sp = surf_pos[c]
            assert sp >= 0
            ce_slots = slot_of(np.array([c]))[0]
            ee = int(np.nonzero(ce_slots == slot)[0][0])
            quad_vd[q, t] = owner[sp, ee]
    flipm = (sdf.detach()[torch.from_numpy(ef[eid])] > 0).numpy()
    quad_vd = np.concatenate((quad_vd[flipm][:, [0, 1, 3, 2]], quad_vd[~flipm][:, [2, 3, 1, 0]]))
    qt = torch.from_numpy(quad_vd)
    qg = vd_gamma[qt.reshape(-1)].reshape(-1, 4)
    g02 = (qg[:, 0] * qg[:, 2])[:, None]; g13 = (qg[:, 1] * qg[:, 3])[:, None]
    vq = vd[qt.reshape(-1)].reshape(-1, 4, 3)
    v02 = (vq[:, 0] + vq[:, 2]) / 2; v13 = (vq[:, 1] + vq[:, 3]) / 2
    centre = (v02 * g02 + v13 * g13) / ((g02 + g13) + 1e-8)
    nq = quad_vd.shape[0]
    cidx = torch.arange(nq) + Q
    faces = torch.stack((qt[:, [0, 1, 2, 3]], qt[:, [1, 2, 3, 0]], cidx[:, None].expand(nq, 4)), -1).reshape(-1, 3)
    out = (torch.cat((vd, centre)), faces, L_dev)
    if return_aux:
        return out + (dict(case=scase, surf_cubes=sid, num_vd=nvd, Q=Q, E=E, num_quads=nq),)
    return out


def entropy(sdf: torch.Tensor, res) -> torch.Tensor:
    """`compute_entropy` (:715-725): BCE-with-logits across every sign-changing grid edge, both directions."""
    R0, R1, R2 = (int(r) for r in res)
    s1, s2 = R0 + 1, (R0 + 1) * (R1 + 1)
    Vg = s2 * (R2 + 1)
    sdf = sdf.reshape(-1)
    f = np.arange(Vg)
    second = np.stack((f - s1, f + 1, f + s2), -1)
    i0, i1, i2 = f % s1, (f // s1) % (R1 + 1), f // s2
    exists = np.stack((i1 >= 1, i0 < R0, i2 < R2), -1)
    occ = (sdf.detach() < 0).numpy()
    cross = (exists & (occ[f][:, None] != occ[np.where(exists, second, 0)])).reshape(-1)
    a = sdf[torch.from_numpy(np.repeat(f, 3)[cross])]; b = sdf[torch.from_numpy(second.reshape(-1)[cross])]
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    return bce(a, (b > 0).to(sdf.dtype)) + bce(b, (a > 0).to(sdf.dtype))
