"""The render-and-backward step of the path for a batch of views -- the counterpart of
``GeoSplatter.render_report`` (rfstudio/model/geosplat.py:856-927: prefilter once, then a loop of
``attrs.splat`` over the views) followed by the autograd backward the trainer triggers
(rfstudio/optim/optimizer.py:107), with the data-parallel gradient sum of geosplatting_amd.parallel.

Order of work per step (all on the current HIP stream, kernels in libgeosplat_hip.so):
    as_splitsum(cubemap)                                   S5 forward   (once per step)
    for each local view:  shade -> rasterize -> tone-map    S1-S4, A1-A5
                          backward of the same              A6, A7, S1-S3 backward
    prefilter backward with the texel gradients summed over the views   S5 backward (once per step)
    flat-bucket all-reduce over ranks                       RCCL
"""
from __future__ import annotations

import collections
import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import Tensor

import ctypes as C

from . import _lib as L
from .cameras import Camera
from .parallel import GradBucket, collectives_active
from . import front as F
from .shading import _MODE, _TONE, RenderableAttrs, _make_env, get_fg_lut
from .splitsum import (TextureSplitSum, as_splitsum, as_splitsum_backward, as_splitsum_backward_sharded,
                       as_splitsum_sharded, can_shard_prefilter)
from .splats import SplatSet

PARAM_NAMES = ("means", "scales", "quats", "opacities", "normals", "kd", "ks", "cubemap", "exposure")


@dataclass
class PathParams:
    """Leaf tensors of the path (replicated on every rank)."""
    means: Tensor; scales: Tensor; quats: Tensor; opacities: Tensor       # raw Splats fields
    normals: Tensor; kd: Tensor; ks: Tensor                                 # RenderableAttrs
    cubemap: Tensor                                                         # [6,R,R,3]
    exposure: Tensor                                                        # scalar (already exp'ed)

    def named(self) -> Dict[str, Tensor]:
        return {k: getattr(self, k) for k in PARAM_NAMES}

    def shapes(self) -> Dict[str, Sequence[int]]:
        return {k: tuple(v.shape) if v.dim() > 0 else (1,) for k, v in self.named().items()}


def _auto_tail_schedule(n_views: int):
    """Views per tail launch of a step.  A background launch (half of the CUs) works a view off in about half the time the
    compositor needs for one, and the batch before the last has to be done when the last compositor backward ends: the LAST batch
    takes two views, the ones before it up to three (8 views: 3 + 3 + 2; 7: 3 + 2 + 2; 6: 2 + 2 + 2; 5: 3 + 2).  With four views or
    fewer (the strong split over 2 / 4 GPUs) the measured best ends on single views: 4: 2 + 1 + 1 (583 views/s against 571 for
    2 + 2, 552 for one launch), 3: 2 + 1 (519 / 499), 2: 1 + 1 (429 / 418)."""
    if n_views <= 4:
        return {0: [1], 1: [1], 2: [1, 1], 3: [2, 1], 4: [2, 1, 1]}[max(n_views, 0)]
    r, sched = n_views - 2, []
    while r > 0:
        k = 3 if (r >= 3 and r != 4) else min(r, 2)
        sched.append(k)
        r -= k
    return sched + [2]


class RenderStep:
    def __init__(self, params: PathParams, min_roughness: float = 0.1, max_metallic: float = 1.0, mode: str = "pbr",
                 tone_type: str = "naive", prefilter: bool = True, fused: bool = True, fg_lut: Optional[Tensor] = None):
        self.p = params
        self.fg_lut = None if fg_lut is None else fg_lut.to(params.means.device, torch.float32).contiguous()   # default: the packaged table
        self.min_roughness, self.max_metallic, self.mode, self.tone_type = min_roughness, max_metallic, mode, tone_type
        self.prefilter = prefilter
        self.fused = fused
        self.bucket = GradBucket(params.shapes(), params.means.device)
        self._static_env: Optional[TextureSplitSum] = None
        self._cam_cache: Dict[tuple, tuple] = {}
        self._side_stream = None
        self._tail_stream = None
        # capacity protocol state (see _step_fused / poll_capacity)
        self._use_capacity = os.environ.get("GEOSPLAT_CAPACITY", "1") != "0"
        self._key_lo = self._key_hi = None     # depth-bit range seen so far (asynchronous read-back): sizes the 24-bit binning keys
        self._key32 = os.environ.get("GEOSPLAT_KEY_BITS", "24") == "32"   # forced by a key-range overflow, or by the environment
        self._cap_margin = float(os.environ.get("GEOSPLAT_CAPACITY_MARGIN", "1.25"))   # capacity = margin x the largest count seen
        self._i_cap = None                 # intersection capacity per view; None = exact mode (one (V, I) read-back per view)
        self._status = None                # device int64[4]: {GS_ENOSPC or 0, max required I, max required V, depth outside the 24-bit key range}
        self._status_host = None           # pinned snapshot used by the graph-capture paths (one buffer, refreshed by every replay)
        self._status_event = None
        self._status_pending = collections.deque()   # eager steps: (pinned snapshot of the word, event), one per step, each looked at ONCE
        self._status_pool = []             # pinned int64[4] buffers waiting for reuse
        self.kernel_events = None          # set to a list: (name, start event, end event) around the compositor launches (bench.py)
        self.tail_events = None            # set to a list: (start, end of the shading half, end of the projection half) of the LAST tail launch
        self._seen_counts = []
        self._exact_max_i = 0              # largest intersection count read back by an exact-mode step
        self._pre_group = None
        self.truncated_steps = 0           # steps whose overflow word was seen set (each was composited from a truncated list)
        self.chunked_tails = 0             # steps whose last tail ran in Gaussian-range chunks under their all-reduce (_chunked_tail)
        self._overflow_unreported = False  # an overflow seen by the poll inside __call__ that no caller has been told about yet
        n_fly = int(os.environ.get("GEOSPLAT_STEPS_IN_FLIGHT", "3"))
        self._max_in_flight = max(1, n_fly)
        self._in_flight = collections.deque() if n_fly > 0 else None     # end-of-step events of the steps the GPU has not finished (None: unbounded)

    def _prefilter_group(self):
        """A second communicator for the prefilter exchange (25 MB all-reduce + two rounds of small all-gathers), so that it
        never queues behind the 149 MB per-Gaussian all-reduce on the default one.  Created collectively on first use."""
        import torch.distributed as dist
        if self._pre_group is None:
            self._pre_group = dist.new_group()
        return self._pre_group

    def rebind(self, params: PathParams) -> None:
        """New parameter tensors for the next step (a stage-1 loop extracts a different number of Gaussians every
        iteration): the gradient bucket is re-laid out when a shape changed; streams and camera tensors are kept."""
        self.p = params
        if {k: tuple(v) for k, v in params.shapes().items()} != self.bucket.shapes:
            self.bucket = GradBucket(params.shapes(), params.means.device, comm_stream=self.bucket.comm_stream)

    # ------------------------------------------------------------------------------------------------- fused path
    def _camera_tensors(self, cam: Camera):
        """Device copies of (view matrix, K, camera position) of a camera, cached per POSE TENSOR: the entry keeps a reference to
        `cam.c2w` (so CPython cannot recycle its id while the entry lives) together with the tensor's version counter (an
        in-place update of the pose bumps it) and the intrinsics -- a hit therefore costs no device-to-host copy and no
        synchronisation, also for device-resident cameras and inside a graph capture; a training loop that builds new Camera
        objects every step simply misses.  Bounded, oldest entry evicted."""
        c2w = cam.c2w
        key = id(c2w)
        intr = (float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), int(cam.width), int(cam.height))
        # host-resident poses also carry a content fingerprint (48 bytes): an in-place change that does not bump the version counter
        # (`cam.c2w.data.copy_()`, a numpy-backed tensor written through numpy) cannot return stale matrices.  Device-resident poses
        # rely on the version counter alone (a fingerprint would be a device-to-host copy); invalidate_cameras() drops everything.
        fp = c2w.detach().numpy().tobytes() if c2w.device.type == "cpu" else None
        hit = self._cam_cache.get(key)
        if hit is not None and hit[0] is c2w and hit[1] == c2w._version and hit[2] == intr and hit[4] == fp:
            return hit[3]
        dev = self.p.means.device
        if len(self._cam_cache) >= 1024:
            self._cam_cache.pop(next(iter(self._cam_cache)))
        tensors = (cam.view_matrix.to(dev, torch.float32).contiguous(), cam.intrinsic_matrix.to(dev, torch.float32).contiguous(),
                   c2w.detach()[:, 3].to(dev, torch.float32).contiguous())
        self._cam_cache[key] = (c2w, c2w._version, intr, tensors, fp)
        return tensors

    def invalidate_cameras(self) -> None:
        """Forget every cached (view matrix, K, position) triple: for callers that rewrite DEVICE-resident poses behind autograd's back."""
        self._cam_cache.clear()

    def _step_fused(self, cameras, upstream, all_reduce, keep_images, _env=None, _stop_after_views=False, _geo_only=False, _geo=None,
                    _rec_only=False, _rec=None):
        """Same arithmetic as the autograd path, driven directly through the C-ABI: every tail launch ADDS into the flat gradient
        bucket, the exp / sigmoid activations of GSplatter.render_rgba (rfstudio/model/gsplat.py:336-339) are applied once per step
        and chained once per step (their Jacobians do not depend on the view), and the prefilter backward runs once on the texel
        gradients summed over the views.

        HIP streams of a step: two FRONT streams run the memory / latency-bound front of every view (gs_front_fwd, the binning, the
        record stream: ~22 short kernels, none of which fills the chip), the fronts of consecutive views alternating between them
        (one: 638 views/s, two: 690, three: 655); the caller's stream runs the VALU-bound compositor forward / backward -- the step's
        critical path; a tail stream runs the gradient kernels of the views in batches.  In capacity mode nothing makes the host wait,
        so the fronts run as far ahead of the compositor as their inputs allow."""
        lib = L.lib()
        p = self.p
        dev = p.means.device
        N = p.means.shape[0]
        f32 = torch.float32
        st = L.stream
        mode, tone = _MODE[self.mode], _TONE[self.tone_type]
        cubemap = p.cubemap.detach()
        import torch.distributed as dist
        world = dist.get_world_size() if (all_reduce and dist.is_available() and dist.is_initialized()) else 1
        # S5 sharded over the ranks (each applies 1/world of every level's tiles, all-reduce of the disjoint pieces): splitsum.py
        # (collectives_active(): more than one rank, or the one-rank RCCL run of GEOSPLAT_COLLECTIVES_AT_WORLD1=1)
        sharded = (self.prefilter and all_reduce and collectives_active() and os.environ.get("GEOSPLAT_SHARD_PREFILTER", "1") != "0"
                   and can_shard_prefilter(int(cubemap.shape[1]), world))
        # ---- what does not need the pyramid comes first: activations, streams, key width -- and, in capacity mode, the GEOMETRY of
        # the first view (projection, keys, binning: gs_front_fwd without records) on a front stream, so that it runs UNDER the
        # prefilter forward instead of behind it (the step used to start its first compositor 0.9 ms after the prefilter ended)
        if _geo is not None:                                 # (capture_views: the geometry phase of this step ran as its own graph)
            scales_act, opac_act = _geo["scales_act"], _geo["opac_act"]
        else:
            scales_act = p.scales.detach().exp()
            opac_act = torch.sigmoid(p.opacities.detach()).squeeze(-1).contiguous()
        means, quats = p.means.detach(), p.quats.detach()
        normals, kd, ks = p.normals.detach(), p.kd.detach(), p.ks.detach()
        main = torch.cuda.current_stream(dev)
        if self._use_capacity and self._status is None:      # zero-filled on the main stream BEFORE the side streams fork from it
            self._status = torch.zeros(4, dtype=torch.int64, device=dev)
        if self._side_stream is None:
            self._side_stream = [L.shared_stream(dev, "front0"), L.shared_stream(dev, "front1")]
        sides = self._side_stream
        # tile rectangles clipped to the {alpha >= 1/255} extents (GEOSPLAT_TIGHT_TILES=0: gsplat's squares): 18 % fewer intersections
        # on the bench scene, identical pixels (gs_front_fwd; the engine never returns gsplat's `meta`)
        tight = os.environ.get("GEOSPLAT_TIGHT_TILES", "1") != "0"
        # binning keys: 24 bits (three depth passes instead of four) once the depth range of earlier views is known -- key = depth
        # bits - key_base with half an octave of room below the smallest depth seen; a view outside the range is reported through
        # the status word (poll_capacity) and the engine falls back to 32-bit keys
        key_bits, key_base = 32, 0
        if self._use_capacity and self._i_cap is not None and not self._key32 and self._key_lo is not None:
            base = max(0, self._key_lo - (1 << 22))
            if self._key_hi - base < (1 << 24) - (1 << 21):
                key_bits, key_base = 24, base
        i_cap = self._i_cap if self._use_capacity else None
        status = self._status if (key_bits == 24 or self._use_capacity) else None
        seen = []                                            # (pinned counts, event) of this step's views
        early = {}                                           # view index -> (flatten_ids, isect_offsets) binned under the prefilter
        if _geo is not None:
            early = _geo["early"]
        elif i_cap is not None and (_env is None or _geo_only) and self.prefilter and len(cameras) > 0:
            # (the geometry of ONE view: of two, beside the prefilter, costs more than it hides -- 563 against 580 views/s)
            ev_a = torch.cuda.Event(); ev_a.record(main)
            vm_0, K_0, cp_0 = self._camera_tensors(cameras[0])
            sides[0].wait_event(ev_a)
            with torch.cuda.stream(sides[0]):
                fr_g = F.front_stage(means, quats, scales_act, opac_act, normals, kd, ks, vm_0, K_0, cp_0, None, cameras[0].width,
                                     cameras[0].height, self.min_roughness, self.max_metallic, mode, key_base, key_bits, status,
                                     records=False, tight_tiles=tight)
                gstate, _, _ = F.bin_stage(fr_g, i_cap, self._status, prepare=False)
            seen.append((fr_g.host_counts, fr_g.event))
            early[0] = (gstate["flatten_ids"], gstate["isect_offsets"])
            for t in (scales_act, opac_act):
                t.record_stream(sides[0])
        if _geo_only:
            for sd in sides:
                main.wait_stream(sd)                         # (a captured phase: every forked stream joins before the capture ends)
            self._seen_counts.extend(seen)
            return dict(early=early, scales_act=scales_act, opac_act=opac_act)
        if _rec_only:
            # capture_views: the RECORDS (shading) of the views whose geometry came from the geometry graph, as a graph of their own --
            # it needs the pyramid but not the binning, so it replays BESIDE the rest of the geometry graph (projection keys, five
            # radix passes, emission: 0.25 ms per view) instead of behind it
            env_r = TextureSplitSum(_env.base.detach(), [l.detach().contiguous() for l in _env.levels], _env.min_roughness, _env.max_roughness)
            e_r = _make_env(get_fg_lut(dev) if self.fg_lut is None else self.fg_lut, env_r)
            recs = {}
            for j in sorted(early):
                cam = cameras[j]
                vm, K, cam_pos = self._camera_tensors(cam)
                recs[j] = F.front_stage(means, quats, scales_act, opac_act, normals, kd, ks, vm, K, cam_pos, e_r, cam.width, cam.height,
                                        self.min_roughness, self.max_metallic, mode, want_packed_index=True, binning=False, tight_tiles=tight)
            return dict(recs=recs, e=e_r, env_d=env_r)
        # ONE fill, issued before the prefilter: the bucket, the activation-gradient accumulators and the texel-gradient accumulators
        # share an allocation
        cres = int(p.cubemap.shape[1])
        tex_res = [cres]
        while tex_res[-1] > 16:
            tex_res.append(tex_res[-1] // 2)
        tex_sizes = [6 * tex_res[-1] * tex_res[-1] * 3] + [6 * r * r * 3 for r in tex_res]        # base map, then the levels
        sc = self.bucket.zero_with_scratch([3 * N, N, sum(tex_sizes)])
        b = self.bucket.unpack()
        g_scales_act = sc[0].view(N, 3); g_opac_act = sc[1]
        exposure = p.exposure.detach().reshape(1).contiguous()
        if _env is not None:
            env = _env                                       # the pyramid of this step, already filtered (capture_views)
        elif self.prefilter:
            if sharded:
                env = as_splitsum_sharded(cubemap, dist.get_rank(), world, self._prefilter_group())
            else:
                with torch.no_grad():
                    env = as_splitsum(cubemap)
        else:
            if self._static_env is None:
                with torch.no_grad():
                    self._static_env = as_splitsum(p.cubemap)
            env = self._static_env
        env_d = TextureSplitSum(env.base.detach(), [l.detach().contiguous() for l in env.levels], env.min_roughness,
                                env.max_roughness)
        lut = get_fg_lut(dev) if self.fg_lut is None else self.fg_lut
        e = _make_env(lut, env_d)
        # texel-gradient accumulators: one flat buffer behind the base + level gradients (the sharded prefilter sums them over the
        # ranks in ONE all-reduce).  (Two sets with the first half's prefilter backward under the remaining compositor work were
        # measured: the backward is linear, splitting it doubles it, the overlap does not pay that back.)
        sizes = [env_d.base.numel()] + [l.numel() for l in env_d.levels]
        g_flat = sc[2] if sizes == tex_sizes else torch.zeros(sum(sizes), dtype=f32, device=dev)
        parts = torch.split(g_flat, sizes)
        g_base = parts[0].view_as(env_d.base); g_levels = [q.view_as(l) for q, l in zip(parts[1:], env_d.levels)]
        eg = L.GsEnvGrad(); eg.base = g_base.data_ptr()
        for i, g in enumerate(g_levels):
            eg.levels[i] = g.data_ptr()
        self.last_texel_grads = (g_base, g_levels)           # d loss / d pyramid of this step (summed over the local views): tests read it
        images = []
        tail = self._tail_stream
        if tail is None:
            tail = self._tail_stream = L.shared_stream(dev, "tail")
        for sd in sides:
            sd.wait_stream(main)                             # prefilter pyramid, activations, zeroed buckets
        tail.wait_stream(main)

        # the tails of the views in ONE call per batch of views (gs_tail_bwd_multi_parts: the views of a Gaussian on adjacent lanes, its
        # gradients stored once).  The launches that are not the last run as BACKGROUND launches (parts + 4: half of the CUs) beside the
        # compositor of the following views, and the last one, alone on the GPU, covers 2 views instead of 8 (_auto_tail_schedule).
        tb = os.environ.get("GEOSPLAT_TAIL_BATCH", "auto")                                       # "auto" or a schedule "3,4,1" (last repeats)
        tail_sched = _auto_tail_schedule(len(cameras)) if tb == "auto" else [max(1, int(x)) for x in tb.split(",")]
        pending_tails = []
        n_tail_launches = 0
        pstream = None
        # with collectives in the step the LAST tail launch is left to _finish, which runs it in Gaussian-range chunks and sums chunk k
        # over the ranks while chunk k + 1 is still being computed (_chunked_tail)
        n_chunks = max(1, int(os.environ.get("GEOSPLAT_TAIL_CHUNKS", "4")))
        defer_tail = all_reduce and collectives_active() and n_chunks > 1 and N >= 4096
        tail_job = None

        def start_view(cam, j):                              # S1-S3 + A1; (V, I) travel to the host asynchronously
            vm, K, cam_pos = self._camera_tensors(cam)
            side = sides[j % len(sides)]
            if _rec is not None and j in _rec["recs"]:       # (capture_views: the records graph has run)
                return _rec["recs"][j], j, side
            with torch.cuda.stream(side):
                if j in early:                               # binned under the prefilter: only the records (shading) are still missing
                    fr = F.front_stage(means, quats, scales_act, opac_act, normals, kd, ks, vm, K, cam_pos, e, cam.width, cam.height,
                                       self.min_roughness, self.max_metallic, mode, want_packed_index=True, binning=False, tight_tiles=tight)
                else:
                    fr = F.front_stage(means, quats, scales_act, opac_act, normals, kd, ks, vm, K, cam_pos, e, cam.width, cam.height,
                                       self.min_roughness, self.max_metallic, mode, key_base, key_bits, status, want_packed_index=True,
                                       tight_tiles=tight)
            return fr, j, side

        # Capacity protocol (include/geosplat_hip.h): once the engine has seen the intersection counts of a step, the later
        # steps size every per-view buffer by (N, I_cap) and leave (V, I) on the device -- no read-back, no host wait inside
        # the step.  The counts still travel to pinned memory asynchronously; poll_capacity() looks at them (and at the
        # overflow status word) without blocking.
        def bin_view(item):                                  # A2-A4 + record stream on the front stream (exact mode: the host waits for that view's counts)
            fr, j, side = item
            with torch.cuda.stream(side):
                if i_cap is not None:
                    seen.append((fr.host_counts, fr.event))
                state, V, I = F.bin_stage(fr, i_cap, self._status, binned=early.get(j))
                if i_cap is None:
                    self._exact_max_i = max(self._exact_max_i, I)
                    rng = F.depth_range(fr.host_counts)
                    if rng is not None:
                        self._key_lo = rng[0] if self._key_lo is None else min(self._key_lo, rng[0])
                        self._key_hi = rng[1] if self._key_hi is None else max(self._key_hi, rng[1])
                    F.release_counts4(fr.host_counts)
                # the packed gradient records of this view, zeroed HERE (front stream, under the compositor of the previous view)
                state["v_packed"] = torch.zeros(max(V, 1), lib.gs_raster_grad_stride(3), dtype=f32, device=dev)
                ev = torch.cuda.Event(); ev.record(side)
            for t in state.values():
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
            return state, V, I, ev

        n_views = len(cameras)
        proj = [start_view(cameras[j], j) for j in range(min(2, n_views))]   # prologue: A(0), A(1), B1(0)
        binned = bin_view(proj.pop(0)) if n_views else None
        for i, cam in enumerate(cameras):
            vm, K, cam_pos = self._camera_tensors(cam)
            W, H = cam.width, cam.height
            s, V, I, ev = binned
            main.wait_event(ev)
            kev = self.kernel_events
            if kev is not None:
                k0 = torch.cuda.Event(enable_timing=True); k0.record(main)
            # compositor with S4 in its epilogue (`img` comes out of the same launch); it leaves its cull log (stream index + pixel mask
            # of every record that entered a dense batch) for the backward of this view, which then neither culls nor builds masks
            render = torch.empty(H, W, 3, dtype=f32, device=dev); alphas = torch.empty(H, W, dtype=f32, device=dev)
            last_ids = torch.empty(H, W, dtype=torch.int32, device=dev)
            img = torch.empty(H, W, 4, dtype=f32, device=dev)
            rws = s["raster_ws"]
            counts = L.ptr(s["counts"]) if i_cap is not None else None
            log_ws = torch.empty(lib.gs_raster_log_ws_bytes(L.i64(I), W, H, 16), dtype=torch.uint8, device=dev)
            L.check(lib.gs_raster_composite_tone_log(W, H, 16, V, None, L.i64(I), counts, L.ptr(s["isect_offsets"]), L.ptr(render),
                                                     L.ptr(alphas), L.ptr(last_ids), tone, L.ptr(exposure), L.ptr(img), L.ptr(rws),
                                                     C.c_size_t(rws.numel()), L.ptr(log_ws), C.c_size_t(log_ws.numel()), st()),
                    "gs_raster_composite_tone_log")
            if kev is not None:
                k1 = torch.cuda.Event(enable_timing=True); k1.record(main)
                kev.append(("raster_fwd_kernel", k0, k1))
            v_img = upstream(i, img).contiguous()
            v_packed = s["v_packed"]
            if kev is not None:
                k2 = torch.cuda.Event(enable_timing=True); k2.record(main)
            L.check(lib.gs_raster_bwd_tone_log_acc(W, H, 16, V, None, L.i64(I), counts, L.ptr(s["isect_offsets"]), L.ptr(render),
                                                   L.ptr(alphas), L.ptr(last_ids), tone, L.ptr(exposure), L.ptr(v_img), L.ptr(v_packed),
                                                   L.ptr(b["exposure"]), L.ptr(rws), C.c_size_t(rws.numel()), L.ptr(log_ws),
                                                   C.c_size_t(log_ws.numel()), st()), "gs_raster_bwd_tone_log_acc")
            if kev is not None:
                k3 = torch.cuda.Event(enable_timing=True); k3.record(main)
                kev.append(("raster_bwd_kernel", k2, k3))
            # keep the front streams two views ahead: A(i+2), then B1(i+1) -- AFTER this view's main-stream chain has been enqueued
            # (their ~35 launches cost the host 0.2-0.3 ms; issued first, the compositor stream idled that long per view)
            if i + 2 < n_views:
                proj.append(start_view(cameras[i + 2], i + 2))
            binned = bin_view(proj.pop(0)) if i + 1 < n_views else None
            # gradient tail (A7 + S1-S3 backward: HBM / atomic-rate bound) on the tail stream, beside the compositor of the next views
            pending_tails.append((vm, K, cam_pos, s["vis_records"], v_packed, s["packed_index"], W, H))
            last = i == n_views - 1
            if len(pending_tails) == tail_sched[min(n_tail_launches, len(tail_sched) - 1)] or last:
                ev_r = torch.cuda.Event(); ev_r.record(main)
                targs = (pending_tails, means, quats, scales_act, opac_act, normals, kd, ks, e, eg, self.min_roughness, self.max_metallic,
                         mode, b["means"], b["quats"], g_scales_act, g_opac_act, b["normals"], b["kd"], b["ks"])
                if last and defer_tail:
                    tail_job = dict(targs=targs, accumulate=n_tail_launches > 0, n_chunks=n_chunks)
                    if keep_images:
                        images.append(img)
                    continue
                tev = self.tail_events if last else None
                with torch.cuda.stream(tail):
                    tail.wait_event(ev_r)
                    if tev is not None:
                        t0 = torch.cuda.Event(enable_timing=True); t0.record(tail)
                    # (the LAST launch: its projection half on a front stream -- idle by then -- beside the prefilter backward, which
                    #  needs only the shading half's texel gradients)
                    F.tail_multi_stage(*targs, accumulate=n_tail_launches > 0, parts=1 if last else 7)
                used = [tail]
                if last:
                    ev_sh = torch.cuda.Event(enable_timing=tev is not None); ev_sh.record(tail)
                    pstream = sides[0]
                    with torch.cuda.stream(pstream):
                        pstream.wait_event(ev_sh)              # (v_means: the projection half adds to what the shading half stored)
                        F.tail_multi_stage(*targs, accumulate=n_tail_launches > 0, parts=2)
                        if tev is not None:
                            t2 = torch.cuda.Event(enable_timing=True); t2.record(pstream)
                            tev.append((t0, ev_sh, t2))
                    used.append(pstream)
                n_tail_launches += 1
                for tv in pending_tails:
                    for t in tv[:6]:
                        for u in used:
                            t.record_stream(u)
                pending_tails = []
            if keep_images:
                images.append(img)
        main.wait_stream(tail)
        self._seen_counts.extend(seen)                       # entries of earlier steps the host has not looked at yet stay pending
        if i_cap is not None:                                # the overflow word follows the step to the host, asynchronously
            if torch.cuda.is_current_stream_capturing():      # a replayed graph refreshes ONE pinned buffer; replay.check() reads it
                if self._status_host is None:
                    self._status_host = torch.zeros(4, dtype=torch.int64).pin_memory()
                self._status_host.copy_(self._status, non_blocking=True)
                self._status_event = torch.cuda.Event(); self._status_event.record()
            else:
                # eager: every step gets its OWN snapshot and clears the word behind it (all of this step's writers are upstream of
                # the main stream here, the next step's fronts fork from it), so that an overflow is attributed to exactly one step
                # however many steps are in flight
                snap = self._status_pool.pop() if self._status_pool else torch.zeros(4, dtype=torch.int64).pin_memory()
                snap.copy_(self._status, non_blocking=True)
                self._status.fill_(0)
                ev_s = torch.cuda.Event(); ev_s.record()
                self._status_pending.append((snap, ev_s))
        # chain the once-per-step activations (behind the projection half of the last tail); the per-Gaussian gradients are then
        # final, so their all-reduce (RCCL on the communication stream) overlaps the prefilter backward
        if tail_job is None:
            with torch.cuda.stream(pstream if pstream is not None else main):
                L.check(lib.gs_activation_chain(L.i64(N), L.ptr(g_scales_act), L.ptr(scales_act), L.ptr(g_opac_act), L.ptr(opac_act),
                                                L.ptr(b["scales"]), L.ptr(b["opacities"]), st()), "gs_activation_chain")
            if pstream is not None:
                for t in (scales_act, opac_act):
                    t.record_stream(pstream)
        ctx = dict(b=b, images=(images if keep_images else None), g_base=g_base, g_levels=g_levels, g_flat=g_flat, env=env, sharded=sharded,
                   world=world, all_reduce=all_reduce, main=main, pstream=pstream, tail_job=tail_job)
        if _stop_after_views:                                 # (a captured views segment: every forked stream joins before the capture ends)
            if pstream is not None:
                main.wait_stream(pstream)
                ctx["pstream"] = None
            return ctx
        return self._finish(ctx)

    def _finish(self, ctx):
        """What follows the views of a step: gradient all-reduce (its per-Gaussian part under the prefilter backward) and the
        prefilter backward on the texel gradients summed over the views (and, sharded, over the ranks)."""
        import torch.distributed as dist
        b, images, g_base, g_levels, g_flat, env = (ctx[k] for k in ("b", "images", "g_base", "g_levels", "g_flat", "env"))
        sharded, world, all_reduce, main = (ctx[k] for k in ("sharded", "world", "all_reduce", "main"))
        pstream = ctx.get("pstream")                          # the per-Gaussian gradients become final on this stream (None: on main)
        chunked = ctx.get("tail_job") is not None
        if chunked:                                           # last tail in Gaussian-range chunks, each summed over the ranks as it ends
            pstream = self._chunked_tail(ctx)
            main.wait_stream(self._tail_stream)               # the texel gradients are complete behind the last chunk's shading half
        head_stream = pstream if pstream is not None else main
        if sharded:
            grp = self._prefilter_group()
            dist.all_reduce(g_flat, op=dist.ReduceOp.SUM, group=grp)             # texel gradients of ALL views: needed by every share
            if not chunked:
                start_head, _ = self.bucket.all_reduce_split("cubemap")
                with torch.cuda.stream(head_stream):
                    start_head()                                # per-Gaussian segments: communication stream, default communicator
            g_cube = as_splitsum_backward_sharded(g_base, g_levels, dist.get_rank(), world, grp, min_roughness=env.min_roughness,
                                                  max_roughness=env.max_roughness)
            b["cubemap"].copy_(g_cube)                          # identical on every rank: not reduced again
            if pstream is not None:
                main.wait_stream(pstream)
            self.bucket.all_reduce_names(["exposure"])          # queues behind the head on the communication stream, then joins it
            return b, images
        start_head, finish = self.bucket.all_reduce_split("cubemap") if all_reduce else ((lambda: None), (lambda: None))
        if not chunked:
            with torch.cuda.stream(head_stream):
                start_head()
        if self.prefilter:                                      # (the finest level's transposed apply writes the bucket's slice itself)
            as_splitsum_backward(g_base, g_levels, min_roughness=env.min_roughness, max_roughness=env.max_roughness, out=b["cubemap"])
        if pstream is not None:
            main.wait_stream(pstream)
        finish()
        return b, images

    def _chunked_tail(self, ctx):
        """The last tail launch of a step with collectives, in `n_chunks` Gaussian ranges: shading half of chunk k on the tail stream,
        its projection half + the activation chain of those rows on a front stream behind it, and -- as soon as they are final -- the
        all-reduce of exactly those rows of the seven per-Gaussian segments on the communication stream (GradBucket.all_reduce_rows),
        while the tail stream already computes chunk k + 1.  The tail kernels index every per-Gaussian array by the Gaussian, so a
        chunk is the same launch on pointers moved to the range's first row.  With one view per rank (BASELINE config 4) the 149 MB
        all-reduce is otherwise fully exposed behind a 0.33 ms tail; here all but the last quarter of it runs under the tail.
        Returns the stream on which the per-Gaussian gradients became final."""
        job = ctx["tail_job"]
        lib = L.lib()
        b = ctx["b"]
        self.chunked_tails += 1
        (views, means, quats, scales_act, opac_act, normals, kd, ks, e, eg, mr, mm, mode, g_means, g_quats, g_scales_act, g_opac_act,
         g_normals, g_kd, g_ks) = job["targs"]
        N = means.shape[0]
        C_ = job["n_chunks"]
        bounds = [min(N, ((N * k // C_ + 1023) // 1024) * 1024) for k in range(C_)] + [N]
        tail, pstream = self._tail_stream, self._side_stream[0]
        names = [k for k in PARAM_NAMES if k not in ("cubemap", "exposure")]
        tail.wait_stream(ctx["main"])                         # (behind the last compositor backward; a captured views segment: behind its replay)
        captured = ctx.get("captured", False)
        for c in range(C_):
            n0, n1 = bounds[c], bounds[c + 1]
            if n1 <= n0:
                continue
            sl = lambda t: t[n0:n1]
            vc = [(vm, K, cp, vis, vp, pidx[n0:n1], W, H) for (vm, K, cp, vis, vp, pidx, W, H) in views]
            args = (vc, sl(means), sl(quats), sl(scales_act), sl(opac_act), sl(normals), sl(kd), sl(ks), e, eg, mr, mm, mode, sl(g_means),
                    sl(g_quats), sl(g_scales_act), sl(g_opac_act), sl(g_normals), sl(g_kd), sl(g_ks))
            with torch.cuda.stream(tail):
                F.tail_multi_stage(*args, accumulate=job["accumulate"], parts=1)
                ev_sh = torch.cuda.Event(); ev_sh.record(tail)
            with torch.cuda.stream(pstream):
                pstream.wait_event(ev_sh)                      # (v_means: the projection half adds to what the shading half stored)
                F.tail_multi_stage(*args, accumulate=job["accumulate"], parts=2)
                L.check(lib.gs_activation_chain(L.i64(n1 - n0), L.ptr(sl(g_scales_act)), L.ptr(sl(scales_act)), L.ptr(sl(g_opac_act)),
                                                L.ptr(sl(opac_act)), L.ptr(b["scales"][n0:n1]), L.ptr(b["opacities"][n0:n1]), L.stream()),
                        "gs_activation_chain")
                ev_c = torch.cuda.Event(); ev_c.record(pstream)
            self.bucket.all_reduce_rows(names, n0, n1, after=ev_c)
        if not captured:                                      # (a captured segment's tensors live in the graph's private pool)
            for tv in views:
                for t in tv[:6]:
                    t.record_stream(tail); t.record_stream(pstream)
            for t in (scales_act, opac_act, g_scales_act, g_opac_act):
                t.record_stream(tail); t.record_stream(pstream)
        return pstream

    def capture(self, cameras: List[Camera], upstream: Callable[[int, Tensor], Tensor], keep_images: bool = False):
        """One fused step over a FIXED camera list recorded into a HIP graph (torch.cuda.CUDAGraph): possible because in
        capacity mode a step has fixed launch shapes and no host synchronisation.  Returns `replay() -> (grads, images)`:
        the gradients land in this engine's bucket, exactly as after __call__(..., all_reduce=False); the caller reduces
        them over the ranks itself (collectives stay outside the graph).  `upstream` is traced once: it must compute
        d(loss)/d(image) with device-side operations on buffers that keep their addresses (update ground-truth images in
        place between replays).  Parameters are read from the tensors bound at capture time (update them in place).
        replay.check() synchronises and returns False if a view exceeded the capacity OR left the depth range of the 24-bit binning
        keys baked into the graph: the graph is then stale (it would keep replaying with clamped keys, i.e. a wrong depth order) --
        call poll_capacity() and capture() AGAIN before the next replay; a replay after a False check() raises."""
        if not (self.fused and self.mode == "pbr"):
            raise RuntimeError("capture() needs the fused path")
        if self._i_cap is None:
            raise RuntimeError("capture() needs a known capacity: run one eager step and poll_capacity(wait=True) first")
        from .rasterization import _pinned_pool
        dev = self.p.means.device
        while len(_pinned_pool) < 2 * len(cameras) + 2:       # pinned buffers cannot be allocated while a stream is capturing
            _pinned_pool.append(torch.empty(2, dtype=torch.int64).pin_memory())
        F.reserve_pinned(2 * len(cameras) + 2)
        if self._status is None:
            self._status = torch.zeros(4, dtype=torch.int64, device=dev)
        if self._status_host is None:
            self._status_host = torch.zeros(4, dtype=torch.int64).pin_memory()
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):                          # (allocator warm-up on a side stream, as torch.cuda.graph asks)
            self._step_fused(cameras, upstream, False, keep_images)
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        self.poll_capacity(_internal=True)                    # an overflow of the warm-up step stays pending for the caller's next poll
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._step_fused(cameras, upstream, False, keep_images)
        counts = [hc for hc, _ in self._seen_counts]          # refreshed by every replay (D2H copies are graph nodes)
        self._seen_counts = []
        self._status_event = None
        cap = self._i_cap
        status_host = self._status_host

        stale = [False]

        def replay():
            if stale[0]:
                raise RuntimeError("this captured step overflowed its capacity / key range (check() returned False): capture() again")
            graph.replay()
            return out

        def check() -> bool:
            torch.cuda.synchronize(dev)
            worst = max([int(hc[1]) for hc in counts] + [0])
            overflow = int(status_host[0]) != 0 or int(status_host[3]) != 0 or worst > cap
            if overflow:
                stale[0] = True
                self.truncated_steps += 1
                self._key32 = self._key32 or int(status_host[3]) != 0
                self._exact_max_i = max(self._exact_max_i, worst, int(status_host[1]))
                self._status.zero_(); status_host.zero_()
            return not overflow
        replay.check = check
        replay.graph = graph
        return replay

    def capture_views(self, cameras: List[Camera], upstream: Callable[[int, Tensor], Tensor], all_reduce: bool = True,
                      keep_images: bool = False):
        """The VIEWS of a step -- everything between the pyramid and the texel gradients: shading, projection, binning, compositor,
        tone map, upstream and the whole per-view backward -- recorded into one HIP graph, with the prefilter and every collective
        OUTSIDE of it.  This is the shape BASELINE config 4 needs (one view per GPU): a single view is ~45 launches, 2.3-4.5 ms from
        Python depending on the host core against 2.1 ms of kernel time, while the sharded prefilter, the all-reduce of the texel
        gradients and the flat gradient all-reduce (RCCL) cannot be captured together with it.  Returns `step() -> (grads, images)`
        = prefilter forward (eager, sharded over the ranks when all_reduce and world > 1) -> copy into the graph's pyramid buffers ->
        graph replay -> prefilter backward + all-reduces (eager); same results as __call__.  `upstream` as in capture();
        step.check() as replay.check()."""
        if not (self.fused and self.mode == "pbr" and self.prefilter):
            raise RuntimeError("capture_views() needs the fused path with the prefilter in the step")
        if self._i_cap is None:
            raise RuntimeError("capture_views() needs a known capacity: run one eager step and poll_capacity(wait=True) first")
        import torch.distributed as dist
        from .rasterization import _pinned_pool
        dev = self.p.means.device
        while len(_pinned_pool) < 2 * len(cameras) + 2:       # pinned buffers cannot be allocated while a stream is capturing
            _pinned_pool.append(torch.empty(2, dtype=torch.int64).pin_memory())
        F.reserve_pinned(2 * len(cameras) + 2)
        if self._status_host is None:
            self._status_host = torch.zeros(4, dtype=torch.int64).pin_memory()
        world = dist.get_world_size() if (all_reduce and dist.is_available() and dist.is_initialized()) else 1
        sharded = (all_reduce and collectives_active() and os.environ.get("GEOSPLAT_SHARD_PREFILTER", "1") != "0"
                   and can_shard_prefilter(int(self.p.cubemap.shape[1]), world))

        def filter_env():
            cubemap = self.p.cubemap.detach()
            if sharded:
                return as_splitsum_sharded(cubemap, dist.get_rank(), world, self._prefilter_group())
            with torch.no_grad():
                return as_splitsum(cubemap)
        first = filter_env()
        pyramid = TextureSplitSum(first.base.detach().clone(), [l.detach().clone() for l in first.levels], first.min_roughness,
                                  first.max_roughness)
        slots = [pyramid.base] + list(pyramid.levels)
        # THREE graphs (GEOSPLAT_GEO_GRAPH=0: one).  With one view per GPU the chain front -> binning -> record stream -> compositor ->
        # tail is serial; two pieces of it need less than the whole chain has:
        #   geometry graph: projection, depth keys, binning of the views -- nothing that needs the pyramid -- on its own stream BESIDE the
        #                   eager (sharded) prefilter forward and its all-reduce;
        #   records graph : the shading of those views -- needs the pyramid, NOT the binning -- behind the prefilter, beside whatever is
        #                   left of the geometry graph (0.39 ms of front + binning against 0.1 ms of sharded prefilter at 8 GPUs);
        #   views graph   : record stream, compositor, loss cotangent, compositor backward, background tails -- behind both.
        # The last tail launch and every collective stay eager (_finish): its Gaussian-range chunks are summed over the ranks while the
        # next chunk is computed.
        two = os.environ.get("GEOSPLAT_GEO_GRAPH", "1") != "0"
        F.reserve_pinned(6 * len(cameras) + 2)                 # (the geometry and records fronts carry their own count read-backs)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):                          # (allocator warm-up on a side stream, as torch.cuda.graph asks)
            geo = self._step_fused(cameras, upstream, all_reduce, keep_images, _geo_only=True) if two else None
            rec = self._step_fused(cameras, upstream, all_reduce, keep_images, _env=pyramid, _geo=geo, _rec_only=True) if two else None
            self._step_fused(cameras, upstream, all_reduce, keep_images, _env=pyramid, _stop_after_views=True, _geo=geo, _rec=rec)
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        self.poll_capacity(_internal=True)                    # an overflow of the warm-up step stays pending for the caller's next poll
        geo_graph, geo, rec_graph, rec = None, None, None, None
        if two:
            geo_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(geo_graph):
                geo = self._step_fused(cameras, upstream, all_reduce, keep_images, _geo_only=True)
            rec_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(rec_graph, pool=geo_graph.pool()):
                rec = self._step_fused(cameras, upstream, all_reduce, keep_images, _env=pyramid, _geo=geo, _rec_only=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, pool=geo_graph.pool() if two else None):
            ctx = self._step_fused(cameras, upstream, all_reduce, keep_images, _env=pyramid, _stop_after_views=True, _geo=geo, _rec=rec)
        geo_stream = L.shared_stream(dev, "geo") if two else None
        counts = [hc for hc, _ in self._seen_counts]          # refreshed by every replay (D2H copies are graph nodes)
        self._seen_counts = []
        self._status_event = None
        cap = self._i_cap
        status_host = self._status_host
        ctx["main"] = None
        ctx["captured"] = True

        stale = [False]

        def step():
            if stale[0]:
                raise RuntimeError("these captured views overflowed their capacity / key range (check() returned False): capture_views() again")
            cur = torch.cuda.current_stream(dev)
            if two:
                geo_stream.wait_stream(cur)                    # (the parameters of this step are final on the caller's stream)
                with torch.cuda.stream(geo_stream):
                    geo_graph.replay()
            env = filter_env()
            torch._foreach_copy_(slots, [env.base] + list(env.levels))
            if two:
                rec_graph.replay()                             # shading of the views: beside the geometry graph
                cur.wait_stream(geo_stream)
            graph.replay()
            ctx["main"] = cur
            return self._finish(ctx)

        def check() -> bool:
            torch.cuda.synchronize(dev)
            worst = max([int(hc[1]) for hc in counts] + [0])
            overflow = int(status_host[0]) != 0 or int(status_host[3]) != 0 or worst > cap
            if overflow:
                stale[0] = True
                self.truncated_steps += 1
                self._key32 = self._key32 or int(status_host[3]) != 0
                self._exact_max_i = max(self._exact_max_i, worst, int(status_host[1]))
                self._status.zero_(); status_host.zero_()
            return not overflow
        step.check = check
        step.graph = graph
        step.geo_graph = geo_graph
        step.rec_graph = rec_graph
        return step

    def _throttle(self) -> None:
        """Bound how far the host runs ahead of the GPU.  The step has no host synchronisation, the host enqueues it in 3-5 ms
        against 15 ms on the GPU, and HIP queues without limit: after 400 steps the host was 200 steps ahead and the caching
        allocator -- whose blocks are only reusable once the streams they were recorded on have passed them -- held 186 GiB for
        a step whose live memory is 3.8 GiB (scripts/soak.py).  At most GEOSPLAT_STEPS_IN_FLIGHT (default 3) steps are enqueued
        beyond the one the GPU is working on; the wait polls the event (no blocking call: the step stays clean under
        torch.cuda.set_sync_debug_mode) and never leaves the GPU idle, since one whole step is still queued behind it."""
        if self._in_flight is None:
            return
        import time
        while len(self._in_flight) >= self._max_in_flight:
            ev = self._in_flight[0]
            while not ev.query():
                time.sleep(5e-5)
            self._in_flight.popleft()

    def poll_capacity(self, wait: bool = False, _internal: bool = False) -> bool:
        """Host side of the capacity protocol; never blocks unless `wait`.  Looks at what the earlier steps left in pinned memory:
        the per-view (V, I) counts set / raise the intersection capacity (1.25 x the largest count seen, rounded up to 64 Ki),
        and the overflow word tells whether a view exceeded the capacity it ran with.  Returns False in that case -- that step's
        gradients are incomplete (memory-safe, a truncated view) and the caller should repeat it; the capacity has already been
        raised.  An overflow is never lost: RenderStep.__call__ polls before every step (non-blocking) and, if IT sees the
        word set, counts the step in `truncated_steps` and keeps the fact until the next poll_capacity() of the caller, which
        then returns False.  A trainer that must not consume a truncated step calls poll_capacity(wait=True) before its optimiser
        step (stage1.py does); bench.py reports `truncated_steps`."""
        ok = True
        max_i, self._exact_max_i = self._exact_max_i, 0
        for hc, ev in self._seen_counts:
            if hc is None:
                continue
            if wait:
                ev.synchronize()
            if ev.query():
                max_i = max(max_i, int(hc[1]))
                if hc.numel() == 4:                             # fused front: {V, I, ~min depth bits, max depth bits}
                    rng = F.depth_range(hc)
                    if rng is not None:
                        self._key_lo = rng[0] if self._key_lo is None else min(self._key_lo, rng[0])
                        self._key_hi = rng[1] if self._key_hi is None else max(self._key_hi, rng[1])
        while self._status_pending:                           # eager steps, oldest first; each snapshot is inspected exactly once
            snap, ev_s = self._status_pending[0]
            if wait:
                ev_s.synchronize()
            if not ev_s.query():
                break
            self._status_pending.popleft()
            if int(snap[0]) != 0 or int(snap[3]) != 0:
                ok = False
                self.truncated_steps += 1
                max_i = max(max_i, int(snap[1]))
                if int(snap[3]) != 0:                          # a depth outside the 24-bit key range: 32-bit keys from now on
                    self._key32 = True
            self._status_pool.append(snap)
        if self._status_event is not None:                    # the single buffer of a captured step (see capture / capture_views)
            if wait:
                self._status_event.synchronize()
            if self._status_event.query() and (int(self._status_host[0]) != 0 or int(self._status_host[3]) != 0):
                ok = False
                self.truncated_steps += 1
                max_i = max(max_i, int(self._status_host[1]))
                if int(self._status_host[3]) != 0:
                    self._key32 = True
                self._status.zero_(); self._status_host.zero_()
            if self._status_event.query():
                self._status_event = None
        from .rasterization import _pinned_pool
        still = []
        for hc, ev in self._seen_counts:
            if hc is None:
                continue
            if ev.query():
                (F.release_counts4 if hc.numel() == 4 else _pinned_pool.append)(hc)
            else:
                still.append((hc, ev))
        self._seen_counts = still
        if self._use_capacity and max_i > 0:
            want = ((int(max_i * self._cap_margin) + 65535) // 65536) * 65536
            if self._i_cap is None or want > self._i_cap or max_i > self._i_cap:
                self._i_cap = max(want, self._i_cap or 0)
        if _internal:
            self._overflow_unreported = self._overflow_unreported or not ok
        elif self._overflow_unreported:
            ok, self._overflow_unreported = False, False
        return ok

    def __call__(self, cameras: List[Camera], upstream: Callable[[int, Tensor], Tensor], all_reduce: bool = True,
                 keep_images: bool = False):
        """Forward + backward for `cameras`; `upstream(i, image)` returns d(loss)/d(image) for local view i.
        Returns (grads dict of views into the flat bucket, images or None)."""
        if self.fused and self.mode == "pbr":
            self._throttle()                                 # first: the step that just left the window has finished, its word is readable
            self.poll_capacity(_internal=True)               # non-blocking: counts / overflow words of the finished steps (an overflow
                                                             # seen here is kept for the caller's next poll_capacity())
            out = self._step_fused(cameras, upstream, all_reduce, keep_images)
            if self._in_flight is not None:
                ev = torch.cuda.Event(); ev.record()
                self._in_flight.append(ev)
            return out
        p = self.p
        leaves = {k: v.detach().requires_grad_(True) for k, v in p.named().items()}
        if self.prefilter:
            env = as_splitsum(leaves["cubemap"])
        else:
            if self._static_env is None:
                with torch.no_grad():
                    self._static_env = as_splitsum(p.cubemap)
            env = self._static_env
        # cut the graph at the pyramid so that the prefilter backward runs ONCE with the summed texel gradients
        base_leaf = env.base.detach().requires_grad_(True)
        level_leaves = [l.detach().requires_grad_(True) for l in env.levels]
        env_leaf = TextureSplitSum(base_leaf, level_leaves, env.min_roughness, env.max_roughness)

        class _G:  # raw Splats-like view
            means = leaves["means"]; scales = leaves["scales"]; quats = leaves["quats"]; opacities = leaves["opacities"]
        attrs = RenderableAttrs(kd=leaves["kd"], ks=leaves["ks"], normals=leaves["normals"])
        images = []
        for i, cam in enumerate(cameras):
            img = attrs.splat(_G, [cam], exposure=leaves["exposure"], envmap=env_leaf, min_roughness=self.min_roughness,
                              max_metallic=self.max_metallic, mode=self.mode, tone_type=self.tone_type, fg_lut=self.fg_lut)
            img.backward(upstream(i, img.detach()))
            if keep_images:
                images.append(img.detach())
        if self.prefilter:
            pairs = [(env.base, base_leaf.grad)] + [(l, ll.grad) for l, ll in zip(env.levels, level_leaves)]
            outs = [o for o, g in pairs if g is not None]
            gouts = [g for o, g in pairs if g is not None]
            if outs:
                torch.autograd.backward(outs, gouts)
        grads = {k: leaves[k].grad for k in PARAM_NAMES}
        self.bucket.pack(grads)
        if all_reduce:
            self.bucket.all_reduce()
        return self.bucket.unpack(), (images if keep_images else None)


class MeshStep:
    """mesh -> Gaussians -> render -> backward to the mesh: the geometry half of the reference's stage-1 loop
    (GeoSplatter.get_gaussians_from_mesh, rfstudio/model/geosplat.py:620-672, then render_report) with the
    FlexiCubes extraction and the hash-grid field replaced by explicit leaves:
    `vertices` [V,3] and per-Gaussian `kd` [6F,3] / `ks` [6F,2].  Per step:
        vertex normals -> MGAdapter (HIP, mesh.py) -> RenderStep (fused C-ABI path, per-Gaussian gradient bucket,
        all-reduce at that cut) -> MGAdapter backward -> vertex-normal backward -> d loss / d vertices."""

    def __init__(self, vertices: Tensor, faces: Tensor, kd: Tensor, ks: Tensor, cubemap: Tensor, exposure: Tensor,
                 **render_kwargs):
        from .mesh import mesh_to_splats, vertex_normals
        self._m2s, self._vn = mesh_to_splats, vertex_normals
        self.vertices, self.faces, self.kd, self.ks, self.cubemap, self.exposure = vertices, faces, kd, ks, cubemap, exposure
        self.render_kwargs = render_kwargs
        self.step: Optional[RenderStep] = None

    def __call__(self, cameras: List[Camera], upstream: Callable[[int, Tensor], Tensor], all_reduce: bool = True,
                 keep_images: bool = False):
        v = self.vertices.detach().requires_grad_(True)
        sp, nrm = self._m2s(v, self.faces, self._vn(v, self.faces))
        params = PathParams(sp.means.detach(), sp.scales.detach(), sp.quats.detach(), sp.opacities.detach(), nrm.detach(),
                            self.kd.detach(), self.ks.detach(), self.cubemap.detach(), self.exposure.detach())
        if self.step is None:
            self.step = RenderStep(params, **self.render_kwargs)
        else:
            self.step.rebind(params)
        grads, images = self.step(cameras, upstream, all_reduce=all_reduce, keep_images=keep_images)
        # shading normals == splat colours of the adapter: their gradient rides on `nrm`
        torch.autograd.backward([sp.means, sp.scales, sp.quats, nrm],
                                [grads["means"], grads["scales"], grads["quats"], grads["normals"]])
        out = {"vertices": v.grad, "kd": grads["kd"], "ks": grads["ks"], "cubemap": grads["cubemap"],
               "exposure": grads["exposure"]}
        return out, images


def params_from_scene(scene, device, exposure: float = 1.0) -> PathParams:
    sp: SplatSet = scene.splats
    d = lambda t: t.to(device).contiguous()
    return PathParams(d(sp.means), d(sp.scales), d(sp.quats), d(sp.opacities), d(scene.normals), d(scene.kd),
                      d(scene.ks), d(scene.cubemap), torch.tensor(exposure, dtype=torch.float32, device=device))
