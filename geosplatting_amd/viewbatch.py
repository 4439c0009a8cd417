"""The fused per-view kernels (csrc/gs_front.hip, the cull-log compositor pair, the batched tails) BEHIND THE REFERENCE'S CALL SHAPE.

The reference's training step is a Python loop of ``attrs.splat(gsplat, camera, exposure=, envmap=, ...)`` over the views of a
batch (rfstudio/model/geosplat.py:863-879) followed by ONE ``loss.backward()`` (rfstudio/optim/optimizer.py:107).  This module is
what ``geosplatting_amd.RenderableAttrs.splat`` runs for that shape -- no engine object, no callback, nothing the trainer has to
know about:

  * every ``splat()`` call is ONE autograd node (``_SplatView``): forward = gs_front_fwd -> gs_isect_bin_front ->
    gs_raster_prepare_vis (a front stream) -> gs_raster_composite_tone_log (the caller's stream); backward =
    gs_raster_bwd_tone_log_acc into the view's 64-byte gradient records;
  * the views that read the SAME parameter tensors (same objects, same version counters) share a ``_Step``: activations once,
    one set of gradient buffers, and one ``_Gather`` node between the parameters and the views.  Autograd runs a node when all
    nodes that depend on it have run, so ``_Gather.backward`` executes exactly once per ``backward()``, AFTER the compositor
    backward of every view that took part: there the tails of the views (projection + shading backward, gs_tail_bwd_multi_parts:
    the views of a Gaussian on adjacent lanes) go out in the engine's batches -- the early batches as background launches from the
    view nodes themselves, beside the compositor backward of the following views -- and the SUMS over the views are returned as
    the gradients of the parameters.  The view nodes return no per-Gaussian gradient at all (autograd would add eight 149 MB
    tensors otherwise).  A backward over a subset of the views, several backward calls, views that never get one: all correct
    (whatever is pending when the gather node runs is flushed; nothing pending, nothing returned).

Capacity protocol (include/geosplat_hip.h): EVERY view of the first step at an image size (W, H) reads its counts back (exact mode,
as gsplat does); later steps size their buffers by 1.5 x the largest intersections-per-Gaussian RATIO seen at that image size, times
their own N -- a stage-1 loop that extracts a different number of Gaussians every iteration keeps its capacity -- and leave the
counts on the device.  The counts of every view still travel to pinned memory; they are checked when the gather node runs (by then
the fronts have long finished: the wait costs nothing) and a view that exceeded its capacity raises GeoSplatCapacityError THERE --
from the ``backward()`` of the step that owns the view, before any optimizer can consume it -- with the capacity already raised for
the retry (``retry_on_capacity`` wraps a step function accordingly; stage1.train_step uses it).  A truncated view that never gets a
backward is reported by a warning.  GEOSPLAT_CAPACITY=0 keeps every call exact.  Calls under ``torch.no_grad()`` (or on tensors that
need no gradient) are always exact.

Parameter identity: views join a step when they read the same tensor OBJECTS at the same storage address with the same version
counters.  An in-place write that bypasses the version counter (``p.data.copy_()`` / a checkpoint load through ``.data``) between
two calls with NO version bump in between is not seen -- call ``viewbatch.reset()`` after such a load.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
import warnings
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import front as F
from .cameras import Camera
from .splitsum import TextureSplitSum


class GeoSplatCapacityError(L.GeoSplatHipError):
    """A view had more tile intersections (or a depth range wider) than the capacity it was launched with: its image and gradients
    are those of a truncated list.  The capacity has been raised; repeat the step."""


# ------------------------------------------------------------------------------------------------------------ per-device state
class _DeviceState:
    def __init__(self, dev: torch.device):
        self.dev = dev
        # two front streams (+ the caller's stream + the tail stream = the four hardware queues HIP maps its streams onto by default:
        # a third front stream is 2 % faster alone in a process -- 608 against 596 views/s -- and 13 % SLOWER, 520 against 595, in a
        # process that holds other streams, where two of them then share a queue)
        self.fronts = [L.shared_stream(dev, "front0"), L.shared_stream(dev, "front1")]
        self.tail = L.shared_stream(dev, "tail")
        self.caps: "collections.OrderedDict[Tuple[int, int], _Capacity]" = collections.OrderedDict()   # by image size, least recently used first
        self.current: Optional["_Step"] = None   # the step new splat() calls may join
        self.unchecked = collections.deque()     # capacity-mode views whose counts nobody has looked at yet
        self.cam_cache: Dict[tuple, tuple] = {}
        self.n_views = 0                         # views rendered so far (front stream round robin)
        self.lock = threading.RLock()            # forward runs on the caller's thread, backward on autograd's: both touch `unchecked`


_MAX_CAPS = 64


class _Capacity:
    """What the earlier views at an image size (W, H) taught: intersections per Gaussian (the capacity of a later view is that
    ratio x its own N x margin) and the depth-bit range for 24-bit keys.  `first_step`: key of the step whose views taught it --
    all of them run exact, so the capacity rests on a whole batch of views, not on one."""
    __slots__ = ("ratio", "key_lo", "key_hi", "key32", "max_i", "first_step")

    def __init__(self):
        self.ratio = None; self.key_lo = None; self.key_hi = None
        self.key32 = os.environ.get("GEOSPLAT_KEY_BITS", "24") == "32"
        self.max_i = 0
        self.first_step = None

    def learn(self, host_counts: Tensor, n: int) -> None:
        i = int(host_counts[1])
        self.max_i = max(self.max_i, i)
        r = max(i, 1) / float(max(n, 1))
        if self.ratio is None or r > self.ratio:
            self.ratio = r
        rng = F.depth_range(host_counts)
        if rng is not None:
            self.key_lo = rng[0] if self.key_lo is None else min(self.key_lo, rng[0])
            self.key_hi = rng[1] if self.key_hi is None else max(self.key_hi, rng[1])

    def i_cap(self, n: int) -> Optional[int]:
        if self.ratio is None:
            return None
        margin = float(os.environ.get("GEOSPLAT_CAPACITY_MARGIN", "1.5"))
        want = int(self.ratio * max(n, 1) * margin) + 1
        gran = 65536 if want >= (1 << 20) else 4096             # (stable buffer sizes for the caching allocator)
        return ((want + gran - 1) // gran) * gran

    def keys(self) -> Tuple[int, int]:
        """(key_bits, key_base) for the next view: 24-bit keys (three depth passes instead of four) with half an octave of room
        below the smallest depth seen, while the range seen fits."""
        if self.key32 or self.key_lo is None:
            return 32, 0
        base = max(0, self.key_lo - (1 << 22))
        if self.key_hi - base < (1 << 24) - (1 << 21):
            return 24, base
        return 32, 0


_states: Dict[int, _DeviceState] = {}
_lock = threading.Lock()


def _state(dev: torch.device) -> _DeviceState:
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        st = _states.get(idx)
        if st is None:
            st = _states[idx] = _DeviceState(torch.device("cuda", idx))
        return st


def _use_capacity() -> bool:
    return os.environ.get("GEOSPLAT_CAPACITY", "1") != "0"


def camera_tensors(st: _DeviceState, cam: Camera, stream: torch.cuda.Stream):
    """Device (view matrix, K, position) of a camera, cached by the pose's STORAGE ADDRESS + version counter + intrinsics: the entry
    keeps the pose tensor alive (its address cannot be recycled), an in-place update bumps the version, and `cameras[i]` of a
    resident dataset -- a new tensor object every step over the same storage -- hits.  A miss builds the three tensors on `stream`
    (a front stream: nothing on the caller's stream has to finish first) when the pose is host-resident; a device-resident pose
    may still be in flight on the caller's stream, so its miss is ordered behind that stream."""
    c2w = cam.c2w
    intr = (float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), int(cam.width), int(cam.height))
    host = c2w.device.type == "cpu"
    fp = c2w.detach().numpy().tobytes() if host else None              # 48 bytes: catches writes that bypass the version counter
    key = (c2w.data_ptr(), tuple(c2w.shape), tuple(c2w.stride()), intr)
    hit = st.cam_cache.get(key)
    if hit is not None and hit[1] == c2w._version and hit[3] == fp:
        return hit[2]
    if len(st.cam_cache) >= 4096:
        st.cam_cache.pop(next(iter(st.cam_cache)))
    if not host:
        stream.wait_stream(torch.cuda.current_stream(st.dev))
    with torch.cuda.stream(stream):
        tensors = (cam.view_matrix.to(st.dev, torch.float32).contiguous(), cam.intrinsic_matrix.to(st.dev, torch.float32).contiguous(),
                   c2w.detach()[:, 3].to(st.dev, torch.float32).contiguous())
        ev = torch.cuda.Event(); ev.record(stream)
    for f in st.fronts:                                                 # whichever front stream uses the entry later: ordered behind its upload
        if f is not stream:
            f.wait_event(ev)
    st.cam_cache[key] = (c2w, c2w._version, tensors, fp)
    return tensors


# ------------------------------------------------------------------------------------------------------------ the step
class _View:
    __slots__ = ("cam", "W", "H", "V", "I", "state", "render", "alphas", "last_ids", "log_ws", "tone", "exposure", "exact", "cap",
                 "cap_used", "key_bits", "key_base", "host_counts", "host_status", "event", "done", "index", "N", "status", "over", "i_seen", "reported")


class _Step:
    """The views that read one set of parameter tensors.  Holds detached handles of them (which also keeps their ids unique while
    the step is alive), the activations, and -- between the first view backward and the gather node -- the gradient buffers."""

    def __init__(self, st: _DeviceState, key, tensors: Dict[str, Tensor], env: TextureSplitSum, lut: Tensor, cfg):
        self.st, self.key = st, key
        self.t = tensors                                   # detached, contiguous fp32
        self.env, self.lut, self.cfg = env, lut, cfg       # cfg = (min_roughness, max_metallic, mode)
        from .shading import _make_env
        self.e = _make_env(lut, env)
        with torch.no_grad():
            self.scales_act = tensors["scales"].exp()
            self.opac_act = torch.sigmoid(tensors["opacities"]).reshape(-1).contiguous()
        self.ready = torch.cuda.Event(); self.ready.record()     # parameters, pyramid, activations: final on the caller's stream here
        self.views: List[_View] = []
        self.gathered = None                               # outputs of this step's _Gather node while the step can still be joined
        self.has_gather = False                            # a _Gather node sits in front of this step's views (the parameters need gradients)
        self.pending: List[_View] = []                     # compositor backward done, tail not launched
        self.g = None                                      # gradient buffers of the backward pass in progress
        self.n_tail = 0
        self.sched = None

    # gradient buffers: allocated on the stream of the first view backward, handed out (and forgotten) by the gather node
    def grads(self):
        if self.g is None:
            t, dev = self.t, self.st.dev
            N = t["means"].shape[0]
            z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            sizes = [self.env.base.numel()] + [l.numel() for l in self.env.levels]
            g_flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)     # texel gradients: float atomics add into them
            parts = torch.split(g_flat, sizes)
            g_base = parts[0].view_as(self.env.base)
            g_levels = [q.view_as(l) for q, l in zip(parts[1:], self.env.levels)]
            eg = L.GsEnvGrad(); eg.base = g_base.data_ptr()
            for i, gl in enumerate(g_levels):
                eg.levels[i] = gl.data_ptr()
            self.g = dict(means=z(N, 3), quats=z(N, 4), scales_act=z(N, 3), opac_act=z(N), normals=z(N, 3), kd=z(N, 3), ks=z(N, 2),
                          g_flat=g_flat, base=g_base, levels=g_levels, eg=eg)
            self.n_tail = 0
            tb = os.environ.get("GEOSPLAT_TAIL_BATCH", "auto")
            n = sum(1 for v in self.views if not v.done)
            # batches of TWO views in this call shape (8 views: 2 + 2 + 2 in the background, 2 at the gather node): 612 views/s
            # against 598 for the engine's 3 + 3 + 2, 603 for one view per launch, 586 for one launch of 8 -- here the whole
            # compositor backward phase runs back to back, so smaller background launches keep up with it better
            self.sched = [2] if tb == "auto" else [max(1, int(x)) for x in tb.split(",")]
            ev = torch.cuda.Event(); ev.record()
            self.st.tail.wait_event(ev)                    # (the zero fill above)
            for f in self.st.fronts:
                f.wait_event(ev)
        return self.g

    def discard_pass(self) -> None:
        """Forget the gradient buffers of a backward pass that did not reach the gather node; the launches that wrote them are joined
        to the caller's stream first (their buffers go back to the allocator)."""
        if self.g is None:
            return
        main = torch.cuda.current_stream(self.st.dev)
        main.wait_stream(self.st.tail)
        for f in self.st.fronts:
            main.wait_stream(f)
        for v in self.pending:
            v.state = v.render = v.alphas = v.last_ids = v.log_ws = None
        self.g, self.pending, self.n_tail = None, [], 0

    def end_of_backward(self) -> None:
        if self.g is not None:                             # the gather node did not run in this pass
            self.discard_pass()

    def launch_tail(self, final: bool) -> None:
        """A7 + S1-S3 backward of the pending views in one call pair on the tail stream; `final`: the last launch of this backward
        pass (all CUs, its projection half on a front stream -- idle by now -- beside whatever the caller runs next, e.g. the
        prefilter backward, which needs only the texel gradients)."""
        views, self.pending = self.pending, []
        if not views:
            return
        st, t, g = self.st, self.t, self.g
        mr, mm, mode = self.cfg
        main = torch.cuda.current_stream(st.dev)
        tail = st.tail
        ev = torch.cuda.Event(); ev.record(main)
        tail.wait_event(ev)
        tv = [(v.cam[0], v.cam[1], v.cam[2], v.state["vis_records"], v.state["v_packed"], v.state["packed_index"], v.W, v.H) for v in views]
        split = final
        args = (tv, t["means"], t["quats"], self.scales_act, self.opac_act, t["normals"], t["kd"], t["ks"], self.e, g["eg"], mr, mm, mode,
                g["means"], g["quats"], g["scales_act"], g["opac_act"], g["normals"], g["kd"], g["ks"])
        with torch.cuda.stream(tail):
            F.tail_multi_stage(*args, accumulate=self.n_tail > 0, parts=1 if split else (3 if final else 7))
        used = [tail]
        if split:
            ev_sh = torch.cuda.Event(); ev_sh.record(tail)
            ps = st.fronts[0]
            ps.wait_event(ev_sh)                           # (v_means: the projection half adds to what the shading half stored)
            with torch.cuda.stream(ps):
                F.tail_multi_stage(*args, accumulate=self.n_tail > 0, parts=2)
            used.append(ps)
        self.n_tail += 1
        for v in views:
            for x in (v.state["vis_records"], v.state["v_packed"], v.state["packed_index"]) + tuple(v.cam):
                for s in used:
                    x.record_stream(s)
            v.state = v.render = v.alphas = v.last_ids = v.log_ws = None      # the view's ~1 GB of buffers go back to the allocator
        for name in ("means", "quats", "scales_act", "opac_act", "normals", "kd", "ks", "g_flat"):
            for s in used:
                g[name].record_stream(s)


def _capacity_error(v: "_View") -> GeoSplatCapacityError:
    return GeoSplatCapacityError(
        f"splat(): view {v.index} of a step had {v.i_seen} tile intersections (capacity {v.cap_used}) or left the 24-bit depth-key "
        f"range: its image and gradients come from a truncated list.  The capacity is now {v.cap.i_cap(v.N)}; repeat the step "
        "(viewbatch.retry_on_capacity does; GEOSPLAT_CAPACITY=0 reads the exact counts back for every view instead).")


def _poll_unchecked(st: _DeviceState, wait_views=None) -> None:
    """Look at the counts of capacity-mode views that have finished (all of `wait_views`: waited for): the capacity learns from them,
    a truncated view is MARKED (`over`).  Nothing is raised here: the step that owns the view raises from its gather node."""
    with st.lock:
        keep = collections.deque()
        while st.unchecked:
            v = st.unchecked.popleft()
            if wait_views is not None and any(v is w for w in wait_views):
                v.event.synchronize()
            if not v.event.query():
                keep.append(v)
                continue
            hc = v.host_counts
            v.i_seen = int(hc[1])
            cap = v.cap
            over = v.i_seen > v.cap_used
            if v.key_bits == 24:
                rng = F.depth_range(hc)
                if rng is not None and (rng[0] < v.key_base or rng[1] >= v.key_base + (1 << 24)):
                    over = True
                    cap.key32 = True
            hs = v.host_status
            if hs is not None:                               # the view's OWN device word (nothing shared, nothing to clear)
                if int(hs[0]) != 0 or int(hs[3]) != 0:
                    over = True
                    cap.key32 = cap.key32 or int(hs[3]) != 0
                F.release_counts4(hs)
                v.host_status = None
            cap.learn(hc, v.N)
            F.release_counts4(hc)
            v.host_counts = None
            v.status = None
            v.over = over
        st.unchecked = keep


def _warn_unreported(step: "_Step") -> None:
    """A step that can no longer be joined: truncated views of it that no backward has reported (and none may ever) get a warning."""
    for v in step.views:
        if v.over and not v.reported:
            v.reported = True
            warnings.warn(str(_capacity_error(v)), RuntimeWarning)


def _view_forward(step: _Step, cam: Camera, exposure: Tensor, tone: int, want_grad: bool) -> Tuple[Tensor, _View]:
    st, t = step.st, step.t
    lib = L.lib()
    dev = st.dev
    mr, mm, mode = step.cfg
    W, H = int(cam.width), int(cam.height)
    N = t["means"].shape[0]
    main = torch.cuda.current_stream(dev)
    side = st.fronts[st.n_views % len(st.fronts)]
    st.n_views += 1
    _poll_unchecked(st)                                    # non-blocking, never raises
    with st.lock:
        cap = st.caps.get((W, H))
        if cap is None:
            cap = st.caps[(W, H)] = _Capacity()
            while len(st.caps) > _MAX_CAPS:
                st.caps.popitem(last=False)
        else:
            st.caps.move_to_end((W, H))
        if cap.first_step is None and want_grad:
            cap.first_step = step.key
    # exact: no gradient, protocol off, nothing learnt yet -- or a view of the very step that is teaching the capacity
    exact = (not want_grad) or (not _use_capacity()) or cap.ratio is None or cap.first_step == step.key
    i_cap = None if exact else cap.i_cap(N)
    key_bits, key_base = (32, 0) if exact else cap.keys()
    status = None
    if not exact:
        with torch.cuda.stream(side):                      # the view's own status word: an overflow is attributed to exactly this view
            status = torch.zeros(4, dtype=torch.int64, device=dev)
    tight = os.environ.get("GEOSPLAT_TIGHT_TILES", "1") != "0"
    side.wait_event(step.ready)                            # parameters / pyramid / activations -- NOT the previous view's compositor
    cam_t = camera_tensors(st, cam, side)
    v = _View()
    v.cam, v.W, v.H, v.tone, v.exact, v.cap, v.done, v.index = cam_t, W, H, tone, exact, cap, (False if want_grad else None), len(step.views)
    v.N, v.status, v.over, v.i_seen, v.reported = N, status, False, 0, False
    with torch.cuda.stream(side):
        fr = F.front_stage(t["means"], t["quats"], step.scales_act, step.opac_act, t["normals"], t["kd"], t["ks"], cam_t[0], cam_t[1],
                           cam_t[2], step.e, W, H, mr, mm, mode, key_base, key_bits, status,
                           want_packed_index=want_grad, tight_tiles=tight)
        # the record stream of the compositor (gs_raster_prepare_vis: 0.1 ms, HBM gather) is built on the CALLER's stream, whose
        # compositor forward (0.29 ms per view) leaves it idle for more than half of the forward phase, while the front streams
        # (0.47 ms of front + binning per view) are its critical path (measured: the same 13.3 ms either way)
        split = True
        state, V, I = F.bin_stage(fr, i_cap, status, prepare=not split)
        v_packed = torch.zeros(max(V, 1), lib.gs_raster_grad_stride(3), dtype=torch.float32, device=dev) if want_grad else None
        log_ws = torch.empty(lib.gs_raster_log_ws_bytes(L.i64(I), W, H, 16), dtype=torch.uint8, device=dev)
        v.host_status = None
        if not exact:
            # the device's own word (word 0: a view truncated at its capacity, or a look-back of the front / emission that gave up;
            # word 3: a depth outside the 24-bit key range) follows the view to the host with its counts
            v.host_status = F.pinned_counts4()
            v.host_status.copy_(status, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(side)
    if exact:
        cap.learn(fr.host_counts, N)                       # (bin_stage waited for them)
        F.release_counts4(fr.host_counts)
        v.host_counts = None
    else:
        v.host_counts, v.event, v.cap_used, v.key_bits, v.key_base = fr.host_counts, ev, i_cap, key_bits, key_base
        with st.lock:
            st.unchecked.append(v)
    for x in list(state.values()) + [log_ws, v_packed, step.scales_act, step.opac_act] + list(cam_t):
        if isinstance(x, Tensor):
            x.record_stream(main)
    main.wait_event(ev)
    if split:
        for x in (fr.vis, fr.counts, fr.packed_index):
            if x is not None:
                x.record_stream(main)
        state, V, I = F.bin_stage(fr, i_cap, status, binned=(state["flatten_ids"], state["isect_offsets"]))
    if v_packed is not None:
        state["v_packed"] = v_packed
    f32 = torch.float32
    render = torch.empty(H, W, 3, dtype=f32, device=dev); alphas = torch.empty(H, W, dtype=f32, device=dev)
    last_ids = torch.empty(H, W, dtype=torch.int32, device=dev); img = torch.empty(H, W, 4, dtype=f32, device=dev)
    rws = state["raster_ws"]
    L.check(lib.gs_raster_composite_tone_log(W, H, 16, V, None, L.i64(I), L.ptr(state["counts"]), L.ptr(state["isect_offsets"]),
                                             L.ptr(render), L.ptr(alphas), L.ptr(last_ids), tone, L.ptr(exposure), L.ptr(img), L.ptr(rws),
                                             C.c_size_t(rws.numel()), L.ptr(log_ws), C.c_size_t(log_ws.numel()), L.stream()),
            "gs_raster_composite_tone_log")
    v.V, v.I, v.state, v.render, v.alphas, v.last_ids, v.log_ws, v.exposure = V, I, state, render, alphas, last_ids, log_ws, exposure
    return img, v


def _view_backward(step: _Step, v: _View, v_img: Tensor) -> Tensor:
    """A6 + S4 backward of one view on the current stream; queues the view for a tail launch.  Returns d loss / d exposure."""
    lib = L.lib()
    s = v.state
    if s is None:
        raise L.GeoSplatHipError("splat(): backward through a view a second time (its buffers were released after the first; "
                                 "render the view again instead of retain_graph)")
    v_img = v_img.contiguous().float()
    g_exp = torch.zeros(1, dtype=torch.float32, device=step.st.dev)
    rws = s["raster_ws"]
    gather = step.has_gather                               # (False: only the exposure needs a gradient -- no tail, no gather node)
    if gather:
        if step.g is None:
            # the gradient buffers belong to THIS backward pass: if its gather node never runs (autograd.grad w.r.t. the exposure only,
            # an exception in a later node) the end-of-pass callback drops what was accumulated, so that the next pass starts clean
            torch.autograd.Variable._execution_engine.queue_callback(step.end_of_backward)
        step.grads()
    try:
        return _view_backward_launch(step, v, v_img, g_exp, gather)
    except BaseException:
        step.discard_pass()
        raise


def _view_backward_launch(step: _Step, v: _View, v_img: Tensor, g_exp: Tensor, gather: bool) -> Tensor:
    lib = L.lib()
    s = v.state
    rws = s["raster_ws"]
    L.check(lib.gs_raster_bwd_tone_log_acc(v.W, v.H, 16, v.V, None, L.i64(v.I), L.ptr(s["counts"]), L.ptr(s["isect_offsets"]),
                                           L.ptr(v.render), L.ptr(v.alphas), L.ptr(v.last_ids), v.tone, L.ptr(v.exposure), L.ptr(v_img),
                                           L.ptr(s["v_packed"]), L.ptr(g_exp), L.ptr(rws), C.c_size_t(rws.numel()), L.ptr(v.log_ws),
                                           C.c_size_t(v.log_ws.numel()), L.stream()), "gs_raster_bwd_tone_log_acc")
    v.done = True
    if not gather:
        v.state = v.render = v.alphas = v.last_ids = v.log_ws = None
        return g_exp
    step.pending.append(v)
    k = step.sched[min(step.n_tail, len(step.sched) - 1)]
    remaining = sum(1 for w in step.views if not w.done)
    if len(step.pending) >= k and remaining > 0:           # (the last batch belongs to the gather node: it knows that it IS the last)
        step.launch_tail(final=False)
    return g_exp


_PARAMS = ("means", "scales", "quats", "opacities", "normals", "kd", "ks")


class _Gather(torch.autograd.Function):
    """Identity on the parameter tensors of a step; its backward is the one place where the per-view tails end and the sums over
    the views become the gradients of the parameters (module docstring)."""

    @staticmethod
    def forward(ctx, step: _Step, *tensors: Tensor):
        ctx.step = step
        ctx.shapes = [tuple(x.shape) for x in tensors]
        ctx.set_materialize_grads(False)
        return tuple(x.detach() for x in tensors)

    @staticmethod
    def backward(ctx, *unused):
        step: _Step = ctx.step
        st = step.st
        if step.g is None:                                  # no view of this step took part in this backward pass
            return (None,) * (1 + len(ctx.shapes))
        step.launch_tail(final=True)
        g, step.g = step.g, None
        main = torch.cuda.current_stream(st.dev)
        done = [v for v in step.views if v.done]
        tail_stream = st.fronts[0]
        tail_stream.wait_stream(st.tail)
        with torch.cuda.stream(tail_stream):                # behind the projection half; chains the once-per-step activations
            g_scales = g["scales_act"] * step.scales_act
            g_opac = (g["opac_act"] * step.opac_act * (1.0 - step.opac_act)).reshape(ctx.shapes[3])
        main.wait_stream(st.tail)
        main.wait_stream(tail_stream)
        for x in (g_scales, g_opac):
            x.record_stream(main)
        for x in (g["scales_act"], g["opac_act"], step.scales_act, step.opac_act):
            x.record_stream(tail_stream)
        # the fronts of this step finished long ago: looking at their counts now costs no GPU idle time, and a truncated view
        # stops the step HERE, before an optimizer can consume it
        _poll_unchecked(st, wait_views=[v for v in done if v.host_counts is not None])
        bad = [v for v in done if v.over and not v.reported]
        if bad:
            for v in bad:
                v.reported = True
            raise _capacity_error(bad[0])
        out = [g["means"], g_scales, g["quats"], g_opac, g["normals"], g["kd"], g["ks"], g["base"]] + list(g["levels"])
        return (None, *[x.reshape(s) for x, s in zip(out, ctx.shapes)])


class _SplatView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, step: _Step, cam: Camera, tone: int, exposure: Tensor, *gathered: Tensor):
        exposure_d = exposure.detach().reshape(1).contiguous().float()
        img, view = _view_forward(step, cam, exposure_d, tone, True)
        step.views.append(view)
        ctx.step, ctx.view, ctx.n, ctx.exp_shape = step, view, len(gathered), tuple(exposure.shape)
        ctx.set_materialize_grads(False)
        return img

    @staticmethod
    def backward(ctx, v_img):
        none = (None,) * (4 + ctx.n)
        if v_img is None:
            return none
        g_exp = _view_backward(ctx.step, ctx.view, v_img)
        return (None, None, None, g_exp.reshape(ctx.exp_shape)) + (None,) * ctx.n


def splat_view(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, normals: Tensor, kd: Tensor, ks: Tensor, camera: Camera,
               exposure: Tensor, envmap: TextureSplitSum, lut: Tensor, min_roughness: float, max_metallic: float, mode: int,
               tone: int) -> Tensor:
    """One view of ``RenderableAttrs.splat`` (rfstudio/model/geosplat.py:53-132) -> tone-mapped RGBA [H, W, 4], differentiable
    w.r.t. the seven per-Gaussian tensors, the exposure and the pyramid (base + levels)."""
    raw = dict(zip(_PARAMS, (means, scales, quats, opacities, normals, kd, ks)))
    L.require_cuda(*raw.values())
    st = _state(means.device)
    srcs = list(raw.values()) + [envmap.base] + list(envmap.levels)
    cfg = (float(min_roughness), float(max_metallic), int(mode))
    key = (tuple((id(x), x._version, x.data_ptr()) for x in srcs), cfg, id(lut), float(envmap.min_roughness), float(envmap.max_roughness),
           torch.is_grad_enabled() and any(x.requires_grad for x in srcs))
    exposure = torch.as_tensor(exposure, dtype=torch.float32, device=st.dev)
    params_grad = key[-1]
    want_grad = torch.is_grad_enabled() and (params_grad or exposure.requires_grad)
    step = st.current
    if step is None or step.key != key:
        # contiguous fp32 handles; .contiguous()/.float() are autograd ops when they copy, so the gather node sits behind them
        conv = [x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous() for x in srcs]
        det = dict(zip(_PARAMS, (x.detach() for x in conv[:7])))
        env_d = TextureSplitSum(conv[7].detach(), [x.detach() for x in conv[8:]], envmap.min_roughness, envmap.max_roughness)
        if step is not None:
            _warn_unreported(step)
            # the step that can no longer be joined lets go of its gather outputs: step -> outputs -> grad_fn -> ctx -> step is a cycle
            # through C++ ownership that Python's collector cannot see (56 MB per step -- activations + pyramid -- leaked without this;
            # scripts/soak_callshape.py).  Its view nodes keep their own edges to the gather node.
            step.gathered = None
            step.srcs = None
        step = _Step(st, key, det, env_d, lut, cfg)
        step.srcs = srcs                                   # keeps the ids in `key` unique while the step can be joined
        step.has_gather = params_grad
        step.gathered = _Gather.apply(step, *conv) if params_grad else None
        st.current = step
    if not want_grad:
        img, _ = _view_forward(step, camera, exposure.detach().reshape(1).contiguous(), tone, False)
        return img
    return _SplatView.apply(step, camera, tone, exposure, *(step.gathered or ()))


def reset() -> None:
    """Forget the joinable step and the learnt capacities (tests; after a checkpoint load through `.data`)."""
    for st in _states.values():
        with st.lock:
            if st.current is not None:
                st.current.gathered = None
            st.current = None
            st.caps.clear()


def retry_on_capacity(step_fn, max_retries: int = 2):
    """Wrap a training-step function `step_fn(*a, **k)` (forward: the loop of `splat()` calls; `loss.backward()`; NO optimizer step
    inside, or only behind the backward) so that a GeoSplatCapacityError -- a view had more tile intersections than the capacity the
    earlier steps taught -- repeats the step: the capacity has been raised when the error is raised, gradients accumulated by the
    failed attempt must be dropped by the caller's usual `zero_grad` inside `step_fn`."""
    def wrapped(*a, **k):
        for attempt in range(max_retries + 1):
            try:
                return step_fn(*a, **k)
            except GeoSplatCapacityError:
                if attempt == max_retries:
                    raise
    return wrapped
