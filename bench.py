#!/usr/bin/env python3
"""bench.py -- forward+backward views/sec of the GeoSplatting render path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
prints ONE JSON line on rank 0.

Workload (BASELINE.json `metric`: "fwd+bwd views/sec at 2M Gaussians, 800x800"):
  synthetic surface splats = MGAdapter restatement on a bumpy icosphere level 7 -> 1 966 080 Gaussians, 800x800,
  Blender-style cameras, seeded 6x512^2 HDR cubemap.  One STEP = what one training step of the
  reference renders and back-propagates on one device (rfstudio/model/geosplat.py:856-879, batch_size 8,
  tests/model/test_geosplat.py:28):
      split-sum prefilter forward (S5) -> 8 x [shade -> project/bin/sort/composite -> tone-map, then the backward of
      all of it] -> prefilter backward -> (N>1) one flat RCCL all-reduce of all parameter gradients.
  DEFAULT = the north star's split (BASELINE.json north_star / config 4): STRONG scaling -- the reference batch of 8 views is
  spread over the N GPUs, view i -> rank i mod N (one view per GPU at N = 8); value = 8 / (max-over-ranks step time); a rank with
  at most two views replays them as one HIP graph, prefilter and collectives stay eager around it.  At N = 1 this is the same
  workload as before (8 views on the one GPU).  --views-total V changes the batch; --weak restores weak scaling (every GPU
  renders its own --views views, value = views * N / step time).  With N > 1 the split-sum prefilter is sharded over the ranks
  (geosplatting_amd/splitsum.py) in both modes.  `scale_model` in the line is the prediction for N = 1, 2, 4, 8 from the
  single-GPU measurements (view alone, prefilter alone, bytes per collective over xGMI) that a hardware curve is to be read against.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed steps for this long BEFORE the W warm-up steps: a GPU that has idled starts a process at low clocks "
                         "(the first bench run on a fresh box read 10 %% below every later one); part of initialisation, like the table build")
    ap.add_argument("--level", type=int, default=7, help="icosphere level: 7 -> 1 966 080 Gaussians, 6 -> 491 520")
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--views", type=int, default=8, help="views per step per GPU (reference batch_size = 8)")
    ap.add_argument("--views-total", type=int, default=0,
                    help="strong scaling (default): this many views per step over ALL GPUs (view i -> rank i mod N); 0 = --views (8)")
    ap.add_argument("--weak", action="store_true", help="weak scaling instead: every GPU renders its own --views views per step")
    ap.add_argument("--cubemap-res", type=int, default=512)
    ap.add_argument("--no-prefilter", action="store_true", help="diagnostic only: keep the pyramid fixed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-call-shaped", action="store_true", help="skip the `call_shaped` leg (the same step through RenderableAttrs.splat + autograd)")
    ap.add_argument("--graph", type=int, default=-1, help="-1 (default): 2 when a rank renders at most two views per step, else 0; 1: replay each step as one HIP graph (--gpus 1 only; measured 12 % slower than eager launches at 8 views per GPU: a graph serialises what the eager queues overlap); 2: only the VIEWS of a step as a graph, prefilter and collectives eager (any --gpus; for few views per GPU, e.g. --views-total 8 on 8 GPUs); 0 (default): eager launches")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--rccl-world1", action="store_true",
                    help="diagnostic: run the step with EVERY collective of the multi-GPU path through RCCL on a one-rank process group "
                         "(sharded prefilter + its communicator, two-phase gradient all-reduce, graph replay beside them)")
    return ap.parse_args()


def algorithmic_bytes_per_view(N, V, I, P, E):
    """228 N + 184 V + 120 I + 44 P + 2 E (fp32, D=3, ideal single-pass sort)."""
    return 228 * N + 184 * V + 120 * I + 44 * P + 2 * E


FLOPS_FWD_PER_PAIR = 37      # dx,dy 2 | sigma 8 | canonical exp 16 | alpha 2 | T 3 | colour FMAs 6
FLOPS_BWD_PER_PAIR = 76      # sigma/exp/alpha 28 | 1/(1-alpha), T, fac 3 | v_alpha 18 | v_sigma 2 | buffer 6 | moment reduction 19
VALU_PEAK_TFLOPS = 157.3     # FP32 vector peak, MI355X_MICROARCH.md


def time_dominant_kernels(params, env, cam, res, iters):
    """Durations of the compositor kernels the ENGINE launches, each ALONE, on the engine's own inputs (fused front with its tile
    rectangles, binning from its outputs), with HIP events on the stream they are launched on (torch's current stream):
    `gs_raster_composite_tone_log` (raster_fwd_window_kernel with S4 in its epilogue, writing the cull log) on a prepared workspace
    and `gs_raster_bwd_tone_log_acc` (raster_bwd_log_kernel); the workspace preparation (sorted record stream, tile order) is timed
    as its own entry.  Returns (times, I of the engine's list)."""
    import geosplatting_amd as gs
    import geosplatting_amd._lib as L
    from geosplatting_amd import front as F
    from geosplatting_amd.shading import _MODE, _make_env
    lib = L.lib()
    dev = params.means.device
    W = H = res
    f32 = torch.float32
    use_log = True
    tight = os.environ.get("GEOSPLAT_TIGHT_TILES", "1") != "0"
    env_d = gs.TextureSplitSum(env.base.detach(), [l.detach().contiguous() for l in env.levels], env.min_roughness, env.max_roughness)
    e = _make_env(gs.get_fg_lut(dev), env_d)
    d = lambda t: t.to(dev, f32).contiguous()
    fr = F.front_stage(params.means, params.quats, params.scales.exp(), torch.sigmoid(params.opacities).squeeze(-1).contiguous(), params.normals,
                       params.kd, params.ks, d(cam.view_matrix), d(cam.intrinsic_matrix), d(cam.c2w[:, 3]), e, W, H, 0.1, 1.0, _MODE["pbr"],
                       tight_tiles=tight)
    state, V, I = F.bin_stage(fr, None, None)
    offsets, flat, rws = state["isect_offsets"], state["flatten_ids"], state["raster_ws"]
    rws_bytes = rws.numel()
    render = torch.empty(H, W, 3, dtype=f32, device=dev); alphas = torch.empty(H, W, dtype=f32, device=dev)
    last = torch.empty(H, W, dtype=torch.int32, device=dev); img = torch.empty(H, W, 4, dtype=f32, device=dev)
    v_img = torch.rand(H, W, 4, device=dev) * 2 - 1
    exposure = torch.ones(1, device=dev); v_exp = torch.zeros(1, device=dev)
    v_packed = torch.zeros(max(V, 1), lib.gs_raster_grad_stride(3), dtype=f32, device=dev)
    s = L.stream()
    log_ws = torch.empty(lib.gs_raster_log_ws_bytes(L.i64(I), W, H, 16), dtype=torch.uint8, device=dev)

    def prep():
        L.check(lib.gs_raster_prepare_vis(W, H, 16, 3, V, L.ptr(fr.vis), L.i64(I), L.ptr(offsets), L.ptr(flat), L.ptr(rws), C.c_size_t(rws_bytes), s),
                "raster_prepare_vis")

    def fwd():
        if use_log:
            L.check(lib.gs_raster_composite_tone_log(W, H, 16, V, None, L.i64(I), None, L.ptr(offsets), L.ptr(render), L.ptr(alphas),
                                                     L.ptr(last), 1, L.ptr(exposure), L.ptr(img), L.ptr(rws), C.c_size_t(rws_bytes),
                                                     L.ptr(log_ws), C.c_size_t(log_ws.numel()), s), "raster_composite_tone_log")
        else:
            L.check(lib.gs_raster_composite_tone(W, H, 16, V, None, L.i64(I), None, L.ptr(offsets), L.ptr(render), L.ptr(alphas),
                                                 L.ptr(last), 1, L.ptr(exposure), L.ptr(img), L.ptr(rws), C.c_size_t(rws_bytes), s),
                    "raster_composite_tone")

    def bwd():
        if use_log:
            L.check(lib.gs_raster_bwd_tone_log_acc(W, H, 16, V, None, L.i64(I), None, L.ptr(offsets), L.ptr(render), L.ptr(alphas),
                                                   L.ptr(last), 1, L.ptr(exposure), L.ptr(v_img), L.ptr(v_packed), L.ptr(v_exp), L.ptr(rws),
                                                   C.c_size_t(rws_bytes), L.ptr(log_ws), C.c_size_t(log_ws.numel()), s),
                    "raster_bwd_tone_log_acc")
        else:
            L.check(lib.gs_raster_bwd_tone_acc(W, H, 16, V, None, L.i64(I), None, L.ptr(offsets), L.ptr(render), L.ptr(alphas),
                                               L.ptr(last), 1, L.ptr(exposure), L.ptr(v_img), L.ptr(v_packed), L.ptr(v_exp), L.ptr(rws),
                                               C.c_size_t(rws_bytes), s), "raster_bwd_tone_acc")
    out = {}
    for name, fn in (("raster_prepare (stream build)", prep), ("raster_fwd_kernel", fwd), ("raster_bwd_kernel", bwd)):
        if fn is bwd:
            prep(); fwd()                              # one forward behind one prepare: the state the backward reads (log, tile order)
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters       # ms per launch
    return out, I


def time_view_without_prefilter(params, cam, up, iters):
    """GPU time of ONE view fwd+bwd with the pyramid held fixed (shade + project + bin + sort + composite + tone map and the
    whole backward): the same work the cpu_baseline sample times, so the two are comparable.  Measured twice: launched eagerly
    (~45 launches from Python: host-bound on a slow host core) and as ONE HIP graph replay (engine.RenderStep.capture, possible
    because the capacity protocol leaves no host synchronisation in a step) -- the BASELINE config 4 case, one view per GPU."""
    from geosplatting_amd.engine import RenderStep
    step = RenderStep(params, prefilter=False)
    fn = lambda: step([cam], lambda i, img: up, all_reduce=False)

    def clock(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    for _ in range(2):
        fn()
    out = {"eager_ms": clock(fn), "graph_ms": None}
    try:
        if step.poll_capacity(wait=True) and step._i_cap is not None:
            # the LAST (here: only) tail launch alone: shading half, then projection half -- what the multi-GPU step chunks and overlaps
            # with its all-reduce (engine._chunked_tail)
            step.tail_events = []
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            tev, step.tail_events = step.tail_events, None
            out["tail_shade_ms"] = sum(a.elapsed_time(b) for a, b, _ in tev) / len(tev)
            out["tail_proj_ms"] = sum(b.elapsed_time(c) for _, b, c in tev) / len(tev)
            replay = step.capture([cam], lambda i, img: up)
            replay(); replay()
            out["graph_ms"] = clock(replay)
            if not replay.check():
                out["graph_ms"] = None
    except Exception as e:                                   # the diagnostic must never take the bench line down
        out["graph_error"] = repr(e)[:200]
    # the three graphs of the multi-GPU shape (engine.capture_views: geometry | records | views), each replayed ALONE: the terms of the
    # strong-scaling model's critical path
    try:
        if out.get("graph_ms"):
            step2 = RenderStep(params, prefilter=True)
            for _ in range(2):
                step2([cam], lambda i, img: up, all_reduce=False)
            if step2.poll_capacity(wait=True) and step2._i_cap is not None:
                seg = step2.capture_views([cam], lambda i, img: up, all_reduce=False)
                seg(); seg()
                torch.cuda.synchronize()
                if seg.geo_graph is not None:
                    out["geo_graph_ms"] = clock(seg.geo_graph.replay)
                    out["rec_graph_ms"] = clock(seg.rec_graph.replay)
                    out["views_graph_ms"] = clock(seg.graph.replay)
    except Exception as e:
        out["segments_error"] = repr(e)[:200]
    return out


def time_call_shaped(params, cams, ups, steps, warmup):
    """The SAME step through the reference's call shape (rfstudio/model/geosplat.py:863-879 + rfstudio/optim/optimizer.py:107): the
    prefilter through autograd, a Python loop of RenderableAttrs.splat() over the views, a loss, ONE backward() into the leaves'
    .grad -- no engine object, no callback (geosplatting_amd/viewbatch.py is what splat() runs).  The loss is sum_i <image_i, w_i>
    with the cotangents w_i the engine's timed steps use, so both paths do the same work."""
    import geosplatting_amd as gs
    leaf = lambda t: t.detach().clone().requires_grad_(True)

    class G:
        pass
    g = G(); g.means, g.scales, g.quats, g.opacities = leaf(params.means), leaf(params.scales), leaf(params.quats), leaf(params.opacities)
    attrs = gs.RenderableAttrs(kd=leaf(params.kd), ks=leaf(params.ks), normals=leaf(params.normals))
    cubemap, exposure = leaf(params.cubemap), leaf(params.exposure)
    leaves = [g.means, g.scales, g.quats, g.opacities, attrs.kd, attrs.ks, attrs.normals, cubemap, exposure]

    def step():
        for t in leaves:
            t.grad = None                                    # optimizer.zero_grad(set_to_none=True)
        env = gs.as_splitsum(cubemap)
        images = [attrs.splat(g, [cam], exposure=exposure, envmap=env, min_roughness=0.1, max_metallic=1.0) for cam in cams]
        loss = images[0].new_zeros(())
        for img, w in zip(images, ups):
            loss = loss + torch.dot(img.reshape(-1), w.reshape(-1))      # <img, w>: one launch forward, one backward
        loss.backward()
    for _ in range(max(2, warmup)):                          # (first step: exact counts; from the second on the capacity protocol)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cap = gs.viewbatch._state(params.means.device).caps.get((cams[0].width, cams[0].height))
    return {"views_per_s": len(cams) * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "views_per_step": len(cams),
            "n_isects_cap": None if cap is None else cap.i_cap(params.means.shape[0]),
            "what": "as_splitsum(cubemap) [autograd] -> for cam in batch: RenderableAttrs.splat(...) -> loss = sum <img, w> -> ONE "
                    "loss.backward() into .grad of means/scales/quats/opacities/kd/ks/normals/cubemap/exposure; same kernels as `value` "
                    "(fused front, cull-log compositor, batched tails), driven by autograd instead of engine.RenderStep"}


def time_d14(params, cam, res, iters):
    """D = 14 deferred rasterization (rfstudio/model/geosplat.py:276-295: 14 feature channels through gsplat.rasterization) forward +
    backward of one view through `geosplatting_amd.rasterization` -- the D > 3 path (the same compositor pair as D <= 3 since round 5:
    raster_fwd_window_kernel<16> / raster_bwd_lanes2_kernel<16> with the colours of a dense batch staged in LDS planes), never on the
    headline path; timed once per bench run so that it has a number."""
    import geosplatting_amd as gs
    dev = params.means.device
    g = torch.Generator().manual_seed(14)
    feats = torch.rand(params.means.shape[0], 14, generator=g).to(dev).requires_grad_(True)
    means = params.means.detach().clone().requires_grad_(True)
    scales = params.scales.exp(); opac = torch.sigmoid(params.opacities).squeeze(-1)
    vm, K = cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None]

    def one():
        r, a, _ = gs.rasterization(means, params.quats, scales, opac, feats, vm, K, res, res)
        (r.sum() + a.sum()).backward()
        means.grad = None; feats.grad = None
    one(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms_per_view_fwd_bwd": ms, "views_per_s": 1e3 / ms, "D": 14, "iters": iters,
            "what": "rasterization(colors=[N,14]) + backward, one view, exact counts (one read-back), op-by-op path"}


def file_sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


LAUNCHED_AS = {"raster_fwd_kernel": "raster_fwd_window_kernel<3>",
               "raster_bwd_kernel": "raster_bwd_log_kernel<3>"}


def committed_profile(name, kernel_source):
    """A committed profile summary (profiles/<name>) is only quoted while the kernel source it was measured on is unchanged:
    the file records the sha256[:16] of that source."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        j = json.load(open(path))
        if j.get("source_sha16") != file_sha16(os.path.join(ROOT, "geosplatting_amd", "csrc", kernel_source)):
            return None
        return j
    except Exception:
        return None


def prefilter_report(cubemap, iters):
    """S5 timed alone (forward and explicit backward of the whole pyramid, HIP events on the launch stream) and the bytes it moves:
    the tiled pair-weight tables are read once per direction and step -- every row by the eight mirror workgroups of its tile
    (L2 side), once from HBM."""
    import geosplatting_amd as gs
    from geosplatting_amd import splitsum as ss
    dev = cubemap.device
    with torch.no_grad():
        env = gs.as_splitsum(cubemap)
    gb = torch.rand_like(env.base); gl = [torch.rand_like(l) for l in env.levels]
    out = {}
    for name, fn in (("fwd_ms", lambda: gs.as_splitsum(cubemap)), ("bwd_ms", lambda: ss.as_splitsum_backward(gb, gl))):
        with torch.no_grad():
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters
    n = len(env.levels)
    rows = l2 = resident = pairs = 0
    for lvl, rough in zip(env.levels, ss._level_roughness(n, env.min_roughness, env.max_roughness)):
        e = ss.specular_tiles(int(lvl.shape[1]), rough, 0.99, dev)
        if e is None:
            continue
        for d in ("fwd", "bwd"):
            rows += e[d]["rows"]
            l2 += e[d]["rows"] * 260 * e["n_mirrors"]                    # 64 weights + one descriptor per row, per mirror workgroup
            resident += e[d]["weights"].numel() * 4 + e[d]["desc"].numel() * 4
            pairs += e[d]["pairs"] * e["n_mirrors"]
    pyramid = sum(l.numel() * 4 for l in env.levels) + env.base.numel() * 4
    out.update({"table_bytes_resident": resident, "table_bytes_from_hbm_per_step": rows * 260, "table_bytes_through_l2_per_step": l2,
                "pairs_per_step": pairs, "pyramid_bytes": pyramid,
                "note": "round 2 streamed 11.8 GB of per-texel weight tables per step (13 GB resident)"})
    return out


XGMI_LINK_GBS = 153.0       # one xGMI link, one direction (MI355X: 7 links per GPU, fully connected 8-GPU node)


def scale_model(view_ms, pre, n_gauss, cubemap_res, views_total, measured_1gpu_ms, detail=None, tail_chunks=4):
    """Prediction of the STRONG-scaling step (views_total views over G GPUs, view i -> rank i mod G) from single-GPU measurements of
    THIS run, to be read beside a hardware curve that the build loop never had: every term is a timed quantity or bytes / link rate.
    all-reduce of S bytes over G fully connected GPUs as reduce-scatter + all-gather with every peer link in use:
    2 (G-1)/G S / ((G-1) x 153 GB/s); on ONE link (a ring): x (G-1).
    Critical path of a rank with ONE view (engine.capture_views + _finish, what the code does):
        max( geometry graph ,  prefilter forward / G + pyramid all-reduce + records graph )        <- three graphs, two streams
      + views graph without its last tail                                                           <- stream build .. compositor backward
      + last tail in `tail_chunks` Gaussian ranges, chunk k's all-reduce under chunk k + 1's tail   <- _chunked_tail
        beside: (behind the last shading half) texel all-reduce + prefilter backward / G + cubemap-piece all-reduce
    Ranks with several views: views x (one view alone), the same ends (pessimistic: the fronts of a rank's views overlap its
    compositor, as on one GPU)."""
    if view_ms is None:
        return None
    d = detail or {}
    res, tiled = cubemap_res, []
    while res >= 16:
        tiled.append(res); res //= 2
    pyramid = sum(6 * r * r * 3 * 4 for r in tiled)                          # every level goes through the tiled operator (R >= 16)
    texel = pyramid + 6 * 16 * 16 * 3 * 4                                    # base + every level: one flat buffer
    per_gauss = 76 * n_gauss
    pre_fwd = pre["fwd_ms"] if pre else 0.0
    pre_bwd = pre["bwd_ms"] if pre else 0.0
    coll = {"pyramid_allreduce_bytes": pyramid, "texel_grad_allreduce_bytes": texel, "per_gaussian_grad_allreduce_bytes": per_gauss,
            "cubemap_grad_pieces_allreduce_bytes": pyramid, "exposure_allreduce_bytes": 4, "tail_chunks": tail_chunks}
    t_geo, t_rec, t_views = d.get("geo_graph_ms"), d.get("rec_graph_ms"), d.get("views_graph_ms")
    t_sh, t_pj = d.get("tail_shade_ms"), d.get("tail_proj_ms")
    have = all(x is not None for x in (t_geo, t_rec, t_views, t_sh, t_pj))
    rows = []
    for G in (1, 2, 4, 8):
        vpr = -(-views_total // G)
        if G == 1:
            rows.append({"gpus": 1, "views_per_gpu": vpr, "step_ms": measured_1gpu_ms,
                         "views_per_s": None if not measured_1gpu_ms else views_total / measured_1gpu_ms * 1e3,
                         "basis": "measured (this run)" if measured_1gpu_ms else "not measured in this run"})
            continue
        row = {"gpus": G, "views_per_gpu": vpr}
        for links, tag in ((G - 1, ""), (1, "_one_link_ring")):
            t = lambda S: 2.0 * (G - 1) / G * S / (links * XGMI_LINK_GBS * 1e9) * 1e3        # ms
            if have:
                tail = t_sh + t_pj
                head = max(t_geo, pre_fwd / G + t(pyramid) + t_rec)
                body = (t_views - tail) + (vpr - 1) * view_ms
                a, r = tail / tail_chunks, t(per_gauss) / tail_chunks
                tail_comm = (tail + r) if r <= a else (a + tail_chunks * r)                    # chunk pipeline: tail k + 1 over all-reduce k
                ends = max(tail_comm, t_sh + t(texel) + pre_bwd / G + t(pyramid))
                step = head + body + ends
                row.update({"head_ms" + tag: head, "body_ms" + tag: body, "ends_ms" + tag: ends})
            else:                                                                               # (no segment timings: the serial sum)
                step = vpr * view_ms + pre_fwd / G + 2 * t(pyramid) + t(texel) + max(pre_bwd / G, t(per_gauss))
            row.update({"step_ms" + tag: step, "views_per_s" + tag: views_total / step * 1e3,
                        "speedup_vs_measured_1gpu" + tag: None if not measured_1gpu_ms else measured_1gpu_ms / step})
        rows.append(row)
    return {"link_GBs": XGMI_LINK_GBS, "collectives": coll, "view_ms_alone": view_ms, "prefilter_fwd_ms": pre_fwd, "prefilter_bwd_ms": pre_bwd,
            "segments_ms": {k: d.get(k) for k in ("geo_graph_ms", "rec_graph_ms", "views_graph_ms", "tail_shade_ms", "tail_proj_ms")},
            "critical_path_model": have,
            "rows": rows, "note": "prediction from this run's single-GPU timings, not a measurement (no multi-GPU box was ever available to the "
                                  "build loop): latency of the collectives (tens of microseconds each), rank skew and the host's eager launches "
                                  "around the graph replays are not modelled"}


def cpu_baseline_cfg1():
    """BASELINE.json configs[0]: 10k random Gaussians, 256x256, 4 orbit views, fwd+bwd on the CPU oracle (rasterizer only, as
    that config has no shading); median of 5 runs after one warm-up."""
    import numpy as np
    import oracle
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.cameras import orbit_cameras
    sp = syn.random_splats(10000, seed=1)
    means, quats = sp.means.numpy(), sp.quats.numpy()
    scales, opac = sp.scales.exp().numpy(), torch.sigmoid(sp.opacities).squeeze(-1).numpy()
    col = sp.colors.numpy()
    cams = orbit_cameras(4, 3.0, 30.0, 256, 256, hfov_degree=40.0)
    g = torch.Generator().manual_seed(0)
    v = (torch.rand(256, 256, 4, generator=g) * 2 - 1).numpy()
    times = []
    for rep in range(6):
        t0 = time.perf_counter()
        for cam in cams:
            vm, K = cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy()
            m = oracle.rasterization(means, quats, scales, opac, col, vm, K, 256, 256)
            oracle.rasterization_bwd(means, quats, scales, opac, col, vm, K, 256, 256, m, v[..., :3], v[..., 3])
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[2]
    return {"value": 4.0 / med, "unit": "views/s", "views": 4, "N": 10000, "res": 256, "median_s_per_4_views": med,
            "sample": "BASELINE.json configs[0] in full: 4 views fwd+bwd, median of 5 after 1 warm-up"}


def cpu_baseline(scene, cam, res, budget_note):
    """The CPU oracle (plain-C restatement, OpenMP over host cores) timed on ONE view fwd+bwd of the same workload."""
    import numpy as np
    import geosplatting_amd as gs
    import oracle
    cores = os.cpu_count() or 1
    sp = scene.splats
    means, quats = sp.means.numpy(), sp.quats.numpy()
    # activations evaluated by the same torch-on-GPU ops the product runs (a host libm exp differs in the last bit of a few
    # scales, and the parity figure below is about the path, not about two exp implementations)
    adev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    scales = sp.scales.to(adev).exp().cpu().numpy()
    opac = torch.sigmoid(sp.opacities.to(adev)).squeeze(-1).cpu().numpy()
    lut = gs.get_fg_lut(torch.device("cpu"))[0].numpy()
    # pyramid: the prefilter is excluded from the CPU sample (it is once per 8 views and its oracle is O(minutes) at 512^2);
    # a seeded random pyramid of the right shape stands in -- texture values do not change the work done.
    g = torch.Generator().manual_seed(0)
    levels = [torch.rand(6, r, r, 3, generator=g).numpy() for r in (512, 256, 128, 64, 32, 16)]
    base = torch.rand(6, 16, 16, 3, generator=g).numpy()
    vm, K = cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy()
    v = (torch.rand(res, res, 4, generator=g) * 2 - 1).numpy()
    t0 = time.time()
    col = oracle.shade_fwd(means, scene.normals.numpy(), scene.kd.numpy(), scene.ks.numpy(), cam.c2w[:, 3].numpy(), lut,
                           base, levels)
    m = oracle.rasterization(means, quats, scales, opac, col, vm, K, res, res)
    rgba = np.concatenate([m["render"], m["alphas"][..., None]], -1)
    img_ref = oracle.tonemap_fwd(rgba, 1.0, "naive")
    v_rgba, _ = oracle.tonemap_bwd(rgba, 1.0, v, "naive")
    gr = oracle.rasterization_bwd(means, quats, scales, opac, col, vm, K, res, res, m, v_rgba[..., :3], v_rgba[..., 3])
    oracle.shade_bwd(means, scene.normals.numpy(), scene.kd.numpy(), scene.ks.numpy(), cam.c2w[:, 3].numpy(), lut, base,
                     levels, gr["v_colors"])
    dt = time.time() - t0
    # "PSNR vs ref" half of the metric: the HIP path on the same view / same pyramid against the oracle's image
    parity = None
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        with torch.no_grad():
            env = gs.TextureSplitSum(torch.from_numpy(base).to(dev), [torch.from_numpy(l).to(dev) for l in levels])
            attrs = gs.RenderableAttrs(kd=scene.kd.to(dev), ks=scene.ks.to(dev), normals=scene.normals.to(dev))
            img = attrs.splat(sp.to(dev), [cam], exposure=torch.tensor(1.0, device=dev), envmap=env, min_roughness=0.1,
                              max_metallic=1.0).reshape(res, res, 4).cpu().numpy()
        ref = np.asarray(img_ref, dtype=np.float64).reshape(res, res, 4)
        mse = float(np.mean((img[..., :3].astype(np.float64) - ref[..., :3]) ** 2))
        parity = {"psnr_vs_oracle_db": (10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")),
                  "rgb_max_abs_err": float(np.abs(img[..., :3] - ref[..., :3]).max()),
                  "view": "the cpu_baseline view, tone-mapped RGB, peak 1.0"}
    except Exception as e:                              # never take the bench line down
        parity = {"psnr_vs_oracle_db": None, "error": str(e)}
    return {"parity": parity, "value": 1.0 / dt, "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"1 view fwd+bwd (shade+project+bin+sort+composite+tonemap and backward; prefilter excluded), "
                      f"N={means.shape[0]}, {res}x{res}, oracle/libgs_oracle.so with OpenMP on {cores} host threads; {budget_note}",
            "pairs_evaluated_by_oracle": m["pairs"], "pairs_valid": m["pairs_valid"],
            "V": int(len(m["gaussian_ids"])), "I": int(len(m["flatten_ids"]))}


def main():
    args = parse()
    # ONE JSON line on stdout, whatever the libraries print: RCCL writes its version banner ("RCCL version : ...", five lines) to
    # file descriptor 1 when its first communicator comes up.  Everything else that reaches fd 1 during the run goes to stderr; the
    # result line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import geosplatting_amd as gs
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.engine import RenderStep, params_from_scene
    from geosplatting_amd.parallel import init_distributed_from_env
    import torch.distributed as dist

    if args.rccl_world1:
        os.environ["GEOSPLAT_COLLECTIVES_AT_WORLD1"] = "1"
    rank, world, dev = init_distributed_from_env("cuda")
    use_coll = world > 1 or args.rccl_world1
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.manual_seed(1)
    scene = syn.sphere_scene(args.level, seed=1, cubemap_res=args.cubemap_res, device=dev)
    N = scene.splats.num
    strong = not args.weak
    views_total = (args.views_total or args.views) if strong else args.views * world
    all_cams = syn.blender_cameras(num=views_total, width=args.res, height=args.res)
    cams = [all_cams[i] for i in range(rank, views_total, world)]                # view i -> rank i mod world
    params = params_from_scene(scene, dev)
    step = RenderStep(params, prefilter=not args.no_prefilter)
    g = torch.Generator().manual_seed(100 + rank)
    ups = [(torch.rand(args.res, args.res, 4, generator=g) * 2 - 1).to(dev) for _ in range(max(1, len(cams)))]

    def one_step():
        step(cams, lambda i, img: ups[i], all_reduce=use_coll)

    n_settle = 0
    if args.settle_seconds > 0 and len(cams) > 0:          # clocks / allocator / capacity settle; identical on every rank (fixed step count)
        one_step(); torch.cuda.synchronize()               # (the first step builds the prefilter tables and learns the capacity)
        step.poll_capacity(wait=True)
        t_s = time.perf_counter()
        one_step(); torch.cuda.synchronize()
        per = max(time.perf_counter() - t_s, 1e-3)
        n_settle = int(min(200, max(40, args.settle_seconds / per)))    # (scripts/xp/ramp.py: a cold process reaches its steady rate after ~20 steps)
        if world > 1:
            t_n = torch.tensor([n_settle], device=dev); dist.all_reduce(t_n, op=dist.ReduceOp.MAX); n_settle = int(t_n.item())
        for _ in range(n_settle):
            one_step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step()
    graphed = None
    if args.graph < 0:                     # few views per rank: the launch chain of a view is host-bound, the views go into a graph
        args.graph = 2 if 0 < len(cams) <= 2 else 0
    if args.graph == 2 and len(cams) > 0 and not args.no_prefilter:
        torch.cuda.synchronize()
        ok_local = step.poll_capacity(wait=True) and step._i_cap is not None
        if ok_local:                       # (every rank runs the same workload shape: the capacity is known everywhere after the warm-up)
            graphed = step.capture_views(cams, lambda i, img: ups[i], all_reduce=use_coll)
            one_step = graphed
            one_step()
    elif args.graph == 1 and world == 1 and len(cams) > 0:
        # the whole step as ONE HIP graph (engine.RenderStep.capture): same kernels, same streams, no per-launch host work.
        # Single-GPU only: the sharded prefilter and the gradient all-reduce (RCCL) stay eager.
        torch.cuda.synchronize()
        if step.poll_capacity(wait=True) and step._i_cap is not None:
            graphed = step.capture(cams, lambda i, img: ups[i])
            one_step = graphed
            one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # capacity protocol: the timed steps ran without any (V, I) read-back if a capacity was known before them (default: learnt in
    # the warm-up); an overflow in any of them would make the number invalid -- checked here, outside the timed region
    cap_ok = graphed.check() if graphed is not None else step.poll_capacity(wait=True)
    capacity = {"mode": "device-side counts, no host synchronisation inside a step" if step._i_cap is not None else "exact (one read-back per view)",
                "n_isects_cap": step._i_cap, "overflow_in_timed_steps": (not cap_ok), "truncated_steps": step.truncated_steps,
                "untimed_settle_steps_before_warmup": n_settle,
                "hip_graph": ({1: "whole step", 2: "views segment (prefilter and collectives eager)"}[args.graph] if graphed is not None else False)}
    # the compositor launches INSIDE the step (they share the CUs with the front / tail streams there): HIP events on the stream they
    # are launched on, three extra untimed steps; not available when the views are replayed from a graph
    engine_ms = None
    if graphed is None and len(cams) > 0 and world == 1:      # (single process only: extra steps would have to be agreed on by all ranks)
        step.kernel_events = []
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        acc = {}
        for name, a, b in step.kernel_events:
            acc.setdefault(name, []).append(a.elapsed_time(b))
        # the step's ENDS: from the end of the last view's compositor backward to the start of the next step's first compositor forward
        # (last tail, prefilter backward + mip links, zero fill / activations, mip chain, diffuse map, prefilter forward, first view's
        # shaded front + record stream) -- the part of a step in which the compositor chain, its critical path, does not run
        ev = step.kernel_events
        nv = len(cams)
        bounds = [ev[2 * nv * (k + 1) - 1][2].elapsed_time(ev[2 * nv * (k + 1)][1]) for k in range(len(ev) // (2 * nv) - 1)
                  if ev[2 * nv * (k + 1) - 1][0] == "raster_bwd_kernel" and ev[2 * nv * (k + 1)][0] == "raster_fwd_kernel"]
        step.kernel_events = None
        engine_ms = {k: sum(v) / len(v) for k, v in acc.items()} or None
        if engine_ms is not None and bounds:
            engine_ms["step_boundary_last_bwd_end_to_next_first_fwd_start"] = sum(bounds) / len(bounds)
    # the same timed loop with gsplat's SQUARE tile rectangles (GEOSPLAT_TIGHT_TILES=0): `value` runs on rectangles clipped to the
    # alpha >= 1/255 extents -- a pixel-neutral subsequence of gsplat's intersection list, not the list itself (config.I_engine vs I)
    square = None
    if world == 1 and len(cams) > 0 and graphed is None and os.environ.get("GEOSPLAT_TIGHT_TILES", "1") != "0":
        try:
            os.environ["GEOSPLAT_TIGHT_TILES"] = "0"
            for _ in range(3):
                one_step()
                torch.cuda.synchronize()
                step.poll_capacity(wait=True)               # (the square list is 23 % longer: the capacity follows)
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts0
            ok_sq = step.poll_capacity(wait=True)
            square = {"views_per_s": views_total * args.steps / dts, "ms_per_step": dts / args.steps * 1e3, "overflow": not ok_sq,
                      "what": "GEOSPLAT_TIGHT_TILES=0: the engine on gsplat's own (tile, Gaussian) list, bit-exact tile / sort indices"}
        except Exception as e:
            square = {"error": repr(e)[:300]}
        finally:
            os.environ["GEOSPLAT_TIGHT_TILES"] = "1"
            for _ in range(2):
                one_step()
            torch.cuda.synchronize()
    call_shaped = None
    if world == 1 and len(cams) > 0 and not args.no_prefilter and not args.no_call_shaped:
        try:
            call_shaped = time_call_shaped(params, cams, ups, args.steps, args.warmup)
            call_shaped["frac_of_value"] = call_shaped["ms_per_step"] and (dt / args.steps * 1e3) / call_shaped["ms_per_step"]
        except Exception as e:                               # never take the bench line down
            call_shaped = {"error": repr(e)[:300]}
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    views_per_s = views_total * args.steps / dt

    # ---------------- measurements for the roofline objects (rank 0, one representative view)
    result = None
    if rank == 0:
        cam = all_cams[0]
        with torch.no_grad():
            env = step._static_env if args.no_prefilter else gs.as_splitsum(params.cubemap)
            colors = gs.shade(params.means, params.normals, params.kd, params.ks, cam.c2w[:, 3].to(dev).contiguous(), env,
                              min_roughness=0.1, max_metallic=1.0)
            _, _, meta = gs.rasterization(params.means, params.quats, params.scales.exp(),
                                          torch.sigmoid(params.opacities).squeeze(-1), colors,
                                          cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], args.res, args.res)
        V, I = int(meta["radii"].shape[0]), int(meta["flatten_ids"].shape[0])
        P = args.res * args.res
        E = sum(6 * r * r * 3 * 4 for r in [l.shape[1] for l in env.levels]) + 6 * 16 * 16 * 3 * 4
        kt, I_engine = time_dominant_kernels(params, env, cam, args.res, args.kernel_iters)
        view_detail = time_view_without_prefilter(params, cam, ups[0], max(3, args.kernel_iters // 2))
        view_detail["note"] = "one view fwd+bwd, pyramid fixed; the headline figure is the HIP-graph replay when it could be captured, else the eager launch"
        view_ms = view_detail["graph_ms"] if view_detail.get("graph_ms") else view_detail["eager_ms"]
        comp = {k: v for k, v in kt.items() if k.startswith("raster_") and "prepare" not in k}
        dom = max(comp, key=comp.get)
        # algorithmic bytes of the compositor launches:
        #   fwd: sorted record stream 48 I + image write 20 P        bwd: record stream 48 I + image read 24 P + per-visible grad write 36 V
        #   (I_engine: the engine's own list -- tile rectangles clipped to the alpha >= 1/255 extents -- is shorter than gsplat's I)
        kbytes = {"raster_fwd_kernel": 48 * I_engine + 20 * P, "raster_bwd_kernel": 48 * I_engine + 24 * P + 36 * V}      # (+ 12 B per logged record)
        hbm_achieved = kbytes[dom] / (kt[dom] * 1e-3) / 1e9
        view_bytes = algorithmic_bytes_per_view(N, V, I, P, E)
        cb = None
        if not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(scene, cam, args.res, "bounded to one view")
            except Exception as e:       # the baseline must never take the bench line down
                cb = {"value": None, "unit": "views/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}", "parity": None}
        pairs_valid = cb.get("pairs_valid") if cb else None
        # the compositor is FP32-VALU bound: useful flops = composited (pixel, Gaussian) pairs x flops per pair
        t_pair = (kt["raster_fwd_kernel"] + kt["raster_bwd_kernel"]) * 1e-3
        valu_tf = None if not pairs_valid else pairs_valid * (FLOPS_FWD_PER_PAIR + FLOPS_BWD_PER_PAIR) / t_pair / 1e12
        dom_flops = FLOPS_FWD_PER_PAIR if dom == "raster_fwd_kernel" else FLOPS_BWD_PER_PAIR
        dom_tf = None if not pairs_valid else pairs_valid * dom_flops / (kt[dom] * 1e-3) / 1e12
        dom_tf_engine = None if not (pairs_valid and engine_ms and engine_ms.get(dom)) else pairs_valid * dom_flops / (engine_ms[dom] * 1e-3) / 1e12
        # committed counter summaries, quoted only while gs_raster.hip is the source they were measured on
        first = lambda stem: next((f"r{r:02d}_{stem}" for r in (6, 5, 4, 3, 2) if committed_profile(f"r{r:02d}_{stem}", "gs_raster.hip")), f"r06_{stem}")
        stats_name, pmc_name = first("raster_stats.json"), first("pmc_traffic.json")
        stats = committed_profile(stats_name, "gs_raster.hip")
        pmc = committed_profile(pmc_name, "gs_raster.hip")
        engine = {"kernel_ms": engine_ms} if engine_ms else None                     # measured live, inside the step (see above)
        lane_util = None
        if stats and args.level == 7 and args.res == 800:
            key = "fwd" if dom == "raster_fwd_kernel" else "bwd"
            cpt = int(stats.get("candidates_per_trip", 1))
            lane_util = {"valid_pairs": stats[key]["valid_pairs"], "trips": stats[key]["trips"], "candidates_per_lane_and_trip": cpt,
                         "frac_of_candidate_slots_used": stats[key]["valid_pairs"] / max(1, stats[key]["trips"] * 64 * cpt),
                         "source": f"profiles/{stats_name} (scripts/raster_stats.py, -DGS_RASTER_STATS build of the same source)"}
        traffic = traffic_src = None
        if pmc and args.level == 7 and args.res == 800 and dom in pmc.get("kernels", {}):
            traffic = pmc["kernels"][dom]["hbm_bytes"]
            traffic_src = (f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes per launch, "
                           f"measured at commit {pmc.get('commit', '?')} on this kernel source)")
        d14 = None
        try:
            d14 = time_d14(params, cam, args.res, max(3, args.kernel_iters // 4))
        except Exception as e:
            d14 = {"error": repr(e)[:300]}
        # issue side of the dominant kernel from the committed counter passes (measured, not inferred)
        issue = None
        if pmc and args.level == 7 and args.res == 800 and dom in pmc.get("kernels", {}):
            kq = pmc["kernels"][dom]
            wc = kq.get("SQ_WAVE_CYCLES")
            if wc and kq.get("valu_wave_instructions"):
                chip_valu_issue_per_s = 256 * 4 * 2.4e9 / 2.0                      # one wave64 fp32 instruction = 2 cycles on a SIMD-32
                issue = {"valu_wave_instructions": kq["valu_wave_instructions"],
                         "valu_wave_instructions_per_valid_pair": None if not pairs_valid else kq["valu_wave_instructions"] / pairs_valid,
                         "frac_of_chip_valu_issue_peak_alone": kq["valu_wave_instructions"] / (kt[dom] * 1e-3) / chip_valu_issue_per_s,
                         "wave_cycles_waiting_at_waitcnt_or_barrier": kq.get("SQ_WAIT_ANY", 0) / wc,
                         "wave_cycles_issue_stalled": kq.get("SQ_WAIT_INST_ANY", 0) / wc,
                         "wave_cycles_issuing": kq.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                         "lds_bank_conflict_cycles_per_lds_instruction": (None if not kq.get("SQ_INSTS_LDS") else
                                                                         kq.get("SQ_LDS_BANK_CONFLICT", 0) / kq["SQ_INSTS_LDS"]),
                         "source": f"profiles/{pmc_name} (SQ_* passes of scripts/run_pmc_r04.sh, mean per launch, kernel alone)"}
        pre = None if args.no_prefilter else prefilter_report(params.cubemap, max(3, args.kernel_iters // 2))
        result = {
            "metric": "fwd+bwd views/sec at 2M Gaussians, 800x800",
            "value": views_per_s, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"surface splats (HIP MGAdapter on icosphere level {args.level}) N={N}, {args.res}x{args.res}, "
                                   f"split-sum GGX envmap {args.cubemap_res}^2, "
                                   + (f"{views_total} views/step over all GPUs (strong scaling)" if strong else f"{args.views} views/step/GPU")
                                   + f", prefilter fwd+bwd {'in' if not args.no_prefilter else 'EXCLUDED from'} every step"
                                   + f"; `value` is timed over {args.steps} steps that FOLLOW {n_settle} untimed settle steps (--settle-seconds "
                                     f"{args.settle_seconds:g}: clock ramp of an idle GPU, allocator, capacity protocol) + the {args.warmup} warm-up steps",
                       "N": N, "V": V, "I": I, "I_engine": I_engine, "P": P, "views_per_step_total": views_total,
                       "parallelism": f"dp{world} (views sharded, prefilter sharded, flat RCCL all-reduce of per-Gaussian grads)"
                                      + (" -- --rccl-world1: every collective issued through RCCL on a one-rank group" if args.rccl_world1 else "")},
            "roofline": {"bound": "valu", "kernel": dom, "launched_as": LAUNCHED_AS[dom],
                         "achieved": dom_tf_engine if dom_tf_engine is not None else dom_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": None if (dom_tf_engine or dom_tf) is None else (dom_tf_engine or dom_tf) / VALU_PEAK_TFLOPS,
                         "frac_basis": "in-engine launch duration (HIP events inside the step)" if dom_tf_engine is not None else "kernel alone",
                         "achieved_alone": dom_tf, "frac_alone": None if dom_tf is None else dom_tf / VALU_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "pairs_valid": pairs_valid, "flops_per_pair": {"fwd": FLOPS_FWD_PER_PAIR, "bwd": FLOPS_BWD_PER_PAIR},
                         "compositor_fwd_plus_bwd": {"achieved_TFLOPs": valu_tf,
                                                     "frac_of_157.3": None if valu_tf is None else valu_tf / VALU_PEAK_TFLOPS},
                         "lane_utilisation": lane_util,
                         "issue": issue,
                         "flops_per_pair_source": "builder's count of the kernels' own arithmetic (bench.py FLOPS_*_PER_PAIR comments); SURVEY 8d estimates ~20 fwd / ~75 bwd",
                         "kernel_ms": kt,
                         "kernel_ms_in_engine": None if not engine else engine.get("kernel_ms"),
                         "frac_in_engine": None if not (engine and pairs_valid and engine.get("kernel_ms", {}).get(dom)) else
                         pairs_valid * dom_flops / (engine["kernel_ms"][dom] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                         "hbm": {"algorithmic_bytes": kbytes[dom], "achieved_GBs": hbm_achieved, "peak_GBs": HBM_PEAK_GBS,
                                 "frac": hbm_achieved / HBM_PEAK_GBS},
                         "note": "`achieved` / `frac`: the dominant kernel's launches INSIDE the step (HIP events on the compositor stream, where it "
                                 "shares the CUs with the front and tail streams: `kernel_ms_in_engine`) -- the figure that matches `value`; "
                                 "`achieved_alone` / `frac_alone` / `kernel_ms`: the same kernels timed ALONE (gs_raster_composite / gs_raster_bwd_acc: no "
                                 "stream build, no memset); useful flops = composited (pixel, Gaussian) pairs x flops per pair"},
            "view_roofline": {"algorithmic_bytes_per_view": view_bytes,
                              "achieved_GBs": view_bytes * views_per_s / world / 1e9,
                              "frac_of_8TBs": view_bytes * views_per_s / world / 1e9 / HBM_PEAK_GBS,
                              # the step as a whole: the views' algorithmic bytes + what the prefilter really moves from HBM (its weight
                              # tables, streamed once per direction; the reference computes these weights instead) + the pyramid
                              "step_bytes_incl_prefilter": (None if not pre or pre.get("table_bytes_from_hbm_per_step") is None else
                                                            view_bytes * len(cams) + pre["table_bytes_from_hbm_per_step"] + 2 * pre.get("pyramid_bytes", 0)),
                              "step_frac_of_8TBs_incl_prefilter": (None if not pre or pre.get("table_bytes_from_hbm_per_step") is None else
                                                                   (view_bytes * len(cams) + pre["table_bytes_from_hbm_per_step"] + 2 * pre.get("pyramid_bytes", 0))
                                                                   / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS)},
            "value_square_tiles": square,
            "call_shaped": call_shaped,
            "d14_deferred": d14,
            "strong_1gpu_ms": ms_per_step if (world == 1 and strong) else None,
            "scale_model": scale_model(view_ms, pre, N, args.cubemap_res, views_total, ms_per_step if world == 1 else None, view_detail),
            "gpu_view_ms_without_prefilter": view_ms,
            "gpu_view_ms_detail": view_detail,
            "prefilter": pre,
            "capacity_protocol": capacity,
        }
        if cb is not None:
            result["parity"] = cb.pop("parity", None)
            if cb.get("value"):
                cb["gpu_over_cpu_same_work"] = (1e3 / view_ms) / cb["value"]
            try:
                cb["cfg1"] = cpu_baseline_cfg1()
            except Exception as e:
                cb["cfg1"] = {"value": None, "sample": f"failed: {e}"}
            result["cpu_baseline"] = cb
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
