"""N > 1 path on the GPU: two ranks share cuda:0 over gloo (GEOSPLAT_DEBUG_SHARE_GPU=1, the same switch bench.py honours)
and run engine.RenderStep -- fused C-ABI drivers on three streams, view sharding, two-phase gradient all-reduce -- on
their views; the reduced gradients must equal a single-process step over all the views."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_VIEWS, RES, LEVEL = 4, 160, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _step(dev, views, all_reduce):
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.engine import RenderStep, params_from_scene
    from oracle import mesh_ref
    # the CPU scene builder: every process must start from bit-identical Gaussians (the HIP vertex-normal kernel accumulates with
    # float atomics, so two builds differ in the last bit -- and a stored-state backward amplifies that)
    sc = syn.sphere_scene(LEVEL, seed=2, cubemap_res=64, mesh_to_splats_fn=mesh_ref.scene_builder)
    cams = syn.blender_cameras(N_VIEWS, RES, RES)
    step = RenderStep(params_from_scene(sc, dev, exposure=1.1))
    ups = {i: (torch.rand(RES, RES, 4, generator=torch.Generator().manual_seed(50 + i)) * 2 - 1).to(dev) for i in range(N_VIEWS)}
    local = [cams[i] for i in views]
    for _ in range(3):      # the first step is exact (one (V, I) read-back per view); the later ones run on the capacity protocol
        grads, _ = step(local, lambda j, img: ups[views[j]].reshape(img.shape), all_reduce=all_reduce)
        torch.cuda.synchronize()
    assert step._i_cap is not None or not local
    return {k: v.detach().cpu().clone() for k, v in grads.items()}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GEOSPLAT_DEBUG_SHARE_GPU="1")
    import torch.distributed as dist
    from geosplatting_amd.parallel import init_distributed_from_env, shard_views
    r, w, dev = init_distributed_from_env("cuda")
    g = _step(dev, shard_views(N_VIEWS, r, w), all_reduce=True)
    torch.save(g, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    want = _step(torch.device("cuda", 0), list(range(N_VIEWS)), all_reduce=False)
    for k, w in want.items():
        scale = w.abs().max().item() + 1e-30
        for r in range(2):
            err = (got[r][k] - w).abs().max().item() / scale
            assert err < 5e-5, (k, r, err)            # different summation order over the views / of the float atomics, nothing else
        assert torch.equal(got[0][k], got[1][k]), k   # both ranks hold the same reduced buffer


def _engine(dev):
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.engine import RenderStep, params_from_scene
    sc = syn.sphere_scene(LEVEL, seed=2, cubemap_res=64)
    cams = syn.blender_cameras(N_VIEWS, RES, RES)
    step = RenderStep(params_from_scene(sc, dev, exposure=1.1))
    ups = {i: (torch.rand(RES, RES, 4, generator=torch.Generator().manual_seed(50 + i)) * 2 - 1).to(dev) for i in range(N_VIEWS)}

    global step_cams, step_up
    step_cams, step_up = cams, (lambda j, img: ups[j].reshape(img.shape))

    def run():
        grads, images = step(cams, lambda j, img: ups[j].reshape(img.shape), all_reduce=False, keep_images=True)
        torch.cuda.synchronize()
        return {n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images]
    return step, run


def _same(ref, got):
    for a, b in zip(ref[1], got[1]):
        assert torch.equal(a, b)                                      # forward: same kernels on the same inputs
    for k, w in ref[0].items():
        scale = w.abs().max().item() + 1e-30
        assert (got[0][k] - w).abs().max().item() / scale < 5e-5, k   # backward: order of the float atomics


def test_capacity_protocol_equals_exact_mode():
    """Capacity protocol (include/geosplat_hip.h): once the engine has seen the counts of a step it sizes the per-view
    buffers by (N, I_cap) and never reads (V, I) back; images bit-identical to the exact mode, gradients equal up to the order
    of the float atomics."""
    step, run = _engine(torch.device("cuda", 0))
    step._use_capacity = False
    run()                                                             # (builds the cached prefilter tables)
    ref = run()
    assert step._i_cap is None
    step._use_capacity = True
    assert step.poll_capacity(wait=True) and step._i_cap is not None and step._i_cap % 65536 == 0
    torch.cuda.set_sync_debug_mode("error")                           # any synchronising torch call inside the step raises
    try:
        grads, images = step(step_cams, step_up, all_reduce=False, keep_images=True)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    got = ({n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images])
    assert step.poll_capacity(wait=True)                              # no overflow
    _same(ref, got)


def test_capacity_overflow_is_reported_and_recovered():
    """A view that needs more intersections than the capacity is truncated (memory-safe), reported through the sticky status
    word, and the capacity is raised so that the repeated step is complete."""
    step, run = _engine(torch.device("cuda", 0))
    step._use_capacity = False
    run()
    ref = run()
    step._use_capacity = True
    step._seen_counts = []; step._exact_max_i = 0                     # (forget what the exact steps learnt)
    step._i_cap = 4096                                                # far below what a view needs
    bad = run()
    assert not step.poll_capacity(wait=True)                          # overflow seen ...
    assert step._i_cap > 4096                                         # ... and the capacity raised to fit
    assert not all(torch.equal(a, b) for a, b in zip(ref[1], bad[1])) # (the truncated step really was incomplete)
    got = run()
    assert step.poll_capacity(wait=True)
    _same(ref, got)


def test_capacity_overflow_seen_inside_call_is_not_lost():
    """RenderStep.__call__ polls (non-blocking) before every step.  If THAT poll is the one that sees the overflow word of an
    earlier step, the fact must survive until the caller asks: `truncated_steps` counts it and the caller's next
    poll_capacity() returns False once."""
    step, run = _engine(torch.device("cuda", 0))
    step._use_capacity = False
    run(); ref = run()
    step._use_capacity = True
    step._seen_counts = []; step._exact_max_i = 0
    step._i_cap = 4096
    assert step.truncated_steps == 0
    run()                                                             # truncated; run() synchronises, so the word is on the host
    run()                                                             # __call__'s own poll sees it (and raises the capacity) ...
    assert step.truncated_steps >= 1 and step._i_cap > 4096
    assert not step.poll_capacity(wait=True)                          # ... and the caller still learns about it
    got = run()
    assert step.poll_capacity(wait=True)                              # reported once; this step was complete
    _same(ref, got)


def test_step_captured_in_hip_graph():
    """With fixed launch shapes and no host synchronisation a whole step (8 views' worth of kernels on three streams, prefilter
    forward and backward) is one HIP graph: replaying it must reproduce the eager step."""
    step, run = _engine(torch.device("cuda", 0))
    run()
    assert step.poll_capacity(wait=True) and step._i_cap is not None
    ref = run()
    replay = step.capture(step_cams, step_up, keep_images=True)
    for _ in range(3):
        grads, images = replay()
    assert replay.check()
    got = ({n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images])
    _same(ref, got)


@pytest.mark.parametrize("geo_graph", ["1", "0"])
def test_views_segment_captured_in_hip_graph(monkeypatch, geo_graph):
    """RenderStep.capture_views: the views of a step (shading ... compositor ... per-view backward) as HIP graphs with the
    prefilter forward / backward and every collective outside of them -- the shape one view per GPU needs (BASELINE config 4);
    by default TWO graphs, the geometry of the views (projection, keys, binning) replayed beside the eager prefilter forward.
    Same images and gradients as the eager step, also after the environment map AND the Gaussians changed under the graphs' feet."""
    monkeypatch.setenv("GEOSPLAT_GEO_GRAPH", geo_graph)
    step, run = _engine(torch.device("cuda", 0))
    run()
    assert step.poll_capacity(wait=True) and step._i_cap is not None
    graphed = step.capture_views(step_cams, step_up, all_reduce=False, keep_images=True)
    assert (graphed.geo_graph is not None) == (geo_graph == "1")
    for scale in (1.0, 1.7):
        with torch.no_grad():
            step.p.cubemap.mul_(scale)                                # the prefilter is NOT in the graph: it must follow the parameter
            step.p.means.mul_(1.0 + 0.01 * (scale - 1.0))             # (the graphs read the parameters in place)
            step.p.scales.add_(0.02 * (scale - 1.0))
        ref = run()
        for _ in range(2):
            grads, images = graphed()
        assert graphed.check()
        got = ({n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images])
        _same(ref, got)
    assert float(ref[0]["cubemap"].abs().sum()) > 0


def test_graph_replay_behind_a_finished_eager_kernel():
    """Regression: a replay of the captured views launched on an idle device behind a short eager kernel (what a host-synchronous
    collective in front of the replay produces) took 1-20 s and later faulted -- the chained-scan state of the projection was cleared
    by a hipMemsetAsync NODE that the runtime no longer ordered in front of the kernel.  Every clear on a capturable path is a
    kernel now (gs_zero_async); the replay must take its normal time and give the same results."""
    import time
    dev = torch.device("cuda", 0)
    step, run = _engine(dev)
    run()
    assert step.poll_capacity(wait=True) and step._i_cap is not None
    graphed = step.capture_views(step_cams, step_up, all_reduce=False, keep_images=True)
    ref = run()
    graphed(); torch.cuda.synchronize()
    one = torch.zeros(1024, device=dev)
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        one.add_(1.0); time.sleep(0.01)                               # the eager kernel has long finished when the graph is launched
        t0 = time.perf_counter()
        grads, images = graphed()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    assert graphed.check()
    assert max(times) < 0.25, times                                   # (normal: a few milliseconds; the bug: > 1 s at this size)
    _same(ref, ({n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images]))


def test_host_lead_is_bounded():
    """The step never synchronises, so nothing but the engine keeps the host from running hundreds of steps ahead of the GPU -- and the
    caching allocator from holding the buffers of all of them (186 GiB after 400 bench steps before the bound existed).  At most
    GEOSPLAT_STEPS_IN_FLIGHT (3) unfinished steps are outstanding, and the reserved memory stops growing."""
    dev = torch.device("cuda", 0)
    step, run = _engine(dev)
    for _ in range(6):
        run()
    torch.cuda.synchronize()
    base = torch.cuda.memory_reserved(dev)
    for _ in range(60):
        step(step_cams, step_up, all_reduce=False)
        assert len(step._in_flight) <= step._max_in_flight == 3
    grown = torch.cuda.memory_reserved(dev) - base
    torch.cuda.synchronize()
    assert step.poll_capacity(wait=True)
    assert grown <= 0.5 * base + (64 << 20), (grown, base)


def test_capacity_follows_changing_views():
    """Thirty steps over changing camera subsets and resolutions' worth of intersection counts: the capacity only ever grows to
    1.25 x the largest count seen, every step is either complete or reported (and then repeated), the pool of pinned count
    buffers stays bounded, and the last step equals the exact mode on the same views."""
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.engine import RenderStep, params_from_scene
    import importlib
    R = importlib.import_module("geosplatting_amd.rasterization")
    from oracle import mesh_ref
    dev = torch.device("cuda", 0)
    sc = syn.sphere_scene(LEVEL, seed=2, cubemap_res=64, mesh_to_splats_fn=mesh_ref.scene_builder)
    cams = syn.blender_cameras(12, RES, RES)
    step = RenderStep(params_from_scene(sc, dev, exposure=1.1))
    g = torch.Generator().manual_seed(9)
    up = (torch.rand(RES, RES, 4, generator=g) * 2 - 1).to(dev)
    repeats = 0
    for it in range(30):
        k = 1 + int(torch.randint(0, 4, (1,), generator=g))
        views = [cams[int(i)] for i in torch.randperm(12, generator=g)[:k]]
        for attempt in range(3):
            grads, images = step(views, lambda j, img: up, all_reduce=False, keep_images=True)
            if step.poll_capacity(wait=True):
                break
            repeats += 1
        else:
            raise AssertionError("capacity never caught up")
    assert step._i_cap is not None and repeats <= 3
    assert len(R._pinned_pool) <= 16
    got = ({n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images])
    step._use_capacity = False
    grads, images = step(views, lambda j, img: up, all_reduce=False, keep_images=True)
    torch.cuda.synchronize()
    _same(({n: v.detach().cpu().clone() for n, v in grads.items()}, [im.detach().cpu().clone() for im in images]), got)


def _rccl_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      GEOSPLAT_COLLECTIVES_AT_WORLD1="1")
    os.environ.pop("GEOSPLAT_DEBUG_SHARE_GPU", None)
    import torch.distributed as dist
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.engine import RenderStep, params_from_scene
    from geosplatting_amd.parallel import collectives_active, init_distributed_from_env
    from oracle import mesh_ref
    r, w, dev = init_distributed_from_env("cuda")
    assert dist.get_backend() == "nccl" and w == 1 and collectives_active()
    sc = syn.sphere_scene(LEVEL, seed=2, cubemap_res=64, mesh_to_splats_fn=mesh_ref.scene_builder)
    cams = syn.blender_cameras(N_VIEWS, RES, RES)
    step = RenderStep(params_from_scene(sc, dev, exposure=1.1))
    ups = {i: (torch.rand(RES, RES, 4, generator=torch.Generator().manual_seed(50 + i)) * 2 - 1).to(dev) for i in range(N_VIEWS)}
    up = lambda j, img: ups[j].reshape(img.shape)
    for _ in range(3):
        grads, _ = step(cams, up, all_reduce=True)                    # sharded prefilter (own communicator) + two-phase all-reduce: RCCL
        torch.cuda.synchronize()
    assert step._pre_group is not None                                # dist.new_group() of the prefilter exchange really ran
    assert step.chunked_tails == 3                                    # the last tail in Gaussian-range chunks, each chunk's rows summed at once
    out = {"eager": {k: v.detach().cpu().clone() for k, v in grads.items()}}
    assert step.poll_capacity(wait=True)
    graphed = step.capture_views(cams[:2], lambda j, img: ups[j].reshape(img.shape), all_reduce=True)   # graph replay beside the collectives
    for _ in range(2):
        g2, _ = graphed()
        torch.cuda.synchronize()
    assert graphed.check()
    assert graphed.rec_graph is not None and step.chunked_tails == 5  # three graphs + the eager chunked tail behind them
    out["graph2"] = {k: v.detach().cpu().clone() for k, v in g2.items()}
    torch.save(out, os.path.join(out_dir, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_one_rank_executes_the_multi_gpu_path(tmp_path):
    """RCCL itself (torch.distributed backend "nccl") on the GPU box: one rank -- two ranks cannot share a GPU under RCCL -- runs
    the COMPLETE multi-GPU step: sharded prefilter forward / backward with its own communicator (dist.new_group), the all-reduce of
    the texel gradients, the last tail in Gaussian-range chunks with the all-reduce of each chunk's rows on the communication stream
    (engine._chunked_tail), and the views of a step replayed as three HIP graphs between eager collectives (capture_views: geometry,
    records, views).  With one rank every sum is the identity, so the results
    must equal the plain single-process step."""
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = torch.load(os.path.join(tmp_path, "rccl.pt"))
    want = _step(torch.device("cuda", 0), list(range(N_VIEWS)), all_reduce=False)
    want2 = _step(torch.device("cuda", 0), [0, 1], all_reduce=False)
    for name, ref in (("eager", want), ("graph2", want2)):
        for k, w in ref.items():
            scale = w.abs().max().item() + 1e-30
            err = (got[name][k] - w).abs().max().item() / scale
            assert err < 5e-5, (name, k, err)
