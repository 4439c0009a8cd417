"""N>1 path on CPU: world_size-2 gloo processes exercise the view sharding and the flat gradient bucket
all-reduce exactly as bench.py / RenderStep use them (the rendering itself needs a GPU and is replaced by a
deterministic per-view gradient)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_view_grads(view, shapes):
    g = torch.Generator().manual_seed(1000 + view)
    return {k: torch.randn(*v, generator=g) for k, v in shapes.items()}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from geosplatting_amd.parallel import GradBucket, init_distributed_from_env, shard_views
    r, w, dev = init_distributed_from_env("cpu")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    shapes = {"means": (37, 3), "quats": (37, 4), "ks": (37, 2), "cubemap": (6, 4, 4, 3), "exposure": (1,)}
    views = shard_views(8, rank, world)
    assert views == list(range(rank, 8, world))
    local = {k: torch.zeros(*v) for k, v in shapes.items()}
    for v in views:
        for k, g in _fake_view_grads(v, shapes).items():
            local[k] += g
    local["ks"] = None                                   # a parameter without gradient on this rank -> zeros
    bucket = GradBucket(shapes, dev)
    bucket.pack(local)
    wait = bucket.all_reduce(async_op=True)
    wait()
    out = {k: v.clone() for k, v in bucket.unpack().items()}
    # the two-phase reduction RenderStep uses (per-Gaussian head first, cubemap + exposure tail later) gives the same sums
    bucket.pack(local)
    start_head, finish = bucket.all_reduce_split("cubemap")
    start_head()
    assert torch.allclose(bucket.view("means"), out["means"]) and not torch.allclose(bucket.view("cubemap"), out["cubemap"])
    finish()
    for k, v in bucket.unpack().items():
        assert torch.allclose(v, out[k]), k
    # the chunked schedule of engine._chunked_tail: the per-Gaussian segments summed in Gaussian-range chunks (rows [n0, n1) of every
    # per-Gaussian segment per call), cubemap + exposure afterwards -- the same sums again
    bucket.pack(local)
    per_gaussian = ["means", "quats", "ks"]
    bounds = [0, 16, 16, 30, 37]                          # (an empty chunk included)
    for n0, n1 in zip(bounds[:-1], bounds[1:]):
        if n1 > n0:
            bucket.all_reduce_rows(per_gaussian, n0, n1)
            assert torch.allclose(bucket.view("means")[:n1], out["means"][:n1])
            if n1 < 37:
                assert not torch.allclose(bucket.view("means")[n1:], out["means"][n1:])      # rows behind the chunk: still local
    bucket.join_comm()
    _, finish = bucket.all_reduce_split("cubemap")
    finish()
    for k, v in bucket.unpack().items():
        assert torch.allclose(v, out[k]), k
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    shapes = {"means": (37, 3), "quats": (37, 4), "ks": (37, 2), "cubemap": (6, 4, 4, 3), "exposure": (1,)}
    want = {k: torch.zeros(*v) for k, v in shapes.items()}
    for view in range(8):
        for k, g in _fake_view_grads(view, shapes).items():
            want[k] += g
    want["ks"].zero_()
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    for k in shapes:
        assert torch.allclose(outs[0][k], want[k], atol=1e-5), k
        assert torch.equal(outs[0][k], outs[1][k]), k            # every rank holds the identical sum


def test_bucket_layout_single_process():
    from geosplatting_amd.parallel import GradBucket, shard_views
    shapes = {"a": (5, 3), "b": (1,), "c": (2, 2, 2)}
    b = GradBucket(shapes, torch.device("cpu"))
    assert b.numel % 64 == 0 and all(o % 64 == 0 for o in b.offsets.values())
    b.pack({"a": torch.ones(5, 3), "b": torch.tensor(2.0), "c": None})
    assert b.all_reduce() is None                         # no process group -> no-op
    u = b.unpack()
    assert u["a"].sum() == 15 and u["b"].item() == 2 and u["c"].abs().sum() == 0
    assert [shard_views(8, r, 8) for r in range(8)] == [[r] for r in range(8)]
    assert shard_views(3, 2, 4) == [2] and shard_views(3, 3, 4) == []


def _stage1_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from geosplatting_amd.parallel import init_distributed_from_env
    from geosplatting_amd.stage1 import flat_all_reduce
    init_distributed_from_env("cpu")
    g = torch.Generator().manual_seed(7 + rank)
    grads = [torch.randn(5, 3, generator=g), torch.randn(1, generator=g), torch.randn(2, 2, 2, generator=g)[:, :, 0]]   # one non-contiguous
    flat_all_reduce(grads)
    torch.save([t.clone() for t in grads], os.path.join(out_dir, f"s1_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_stage1_flat_gradient_allreduce(tmp_path):
    """the stage-1 trainer step's one flat all-reduce over an arbitrary list of parameter gradients (stage1.flat_all_reduce)"""
    mp.spawn(_stage1_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"s1_rank{r}.pt")) for r in range(2)]
    want = None
    for rank in range(2):
        g = torch.Generator().manual_seed(7 + rank)
        gs_ = [torch.randn(5, 3, generator=g), torch.randn(1, generator=g), torch.randn(2, 2, 2, generator=g)[:, :, 0]]
        want = gs_ if want is None else [a + b for a, b in zip(want, gs_)]
    for a, b, w in zip(outs[0], outs[1], want):
        assert torch.equal(a, b) and torch.allclose(a, w, atol=1e-6)


def test_prefilter_tile_shards_partition_every_level():
    """host logic of the sharded split-sum prefilter: contiguous shares that tile [0, n_tiles) with sizes differing by at most one,
    equal to a round-robin deal of the longest-first tile order"""
    from geosplatting_amd.splitsum import can_shard_prefilter, shard_tiles, tiles_eligible
    for world in (1, 2, 3, 4, 6, 8):
        for n in (12, 48, 96, 192, 384, 768, 1536, 6144, 7):
            edges = [shard_tiles(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == [len(range(r, n, world)) for r in range(world)]
    assert can_shard_prefilter(512, 8) and not can_shard_prefilter(512, 1) and can_shard_prefilter(512, 5)
    assert tiles_eligible(64) and tiles_eligible(96) and tiles_eligible(16) and not tiles_eligible(100) and not tiles_eligible(8)


def test_stage_handoff_files(tmp_path):
    """export_model / checkpoints (rfstudio/model/geosplat.py:839-854, rfstudio/engine/train.py:172-190): key names and shapes as the
    reference writes them (state_dict names probed from the real GeoSplatter in the build container), values round-trip."""
    from geosplatting_amd.stage1 import Stage1Model
    m = Stage1Model(8, light_resolution=16, device="cpu", log2_hashmap_size=12, seed=3)
    with torch.no_grad():
        m.sdf_params.add_(0.25); m.cubemap.mul_(1.5); m.exposure_params.fill_(0.3)
    path = tmp_path / "stage1.pkl"
    m.export_model(path)
    a = torch.load(path, map_location="cpu")
    assert set(a) == {"geom_scale", "resolution", "min_roughness", "max_metallic", "exposure", "cubemap", "deforms", "weights",
                      "sdfs", "ks_enc", "initial_guess"}
    assert a["resolution"] == 8 and a["deforms"].shape == (729, 3) and a["sdfs"].shape == (729, 1) and a["weights"].shape == (512, 21)
    assert a["cubemap"].shape == (6, 16, 16, 3) and a["exposure"].shape == (1,) and a["initial_guess"].shape == (2,)
    assert set(a["ks_enc"]) == {"encoder.params", "mlp.nn_layers.0.weight", "mlp.nn_layers.1.weight"}
    assert a["ks_enc"]["encoder.params"].shape == (16 * 2 ** 12, 2) and a["ks_enc"]["mlp.nn_layers.1.weight"].shape == (2, 32)
    m2 = Stage1Model.from_export(path, device="cpu")
    for k in ("sdf_params", "deform_params", "weight_params", "cubemap", "exposure_params"):
        assert torch.equal(getattr(m2, k), getattr(m, k)), k
    assert torch.equal(m2.field.ks_enc.hash_table, m.field.ks_enc.hash_table)
    # checkpoints
    sd = m.state_dict()
    assert {"exposure_params", "deform_params", "sdf_params", "weight_params", "initial_guess_bias", "cubemap", "latlng"} <= set(sd)
    assert "field.kd_enc.mlp.nn_layers.2.weight" in sd and sd["field.kd_enc.mlp.nn_layers.2.weight"].shape == (3, 32)
    m.save_checkpoint(tmp_path / "ckpts", 7); m.save_checkpoint(tmp_path / "ckpts", 120)
    assert sorted(os.listdir(tmp_path / "ckpts")) == ["0000000007.ckpt", "0000000120.ckpt"]
    m3 = Stage1Model(8, light_resolution=16, device="cpu", log2_hashmap_size=12, seed=9)
    assert m3.load_checkpoint(tmp_path / "ckpts") == 120
    for (k, a_), (_, b_) in zip(m.named_parameters().items(), m3.named_parameters().items()):
        assert torch.equal(a_, b_), k
    assert m3.load_checkpoint(tmp_path / "nowhere") is None


def test_tail_schedule_covers_every_view_count():
    """engine._auto_tail_schedule: the tail launches of a step cover every view once, the last launch takes at most two views
    (it runs alone on the GPU), the background launches before it at most three."""
    from geosplatting_amd.engine import _auto_tail_schedule
    for n in range(1, 33):
        sched = _auto_tail_schedule(n)
        assert sum(sched) == n and all(k >= 1 for k in sched)
        assert sched[-1] <= 2 and all(k <= 3 for k in sched[:-1])
    assert _auto_tail_schedule(8) == [3, 3, 2] and _auto_tail_schedule(4) == [2, 1, 1] and _auto_tail_schedule(2) == [1, 1] and _auto_tail_schedule(1) == [1]
