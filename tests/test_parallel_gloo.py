"""CPU, world_size 2 (and one case at 8) over gloo: the multi-GPU plumbing of gazenerf_amd.parallel.
The render op itself needs no collective (rays/images are independent); these tests cover the
sharding helpers and the one exchange the training path adds: the flat-bucket gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gazenerf_amd import parallel, synth
from gazenerf_amd.render import PARAM_ORDER


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return dict(ret)


def _allreduce_case(rank, world):
    torch.manual_seed(0)
    params = [v.reshape(v.shape[0], -1).clone() if v.dim() == 4 else v.clone()
              for v in synth.hash_mlp_params("face", seed=1).values()]
    expected = []
    for i, p in enumerate(params):
        g = [torch.full_like(p, float(r + 1)) * (i + 1) for r in range(world)]
        p.grad = g[rank].clone()
        expected.append(sum(g) / world)
    red = parallel.GradAllReducer(params, world, bucket_numel=400000)
    assert len(red.buckets) >= 3
    red.all_reduce()
    return all(torch.equal(p.grad, e) for p, e in zip(params, expected))


def test_grad_all_reduce_two_ranks():
    res = _run(_allreduce_case)
    assert res == {0: True, 1: True}


def _shard_case(rank, world):
    xy = synth.pixel_grid(9)                      # 81 rays: not divisible by 2
    local = parallel.shard_rays(xy, rank, world)
    lo, hi = parallel.shard_range(81, rank, world)
    assert local.shape[-1] == hi - lo and torch.equal(local, xy[:, :, lo:hi])
    feat = local[:, :1, :] * 2.0 + rank          # stand-in for a rendered [B,C,N_r_local] slice
    full = parallel.gather_rays(feat, 81, world)
    exp = torch.cat([xy[:, :1, slice(*parallel.shard_range(81, r, world))] * 2.0 + r for r in range(world)], -1)
    return bool(torch.equal(full, exp))


def test_ray_shard_and_gather_two_ranks():
    res = _run(_shard_case)
    assert res == {0: True, 1: True}


def test_shard_ranges_cover_everything():
    for n in (1, 7, 16, 4096, 262144):
        for world in (1, 2, 4, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_images(16, 3, 8) == (6, 8)       # cfg4: 2 images per rank


def test_world_size_one_is_a_noop():
    p = torch.ones(4, requires_grad=True)
    p.grad = torch.full((4,), 3.0)
    parallel.GradAllReducer([p], 1).all_reduce()
    assert torch.equal(p.grad, torch.full((4,), 3.0))
    assert len(PARAM_ORDER) == 24


def _overlap_case(rank, world):
    """Explicit buckets in backward order + hook-driven launch: bucket 0's all-reduce starts from the autograd
    hook of its last parameter, before the rest of the backward has run (cfg4: the NeuralRenderer bucket flies
    during the hot path's backward)."""
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(3, 2))]     # "late" layers: grads first
    b = [torch.nn.Parameter(torch.ones(4))]                                           # "early" layer: grad last
    red = parallel.GradAllReducer([a, b], world)
    assert red.n_buckets == 2 and red.bytes_per_step == 4 * (5 + 6 + 4)
    red.arm_overlap()
    launched_before_b = []
    b[0].register_hook(lambda g: launched_before_b.append(0 in red._inflight) or g)    # runs when b's grad is computed
    ok = True
    for it in range(2):                                                                # hooks re-arm across steps
        for p in a + b:
            p.grad = None
        x = (b[0] * (rank + 1.0)).sum()                      # b is used first -> its gradient arrives last
        y = (a[0] * x).sum() * (it + 1.0) + (a[1] * 2.0 * (rank + 1.0)).sum()
        y.backward()
        red.all_reduce()
        rs = [r + 1.0 for r in range(world)]
        exp_a0 = sum(4.0 * r * (it + 1.0) for r in rs) / world
        exp_a1 = sum(2.0 * r for r in rs) / world
        exp_b = sum(5.0 * (it + 1.0) * r for r in rs) / world
        ok = ok and torch.allclose(a[0].grad, torch.full((5,), exp_a0)) and torch.allclose(a[1].grad, torch.full((3, 2), exp_a1))
        ok = ok and torch.allclose(b[0].grad, torch.full((4,), exp_b))
    return bool(ok) and launched_before_b == [True, True]


def test_hook_driven_bucket_overlap_two_ranks():
    res = _run(_overlap_case)
    assert res == {0: True, 1: True}


def test_whole_network_buckets_cover_all_trainable_floats():
    """cfg4's exchange: NeuralRenderer (incl. bg_featmap) + both MLPs = 5 015 714 floats."""
    from gazenerf_amd import GazeNeRFNetAMD
    net = GazeNeRFNetAMD()
    groups = {pre: [q for n, q in net.named_parameters() if n.startswith(pre)]
              for pre in ("neural_render.", "fg_CD_predictor_face.", "fg_CD_predictor_eyes.")}
    red = parallel.GradAllReducer(list(groups.values()), 8)
    assert red.n_buckets == 3 and red.bytes_per_step == 4 * 5015714
    assert sum(q.numel() for q in groups["neural_render."]) == 1977756
    assert len(red.params) == len(list(net.parameters()))


def _accumulation_case(rank, world):
    """Two backward passes (gradient accumulation) and a backward aborted half-way before ONE all_reduce(): the
    exchange must carry the accumulated gradients, and every rank must issue the same collectives (round-2 advice:
    the hook counters went negative and a stale flat copy overwrote the later micro-batch's gradients)."""
    a = [torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(3, 2))]
    b = [torch.nn.Parameter(torch.ones(4))]
    red = parallel.GradAllReducer([a, b], world)
    red.arm_overlap()
    rs = [r + 1.0 for r in range(world)]

    def loss(scale):
        x = (b[0] * (rank + 1.0)).sum()
        return ((a[0] * x).sum() + (a[1] * 2.0 * (rank + 1.0)).sum()) * scale

    ok = True
    # (1) accumulation: micro-batches with scales 1 and 3, one exchange
    loss(1.0).backward()
    loss(3.0).backward()
    red.all_reduce()
    ok = ok and torch.allclose(a[0].grad, torch.full((5,), sum(4.0 * r * 4.0 for r in rs) / world))
    ok = ok and torch.allclose(a[1].grad, torch.full((3, 2), sum(2.0 * r * 4.0 for r in rs) / world))
    ok = ok and torch.allclose(b[0].grad, torch.full((4,), sum(5.0 * r * 4.0 for r in rs) / world))
    ok = ok and red.relaunched == 2 and not red._inflight
    # (2) a backward that dies after bucket 0 has only been partly counted, then a clean step
    for p in a + b:
        p.grad = None
    red._make_hook(0)(a[0])                   # what an aborted backward leaves behind: one of two hooks fired
    loss(2.0).backward()
    red.all_reduce()
    ok = ok and torch.allclose(a[0].grad, torch.full((5,), sum(4.0 * r * 2.0 for r in rs) / world))
    ok = ok and torch.allclose(b[0].grad, torch.full((4,), sum(5.0 * r * 2.0 for r in rs) / world))
    # (3) begin_step() drops an exchange nobody collected
    for p in a + b:
        p.grad = None
    loss(1.0).backward()
    red.begin_step()
    ok = ok and not red._inflight
    for p in a + b:
        p.grad = None
    loss(5.0).backward()
    red.all_reduce()
    ok = ok and torch.allclose(b[0].grad, torch.full((4,), sum(5.0 * r * 5.0 for r in rs) / world))
    return bool(ok)


def test_gradient_accumulation_and_aborted_backward_two_ranks():
    res = _run(_accumulation_case)
    assert res == {0: True, 1: True}


def _no_sync_case(rank, world):
    """Accumulation the intended way: all micro-steps but the last under no_sync() -- the hooks launch
    nothing, nothing is re-launched, the last backward starts every bucket exactly once, and check_complete finds no
    half-counted bucket."""
    a = [torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(3, 2))]
    b = [torch.nn.Parameter(torch.ones(4))]
    red = parallel.GradAllReducer([a, b], world)
    red.arm_overlap()
    rs = [r + 1.0 for r in range(world)]

    def loss(scale):
        x = (b[0] * (rank + 1.0)).sum()
        return ((a[0] * x).sum() + (a[1] * 2.0 * (rank + 1.0)).sum()) * scale

    with red.no_sync():
        loss(1.0).backward()
        idle = not red._inflight
    loss(3.0).backward()
    launched = sorted(red._inflight) == [0, 1]
    red.all_reduce(check_complete=True)
    ok = idle and launched and red.relaunched == 0
    ok = ok and torch.allclose(a[0].grad, torch.full((5,), sum(4.0 * r * 4.0 for r in rs) / world))
    ok = ok and torch.allclose(b[0].grad, torch.full((4,), sum(5.0 * r * 4.0 for r in rs) / world))
    # a parameter that received no gradient in this backward is reported, not silently exchanged as zeros
    for p in a + b:
        p.grad = None
    red.begin_step()
    (a[0] * (b[0] * 1.0).sum()).sum().backward()           # a[1] unused
    try:
        red.all_reduce(check_complete=True)
        raised = False
    except RuntimeError as e:
        raised = "only part of their parameters" in str(e)
    red.all_reduce()                                        # both ranks still issue the same collectives
    return bool(ok and raised)


def test_no_sync_accumulation_and_incomplete_bucket_check_two_ranks():
    res = _run(_no_sync_case)
    assert res == {0: True, 1: True}


def _forced_case(rank, world):
    p = torch.nn.Parameter(torch.ones(6))
    p.grad = torch.full((6,), 2.0)
    red = parallel.GradAllReducer([[p]], 1, force_collective=True)      # world size 1 *claimed*, collective still issued
    assert red.active
    red.all_reduce()
    return bool(torch.equal(p.grad, torch.full((6,), 2.0 * world)))     # gloo group of 2 sums both, divided by the claimed 1


def test_force_collective_issues_the_exchange_at_claimed_world_size_one():
    res = _run(_forced_case)
    assert res == {0: True, 1: True}


def _eight_rank_case(rank, world):
    """The shapes of the 8-GPU runs (BASELINE cfg4 and the strong-scaled cfg2b render), on CPU: global batch 16 = 2 images per
    rank; one 512 x 512-ray image as eight contiguous blocks of 32 768 rays, gathered back bit for bit; the three-bucket
    exchange of cfg4's parameter groups (scaled down: same bucket structure, hook-driven first bucket) averaged over 8."""
    lo, hi = parallel.shard_images(16, rank, world)
    ok = (hi - lo) == 2 and lo == 2 * rank
    xy = synth.pixel_grid(512)
    local = parallel.shard_rays(xy, rank, world)
    ok = ok and local.shape[-1] == 32768 and torch.equal(local, xy[:, :, 32768 * rank:32768 * (rank + 1)])
    feat = local[:, :1, :] + 1000.0 * rank
    full = parallel.gather_rays(feat, 512 * 512, world)
    ok = ok and full.shape[-1] == 262144 and all(
        torch.equal(full[:, :, 32768 * r:32768 * (r + 1)], xy[:, :1, 32768 * r:32768 * (r + 1)] + 1000.0 * r) for r in range(world))
    nr = [torch.nn.Parameter(torch.ones(7, 3)), torch.nn.Parameter(torch.ones(11))]          # "NeuralRenderer": gradients first
    face, eyes = [torch.nn.Parameter(torch.ones(9))], [torch.nn.Parameter(torch.ones(5, 2))]
    red = parallel.GradAllReducer([nr, face, eyes], world)
    red.arm_overlap()
    for it in range(2):
        for p in nr + face + eyes:
            p.grad = None
        red.begin_step()
        h = (face[0] * (rank + 1.0)).sum() + (eyes[0] * 2.0).sum()
        ((nr[0] * h).sum() + (nr[1] * (rank + 1.0)).sum()).backward()
        first_in_flight = 0 in red._inflight
        red.all_reduce(check_complete=True)
        rs = [r + 1.0 for r in range(world)]
        ok = ok and first_in_flight
        ok = ok and torch.allclose(nr[1].grad, torch.full((11,), sum(rs) / world))
        ok = ok and torch.allclose(nr[0].grad, torch.full((7, 3), sum(9.0 * r + 20.0 for r in rs) / world))
        ok = ok and torch.allclose(face[0].grad, torch.full((9,), sum(21.0 * r for r in rs) / world))
        ok = ok and torch.allclose(eyes[0].grad, torch.full((5, 2), 42.0))
    return bool(ok) and red.n_buckets == 3


def test_eight_ranks_cfg4_and_strong_scaling_shapes():
    res = _run(_eight_rank_case, world=8)
    assert res == {r: True for r in range(8)}
