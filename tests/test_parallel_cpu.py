"""The N>1 path on CPU: world_size-2 gloo.  Checks the flat gradient all-reduce (with permuted-stride parameters and
a rank that has no gradient for one tensor) and that data-parallel U-Net gradients computed by the oracle on two
half-batches, averaged by GradAllReducer, equal the single-process full-batch gradients when the Dice statistics
are all-reduced."""
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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import smsut_amd  # noqa: F401
    from smsut_amd import ops, parallel
    from oracle import recipe, smsut_oracle as O
    r, w, local, group = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and group is not None
    # ---- 1. flat all-reduce over permuted-stride parameters, one tensor without a gradient on rank 1
    p1 = torch.nn.Parameter(ops.new_weight(4, 3, 3, 3)); p2 = torch.nn.Parameter(torch.zeros(5)); p3 = torch.nn.Parameter(torch.zeros(2, 2))
    g1 = ops.new_weight(4, 3, 3, 3); g1.copy_(torch.arange(108.).reshape(4, 3, 3, 3) * (rank + 1)); p1.grad = g1
    g2 = torch.full((5,), float(rank)); p2.grad = g2
    g3 = torch.ones(2, 2) if rank == 0 else None; p3.grad = g3
    red = parallel.GradAllReducer([p1, p2, p3], group)
    red.reduce()
    ok = torch.allclose(p1.grad, torch.arange(108.).reshape(4, 3, 3, 3) * 1.5) and p1.grad.stride() == p1.stride()
    ok = ok and torch.allclose(p2.grad, torch.full((5,), 0.5)) and torch.allclose(p3.grad, torch.full((2, 2), 0.5))
    # ... no unpack copy: the gradients ARE views of the flat bucket now, and the backward's own tensors are untouched
    ok = ok and all(p.grad.untyped_storage().data_ptr() == red._flat.untyped_storage().data_ptr() for p in (p1, p2, p3))
    ok = ok and torch.equal(g2, torch.full((5,), float(rank)))
    # a second step as under hipGraph replay: the backward REWRITES the same gradient tensors while p.grad still names the bucket
    g1.mul_(2.0); g2.add_(10.0)
    red.reduce()
    ok = ok and torch.allclose(p1.grad, torch.arange(108.).reshape(4, 3, 3, 3) * 3.0) and torch.allclose(p2.grad, torch.full((5,), 10.5))
    ok = ok and torch.allclose(p3.grad, torch.full((2, 2), 0.5))
    # ... and a step that re-points .grad (an eager step, a graph re-installing its bindings): the new tensor is the source
    p2.grad = torch.full((5,), 100.0 * (rank + 1))
    red.reduce()
    ok = ok and torch.allclose(p2.grad, torch.full((5,), 150.0)) and torch.allclose(p1.grad, torch.arange(108.).reshape(4, 3, 3, 3) * 3.0)
    # ---- 2. broadcast_parameters
    lin = torch.nn.Linear(3, 3)
    parallel.broadcast_parameters(lin, group)
    t = lin.weight.detach().clone(); dist.broadcast(t, 0)
    ok = ok and torch.equal(t, lin.weight.detach())
    # ---- 3. DP U-Net step == single-process global batch (oracle arithmetic, Dice stats all-reduced)
    shapes = recipe.unet_shapes(1, 3, 4)
    sd = {k: v.clone().requires_grad_(True) for k, v in recipe.fill(shapes, 3).items()}
    x = recipe.synth_images((4, 1, 32, 32), 4); y = recipe.synth_labels(4, 32, 32, 3, 5, block=8)
    xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
    logits = O.unet_forward(sd, xs)
    # global-batch Dice: all-reduce {tp, sum_p, count}; CE is a mean over all pixels -> local mean / world after averaging
    prob = torch.softmax(logits, 1)
    onehot = torch.zeros_like(prob).scatter_(1, ys.unsqueeze(1), 1.0)
    tp, sp, cnt = (prob * onehot).sum((0, 2, 3)), prob.sum((0, 2, 3)), onehot.sum((0, 2, 3))
    stats = torch.stack([tp, sp, cnt]).detach().clone(); dist.all_reduce(stats)
    # surrogate with the right gradient: dc_c = (2 tp + s)/(sp + cnt + s + e) with global denominators
    tp_g, sp_g, cnt_g = stats
    den = sp_g + cnt_g + 1e-5 + 1e-8
    dc_lin = (2 * tp / den - (2 * tp_g + 1e-5) / den ** 2 * sp)[1:]          # first-order expansion = exact gradient
    loss = 0.5 * (-(dc_lin.mean())) * world + 0.5 * torch.nn.functional.cross_entropy(logits, ys)
    loss.backward()
    params = [torch.nn.Parameter(v.detach().clone()) for v in sd.values()]
    for p, v in zip(params, sd.values()):
        p.grad = v.grad.clone()
    parallel.GradAllReducer(params, group).reduce()
    if rank == 0:
        ref = {k: v.clone().requires_grad_(True) for k, v in recipe.fill(shapes, 3).items()}
        O.dice_ce(O.unet_forward(ref, x), y).backward()
        worst = max(((p.grad - r.grad).norm() / (r.grad.norm() + 1e-12)).item() for p, r in zip(params, ref.values()))
        ok = ok and worst < 1e-4
        q.put(("worst", worst))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = dict(out)
    assert res[0] is True and res[1] is True, res
    assert res["worst"] < 1e-4
