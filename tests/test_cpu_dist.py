"""world_size=2 gloo test of the tile-sharded RealESRGANer path (CPU, stand-in network)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Net(torch.nn.Module):
    """Deterministic stand-in with a receptive field, so tile borders matter: blur then nearest x4."""

    def forward(self, x):
        k = torch.ones(x.shape[1], 1, 3, 3) / 9
        y = torch.nn.functional.conv2d(x, k, padding=1, groups=x.shape[1])
        return torch.nn.functional.interpolate(y, scale_factor=4, mode='nearest')


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from super_resolution_amd import RealESRGANer
    img = np.random.RandomState(3).rand(45, 61, 3).astype(np.float32)
    r = RealESRGANer(scale=4, model_path=None, model=_Net(), tile=16, tile_pad=4, pre_pad=3, device=torch.device('cpu'),
                     shard_tiles=True)
    r.pre_process(img)
    r.tile_process()
    out = r.post_process()
    wins = r.tile_windows()
    mine = r.shard_of(wins, rank, world)
    if rank == 0:
        single = RealESRGANer(scale=4, model_path=None, model=_Net(), tile=16, tile_pad=4, pre_pad=3, device=torch.device('cpu'))
        single.pre_process(img)
        single.tile_process()
        ref = single.post_process()
        q.put((bool(torch.equal(out, ref)), len(wins), len(mine)))
    else:
        q.put((None, len(wins), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharding_world2_matches_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    oks = [r[0] for r in res if r[0] is not None]
    assert oks == [True]
    assert sum(r[2] for r in res) == res[0][1] == 12  # 3 x 4 tiles, every tile computed exactly once


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from super_resolution_amd.train import allreduce_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 3, 3, padding=1))
    frozen = torch.nn.Parameter(torch.ones(5))  # a parameter that never receives a gradient on rank 1
    params = list(net.parameters()) + [frozen]
    data = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(10))  # global batch; each rank takes its half
    x = data[2 * rank:2 * rank + 2]
    loss = net(x).square().mean() + (frozen.sum() if rank == 0 else 0.0)
    loss.backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    nb = allreduce_gradients(params, bucket_bytes=600)  # tiny buckets: several collectives
    gathered = [torch.zeros(sum(p.numel() for p in params)) for _ in range(world)]
    mine = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(local, params)])
    dist.all_gather(gathered, mine)
    want = sum(gathered) / world
    got = torch.cat([p.grad.reshape(-1) for p in params])
    ok = bool(torch.allclose(got, want, atol=1e-7))
    # FlatAdam: the same exchange as ONE all-reduce over its flat gradient buffer (a parameter without a gradient on this
    # rank enters as zeros); after the step both ranks hold the same parameters
    from super_resolution_amd.train import FlatAdam
    for p, g in zip(params, local):
        p.grad = None if g is None else g.clone()
    opt = FlatAdam(params, lr=1e-2)
    opt.step()
    ok = ok and bool(torch.allclose(opt.grad, want, atol=1e-7))
    both = [torch.zeros_like(opt.flat) for _ in range(world)]
    dist.all_gather(both, opt.flat)
    ok = ok and bool(torch.equal(both[0], both[1]))
    # BatchNorm statistics stay local: the two ranks saw different halves of the batch
    q.put((rank, ok, nb, net[1].running_mean.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_averages_and_keeps_bn_local():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == res[1][2] and res[0][2] >= 3  # the same number of collectives on both ranks, several buckets
    assert not torch.allclose(res[0][3], res[1][3])   # running statistics differ: not exchanged
