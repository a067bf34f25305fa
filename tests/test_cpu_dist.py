"""world_size=2 gloo test of the tile-sharded RealESRGANer path (CPU, stand-in network)."""
import os
import sys

import numpy as np
import pytest
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


def _worker(rank, world, port, q, tile_batch=1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from super_resolution_amd import RealESRGANer
    import inspect
    # RCCL ignores p2p tags and matches messages per peer in issue order: nothing in the gather may lean on a tag
    assert 'tag' not in inspect.getsource(RealESRGANer.tile_process).replace('no tags', '').replace('neither tags', '')
    img = np.random.RandomState(3).rand(45, 61, 3).astype(np.float32)
    r = RealESRGANer(scale=4, model_path=None, model=_Net(), tile=16, tile_pad=4, pre_pad=3, device=torch.device('cpu'),
                     shard_tiles=True, tile_batch=tile_batch)
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


@pytest.mark.parametrize('tile_batch', [1, 3])
def test_tile_sharding_world2_matches_single_process(tile_batch):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + tile_batch
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, tile_batch)) for r in range(2)]
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
    with torch.no_grad():
        for p in params:
            p.add_(float(rank))   # replicas that start apart (different seeds): the constructor broadcasts rank 0's values
    opt = FlatAdam(params, lr=1e-2)
    start = [torch.zeros_like(opt.flat) for _ in range(world)]
    dist.all_gather(start, opt.flat)
    ok = ok and bool(torch.equal(start[0], start[1])) and bool(torch.equal(params[0].detach().reshape(-1), opt.flat[:params[0].numel()]))
    opt.step()
    ok = ok and bool(torch.allclose(opt.grad, want, atol=1e-7))
    both = [torch.zeros_like(opt.flat) for _ in range(world)]
    dist.all_gather(both, opt.flat)
    ok = ok and bool(torch.equal(both[0], both[1]))
    # BatchNorm statistics stay local: the two ranks saw different halves of the batch
    q.put((rank, ok, nb, net[1].running_mean.tolist()))  # plain lists: a tensor in a Queue needs its sender alive
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
    assert not torch.allclose(torch.tensor(res[0][3]), torch.tensor(res[1][3]))   # running statistics differ: not exchanged


class _FifoDist:
    """In-process stand-in for torch.distributed with RCCL's p2p matching rule: messages between a pair of ranks match
    in ISSUE ORDER, tags are ignored, a size mismatch is an error.  Ranks are run one after the other (peers first:
    their sends wait in the mailbox), which is legal for a gather that has no message from rank 0 to the peers."""

    class P2POp:
        def __init__(self, op, tensor, peer, group=None, tag=0):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Done:
        def wait(self):
            return True

    def __init__(self, world):
        self.world, self.rank, self.box = world, 0, {}

    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.rank

    def isend(self, tensor, dst, group=None, tag=0):
        self.box.setdefault((self.rank, dst), []).append(tensor.clone())
        return self._Done()

    def irecv(self, tensor, src, group=None, tag=0):
        msg = self.box[(src, self.rank)].pop(0)
        assert msg.numel() == tensor.numel(), 'p2p size mismatch: RCCL would hang or fail here'
        tensor.copy_(msg.view_as(tensor))
        return self._Done()

    def batch_isend_irecv(self, ops):
        return [op.op(op.tensor, op.peer) for op in ops]


class _Identity(torch.nn.Module):
    def forward(self, x):
        return x


@pytest.mark.parametrize('world', [2, 4, 8])
@pytest.mark.parametrize('tile_batch', [1, 18])
def test_tile_gather_is_order_safe_at_config3_geometry(world, tile_batch, monkeypatch):
    """BASELINE configs[2] geometry (4096^2, tile 512, pad 10) under RCCL's matching rule.  The order a rank RUNS its tiles
    in may differ from its shard_of() order (here it is reversed on purpose); the gather must paste every core at its own place
    all the same (round 2 paired per-tile sends and receives by tag and mis-pasted edge tiles)."""
    import torch.distributed as real_dist
    from super_resolution_amd import RealESRGANer
    fake = _FifoDist(world)
    for name in ('is_available', 'is_initialized', 'get_world_size', 'get_rank', 'isend', 'irecv', 'batch_isend_irecv', 'P2POp'):
        monkeypatch.setattr(real_dist, name, getattr(fake, name))
    # shard_of() hands out shape-sorted blocks, so batches_of() would run them in shard order: run them backwards
    plain_batches = RealESRGANer.batches_of
    monkeypatch.setattr(RealESRGANer, 'batches_of', staticmethod(
        lambda wins, idxs, tile_batch: [part[::-1] for part in plain_batches(wins, idxs, tile_batch)][::-1]))
    img = torch.rand(4096, 4096, 1, generator=torch.Generator().manual_seed(world)).numpy()
    out, reordered = None, 0
    for rank in list(range(1, world)) + [0]:
        fake.rank = rank
        # scale 1 with an identity network: the assembled image must be the input, pixel for pixel
        r = RealESRGANer(scale=1, model_path=None, model=_Identity(), tile=512, tile_pad=10, pre_pad=0,
                         device=torch.device('cpu'), shard_tiles=True, tile_batch=tile_batch)
        r.pre_process(img)
        wins = r.tile_windows()
        mine = r.shard_of(wins, rank, world)
        ran = [i for part in r.batches_of(wins, mine, tile_batch) for i in part]
        assert sorted(ran) == sorted(mine)
        reordered += ran != mine
        r.tile_process()
        if rank == 0:
            out = r.post_process()
        else:
            assert r.output is None
    assert reordered > 0, 'the geometry no longer exercises run order != shard order'
    assert all(len(v) == 0 for v in fake.box.values()), 'unreceived messages'
    assert torch.equal(out[0, 0], torch.from_numpy(img[:, :, 0]))
    # layouts of the peers partition the tiles and agree in size with what was sent
    seen = sorted(i for rk in range(world) for i, _, _ in RealESRGANer.shard_layout(wins, rk, world, 1, 1, 1)[0])
    assert seen == list(range(64))
