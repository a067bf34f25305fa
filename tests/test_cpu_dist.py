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
