"""CPU-side checks: the C-ABI library loads and exports every symbol include/sr_amd.h declares, host-only
logic (mask resize, argument validation, module construction / state_dict layout, tile geometry, sharding)."""
import os
import re

import numpy as np
import pytest
import torch

import golden_params as gp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    from super_resolution_amd.build import build
    build(verbose=False)
    from super_resolution_amd import _lib
    _lib.lib()
    return _lib


def test_library_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, 'include', 'sr_amd.h')).read()
    declared = set(re.findall(r'\b(sr_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(L.HEADER_SYMBOLS), declared ^ set(L.HEADER_SYMBOLS)
    lib = L.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sr_version() >= 100
    assert lib.sr_strerror(-1).decode().startswith('bad argument')


def test_struct_sizes_match_header(L):
    # ask the C compiler what the header's structs measure, and compare with the ctypes mirrors
    import ctypes as C
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "sr_amd.h"\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(sr_view), ' \
          'sizeof(sr_concat), sizeof(sr_epilogue), sizeof(sr_layer_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'a.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'a.c'), '-o', os.path.join(d, 'a')])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, 'a')]).split()]
    assert sizes == [C.sizeof(L.View), C.sizeof(L.Concat), C.sizeof(L.Epilogue), C.sizeof(L.LayerParams)]


def test_masks_host_vs_reference_golden(L, golden):
    g = golden['fbm_masks']
    for key in g.files:
        if not key.startswith('rows_'):
            continue
        n = int(key[5:])
        rows = np.zeros(3 * n, np.uint8)
        cols = np.zeros(3 * (n // 2 + 1), np.uint8)
        L.check(L.lib().sr_fbm_masks_host(n, n, rows.ctypes.data, cols.ctypes.data))
        assert np.array_equal(rows.reshape(3, n).astype(bool), g['rows_%d' % n]), n
        assert np.array_equal(cols.reshape(3, -1).astype(bool), g['cols_%d' % n]), n


def test_masks_host_vs_torch_interpolate_sweep(L):
    from oracle import fdconv_ref as fr
    for n in list(range(2, 200)) + [1000, 1044, 4095]:
        rows = np.zeros(3 * n, np.uint8)
        cols = np.zeros(3 * (n // 2 + 1), np.uint8)
        L.check(L.lib().sr_fbm_masks_host(n, n, rows.ctypes.data, cols.ctypes.data))
        rm, cm = fr.fbm_masks(n, n)
        assert np.array_equal(rows.reshape(3, n).astype(bool), rm.numpy()), n
        assert np.array_equal(cols.reshape(3, -1).astype(bool), cm.numpy()), n


def test_argument_validation_without_gpu(L):
    import ctypes as C
    lib = L.lib()
    assert lib.sr_pack_nchw(None, 0, None, 0, 1, 1, 1, 1, 1, None) == -1
    assert lib.sr_fbm_workspace_bytes(None, 1, 1) == 0
    assert lib.sr_fdconv_basis_workspace_bytes(32, 64) == 4 * 96 * 97 * 16
    h = C.c_void_p()
    assert lib.sr_rrdbnet_create(3, 3, 3, 64, 23, 32, C.byref(h)) == -1  # scale 3 does not exist
    L.check(lib.sr_rrdbnet_create(1, 1, 4, 64, 23, 32, C.byref(h)))
    assert lib.sr_rrdbnet_num_layers(h) == 351
    lp = L.LayerParams()
    lp.cin, lp.cout = 5, 64
    assert lib.sr_rrdbnet_set_layer(h, 0, C.byref(lp)) == -1  # wrong shape for conv_first
    assert lib.sr_rrdbnet_forward(h, 0, 0, 1, 8, 8, 1, 1, 256, 0, None) == -4  # layers not set
    lib.sr_rrdbnet_destroy(h)


def test_module_state_dict_layout():
    from super_resolution_amd import FDConv, RRDBNet, replace_conv_with_fdconv
    m = FDConv(64, 32, kernel_size=3, padding=1)
    sd = m.state_dict()
    want = gp.fdconv_params(64, 32, 0)
    assert set(sd) == set(want) and len(sd) == 27
    for k in sd:
        assert tuple(sd[k].shape) == tuple(want[k].shape), k
    assert 'indices' not in sd and tuple(m.indices.shape) == (1, 2, 4, 2328)
    assert tuple(m.FBM.cached_masks.shape) == (3, 1, 64, 33)
    assert isinstance(m, torch.nn.Conv2d) and m.in_channels == 64
    net = replace_conv_with_fdconv(RRDBNet(1, 1, 4, 64, 1, 32))
    ref = gp.rrdbnet_state(1, 1, 64, 1, 32, seed=7)
    assert set(net.state_dict()) == set(ref)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    plain = RRDBNet(3, 3, 4, 64, 23, 32)
    assert len(plain.state_dict()) == 702
    assert sum(p.numel() for p in plain.parameters()) == 16697987
    with pytest.raises(NotImplementedError):
        FDConv(64, 32, kernel_size=3, padding=1, kernel_num=8)
    with pytest.raises(RuntimeError):
        plain(torch.zeros(1, 3, 8, 8))  # CPU tensors are refused: there is no fallback path


def test_fdconv_init_matches_reference_statistics():
    """Same init scheme as the reference (FDConv.py:115-170, 285, 370): near-zero heads, kaiming filter_fc."""
    from super_resolution_amd import FDConv
    torch.manual_seed(0)
    m = FDConv(64, 32, kernel_size=3, padding=1)
    g = m.KSM_Global
    assert g.spatial_fc.weight.abs().max() < 1e-5 and g.kernel_fc.weight.abs().max() < 1e-5
    assert g.channel_fc.weight.abs().max() < 1e-5 and g.filter_fc.weight.std() > 0.05
    assert torch.all(m.KSM_Local.conv.weight == 1e-6)
    assert all(c.weight.abs().max() < 1e-5 and float(c.bias) == 0 for c in m.FBM.freq_weight_conv_list)
    assert m.alpha == 64 and tuple(m.dft_weight.shape) == (1, 96, 97, 2)


def test_tile_windows_and_sharding():
    from super_resolution_amd import RealESRGANer
    from oracle.realesrganer_ref import tile_windows

    class Up(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.interpolate(x, scale_factor=4, mode='nearest')

    r = RealESRGANer(scale=4, model_path=None, model=Up(), tile=512, tile_pad=10, pre_pad=0, device=torch.device('cpu'))
    r.img = torch.zeros(1, 3, 4096, 4096)
    wins = r.tile_windows()
    ref = tile_windows(4096, 4096, 512, 10)
    assert len(wins) == 64 and [w[0] for w in wins] == [d['inp'] for d in ref]
    sizes = sorted(((w[0][1] - w[0][0]), (w[0][3] - w[0][2])) for w in wins)
    assert sizes.count((532, 532)) == 36 and sizes.count((522, 522)) == 4  # SURVEY 2.1 tile census
    shards = [r.shard_of(wins, k, 8) for k in range(8)]
    assert sorted(sum(shards, [])) == list(range(64)) and all(len(s) == 8 for s in shards)
    area = lambda i: (wins[i][0][1] - wins[i][0][0]) * (wins[i][0][3] - wins[i][0][2])
    loads = [sum(area(i) for i in s) for s in shards]
    assert max(loads) / min(loads) < 1.03
    # one contiguous shape-sorted block per rank: at most two window shapes (= forwards) on a rank, one on most
    shapes = [len({(wins[i][0][1] - wins[i][0][0], wins[i][0][3] - wins[i][0][2]) for i in s}) for s in shards]
    assert max(shapes) <= 2 and shapes.count(1) >= 6
    for world in (1, 2, 3, 5, 64, 70):  # any world size partitions the tiles (ranks beyond the tile count get none)
        parts = [r.shard_of(wins, k, world) for k in range(world)]
        assert sorted(sum(parts, [])) == list(range(64)) and max(map(len, parts)) - min(map(len, parts)) <= 1
    # whole pipeline on the CPU with a stand-in network: tiled == whole for a pointwise model
    img = gp.image_input((37, 53, 3), 5)
    r = RealESRGANer(scale=4, model_path=None, model=Up(), tile=16, tile_pad=3, pre_pad=5, device=torch.device('cpu'))
    out, mode = r.enhance((img * 255).astype(np.uint8))
    assert mode == 'RGB' and out.shape == (148, 212, 3)
    assert np.array_equal(out, np.repeat(np.repeat((img * 255).astype(np.uint8), 4, 0), 4, 1))


def test_tile_batching_equals_one_by_one():
    """tile_batch > 1 stacks tiles of equal window size along the batch axis: same result as one tile at a time for a
    network whose statistics are per sample (here: a conv plus a per-sample mean, like FDConv's pooled attention)."""
    from super_resolution_amd import RealESRGANer

    class PerSample(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.conv = torch.nn.Conv2d(3, 3, 3, 1, 1)

        def forward(self, x):
            y = self.conv(x) * (1.0 + x.mean(dim=(1, 2, 3), keepdim=True))
            return torch.nn.functional.interpolate(y, scale_factor=4, mode='nearest')

    img = (gp.image_input((70, 90, 3), 9) * 255).astype(np.uint8)
    outs = []
    for tb in (1, 4):
        r = RealESRGANer(scale=4, model_path=None, model=PerSample(), tile=16, tile_pad=4, pre_pad=0, device=torch.device('cpu'),
                         tile_batch=tb)
        outs.append(r.enhance(img)[0])
    # torch's CPU conv picks its algorithm by batch size: allow one uint8 step of rounding difference
    assert np.abs(outs[0].astype(np.int32) - outs[1].astype(np.int32)).max() <= 1
    assert (outs[0] != outs[1]).mean() < 1e-3


def test_checkpoint_conversion_matches_fdconv_init(tmp_path):
    """super_resolution_amd.convert (reference tools/convert_to_fdconv.py:125-181 generalised to RRDBNet keys): the
    spectrum it derives from a spatial weight is the one FDConv's own constructor derives (FDConv.py:604-623), it
    inverts exactly, thin layers stay plain, containers are handled, and a plain checkpoint loads into an FDConv model."""
    from super_resolution_amd import FDConv, RRDBNet, convert, replace_conv_with_fdconv
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(64, 32, 3, 1, 1)
    w = conv.weight.data.clone()
    layer = FDConv(64, 32, 3, 1, 1)
    layer.weight = torch.nn.Parameter(w.clone())  # rerun the constructor's conversion on a known weight
    layer.convert2dftweight()
    d = convert.spatial_to_dft_weight(w)
    assert d.shape == (1, 96, 97, 2) and torch.allclose(d, layer.dft_weight.data, atol=1e-7)
    assert torch.allclose(convert.dft_weight_to_spatial(d, 32, 64), w, atol=1e-6)

    plain = RRDBNet(1, 1, 4, 64, 1, 32)
    sd = {k: v.clone() for k, v in plain.state_dict().items()}
    new, converted = convert.convert_state_dict({'params_ema': sd})
    assert set(new) == {'params_ema'} and len(converted) == 15 + 4  # 3 RDB x 5 convs + conv_body, conv_up1, conv_up2, conv_hr
    assert 'conv_first.weight' in new['params_ema'] and 'conv_last.weight' in new['params_ema']  # thin: 1 channel
    assert 'body.0.rdb1.conv1.dft_weight' in new['params_ema'] and 'body.0.rdb1.conv1.weight' not in new['params_ema']

    fd = replace_conv_with_fdconv(RRDBNet(1, 1, 4, 64, 1, 32))
    before = {k: v.clone() for k, v in fd.state_dict().items()}
    convert.load_plain_into_fdconv(fd, {'params': sd})
    after = fd.state_dict()
    assert torch.allclose(after['body.0.rdb2.conv3.dft_weight'], convert.spatial_to_dft_weight(sd['body.0.rdb2.conv3.weight']))
    assert torch.equal(after['conv_first.weight'], sd['conv_first.weight'])
    assert torch.equal(after['body.0.rdb2.conv3.bias'], sd['body.0.rdb2.conv3.bias'])
    k = 'body.0.rdb2.conv3.KSM_Global.fc.weight'
    assert torch.equal(after[k], before[k])  # modulation parameters untouched

    src, dst = tmp_path / 'plain.pth', tmp_path / 'fd.pth'
    torch.save({'params': sd}, src)
    assert convert.main(['convert', str(src), str(dst)]) == 0
    back = torch.load(dst, map_location='cpu', weights_only=True)
    assert 'body.0.rdb1.conv5.dft_weight' in back['params']


def test_lanczos4_resize_properties():
    """cv2.INTER_LANCZOS4 restatement (utils.py:256-261): identity at equal size, constants preserved, a linear ramp
    stays linear away from the replicated border, integer dtypes are rounded and saturated.  Parity unpinned (no cv2)."""
    from super_resolution_amd.realesrganer import lanczos4_resize
    rng = np.random.RandomState(4)
    img = (rng.rand(23, 31, 3) * 255).astype(np.uint8)
    assert np.array_equal(lanczos4_resize(img, 23, 31), img)
    const = np.full((16, 20), 1234, np.uint16)
    assert np.array_equal(lanczos4_resize(const, 40, 9), np.full((40, 9), 1234, np.uint16))
    ramp = np.tile(np.arange(64, dtype=np.float32)[None, :], (8, 1))
    up = lanczos4_resize(ramp, 8, 128)
    want = (np.arange(128) + 0.5) * 0.5 - 0.5
    assert np.abs(up[:, 16:112] - want[None, 16:112]).max() < 0.02  # windowed sinc: not exact on ramps
    sat = lanczos4_resize(np.array([[0, 255] * 8] * 4, np.uint8), 4, 37)
    assert sat.dtype == np.uint8 and sat.min() == 0 and sat.max() == 255
    down = lanczos4_resize(img, 10, 12)
    assert down.shape == (10, 12, 3)


def test_prefetch_reader_and_io_consumer(tmp_path):
    """PrefetchReader / IOConsumer (utils.py:266-313): images come back in order, BGR like cv2.imread; the consumer
    writes what it is handed until it receives 'quit'."""
    import queue
    from super_resolution_amd.realesrganer import IOConsumer, PrefetchReader
    rng = np.random.RandomState(6)
    imgs = [(rng.rand(9 + i, 7, 3) * 255).astype(np.uint8) for i in range(5)]
    paths = []
    que = queue.Queue()
    worker = IOConsumer(None, que, 0)
    worker.start()
    for i, im in enumerate(imgs):
        p = str(tmp_path / ('im%d.%s' % (i, 'png' if i % 2 else 'npy')))
        paths.append(p)
        que.put({'output': im, 'save_path': p})
    que.put('quit')
    worker.join(30)
    assert not worker.is_alive()
    reader = PrefetchReader(paths, num_prefetch_queue=2)
    reader.start()
    got = list(reader)
    assert len(got) == 5 and all(np.array_equal(a, b) for a, b in zip(got, imgs))


def test_converter_matches_reference_converter(golden_r2):
    """super_resolution_amd.convert against the output of the reference's own tools/convert_to_fdconv.py (fixture
    tests/golden/convert_ref.npz): same spectrum, same skip rule for thin layers, same sorted-frequency table."""
    from super_resolution_amd import convert
    from super_resolution_amd.fdconv import get_fft2freq
    g = golden_r2['convert_ref']
    r = np.random.RandomState(500)
    w2 = torch.from_numpy(r.standard_normal((64, 32, 3, 3)).astype(np.float32))
    assert 'layer1.0.conv2.dft_weight' in list(g['keys']) and 'layer1.0.conv3.weight' in list(g['keys'])
    d = convert.spatial_to_dft_weight(w2)
    assert d.shape == g['layer1.0.conv2.dft_weight'].shape
    assert np.abs(d.numpy() - g['layer1.0.conv2.dft_weight']).max() < 1e-6
    coords, _ = get_fft2freq(96, 192, use_rfft=True)
    assert np.array_equal(coords.numpy(), g['idx_96_97'])


def test_flat_adam_matches_torch_adam():
    """train.FlatAdam (one flat buffer, one gather, five tensor ops) against torch.optim.Adam on the same gradients:
    parameters after several steps, with and without weight decay / gradient clipping; parameter version counters move
    (FDConv keys its cached spectral bases on them)."""
    import copy
    import torch
    import torch.nn as nn
    from super_resolution_amd.train import FlatAdam
    torch.manual_seed(3)
    for wd, clip in ((0.0, None), (0.01, 0.5)):
        a = nn.Sequential(nn.Linear(7, 5), nn.Tanh(), nn.Linear(5, 3), nn.Conv1d(3, 2, 1))
        b = copy.deepcopy(a)
        oa = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=wd)
        ob = FlatAdam(b.parameters(), lr=1e-2, weight_decay=wd)
        v0 = [p._version for p in b.parameters()]
        for step in range(5):
            x = torch.randn(4, 7)
            for net, opt in ((a, oa), (b, ob)):
                opt.zero_grad()
                y = net[:3](x)
                loss = net[3](y[:, :, None]).pow(2).sum() + y.abs().sum()
                loss.backward()
                if opt is oa:
                    if clip is not None:
                        nn.utils.clip_grad_norm_(a.parameters(), clip)
                    opt.step()
                else:
                    opt.step(clip_grad_norm=clip)
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert (pa - pb).abs().max().item() < 2e-6, (wd, clip)
        assert all(p._version > v for p, v in zip(b.parameters(), v0))
        # the parameters are views of the flat buffer
        assert all(p.data_ptr() >= ob.flat.data_ptr() for p in b.parameters())


def test_flat_adam_is_an_optimizer_checkpoints_and_guards_its_aliasing():
    """Round-2 advisor finding: the owner scripts attach ReduceLROnPlateau to the optimizer
    (train/train_realESRGAN_fdconv.py:121) and checkpoint optimizer.state_dict()
    (train/train_realESRGAN_fdconv_OST_1.py:401).  FlatAdam must be a torch.optim.Optimizer, follow a scheduler's rate,
    save / resume its moments in torch.optim.Adam's own layout, and notice a module cast after construction."""
    import copy
    import torch
    import torch.nn as nn
    from super_resolution_amd.train import FlatAdam
    torch.manual_seed(5)
    net = nn.Sequential(nn.Linear(6, 4), nn.Tanh(), nn.Linear(4, 2))
    twin = copy.deepcopy(net)
    opt, ref = FlatAdam(net.parameters(), lr=1e-2), torch.optim.Adam(twin.parameters(), lr=1e-2)
    assert isinstance(opt, torch.optim.Optimizer)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode='min', factor=0.5, patience=0)
    sch_ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref, mode='min', factor=0.5, patience=0)
    xs = [torch.randn(3, 6) for _ in range(6)]

    def run(model, o, x):
        o.zero_grad()
        model(x).pow(2).sum().backward()
        o.step()
    for i in range(3):
        run(net, opt, xs[i])
        run(twin, ref, xs[i])
        sch.step(1.0)        # a plateau: the rate halves from the second call on
        sch_ref.step(1.0)
    assert opt.param_groups[0]['lr'] == ref.param_groups[0]['lr'] < 1e-2
    for pa, pb in zip(net.parameters(), twin.parameters()):
        assert (pa - pb).abs().max().item() < 2e-6
    # checkpoint in Adam's layout; resume into a fresh FlatAdam AND into torch.optim.Adam, continue, compare
    ck = opt.state_dict()
    assert set(ck) == {'state', 'param_groups'} and set(ck['state'][0]) >= {'step', 'exp_avg', 'exp_avg_sq'}
    net2, net3 = copy.deepcopy(net), copy.deepcopy(net)
    opt2 = FlatAdam(net2.parameters(), lr=123.0)
    opt2.load_state_dict(ck)
    opt3 = torch.optim.Adam(net3.parameters(), lr=123.0)
    opt3.load_state_dict(ck)
    assert opt2.steps == 3 and opt2.param_groups[0]['lr'] == opt.param_groups[0]['lr']
    for i in range(3, 6):
        run(net, opt, xs[i])
        run(net2, opt2, xs[i])
        run(net3, opt3, xs[i])
    for pa, pb, pc in zip(net.parameters(), net2.parameters(), net3.parameters()):
        assert torch.equal(pa, pb)
        assert (pa - pc).abs().max().item() < 2e-6
    # zero_grad(set_to_none=False) keeps zeroed gradient tensors
    opt.zero_grad(set_to_none=False)
    assert all(p.grad is not None and not p.grad.any() for p in net.parameters())
    # a cast after construction re-allocates the parameters: step() must refuse instead of updating a dead buffer
    net.double()
    net.float()
    net(xs[0]).sum().backward()
    with pytest.raises(RuntimeError, match='no longer aliases'):
        opt.step()
