#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  It loads the reference's own
`FDConv.py` and `realesrgan/utils.py` from where they lie (nothing is copied) with three
import-time placeholders for packages that are not installed and that the code paths used
here never call into:
  * mmcv.cnn.CONV_LAYERS.register_module()  -- a registry decorator (FDConv.py:12,518)
  * timm.models.layers.trunc_normal_        -- imported, unused (FDConv.py:16)
  * cv2, basicsr.utils.download_util        -- imported by utils.py:1,8; only enhance()/URL paths use them
Parameters and inputs come from golden_params.py (numpy RNG), so the fixtures hold only the
reference OUTPUTS plus the tie-dependent frequency-group maps.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_params as gp  # noqa: E402

REF = '/root/reference'
FDCONV_PY = os.path.join(REF, 'FDConv-main/FDConv-main/FDConv_detection/mmdet_custom/FDConv.py')
UTILS_PY = os.path.join(REF, 'Real-ESRGAN-master/realesrgan/utils.py')


def _placeholder_modules():
    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('mmcv')
    mod('mmcv.cnn', CONV_LAYERS=_Registry())
    mod('timm')
    mod('timm.models')
    mod('timm.models.layers', trunc_normal_=torch.nn.init.trunc_normal_)
    mod('cv2')
    mod('basicsr')
    mod('basicsr.utils')
    mod('basicsr.utils.download_util', load_file_from_url=None)


def load_reference():
    _placeholder_modules()
    out = []
    for name, path in (('ref_fdconv', FDCONV_PY), ('ref_utils', UTILS_PY)):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        out.append(m)
    return out


def to_torch(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def group_map_of(layer):
    """uint8 (3Cout, 3Cin/2+1) group id from the reference's `indices` buffer (1,2,kn,n)."""
    idx = layer.indices[0]
    d1 = layer.out_channels * 3
    d2 = layer.in_channels * 3 // 2 + 1
    gmap = torch.full((d1, d2), 255, dtype=torch.uint8)
    for g in range(idx.shape[1]):
        gmap[idx[0, g], idx[1, g]] = g
    assert int(gmap.max()) == idx.shape[1] - 1
    return gmap.numpy()


def capture_intermediates(layer, x):
    """Run the reference layer, recording what its sub-modules return."""
    rec = {}
    h1 = layer.KSM_Global.register_forward_hook(lambda m, i, o: rec.update(
        ca=o[0].flatten(1), fa=o[1].flatten(1), sa=o[2].flatten(1), ka=o[3].flatten(1)))
    h2 = layer.FBM.register_forward_hook(lambda m, i, o: rec.update(fbm=o))
    y = layer(x)
    h1.remove()
    h2.remove()
    rec['y'] = y
    return {k: v.detach().numpy() for k, v in rec.items()}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_fd, ref_utils = load_reference()
    FDConv = ref_fd.FDConv

    # ---- 1. single FDConv layers, eval mode -------------------------------------------------
    out = {}
    gmaps = {}
    sizes = {(64, 32): [(12, 12), (33, 35), (40, 48), (64, 64)], (64, 64): [(24, 20), (40, 48)]}
    for li, (cin, cout) in enumerate(gp.LAYER_SHAPES):
        layer = FDConv(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
        layer.load_state_dict(to_torch(gp.fdconv_params(cin, cout, seed=100 + li)), strict=True)
        layer.eval()
        gmaps['gmap_%d_%d' % (cin, cout)] = group_map_of(layer)
        for (h, w) in sizes.get((cin, cout), [(24, 20)]):
            x = torch.from_numpy(gp.signed_input((2, cin, h, w), seed=200 + li))
            with torch.no_grad():
                rec = capture_intermediates(layer, x)
            tag = 'c%d_%d_h%d_w%d' % (cin, cout, h, w)
            for k in ('y', 'ca', 'fa', 'sa', 'ka'):
                out[tag + '.' + k] = rec[k]
            if (cin, cout, h, w) in ((64, 32, 33, 35), (64, 32, 12, 12)):
                out[tag + '.fbm'] = rec['fbm']
        if (cin, cout) == (64, 32):
            out['cached_masks'] = layer.FBM.cached_masks.numpy()
            # train mode (batch-statistics BatchNorm), B=4
            layer.train()
            x = torch.from_numpy(gp.signed_input((4, cin, 20, 24), seed=300))
            with torch.no_grad():
                out['train_c64_32.y'] = layer(x).numpy()
            layer.eval()
    np.savez_compressed(os.path.join(HERE, 'fdconv_layers.npz'), **out)
    np.savez_compressed(os.path.join(HERE, 'group_maps.npz'), **gmaps)

    # ---- 2. FBM masks after the nearest resize ----------------------------------------------
    fbm = ref_fd.FrequencyBandModulation(4, k_list=[2, 4, 8])
    masks = {}
    for n in (12, 20, 24, 33, 35, 40, 48, 64, 77, 128, 130, 256, 512, 522, 532, 1024, 1044, 1064, 2048, 2088, 2128):
        m = torch.nn.functional.interpolate(fbm.cached_masks.float(), size=(n, n // 2 + 1), mode='nearest')
        # separable: row mask = column 0 of the 2-D mask, column mask = row 0
        masks['rows_%d' % n] = m[:, 0, :, 0].bool().numpy()
        masks['cols_%d' % n] = m[:, 0, 0, :].bool().numpy()
        assert torch.equal(m[:, 0], m[:, 0, :, 0:1] * m[:, 0, 0:1, :])
    np.savez_compressed(os.path.join(HERE, 'fbm_masks.npz'), **masks)

    # ---- 3. whole network: RRDBNet(1,1,64,1,32) with every conv swapped for the reference FDConv
    from oracle.rrdbnet_ref import RRDBNetRef

    def swap(module):  # what train/train_realESRGAN_fdconv_1.py:82-95 does
        for name, child in module.named_children():
            if isinstance(child, nn.Conv2d):
                setattr(module, name, FDConv(child.in_channels, child.out_channels, kernel_size=child.kernel_size,
                                             stride=child.stride, padding=child.padding, bias=child.bias is not None))
            else:
                swap(child)

    class Net(RRDBNetRef):
        def forward(self, x):  # basicsr RRDBNet.forward, written against modules so the reference FDConv runs
            import torch.nn.functional as F
            lrelu = lambda t: F.leaky_relu(t, 0.2)
            feat = self.conv_first(x)
            body = feat
            for blk in self.body:
                t_in = body
                for rdb in (blk.rdb1, blk.rdb2, blk.rdb3):
                    x0 = body
                    x1 = lrelu(rdb.conv1(x0))
                    x2 = lrelu(rdb.conv2(torch.cat((x0, x1), 1)))
                    x3 = lrelu(rdb.conv3(torch.cat((x0, x1, x2), 1)))
                    x4 = lrelu(rdb.conv4(torch.cat((x0, x1, x2, x3), 1)))
                    x5 = rdb.conv5(torch.cat((x0, x1, x2, x3, x4), 1))
                    body = x5 * 0.2 + x0
                body = body * 0.2 + t_in
            feat = feat + self.conv_body(body)
            feat = lrelu(self.conv_up1(F.interpolate(feat, scale_factor=2, mode='nearest')))
            feat = lrelu(self.conv_up2(F.interpolate(feat, scale_factor=2, mode='nearest')))
            return self.conv_last(lrelu(self.conv_hr(feat)))

    net = Net(1, 1, 4, 64, 1, 32)
    swap(net)
    net.load_state_dict(to_torch(gp.rrdbnet_state(1, 1, 64, 1, 32, seed=7)), strict=True)
    net.eval()
    x = torch.from_numpy(gp.image_input((2, 1, 32, 32), seed=70))
    with torch.no_grad():
        y = net(x)
    np.savez_compressed(os.path.join(HERE, 'rrdbnet_fdconv_1block.npz'), y=y.numpy(),
                        n_keys=np.array(len(net.state_dict())))

    # ---- 4. RealESRGANer pad / tile / crop through the reference class ----------------------
    res = {}
    sd = gp.rrdbnet_state(3, 3, 8, 1, 4, seed=9, fdconv=False)
    ckpt = '/tmp/_golden_rrdb_small.pth'
    torch.save({'params': to_torch(sd)}, ckpt)

    def restorer(scale, tile, tile_pad, pre_pad, model):
        return ref_utils.RealESRGANer(scale=scale, model_path=ckpt, model=model, tile=tile, tile_pad=tile_pad,
                                      pre_pad=pre_pad, half=False, device=torch.device('cpu'))

    img = gp.image_input((50, 70, 3), seed=90)
    for tag, tile, pad, pre in (('whole', 0, 10, 0), ('tile32', 32, 10, 0), ('tile32_pre7', 32, 10, 7), ('tile16_pad4', 16, 4, 3)):
        r = restorer(4, tile, pad, pre, RRDBNetRef(3, 3, 4, 8, 1, 4))
        r.pre_process(img)
        with torch.no_grad():
            r.tile_process() if tile > 0 else r.process()
        res['rgb50x70_' + tag] = r.post_process().numpy()
    # the geometry cases of Real-ESRGAN-master/tests/test_utils.py:33-56
    r = restorer(4, 10, 10, 2, RRDBNetRef(3, 3, 4, 8, 1, 4))
    img12 = gp.image_input((12, 12, 3), seed=91)
    r.pre_process(img12)
    res['t_pre_shape'] = np.array(r.img.shape)
    r.scale = 1
    r.pre_process(img12)
    res['t_pre_mod_shape'] = np.array(r.img.shape)
    res['t_pre_mod_img'] = r.img.numpy()
    r.scale = 4
    r.pre_process(img12)
    with torch.no_grad():
        r.tile_process()
    res['t_tile_shape'] = np.array(r.output.shape)
    res['t_tile_out'] = r.output.numpy()
    # scale-2 network (pixel_unshuffle path, mod-pad to even)
    sd2 = gp.rrdbnet_state(12, 3, 8, 1, 4, seed=10, fdconv=False)
    torch.save({'params_ema': to_torch(sd2), 'params': to_torch(sd)}, ckpt)
    r = restorer(2, 0, 10, 0, RRDBNetRef(3, 3, 2, 8, 1, 4))
    img2 = gp.image_input((21, 33, 3), seed=92)
    r.pre_process(img2)
    with torch.no_grad():
        r.process()
    res['x2_21x33'] = r.post_process().numpy()
    os.remove(ckpt)
    np.savez_compressed(os.path.join(HERE, 'realesrganer.npz'), **res)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('%-32s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == '__main__':
    main()
