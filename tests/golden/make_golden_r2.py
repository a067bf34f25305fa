#!/usr/bin/env python
"""Round-2 golden fixtures from the REFERENCE implementation (build container only; needs /root/reference).

Adds to make_golden.py (whose fixtures are left untouched):
  fdconv_train_grads.npz   reference FDConv in train() mode (batch-statistics BatchNorm): outputs, the updated
                           BatchNorm running statistics and the gradients of a weighted-sum loss wrt the input and
                           EVERY parameter, for two layer shapes;
  rrdbnet_train_step.npz   one training step of the 1-block RRDBNet whose convs are the reference FDConv, shaped as
                           train/train_realESRGAN_fdconv_1.py:112-118 (bilinear resize of the prediction + L1): loss,
                           prediction and the gradients of a handful of parameters;
  convert_ref.npz          the reference tools/convert_to_fdconv.py run on a small checkpoint: its dft_weight output;
  srvgg.npz                reference SRVGGNetCompact outputs (Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py);
  fdconv_layers_r2.npz     the (130, 77) single-layer case SURVEY 8c lists.
The same import-time placeholders as make_golden.py are used (mmcv registry decorator, timm, cv2, basicsr download
helper) plus basicsr.utils.registry.ARCH_REGISTRY.register(), a decorator srvgg_arch.py:1,6 applies.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_params as gp  # noqa: E402
import make_golden as mg  # noqa: E402

REF = '/root/reference'
SRVGG_PY = os.path.join(REF, 'Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py')
CONVERT_PY = os.path.join(REF, 'FDConv-main/FDConv-main/tools/convert_to_fdconv.py')


def load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def weights_like(y, seed):
    return torch.from_numpy(gp.signed_input(tuple(y.shape), seed))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_fd, _ = mg.load_reference()
    FDConv = ref_fd.FDConv

    # ---- 1. train-mode layers with gradients ------------------------------------------------------------------
    out = {}
    for tag, cin, cout, B, h, w, pseed in (('c64_32', 64, 32, 4, 20, 24, 100), ('c192_64', 192, 64, 2, 16, 12, 104)):
        layer = FDConv(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
        layer.load_state_dict(mg.to_torch(gp.fdconv_params(cin, cout, seed=pseed)), strict=True)
        layer.train()
        x = torch.from_numpy(gp.signed_input((B, cin, h, w), seed=300)).requires_grad_(True)
        y = layer(x)
        r = weights_like(y, 301)
        loss = (y * r).sum()
        loss.backward()
        out[tag + '.y'] = y.detach().numpy()
        out[tag + '.loss'] = np.array(loss.item())
        out[tag + '.grad.x'] = x.grad.numpy()
        for name, p in layer.named_parameters():
            out[tag + '.grad.' + name] = p.grad.numpy()
        out[tag + '.bn.running_mean'] = layer.KSM_Global.bn.running_mean.numpy().copy()
        out[tag + '.bn.running_var'] = layer.KSM_Global.bn.running_var.numpy().copy()
        out[tag + '.bn.num_batches_tracked'] = layer.KSM_Global.bn.num_batches_tracked.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'fdconv_train_grads.npz'), **out)

    # ---- 2. one training step of the 1-block network ----------------------------------------------------------
    from oracle.rrdbnet_ref import RRDBNetRef
    import torch.nn as nn

    def swap(module):
        for name, child in module.named_children():
            if isinstance(child, nn.Conv2d):
                setattr(module, name, FDConv(child.in_channels, child.out_channels, kernel_size=child.kernel_size,
                                             stride=child.stride, padding=child.padding, bias=child.bias is not None))
            else:
                swap(child)

    class Net(RRDBNetRef):
        def forward(self, x):
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
    net.load_state_dict(mg.to_torch(gp.rrdbnet_state(1, 1, 64, 1, 32, seed=7)), strict=True)
    net.train()
    wf = torch.from_numpy(gp.image_input((2, 1, 16, 16), seed=400))
    sr = torch.from_numpy(gp.image_input((2, 1, 80, 80), seed=401))
    pred = net(wf)
    pred_r = F.interpolate(pred, size=sr.shape[-2:], mode='bilinear', align_corners=False)  # train_realESRGAN_fdconv_1.py:115
    loss = F.l1_loss(pred_r, sr)                                                              # :116 (nn.L1Loss)
    loss.backward()
    res = {'pred': pred.detach().numpy(), 'loss': np.array(loss.item())}
    params = dict(net.named_parameters())
    for name in ('conv_first.weight', 'conv_first.bias', 'body.0.rdb1.conv1.dft_weight', 'body.0.rdb1.conv1.bias',
                 'body.0.rdb2.conv3.KSM_Global.fc.weight', 'body.0.rdb2.conv3.KSM_Local.conv.weight',
                 'body.0.rdb3.conv5.dft_weight', 'body.0.rdb3.conv5.FBM.freq_weight_conv_list.1.weight',
                 'body.0.rdb3.conv5.KSM_Global.bn.weight', 'body.0.rdb1.conv2.KSM_Global.channel_fc.weight',
                 'conv_body.dft_weight', 'conv_up1.dft_weight', 'conv_up2.KSM_Global.filter_fc.bias',
                 'conv_hr.FBM.freq_weight_conv_list.0.bias', 'conv_last.weight', 'conv_last.bias'):
        res['grad.' + name] = params[name].grad.numpy()
    res['bn_rm.body.0.rdb1.conv1'] = net.body[0].rdb1.conv1.KSM_Global.bn.running_mean.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'rrdbnet_train_step.npz'), **res)

    # ---- 3. reference checkpoint converter --------------------------------------------------------------------
    conv = load_file('ref_convert', CONVERT_PY)
    r = np.random.RandomState(500)
    ck = {'state_dict': {'layer1.0.conv2.weight': torch.from_numpy(r.standard_normal((64, 32, 3, 3)).astype(np.float32)),
                         'layer1.0.conv1.weight': torch.from_numpy(r.standard_normal((48, 96, 1, 1)).astype(np.float32)),
                         'layer1.0.conv3.weight': torch.from_numpy(r.standard_normal((64, 16, 3, 3)).astype(np.float32))}}
    src, dst = '/tmp/_golden_conv_src.pth', '/tmp/_golden_conv_dst.pth'
    torch.save(ck, src)
    argv = sys.argv
    sys.argv = ['convert_to_fdconv.py', '--model_type', 'resnet', '--weight_path', src, '--save_path', dst]
    try:
        conv.main()
    finally:
        sys.argv = argv
    new = torch.load(dst, map_location='cpu', weights_only=True)
    cres = {'keys': np.array(sorted(new.keys()))}
    for k, v in new.items():
        cres[k] = v.numpy()
    cres['idx_96_97'] = conv.get_fft2freq(96, 192, use_rfft=True).numpy()
    os.remove(src)
    os.remove(dst)
    np.savez_compressed(os.path.join(HERE, 'convert_ref.npz'), **cres)

    # ---- 4. SRVGGNetCompact ---------------------------------------------------------------------------------------
    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls
    m = types.ModuleType('basicsr.utils.registry')
    m.ARCH_REGISTRY = _Reg()
    sys.modules['basicsr.utils.registry'] = m
    srvgg = load_file('ref_srvgg', SRVGG_PY)
    sres = {}
    for tag, args, act, shape in (('anime', (3, 3, 64, 16, 4), 'prelu', (2, 3, 24, 20)),
                                  ('x2_relu', (3, 3, 32, 2, 2), 'relu', (1, 3, 17, 13)),
                                  ('gray_leaky', (1, 1, 64, 3, 4), 'leakyrelu', (1, 1, 9, 33))):
        net = srvgg.SRVGGNetCompact(*args, act_type=act)
        net.load_state_dict(mg.to_torch(gp.srvgg_state(*args, seed=600, act_type=act)), strict=True)
        net.eval()
        x = torch.from_numpy(gp.image_input(shape, seed=601))
        with torch.no_grad():
            sres[tag] = net(x).numpy()
        sres[tag + '.n_keys'] = np.array(len(net.state_dict()))
    np.savez_compressed(os.path.join(HERE, 'srvgg.npz'), **sres)

    # ---- 5. the (130, 77) layer case ------------------------------------------------------------------------------
    layer = FDConv(64, 32, kernel_size=3, stride=1, padding=1, bias=True)
    layer.load_state_dict(mg.to_torch(gp.fdconv_params(64, 32, seed=100)), strict=True)
    layer.eval()
    x = torch.from_numpy(gp.signed_input((2, 64, 130, 77), seed=200))
    with torch.no_grad():
        rec = mg.capture_intermediates(layer, x)
    np.savez_compressed(os.path.join(HERE, 'fdconv_layers_r2.npz'), **{'c64_32_h130_w77.y': rec['y']})
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('%-32s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == '__main__':
    main()
