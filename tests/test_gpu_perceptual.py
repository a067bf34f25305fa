"""The perceptual (VGG19-feature L1) term of BASELINE configs[3] on the GPU against the same extractor in torch on the CPU.

Reference: Real-ESRGAN-master/options/train_realesrgan_x4plus.yml:133-149 (basicsr PerceptualLoss; basicsr is not
vendored and no pretrained VGG weights exist offline, so the extractor is a seeded random initialisation of the same
architecture -- PARITY UNPINNED against basicsr itself, pinned against torch's own conv / relu / max_pool / l1 operators).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _torch_loss(vgg, pred, target, weights):
    if pred.shape[-2:] != target.shape[-2:]:
        pred = F.interpolate(pred, size=target.shape[-2:], mode='bilinear', align_corners=False)
    if pred.shape[1] == 1:
        pred, target = pred.expand(-1, 3, -1, -1), target.expand(-1, 3, -1, -1)
    fp, ft = vgg(pred), vgg(target)
    return sum(w * F.l1_loss(fp[k], ft[k]) for k, w in weights.items())


@pytest.mark.parametrize('ch,hp,ht', [(1, 32, 32), (3, 32, 32), (1, 16, 48)])
def test_perceptual_loss_and_gradient_vs_torch_cpu(ch, hp, ht):
    from super_resolution_amd import perceptual as P
    vgg = P.VGG19Features(seed=11)
    assert [k for k in vgg.state_dict()][:2] == ['features.0.weight', 'features.0.bias']  # torchvision's vgg19 key layout
    assert len(vgg.state_dict()) == 32 and vgg.state_dict()['features.34.weight'].shape == (512, 512, 3, 3)
    g = torch.Generator().manual_seed(ch * 100 + hp)
    pred = torch.rand(2, ch, hp, hp, generator=g)
    target = torch.rand(2, ch, ht, ht, generator=g)
    weights = dict(P.DEFAULT_LAYER_WEIGHTS)
    p = pred.clone().requires_grad_(True)
    ref = _torch_loss(vgg, p, target, weights)
    ref.backward()
    vg = P.VGG19Features(seed=11).cuda()
    res = {}
    for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
        q = pred.cuda().requires_grad_(True)
        loss = P.perceptual_loss(q, target.cuda(), vg, compute_dtype=dt)
        (loss * 2.0).backward()
        res[name] = (loss.item(), q.grad.cpu() / 2.0)
    l32, g32 = res['f32']
    lb, gb = res['bf16']
    gmax = p.grad.abs().max().item()
    print('perceptual %dch %d->%d: loss torch %.6f, hip f32 %.6f, bf16 %.6f; grad err f32 %.2e bf16 %.2e of %.2e'
          % (ch, hp, ht, ref.item(), l32, lb, (g32 - p.grad).abs().max().item(), (gb - p.grad).abs().max().item(), gmax))
    assert abs(l32 - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    # The loss is piecewise linear in the prediction: a float32 rounding difference can flip the sign of one |a - b| term
    # or the arg-max of one pooling window, which moves single gradient entries -- compare in the mean, bound the worst.
    assert (g32 - p.grad).abs().mean().item() <= 1e-4 * gmax
    assert (g32 - p.grad).abs().max().item() <= 5e-2 * gmax
    assert abs(lb - ref.item()) <= 2e-2 * abs(ref.item())
    cos = float((gb * p.grad).sum() / (gb.norm() * p.grad.norm()))
    assert cos > 0.97, cos


def test_train_step_with_perceptual_term(tmp_path):
    """L1 + perceptual (BASELINE configs[3]) through train.train_step on a 1-block RRDBNet-FDConv: finite, and different from
    the L1-only loss by the perceptual term."""
    import sys
    import super_resolution_amd as sra
    from super_resolution_amd import perceptual as P
    torch.manual_seed(0)
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 1, 32)).cuda()
    opt = sra.train.FlatAdam(net.parameters(), lr=1e-4)
    vgg = P.VGG19Features(seed=3).cuda()
    g = torch.Generator().manual_seed(5)
    srt = (torch.rand(4, 1, 64, 64, generator=g) * torch.linspace(0.3, 1.0, 4).view(4, 1, 1, 1)).cuda()
    wf = F.avg_pool2d(srt, 4)
    l1 = float(sra.train.train_step(net, opt, wf, srt))
    both = [float(sra.train.train_step(net, opt, wf, srt, perceptual=dict(vgg=vgg))) for _ in range(3)]
    assert all(torch.isfinite(torch.tensor(both))) and both[0] > l1
    assert float(opt.grad.abs().max()) > 0
