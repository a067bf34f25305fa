"""Host-side profile of one training step (cProfile): where the Python time of the step goes.
   usage (on the GPU box): python tools/profile_train_host.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as sra  # noqa: E402
from super_resolution_amd import train  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.manual_seed(0)
    net = sra.replace_conv_with_fdconv(sra.RRDBNet(1, 1, 4, 64, 23, 32)).cuda().train()
    net.compute_dtype = torch.bfloat16
    opt = train.FlatAdam(net.parameters(), lr=1e-4)
    wf = torch.rand(32, 1, 64, 64, device='cuda')
    sr = torch.rand(32, 1, 256, 256, device='cuda')
    for _ in range(2):
        train.train_step(net, opt, wf, sr)
    torch.cuda.synchronize()
    # the backward of the layers runs on autograd's device thread, which the profile of this thread does not see
    prb = cProfile.Profile()
    for cls in (train._FDConvFn, train._PlainConvFn):
        orig = cls.backward

        def wrapped(ctx, *a, _orig=orig):
            prb.enable()
            try:
                return _orig(ctx, *a)
            finally:
                prb.disable()
        cls.backward = staticmethod(wrapped)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        train.train_step(net, opt, wf, sr)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(25)
    print('---- layer backward functions (autograd thread) ----')
    pstats.Stats(prb).sort_stats('tottime').print_stats(40)


if __name__ == '__main__':
    main()
