"""CPU oracle for the RRDBNet-FDConv hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``super_resolution_amd/`` may import this
package.  The only legal callers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- always as the checker / the timed CPU baseline,
never as the thing that is shipped or measured as the product.

The oracle is an independent restatement (plain PyTorch on the CPU, fp32 or fp64) of
what the reference computes on this path.  It is pinned to the reference by the golden
fixtures under ``tests/golden/`` which were produced by importing the reference's own
``FDConv.py`` / ``realesrgan/utils.py`` in the build container (``tests/golden/make_golden.py``).
RRDBNet itself lives in the un-vendored ``basicsr==1.4.2`` package which is not present,
so that part is restated from its published architecture and pinned only by
construction (conv2d / leaky_relu / cat / nearest-2x from torch) -- see DESIGN.md.
"""
