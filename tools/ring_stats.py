#!/usr/bin/env python3
"""Who waits for whom inside the ring kernel (csrc/conv_ring.hip built with -DRK_STATS; point LSS_HIP_LIB at that
build): per wave role the number of flag polls spent waiting and the 100-MHz ticks from kernel entry to the end of the
wave's main loop, for the three big BevEncode convs at batch 4.

    LSS_HIP_LIB=diag_libs/liblss_STATS.so python tools/ring_stats.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    buf = torch.zeros(4096 * 12 * 6, dtype=torch.int64, device="cuda")
    os.environ["LSS_RING_STATS"] = "%x" % buf.data_ptr()
    from lss2_multimodal_nu_amd import ops
    from bench_ring import LAYERS
    torch.manual_seed(0)
    for name, B, H, W, Cx, Cout, C2, up, hn in LAYERS:
        x = torch.randn(B, H, W, Cx, device="cuda").bfloat16()
        x2 = torch.randn(B, H * up, W * up, C2, device="cuda").bfloat16() if C2 else None
        w = torch.randn(Cout, Cx + C2, 3, 3, device="cuda") * ((Cx + C2) * 9) ** -0.5
        sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
        wr = ops.pack_conv_weight_ring(w)
        hw, hb = torch.randn(max(hn, 1), Cout, device="cuda") * Cout ** -0.5, torch.randn(max(hn, 1), device="cuda")

        def run():
            if hn:
                return ops.conv3x3_head_nchw(x, wr, sc, sh, hw, hb, x2=x2, up=up)
            return ops.conv2d_nhwc(x, wr, (3, 3), 1, 1, sc, sh, None, True, x2, up, None, 1)
        for _ in range(20):
            run()
        buf.zero_()
        torch.cuda.synchronize()
        run()
        torch.cuda.synchronize()
        nwg = (B * ((H * up + 3) // 4) * ((W * up + 19) // 20) + 3) // 4 * (Cout // 128)
        st = buf[:nwg * 72].view(nwg, 12, 6).cpu().double()
        t0 = st[:, :, 3].min()
        print("%s: %d workgroups, %d chunks" % (name, nwg, (Cx + C2) // 32))
        for role, sl, n0, n1 in (("consumer", slice(0, 8), "polls on FULL_W (slab late)", "polls on FULL_P (patch late)"),
                                 ("weights ", slice(8, 9), "idle polls on FREE_W (ring full)", "-"),
                                 ("patch   ", slice(9, 12), "polls on FREE_P (consumers late)", "polls on FULL_S / FREE_S")):
            s = st[:, sl]
            print("  %s  %-34s mean %8.1f max %7.0f | %-28s mean %7.1f | main loop %6.1f us (max %6.1f), starts at %5.1f us"
                  % (role, n0, s[..., 0].mean(), s[..., 0].max(), n1, s[..., 1].mean(), s[..., 2].mean() / 100,
                     s[..., 2].max() / 100, (s[..., 3] - t0).mean() / 100))
        end = (st[:, :8, 3] + st[:, :8, 2]).max() - t0
        clk = st[:, :8, 4] / (st[:, :8, 5] / 100.0)  # shader cycles per us of the consumers' main loop
        print("  first entry -> last consumer loop end: %.1f us; in-kernel clock over the consumer loops: median %.0f MHz "
              "(min %.0f, max %.0f)" % (end / 100, float(clk.median()), float(clk.min()), float(clk.max())))


if __name__ == "__main__":
    main()
