"""Layer by layer: the critic's forward pass on the engine in a bf16 mode against the CPU restatement of that mode
(oracle/bf16_emulation.py), on real-like (uniform noise) and fake-like (small) images. Prints, per stored tensor, the
l2 distance between the two relative to the tensor's norm. Diagnostic for tests/test_bf16_emulation_gpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gan-pytorch_amd"), os.path.join(ROOT, "tests")]

import copy  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

import bench_configs as cfg  # noqa: E402
from oracle.bf16_emulation import emulate  # noqa: E402
from vegans.engine import modules, ops  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = 64
    D = cfg.dcgan_init(cfg.dcgan_critic(64, 3, 64), 2)
    emu = emulate(copy.deepcopy(D), mode).train()
    eng = copy.deepcopy(D).to("cuda")
    modules.bind(eng)
    modules.prepare_parameters(eng)
    eng.train()
    for label, X in (("uniform(-1,1) images", torch.from_numpy(cfg.synthetic_images(B, (3, 64, 64), seed=40))),
                     ("small images", 0.03 * torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(1)))):
        print(label)
        # CPU restatement, every module's output
        outs_c, x = [], X
        with torch.no_grad():
            for m in emu:
                x = m(x.clone() if isinstance(m, (nn.LeakyReLU, nn.ReLU)) else x)
                outs_c.append((type(m).__name__, x.clone()))
        # engine: leaf forward methods wrapped to record what each call returns
        rec = []
        saved = []
        for cls in (modules.HipBatchNorm2d, modules.HipConv2d, modules.HipLeakyReLU, modules.HipFlatten):
            plain = cls.forward

            def forward(self, *a, plain=plain, **k):
                y = plain(self, *a, **k)
                rec.append((type(self).__name__ + ("+act" if len(a) > 1 and a[1] else ""), y.detach().float().cpu()))
                return y

            saved.append((cls, plain))
            cls.forward = forward
        previous = ops.set_matmul_precision(mode)
        try:
            with torch.no_grad():
                eng(ops.to_nhwc(X.to("cuda")))
        finally:
            ops.set_matmul_precision(previous)
            for cls, plain in saved:
                cls.forward = plain
        # match engine records to the restatement's stored tensors by shape, in order
        ci = 0
        for name, y in rec:
            while ci < len(outs_c) and outs_c[ci][1].shape != y.shape:
                ci += 1
            cands = []
            for cj in range(ci, min(ci + 6, len(outs_c))):
                if outs_c[cj][1].shape == y.shape:
                    d = float((outs_c[cj][1] - y).norm() / outs_c[cj][1].norm().clamp_min(1e-30))
                    cands.append("#%d %s %.2e" % (cj, outs_c[cj][0], d))
            print("   %-24s %-18s restatement tensors of that shape, rel l2: %s" % (name, tuple(y.shape), "; ".join(cands)))


if __name__ == "__main__":
    main()
