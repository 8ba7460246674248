"""Runs one conv contraction in a loop for a few seconds on random and on all-zero operands and samples rocm-smi
(shader clock, socket power) meanwhile: shows whether operand values change the clock the kernel runs at."""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gan-pytorch_amd"))
import torch  # noqa: E402

from vegans.engine import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [ln.strip() for ln in txt.splitlines() if "sclk" in ln or "Power" in ln]
            out.append(" ; ".join(keep))
        except Exception as exc:  # noqa: BLE001
            out.append("rocm-smi failed: %r" % (exc,))
        time.sleep(0.5)


def main():
    N, H, W, Ci, Co, k, s, p = 256, 32, 32, 64, 128, 4, 2, 1
    g = ops.Geom(N, H, W, Ci, Co, k, k, s, s, p, p)
    for scale in (1.0, 0.0):
        x = (torch.randn(N, Ci, H, W, device="cuda") * scale).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05 * scale).contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            ops.raw_conv_fwd(g, x, w, None)
        torch.cuda.synchronize()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        n = 0
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        while time.time() - t0 < 4.0:
            for _ in range(200):
                ops.raw_conv_fwd(g, x, w, None)
            n += 200
            torch.cuda.synchronize()
        end.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        print("operands x %.0f: %.1f us per launch over %d launches" % (scale, 1e3 * start.elapsed_time(end) / n, n))
        for ln in out[1:6]:
            print("    ", ln)


if __name__ == "__main__":
    main()
