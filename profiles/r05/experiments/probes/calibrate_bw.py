#!/usr/bin/env python3
"""Calibrates what this MI355X sustains for plain streaming patterns, with stock PyTorch kernels
(fill = write-only, copy = read + write, max = read-only), so the extractor's kernels can be read
against the box they ran on rather than against the 8 TB/s datasheet number only."""
import json
import torch


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    n = 1 << 30  # 4 GiB of f32
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    out = {}
    out["fill_write_GBps"] = round(4 * n / timed(lambda: y.fill_(1.0)) / 1e9, 1)
    out["copy_read_plus_write_GBps"] = round(8 * n / timed(lambda: y.copy_(x)) / 1e9, 1)
    out["max_read_GBps"] = round(4 * n / timed(lambda: x.max()) / 1e9, 1)
    out["add_2read_1write_GBps"] = round(12 * n / timed(lambda: torch.add(x, y, out=y)) / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
