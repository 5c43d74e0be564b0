#!/usr/bin/env python3
"""Achievable streaming bandwidth on this MI355X for the sizes the headline kernel works on
(tools/membench/membench.hip): read-only, copy, and the headline's own read:write mix.  Prints one
JSON line per case; DESIGN.md section 6 quotes them next to the 8 TB/s spec peak."""
import ctypes as C
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "membench", "libmembench.so"))
lib.membench_read.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p]
lib.membench_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, n=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


for label, mb in (("headline working set (157 MB read)", 157), ("C5 working set (1 180 MB read)", 1180),
                  ("4 GB read", 4096)):
    n4 = mb * (1 << 20) // 16
    src = torch.empty(n4 * 4, dtype=torch.float32, device=dev).normal_()
    for blocks in (2048, 4096, 8192):
        sink = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
        t = timed(lambda: lib.membench_read(src.data_ptr(), n4, sink.data_ptr(), blocks, stream), 200 if mb < 2000 else 30)
        print(json.dumps(dict(case="read", what=label, blocks=blocks, us=round(t * 1e6, 1), GBps=round(n4 * 16 / t / 1e9))))
    del src
for label, mb in (("copy 85 MB -> 85 MB", 85), ("copy 1 GB -> 1 GB", 1024)):
    n4 = mb * (1 << 20) // 16
    src = torch.empty(n4 * 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    t = timed(lambda: lib.membench_copy(src.data_ptr(), dst.data_ptr(), n4, 8192, stream), 100)
    print(json.dumps(dict(case="copy", what=label, us=round(t * 1e6, 1), GBps=round(2 * n4 * 16 / t / 1e9))))
