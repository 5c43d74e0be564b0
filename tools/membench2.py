#!/usr/bin/env python3
"""Streaming-read ceiling for a working set that is re-read every launch (the headline kernel's situation): sizes
from 19 MB to 600 MB, loads in flight per lane, chunked vs grid-strided, non-temporal loads.  One JSON line per case."""
import ctypes as C, json, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "membench", "libmembench.so"))
lib.membench_read2.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
def timed(fn, n=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n
for mb in (19, 38, 76, 157, 302, 600):
    n4 = mb * (1 << 20) // 16
    src = torch.empty(n4 * 4, dtype=torch.float32, device=dev).normal_()
    best = None
    for (u, ch, nt) in ((4, 0, 0), (8, 0, 0), (16, 0, 0), (4, 1, 0), (8, 1, 0), (16, 1, 0), (4, 0, 1), (8, 0, 1), (8, 1, 1)):
        for blocks in (1024, 2048, 4096, 8192):
            sink = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
            rc = lib.membench_read2(src.data_ptr(), n4, sink.data_ptr(), blocks, u, ch, nt, stream)
            assert rc == 0, rc
            t = timed(lambda: lib.membench_read2(src.data_ptr(), n4, sink.data_ptr(), blocks, u, ch, nt, stream), 100 if mb < 400 else 40)
            r = dict(MB=mb, unroll=u, chunked=ch, nt=nt, blocks=blocks, us=round(t * 1e6, 2), GBps=round(n4 * 16 / t / 1e9))
            if best is None or r["GBps"] > best["GBps"]: best = r
            if blocks == 4096: print(json.dumps(r))
    print(json.dumps(dict(best=best)))
    del src
