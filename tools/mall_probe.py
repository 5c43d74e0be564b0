#!/usr/bin/env python3
"""Probe: which load cache-policy bits let a SMALL buffer that is re-read every iteration (the BCD column sums, 134 MB)
stay in the Infinity Cache while a LARGE stream (h_r, 1.1 GB) goes past it?  Per policy of the large reader: time of the
small (default-policy) reader right after it, and of the large reader itself.  tools/membench kernels only."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "membench", "libmembench.so"))
lib.membench_read_aux.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
big_mb, small_mb = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (1100, 134)))
big = torch.empty(big_mb * (1 << 20) // 4, dtype=torch.float32, device=dev).normal_()
small = torch.empty(small_mb * (1 << 20) // 4, dtype=torch.float32, device=dev).normal_()
sink = torch.empty(4096 * 256, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
names = {0: "default", 1: "sc0", 2: "nt", 3: "sc0 nt", 16: "sc1", 17: "sc0 sc1", 18: "sc1 nt", 19: "sc0 sc1 nt"}

def rd(buf, aux, blocks=2048):
    rc = lib.membench_read_aux(buf.data_ptr(), buf.numel() // 4, sink.data_ptr(), blocks, aux, st)
    assert rc == 0, rc

def timed(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

out = {"big_MB": big_mb, "small_MB": small_mb}
for _ in range(3):
    rd(small, 0)
out["small_alone_us"] = timed(lambda: rd(small, 0))
for aux, name in names.items():
    def pair():
        rd(big, aux); rd(small, 0)
    for _ in range(3):
        pair()
    t_pair = timed(pair, 10)
    t_big = timed(lambda: rd(big, aux), 10)
    out[name] = dict(big_us=round(t_big, 1), small_after_big_us=round(t_pair - t_big, 1), big_GBps=round(big.numel() * 4 / t_big / 1e3, 0))
print(json.dumps(out))
