"""one-off: how fast does a trivial kernel read N interleaved 19.2 MB streams?  (lbl_sum_dev over n_in arrays; run under rocprofv3)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyrad_amd import _native as nat
ctx = nat.Context(0)
n = 2400000
for n_in in (3, 12, 48):
    bufs = [ctx.buffer(n).fill(0.0) for _ in range(n_in)]
    out = ctx.buffer(n)
    for _ in range(30):
        ctx.sum_dev(bufs, n, out)
    ctx.sync()
    for b in bufs:
        b.free()
    out.free()
ctx.close()
