#!/usr/bin/env python
"""The clock the chip holds inside k_w4_gemm64b's K loop, measured on the DIAGNOSTICS build of the library
(libnode_hip_diag.so, build.py --diag): every wave stamps s_memrealtime (100 MHz) and s_memtime (shader clock) around the loop
(NODE_TUNE_W4_STAMPS).  Median over the waves of one launch behind ten warm-up launches.  Prints one JSON line
{"held_clock_ghz": x | null}.  The product library has no stamps: bench.py runs this file as a child process.

    NODE_HIP_DIAG=1 python tools/held_clock.py N,C,side
"""
import ctypes as C
import json
import os
import sys

os.environ['NODE_HIP_DIAG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    N, Cc, side = (int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else (128, 256, 8)))
    nn = N * (4 if side == 16 else 1)
    if Cc >= 512 or nn % 16 != 0 or Cc % 64 != 0:
        print(json.dumps({'held_clock_ghz': None}))
        return
    import torch
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    shape = _lib.NodeShape(N, Cc, side, side, 32, 1e-5)
    x = torch.randn(N, Cc, side, side, device='cuda')
    w = torch.randn(Cc, Cc + 1, 3, 3, device='cuda') / 48
    nbytes = lib.node_conv3x3_w4_workspace_bytes(C.byref(shape))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
    base = (ws.data_ptr() + 255) & ~255
    y = torch.empty_like(x)
    stamps = torch.zeros((nn // 16) * (Cc // 64) * 8 * 4, 16, dtype=torch.int64, device='cuda')
    try:
        for it in range(11):
            if it == 10:
                os.environ['NODE_TUNE_W4_STAMPS'] = hex(stamps.data_ptr())
            _lib.check(lib.node_conv3x3_w4(C.byref(shape), w.data_ptr(), 0, x.data_ptr(), y.data_ptr(), base, nbytes,
                                           torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        os.environ.pop('NODE_TUNE_W4_STAMPS', None)
    raw = stamps.cpu().double()
    wall = (raw[:, 3] - raw[:, 1]) / 100.0           # us
    ok = wall > 0
    clk = float(((raw[:, 11] - raw[:, 9])[ok] / wall[ok]).median()) / 1e3 if int(ok.sum()) else None
    print(json.dumps({'held_clock_ghz': clk}))


if __name__ == '__main__':
    main()
