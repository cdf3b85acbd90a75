"""tools/probes/tile_stream.hip: HBM rate of a persistent tile-by-tile copy of a (rows x N) bf16 matrix against the tile width."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libtile_stream.so"))
lib.mk_probe_tile_copy.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
for rows, N in ((384, 1038240 // 1024 * 1024), (768, 1038240 // 1024 * 1024), (384, 115200 // 1024 * 1024), (768, 115200 // 1024 * 1024)):
    NB = 6
    xs = [torch.randn(rows, N, device=dev).bfloat16() for _ in range(NB)]
    ys = [torch.empty_like(x) for x in xs]
    for write, what in ((1, "copy (read + write)"), (0, "read only")):
        line = []
        for tw in (64, 128, 256, 512, 1024):
            for grid in (256, 512, 1024):
                def run(i):
                    lib.mk_probe_tile_copy(tw, ctypes.c_void_p(xs[i % NB].data_ptr()), ctypes.c_void_p(ys[i % NB].data_ptr()), rows, N, write, grid,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                for i in range(6): run(i)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(24): run(i)
                b.record(); torch.cuda.synchronize()
                us = a.elapsed_time(b) / 24 * 1e3
                gb = rows * N * 2 * (2 if write else 1) / us / 1e3
                line.append(f"tw{tw}/g{grid}: {gb:5.0f}")
        print(f"rows {rows} N {N} {what}: GB/s  " + "  ".join(line), flush=True)
