import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from kvazaar_amd import _lib
from bench_all import timed
dev = torch.device("cuda", 0); L = _lib.init(0); st = L.kvz_hip_stream_create()
g = torch.Generator(device=dev); g.manual_seed(1)
W, H, F = 1920, 1080, 16
picf = torch.randint(0, 256, (F * H, W), dtype=torch.uint8, device=dev, generator=g)
reff = torch.roll(picf, shifts=(1, 2), dims=(0, 1)).contiguous()
for n in (8, 16):
    rows = [(x, f * H + y) for f in range(4) for y in range(0, H - n + 1, n) for x in range(0, W - n + 1, n)]
    pus = np.zeros((len(rows), 16), dtype=np.int32)
    pus[:, 0] = [r[0] for r in rows]; pus[:, 1] = [r[1] for r in rows]; pus[:, 2] = n; pus[:, 3] = n
    pus_d = torch.from_numpy(pus).to(dev); res_d = torch.empty((len(rows), 8), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for label, prm in (("default", (20, 1, -1, 4, 0, 0, 1, 1, 0, 0, 1, 0)), ("fme0", (20, 1, -1, 0, 0, 0, 1, 1, 0, 0, 1, 0)),
                       ("no early term", (20, 0, -1, 4, 0, 0, 1, 1, 0, 0, 1, 0)), ("fme0 + no ET", (20, 0, -1, 0, 0, 0, 1, 1, 0, 0, 1, 0)),
                       ("max_steps 0, fme0, no ET", (20, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 0)), ("dia", (20, 1, -1, 4, 0, 0, 1, 1, 1, 0, 1, 0)),
                       ("fme2", (20, 1, -1, 2, 0, 0, 1, 1, 0, 0, 1, 0))):
        p = np.zeros(24, dtype=np.int32); p[:len(prm)] = prm          # kvz_hip_me_params is 96 bytes
        ms = min(timed(L, st, lambda: _lib.check(L.kvz_hip_search_pu_batch(picf.data_ptr(), W, W, F * H, reff.data_ptr(), W, W, F * H, pus_d.data_ptr(), len(rows), p.ctypes.data, res_d.data_ptr(), st), "x")) for _ in range(3))
        print("%2dx%-2d %-26s %8.1f M/s  %6.1f us" % (n, n, label, len(rows) / ms / 1e3, ms * 1e3))
