import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--config", "c4", "--cost-kind", sys.argv[1], "--xl-rows", "20000", "--xl-cols", "200000", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0", "--batch", "16"]
import bench, torch, numpy as np
args = bench.parse()
torch.cuda.set_device(0)
wl = bench.XLWorkload(args, torch, 0, 0, 1)
for k in range(8):
    wl.step(k)
    st = wl.dev.status()
    print(k, st["rows_in_sweep"][:6], st["remaining"][:6], int(st["row_updates"].sum()))
