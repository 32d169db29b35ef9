import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from ann3depth_amd import models
for prec, B in (('fp32', 32), ('bf16s', 64)):
    dev = torch.device('cuda:0')
    net = models.MSDNReplica(B, device=dev, seed=3000, precision=prec, keep_dense_grads=False)
    img, dep = bench.synth_batch(B, 0, dev)
    masks = bench.keep_masks(B, 8, 0, dev)
    for i in range(10): net.step(img, dep, masks[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50): net.step(img, dep, masks[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(prec, 'enqueue ms/step', (t1 - t0) / 50 * 1e3, 'total ms/step', (t2 - t0) / 50 * 1e3)
