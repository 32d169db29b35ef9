"""Rate of the dataset plugin alone and of `make train`'s whole loop (records -> shuffle queue -> pinned -> H2D ->
step) on a synthetic NYU-shaped shard (480x640x3 + 480x640x1 float32 records, 4.9 MB each).
    python tools/bench_input.py [n_records] [batch] [steps]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import data, models, tfrecord  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
root = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
rng = np.random.default_rng(0)
os.makedirs(os.path.join(root, 'nyu'))
img = rng.integers(0, 256, (480, 640, 3)).astype(np.float32) / np.float32(255) - np.float32(.5)
dep = rng.integers(0, 256, (480, 640, 1)).astype(np.float32) / np.float32(255) - np.float32(.5)
t0 = time.perf_counter()
with tfrecord.TFRecordWriter(os.path.join(root, 'nyu', 'train.tfrecords')) as w:
    for i in range(n):
        w.write_example(img, dep)
t_write = time.perf_counter() - t0
out = {'records': n, 'record_MB': round((img.nbytes + dep.nbytes) / 1e6, 2), 'write_records_per_s': round(n / t_write, 1)}
inp, tgt = data.inputs(root, 'nyu', B, seed=0)
sb = inp.pipeline
sb.next_batch()
t0 = time.perf_counter()
for _ in range(steps):
    sb.next_batch()
out['reader_images_per_s'] = round(B * steps / (time.perf_counter() - t0), 1)
sb.close()
inp, tgt = data.inputs(root, 'nyu', B, seed=0)
op = models.msdn(inp, tgt)
op.run(); op.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    op.run()
torch.cuda.synchronize()
out['train_loop_images_per_s'] = round(B * steps / (time.perf_counter() - t0), 1)
op.pipeline.close()
print(json.dumps(out))
import shutil
shutil.rmtree(root, ignore_errors=True)
