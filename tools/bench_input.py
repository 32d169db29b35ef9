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
# the reader threads alone (CRC + parse + decode into the staging pool; the consumer only returns the slots)
for label, u8 in (('uint8', True), ('float32', False)):
    inp, tgt = data.inputs(root, 'nyu', B, seed=0)
    sb = inp.pipeline
    sb.allocate(None, (lambda shape: np.empty(shape, np.uint8)) if u8 else None)
    sb.release(sb.dequeue())
    t0 = time.perf_counter()
    for _ in range(steps):
        sb.release(sb.dequeue())
    out[f'decode_only_images_per_s_{label}'] = round(B * steps / (time.perf_counter() - t0), 1)
    out['reader_threads'] = len(sb.threads)
    out['usable_cpus'] = data.usable_cpus()
    out['host_threads'] = os.cpu_count()
    sb.close()
# `make train`'s loop on both transfer paths: converter-written records staged and DMA'd as uint8 pixel values (the default
# for such records, data.py), and everything as float32 (A3D_NO_U8_RECORDS=1: what round 2 measured)
for label, env in (('uint8_records', '0'), ('float32_records', '1')):
    os.environ['A3D_NO_U8_RECORDS'] = env
    inp, tgt = data.inputs(root, 'nyu', B, seed=0)
    op = models.msdn(inp, tgt)
    op.run(); op.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        op.run()
    torch.cuda.synchronize()
    out[f'train_loop_images_per_s_{label}'] = round(B * steps / (time.perf_counter() - t0), 1)
    out[f'staged_as_{label}'] = str(op.cur[0][0].dtype)
    op.pipeline.close()
    del op
os.environ.pop('A3D_NO_U8_RECORDS')
# the same step with its batch resident in HBM (what bench.py times): the ceiling of the loop above
net = models.MSDNReplica(B, keep_dense_grads=False)
ti = torch.from_numpy(np.broadcast_to(img + np.float32(.5), (B,) + img.shape).copy()).cuda()
td = torch.from_numpy(np.broadcast_to(dep + np.float32(.5), (B,) + dep.shape).copy()).cuda()
keep = (torch.rand((B, 4096), device='cuda') >= 0.5).to(torch.uint8)
for _ in range(3):
    net.step(ti, td, keep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    net.step(ti, td, keep)
torch.cuda.synchronize()
out['resident_images_per_s'] = round(B * steps / (time.perf_counter() - t0), 1)
out['loop_over_resident'] = round(out['train_loop_images_per_s_uint8_records'] / out['resident_images_per_s'], 3)
print(json.dumps(out))
import shutil
shutil.rmtree(root, ignore_errors=True)
