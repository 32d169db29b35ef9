"""bench.py — images/sec of the MSDN train step (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches it under torch.distributed.run, one rank per GPU; gradients are all-reduced with
RCCL).  Prints ONE JSON line on rank 0: whole-job images/s with inputs resident in HBM, the roofline of the dominant
kernel (HIP-event timed inside the timed region) and the CPU baseline (the numpy oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak = fp32 vector peak
METRIC = 'images/sec (640x480->55x74 MSDN train step)'
# algorithmic fp32 FLOPs per image of one train step (2*M*N*K per GEMM-equivalent; forward of both networks +
# the backward of the phase's variables) and the step's algorithmic HBM bytes per image at B = 32
STEP_GFLOP_PER_IMAGE = {'coarse': 9.357, 'fine': 6.302}


def step_roofline(phase, B, seconds_per_step):
    """The whole step against the fp32 matrix-core bound (the fp32 step is FLOP-bound): coarse 1.904 ms, fine
    1.282 ms at B = 32."""
    gflop = STEP_GFLOP_PER_IMAGE[phase] * B
    tf = gflop / seconds_per_step / 1e3
    return {'bound': 'mfma', 'scope': f'{phase}-phase step, 2*M*N*K of every GEMM-equivalent (SURVEY 8d)',
            'gflop_per_step': round(gflop, 1), 'achieved': round(tf, 1), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(tf / PEAK_F32_MFMA_TFLOPS, 4), 'bound_ms': round(gflop / PEAK_F32_MFMA_TFLOPS, 3)}


def synth_batch(B, rank, device):
    """BASELINE.md synthetic inputs: u8 images /255 (seed 1000+rank); per-image min-max 8-bit depth /255
    (seed 2000+rank), stored-size 480x640."""
    rng = np.random.default_rng(1000 + rank)
    img = (rng.integers(0, 256, (B, 480, 640, 3), dtype=np.uint8).astype(np.float32) / np.float32(255))
    rng = np.random.default_rng(2000 + rank)
    ys, xs = np.mgrid[0:480, 0:640].astype(np.float32)
    dep = np.empty((B, 480, 640, 1), np.float32)
    for b in range(B):
        a, c, p = rng.uniform(0.002, 0.02, 3)
        f = np.sin(a * xs + p) + np.cos(c * ys) + 0.002 * ys
        f = (f - f.min()) / (f.max() - f.min())
        dep[b, :, :, 0] = np.round(f * 255).astype(np.uint8).astype(np.float32) / np.float32(255)
    return torch.from_numpy(img).to(device), torch.from_numpy(dep).to(device)


def keep_masks(B, n, rank, device):
    out = []
    for step in range(n):
        rng = np.random.default_rng(4000 + step + 100000 * rank)
        out.append(torch.from_numpy((rng.random((B, 4096)) >= 0.5).astype(np.uint8)).to(device))
    return out


def collect_timing(lib):
    from ann3depth_amd._lib import TimingRecord
    cap = 1 << 16
    arr = (TimingRecord * cap)()
    n = lib.a3d_timing_collect(arr, cap)
    return [arr[i] for i in range(n)]


def kernel_name(r):
    if r.lds_dma == 5:
        return f'igemm2_kernel<{r.mode}>'
    if r.lds_dma == 4:
        return f'fewch16_bwdf_kernel<{r.bn // 32}>' if r.prec else f'fewch_bwdf_kernel<{r.bn // 32}>'
    if r.lds_dma == 3:
        return f'igemm_ring_kernel<{r.mode}, {r.bm}, {r.bn}>'
    if r.prec:
        return f'igemm_bf16_kernel<{r.mode}, {r.bm}, {r.bn}, {"true" if r.prec == 1 else "false"}>'
    if r.lds_dma == 2:
        return f'conv3_fwd_kernel<{r.bm // 32}, {r.bn // 32}>'
    if r.lds_dma:
        return f'igemm_glds_kernel<{r.mode}, {r.bm}, {r.bn}, {r.waves_m}, {r.nwaves}>'
    return f'igemm_kernel<{r.mode}, {r.bm}, {r.bn}, {r.waves_m}, {r.nwaves}, {r.bk}, {r.avec}, {r.bvec}>'


EVENT_EVERY = 4      # steps between the ones whose roofline-kernel launches carry HIP events (a pair costs ~15 us of stream time)


def dominant_record(recs):
    """The record of the kernel with the largest summed duration (its template fields name it for a3d_timing_select)."""
    total, first = {}, {}
    for r in recs:
        n = kernel_name(r)
        total[n] = total.get(n, 0.0) + r.ms
        first.setdefault(n, r)
    return first[max(total, key=total.get)] if total else None


def warm_then_time(step, settle, steps, warmup, lib, world, timed_kernels):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize.  With timed_kernels every GEMM
    launch of the warm-up steps (but the first) is bracketed by HIP events — the per-kernel table — and inside the timed
    region only the launches of the dominant kernel are, in every EVENT_EVERY-th step: an event pair costs the stream
    ~15 us; 30 of them per step were 3.5 % of the fp32 step (a3d.h, a3d_timing_select).  -> (seconds, records of the timed region, warm-up records)"""
    import torch.distributed as dist
    warm, like = [], None
    for i in range(warmup):
        if timed_kernels and (i == 1 or warmup == 1):
            lib.a3d_timing_select(None)
            lib.a3d_timing_enable(1)
        step(i)
    if timed_kernels:
        torch.cuda.synchronize()
        lib.a3d_timing_enable(0)
        warm = collect_timing(lib)
        like = dominant_record(warm)
        lib.a3d_timing_select(like)          # None (no warm-up to learn from): every launch, as before
        lib.a3d_timing_enable(1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if timed_kernels and like is not None:
            lib.a3d_timing_enable(int(i % EVENT_EVERY == 0))      # the roofline kernel's events: every fourth step
        step(warmup + i)
    settle()            # N > 1: the last step's dense-bucket collective + ApplyAdam belong inside the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.a3d_timing_enable(0)
    lib.a3d_timing_select(None)
    recs = collect_timing(lib) if timed_kernels else []
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, recs, (warm or recs)


def run_phase(net, img, dep, masks, steps, warmup, global_step, lib, world, timed_kernels):
    net.global_step = global_step
    dt, recs, warm = warm_then_time(lambda i: net.step(img, dep, masks[i % len(masks)]), net.settle, steps, warmup, lib, world,
                                    timed_kernels)
    return dt, (recs, warm)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/*_pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 x2 read correction applied); None if not profiled."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            k = json.load(open(path))['kernels'].get('a3d::' + kernel)
        except (OSError, ValueError, KeyError):
            continue
        if k:
            return k['hbm_bytes_per_launch'], os.path.relpath(path, ROOT)
    return None, None


def roofline_from(recs, with_traffic=True):
    groups = {}
    for r in recs:
        g = groups.setdefault(kernel_name(r), {'ms': 0.0, 'flops': 0.0, 'calls': 0})
        g['ms'] += r.ms
        g['flops'] += r.flops
        g['calls'] += 1
    if not groups:
        return None, {}
    name, g = max(groups.items(), key=lambda kv: kv[1]['ms'])
    achieved = g['flops'] / (g['ms'] * 1e-3) / 1e12
    table = {k: {'calls': v['calls'], 'avg_us': round(1e3 * v['ms'] / v['calls'], 2),
                 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)} for k, v in groups.items()}
    roof = {'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            'traffic': pmc_traffic(name)[0] if with_traffic else None,
            'calls': g['calls'], 'avg_launch_us': round(1e3 * g['ms'] / g['calls'], 2),
            'flops_per_launch': g['flops'] / g['calls']}
    if roof['traffic'] is not None:
        # what the counters can and cannot say (profiles/r04_ic_evidence.txt): FETCH_SIZE / WRITE_SIZE count the L2s' fabric
        # requests, Infinity-Cache hits included, and gfx950 has no counter that separates those from HBM reads
        # (TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ on every launch)
        roof['traffic_fabric'] = roof['traffic']
        roof['traffic_hbm'] = None
        roof['traffic_source'] = pmc_traffic(name)[1]      # NOT measured by this run: the committed counter passes of this command
        roof['traffic_note'] = ('fabric-side bytes per launch from the committed PMC passes of this command (profiles/*_pmc_traffic.json), '
                                'Infinity-Cache hits included; no gfx950 counter isolates HBM: see profiles/r04_ic_evidence.txt')
    return roof, table


def cpu_baseline(B=32, seconds_budget=25.0):
    """The oracle (numpy + OpenBLAS, 'port') on this box's host cores: the SAME workload as the GPU line — coarse-phase
    train steps at the same batch, on synth_batch's images and depths and keep_masks' dropout masks — bounded to about
    seconds_budget of CPU work (a step at B=32 takes ~5 s: a handful of steps)."""
    from oracle import msdn as O
    img, dep = (t.numpy() for t in synth_batch(B, 0, 'cpu'))
    tr = O.Trainer(O.init_params(3000), B)
    keeps = [k.numpy().astype(bool) for k in keep_masks(B, 8, 0, 'cpu')]
    t0 = time.perf_counter()
    tr.step(img, dep, keeps[0])                   # warm-up (BLAS threads, page faults)
    first = time.perf_counter() - t0
    n = max(1, min(7, int(seconds_budget / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for i in range(n):
        tr.step(img, dep, keeps[1 + i])
    dt = time.perf_counter() - t0
    threads = os.cpu_count()
    try:                                            # the BLAS pool is what actually does the work (OpenBLAS caps it)
        from threadpoolctl import threadpool_info
        pools = [p['num_threads'] for p in threadpool_info() if p.get('user_api') == 'blas']
        if pools:
            threads = max(pools)
    except Exception:                               # noqa: BLE001 - reporting only
        pass
    return {'value': round(B * n / dt, 3), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{n} coarse-phase train steps at batch {B}, same synthetic inputs and step definition as the GPU line '
                      f'(480x640 stored -> 228x304 net); numpy oracle with {threads} OpenBLAS threads (the pool\'s '
                      f'compiled-in cap) on a {os.cpu_count()}-thread host; CPU restatement, not TF-1.3 Eigen'}


def hbm_bytes_bf16_storage(B, phase='coarse'):
    """Algorithmic HBM bytes of one train step under precision 'bf16s' (BASELINE config 5: 2-byte activations and weight
    copies), by SURVEY 8d's accounting rule: every tensor is counted once per pass that must read or write it — an activation
    once when produced and once per consumer pass, weights once per pass that uses them, the frozen ApplyAdam as read g + read
    m + write m (conv groups) or read m + write m (dense kernels, gradient never stored).  phase 'coarse': both forwards, the
    backward of coarse/*; 'fine': both forwards, the backward of fine/* (src/models.py:326-344)."""
    f4, b2 = 4, 2
    px = lambda h, w, c, e: h * w * c * e
    forward = (
        px(480, 640, 3, f4) + px(480, 640, 1, f4)                  # stored record, read by the resize
        + 3 * px(228, 304, 3, f4) + 3 * px(55, 74, 1, f4)          # x: written, read by conv2d_0 fwd and by the 4-channel bf16 copy; t: written, read by both losses
        + 2 * px(228, 304, 4, b2)                                  # fine/first on the bf16 pipe: x4 written + read (conv + pool fused: f1 never reaches HBM)
        # forward activations, written once and read once by the next layer (bf16): p0, p1, c2, c3, c4, cat; f2 fp32.  c0 / c1
        # never reach HBM (the pools are in the conv kernels' epilogues): one argmax byte per pool window instead
        + 2 * (px(27, 37, 96, b2) + px(13, 18, 256, b2) + 2 * px(13, 18, 384, b2) + px(6, 8, 256, b2)
               + px(55, 74, 64, b2)) + 2 * px(55, 74, 64, f4) + px(13, 18, 256, 1) + px(27, 37, 96, 1)
        + 3 * 4096 * f4 + 3 * 4070 * f4                            # dense side tensors of the forward (d0, drop, coarse)
    )
    conv_w = 34944 + 614656 + 885120 + 1327488 + 884992            # coarse/conv parameters
    fine_w = 15372 + 102464 + 1601
    d0, d1 = 12288 * 4096 + 4096, 4096 * 4070 + 4070
    forward_w = (conv_w - 34944) * b2 + 34944 * f4 + fine_w * f4 + (d0 + d1) * b2      # bf16 copies (conv2d_0, fine/*: fp32)
    if phase == 'fine':
        backward = (
            px(228, 304, 3, f4)                                    # x again: fine/first's filter gradient
            + px(55, 74, 64, 1)                                    # fine/first's pool: argmax bytes written by the forward ...
            + px(55, 74, 64, 1) + px(55, 74, 64, b2)               # ... and read, with the pooled activations, by its fused pool gradient
            + px(55, 74, 64, f4)                                   # f2 read again by fine/third's backward
            + 3 * px(55, 74, 64, b2)                               # df2 written, read by fine/second's bwd-filter and bwd-data
            + px(55, 74, 64, b2)                                   # cat read again by fine/second's bwd-filter
            + 2 * px(55, 74, 64, b2)                               # dcat written, read by fine/first's filter gradient
            + 2 * px(55, 74, 1, f4)                                # the loss gradient
        )
        backward_w = 102464 * b2 + fine_w * f4 * (1 + 3)           # fine/second's copy in bwd-data; dW written; ApplyAdam: g, m read, m written
        return B * (forward + backward) + forward_w + backward_w
    backward = (
        px(228, 304, 3, f4)                                        # x again: conv2d_0's filter gradient
        # each stored activation read again (bwd-filter A operand / ReLU mask), each activation gradient written once and
        # read by bwd-filter and bwd-data of the layer below (conv2d_1's pool gradient reads the argmax bytes and p1)
        + (px(27, 37, 96, b2) + px(13, 18, 256, 1) + 2 * px(13, 18, 256, b2) + 2 * px(13, 18, 384, b2) + px(6, 8, 256, b2))
        + 3 * (px(6, 8, 256, b2) + 2 * px(13, 18, 384, b2) + px(13, 18, 256, b2) + px(27, 37, 256, b2) + px(27, 37, 96, b2))
        + px(27, 37, 96, 1) + px(27, 37, 96, b2) + 2 * px(55, 74, 96, b2)   # argmax bytes and p0 read by the pool gradient; dc0 (bf16) written + read
        + 3 * 4096 * f4 + 3 * 4070 * f4                            # dense side tensors of the backward (dz1, dz0, ...)
    )
    backward_w = (
        (conv_w - 34944) * b2 + 34944 * f4                         # conv kernels again in bwd-data
        + (d0 + d1) * b2                                           # the dense layers' bf16 copies in bwd-data
        + conv_w * f4 * (1 + 3)                                    # conv dW written; ApplyAdam: g, m read, m written
        + (d0 + d1) * f4 * 2                                       # dense dW -> m slot in place: m read, m written
    )
    return B * (forward + backward) + forward_w + backward_w


def hbm_roofline_bf16_storage(B, phase, seconds_per_step):
    nbytes = hbm_bytes_bf16_storage(B, phase)
    return {'bound': 'hbm', 'scope': f'{phase}-phase step, 2-byte activations and weight copies (hbm_bytes_bf16_storage)',
            'achieved': round(nbytes / seconds_per_step / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
            'frac': round(nbytes / seconds_per_step / 8e12, 4), 'traffic': None,
            'bytes_per_step': nbytes, 'bytes_per_image': round(nbytes / B)}


def bench_dcnf(args, lib, device, rank, world):
    """BASELINE config 4: DCNF-lite unary conv stack (src/models.py:50-89), batch 16 -> 768 patches of 100x100x3.
    A step = unary forward (resize, patches, 5 conv / 3 pool / 3 dense) + unary backward from a synthetic dz: 2.672 GFLOP
    per patch forward and twice that backward (no input gradient for the first conv: 2.672 * 2 -
    0.377)."""
    from ann3depth_amd import models
    B = 16 if args.batch == 32 else args.batch
    rng = np.random.default_rng(1000 + rank)
    img = torch.from_numpy((rng.integers(0, 256, (B, 480, 640, 3)) / 255).astype(np.float32)).to(device)
    net = models.DCNFUnary(B, device=device)
    dz = torch.randn((net.P, 1), device=device)

    def step():
        net.forward(img)
        net.backward(dz)
    steps, warm = min(args.steps, 20), min(args.warmup, 3)
    dt, recs, warm_recs = warm_then_time(lambda i: step(), lambda: None, steps, warm, lib, 1, True)
    roof, _ = roofline_from(recs, with_traffic=False)                 # (the committed PMC passes are MSDN's)
    _, table = roofline_from(warm_recs, with_traffic=False)           # every kernel: from the warm-up steps
    gflop = net.P * (2.672 + 2 * 2.672 - 0.377)
    if rank == 0:
        print(json.dumps({
            'metric': 'images/sec (DCNF-lite unary stack, forward + backward)', 'value': round(world * B * steps / dt, 1),
            'unit': 'images/sec', 'n_gpus': world, 'steps': steps, 'warmup': warm, 'ms_per_step': round(1e3 * dt / steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'DCNF-lite unary conv stack, batch {B} -> {net.P} patches 100x100x3, forward + backward '
                                   f'(BASELINE config 4)', 'per_gpu_batch': B},
            'step_tflops': round(gflop / (dt / steps) / 1e3, 1), 'roofline': roof, 'igemm_kernels': table}), flush=True)
    return 0


def comm_report(net, img, dep, masks, args, lib, world, dt):
    """N > 1 only: how long each gradient bucket's RCCL all-reduce takes on its own (the 283 MB of src/ann3depth.py:77-92's
    parameter-server traffic, in the pieces MSDNReplica.step sends them), and how much communication the step does NOT
    hide: ms/step with the reducer minus ms/step of the same replica without one (same kernels, no collectives)."""
    import torch.distributed as dist
    from ann3depth_amd import models
    from ann3depth_amd import dp
    gd, gc = net.groups['CoarseDense'], net.groups['CoarseConv']
    cut = gc.offsets['coarse/conv/conv2d_2/kernel'][0]
    if gd.frozen() and net.dense_exchange == 'reduce_scatter':       # models.MSDNReplica._dense_buckets
        (early,), late = net._dense_buckets()
        buckets = {'dense_1': ('reduce_scatter', gd.grad[early[0]:early[1]]),
                   'dense_0_piece': ('reduce_scatter', gd.grad[late[0][0]:late[0][1]])}
    else:
        d1 = gd.offsets['coarse/dense/dense_1/kernel'][0]
        buckets = {'dense_1': ('all_reduce', gd.grad[d1:]), 'dense_0_piece': ('all_reduce', gd.grad[:d1 // 3])}
    buckets.update({'conv_tail': ('all_reduce', gc.grad[cut:]), 'conv_head': ('all_reduce', gc.grad[:cut])})
    times = {}
    for name, (kind, buf) in buckets.items():
        scratch = buf.clone()
        n = scratch.numel() // world
        own = scratch[dist.get_rank() * n:(dist.get_rank() + 1) * n]
        run = (lambda: dist.all_reduce(scratch)) if kind == 'all_reduce' else (lambda: dist.reduce_scatter_tensor(own, scratch))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / 5], device=img.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times[name] = {'ms': round(1e3 * float(t.item()), 3), 'mbytes': round(scratch.numel() * 4 / 1e6, 1),
                       'collective': kind}
        del scratch, own
    # the same rank without its collectives: same kernels (dense gradients materialised, ApplyAdam of its own slices)
    solo = models.MSDNReplica(net.B, device=img.device, seed=3000, reducer=dp.DetachedReducer(world, dist.get_rank()),
                              precision=args.precision)
    n = max(3, args.steps // 2)
    dts, _ = run_phase(solo, img, dep, masks, n, 2, 0, lib, world, timed_kernels=False)
    return {'rccl_ranks': world, 'dense_exchange': net.dense_exchange, 'allreduce_ms': times,
            'ms_per_step_without_comm': round(1e3 * dts / n, 3),
            'exposed_comm_ms': round(1e3 * (dt / args.steps - dts / n), 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (BASELINE config 2/3: 32)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x3', 'bf16', 'bf16s'],
                    help='conv arithmetic; the headline (and parity) mode is fp32')
    ap.add_argument('--also', default='bf16x3', help='comma list of extra precisions measured after the headline run')
    ap.add_argument('--no-fine', action='store_true', help='skip the additional fine-phase measurement')
    ap.add_argument('--phase', default='coarse', choices=['coarse', 'fine'],
                    help='which phase the main timed (and profiled) run executes; the driver-facing line is coarse. '
                         'fine = global_step 2 000 000 // B (src/models.py:302-305): a profiling aid, prints a fine-phase line')
    ap.add_argument('--no-dp-rank', action='store_true', help='N = 1: skip timing the step a data-parallel rank would run')
    ap.add_argument('--dp-world', type=int, default=8, help='world size assumed by the N = 1 dp_rank measurement')
    ap.add_argument('--no-dp-rank-standin', action='store_true',
                    help='N = 1: skip the dp_rank measurement WITH a stand-in for the collectives (dp.StandinReducer)')
    ap.add_argument('--standin-gbps', type=float, default=200.0,
                    help='rate the stand-in collective is paced to, reads + writes (xGMI: 7 links x ~153 GB/s per GPU at best, one '
                         'ring ~150 GB/s)')
    ap.add_argument('--standin-workgroups', type=int, default=24, help='CUs the stand-in collective occupies (RCCL: 16-32)')
    ap.add_argument('--standin-one-stream', action='store_true',
                    help='every stand-in on one stream (one communicator: the conv buckets queue behind the dense reduce-scatter)')
    ap.add_argument('--model', default='msdn', choices=['msdn', 'dcnf'],
                    help="msdn = the headline (BASELINE config 2/3/5); dcnf = BASELINE config 4, the DCNF-lite unary stack "
                         "at batch 16 (768 patches): a separate line, 'step' = unary forward + backward")
    args = ap.parse_args()

    from ann3depth_amd import _lib, dp, models
    lib = _lib.load()
    rank, local_rank, world = dp.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run',
                  file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        backend = dist.get_backend()
        if dist.get_world_size() != args.gpus or (backend != 'nccl' and not os.environ.get('A3D_DIST_BACKEND')):
            # the line below would claim N RCCL ranks: refuse instead
            if rank == 0:
                print(f'bench.py: --gpus {args.gpus} needs {args.gpus} RCCL ranks, have {dist.get_world_size()} over {backend}',
                      file=sys.stderr)
            sys.exit(3)
    device = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    B = args.batch
    if args.model == 'dcnf':
        return bench_dcnf(args, lib, device, rank, world)
    reducer = dp.GradReducer() if world > 1 else None
    net = models.MSDNReplica(B, device=device, seed=3000, reducer=reducer, precision=args.precision,
                             keep_dense_grads=False)      # as models.msdn builds it for `make train`
    img, dep = synth_batch(B, rank, device)
    masks = keep_masks(B, 8, rank, device)

    # headline: coarse-phase train step (what `make train` executes from global_step 0; the heaviest real phase)
    gs0 = 0 if args.phase == 'coarse' else models.SAMPLES_COARSE // B
    dt, (recs, warm_recs) = run_phase(net, img, dep, masks, args.steps, args.warmup, gs0, lib, world, timed_kernels=True)
    value = world * B * args.steps / dt
    roof, _ = roofline_from(recs)                                     # the dominant kernel, event-timed in the timed region
    _, table = roofline_from(warm_recs, with_traffic=False)           # every kernel: event-timed during the warm-up steps
    if args.precision == 'bf16s':
        # the bf16 matrix cores leave this step HBM-bound: price the whole step of the phase that ran against the HBM roofline
        roof = hbm_roofline_bf16_storage(B, args.phase, dt / args.steps)
    extra = {}
    if not args.no_fine and args.phase == 'coarse':
        dtf, _ = run_phase(net, img, dep, masks, args.steps, min(args.warmup, 2), models.SAMPLES_COARSE // B, lib,
                           world, timed_kernels=False)
        extra['fine_phase'] = {'value': round(world * B * args.steps / dtf, 1), 'ms_per_step': round(1e3 * dtf / args.steps, 3)}
        if args.precision == 'fp32':
            extra['fine_phase']['roofline'] = step_roofline('fine', B, dtf / args.steps)
        elif args.precision == 'bf16s':
            extra['fine_phase']['roofline'] = hbm_roofline_bf16_storage(B, 'fine', dtf / args.steps)
    for prec in [p for p in args.also.split(',') if p and p != args.precision]:
        alt = models.MSDNReplica(B, device=device, seed=3000, reducer=reducer, precision=prec, keep_dense_grads=False)
        dta, _ = run_phase(alt, img, dep, masks, args.steps, min(args.warmup, 3), 0, lib, world, timed_kernels=False)
        extra.setdefault('other_precisions', {})[prec] = {
            'value': round(world * B * args.steps / dta, 1), 'ms_per_step': round(1e3 * dta / args.steps, 3),
            'note': ('the reference arithmetic (fp32 MFMA), same batch' if prec == 'fp32' else
                     'conv contractions on the bf16 matrix cores; NOT the headline: parity is stated for fp32')}
        del alt
    comm = {}
    if world > 1:
        comm = comm_report(net, img, dep, masks, args, lib, world, dt)
    elif not args.no_dp_rank:
        # The step a data-parallel rank runs is NOT the fused single-GPU step timed above (it must materialise the 268 MB
        # of dense gradients for the reduce-scatter and applies Adam to its own 1/world of them).  Timed here on this one
        # GPU with the collectives detached: the ceiling of 1 -> 8 scaling before any communication cost is
        # 8 * ms_per_step / ms_per_step_dp_rank.
        from ann3depth_amd import dp as _dp
        dnet = models.MSDNReplica(B, device=device, seed=3000, reducer=_dp.DetachedReducer(args.dp_world, 0),
                                  precision=args.precision)
        dtd, _ = run_phase(dnet, img, dep, masks, args.steps, min(args.warmup, 3), 0, lib, world, timed_kernels=False)
        extra['dp_rank'] = {
            'ms_per_step_dp_rank': round(1e3 * dtd / args.steps, 3), 'world_assumed': args.dp_world,
            'what': 'coarse-phase step of ONE rank of a data-parallel job, collectives detached (dp.DetachedReducer): '
                    'dense dW written (268 MB), ApplyAdam on the rank\'s own slices; no RCCL time included',
            'scaling_ceiling': round(args.dp_world * dt / dtd, 2)}
        del dnet
        if not args.no_dp_rank_standin:
            # ... and with the COST of the collectives on this GPU: a stand-in kernel of each bucket's size on a second stream,
            # paced to an xGMI-like rate, ordered and waited for exactly like the RCCL work handles (dp.StandinReducer)
            sred = _dp.StandinReducer(args.dp_world, 0, args.standin_gbps, args.standin_workgroups,
                                      urgent_stream=not args.standin_one_stream)
            snet = models.MSDNReplica(B, device=device, seed=3000, reducer=sred, precision=args.precision)
            dts, _ = run_phase(snet, img, dep, masks, args.steps, min(args.warmup, 3), 0, lib, world, timed_kernels=False)
            per_step = sred.launched_bytes / (args.steps + min(args.warmup, 3))
            extra['dp_rank'].update({
                'ms_per_step_dp_rank_with_standin': round(1e3 * dts / args.steps, 3),
                'standin': {'gbytes_per_s': args.standin_gbps, 'workgroups': args.standin_workgroups,
                            'communicators': 1 if args.standin_one_stream else 2,
                            'mbytes_per_step': round(per_step / 1e6, 1),
                            'ms_of_standin_per_step': round(per_step / args.standin_gbps / 1e6, 3),
                            'what': 'one kernel per bucket on a second stream wherever the rank starts a collective: reads the '
                                    'bucket and, keeping pace with its reads, writes 1/world of it (reduce-scatter) or all of it '
                                    '(all-reduce), paced to the rate; no communication happens'},
                'scaling_ceiling_with_standin': round(args.dp_world * dt / dts, 2)})
            del snet
            # the same with the FALLBACK exchange: what a rank does when the start-up self-check of the in-place collectives
            # answers False over RCCL (dp.GradReducer.inplace_ok) — the dense bucket as all-reduces (each rank reads and
            # writes ALL of it, keeps every dense gradient and applies Adam to all of it), not one reduce-scatter
            fred = _dp.StandinReducer(args.dp_world, 0, args.standin_gbps, args.standin_workgroups,
                                      urgent_stream=not args.standin_one_stream, inplace=False)
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # (the replica announces the fallback on stdout: this line stays the only one there)
                fnet = models.MSDNReplica(B, device=device, seed=3000, reducer=fred, precision=args.precision)
            dtf2, _ = run_phase(fnet, img, dep, masks, args.steps, min(args.warmup, 3), 0, lib, world, timed_kernels=False)
            per_step_f = fred.launched_bytes / (args.steps + min(args.warmup, 3))
            extra['dp_rank']['fallback_exchange'] = {
                'dense_exchange': fnet.dense_exchange,
                'ms_per_step_dp_rank_with_standin': round(1e3 * dtf2 / args.steps, 3),
                'mbytes_per_step': round(per_step_f / 1e6, 1),
                'ms_of_standin_per_step': round(per_step_f / args.standin_gbps / 1e6, 3),
                'scaling_ceiling_with_standin': round(args.dp_world * dt / dtf2, 2),
                'what': 'inplace_ok() == False: the dense gradients go out as all-reduces (read + write all 268 MB) instead of one '
                        'in-place reduce-scatter (read all, write 1/world); same pacing, same stand-in kernel'}
            del fnet
    if rank == 0:
        line = {
            'metric': METRIC, 'value': round(value, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': {'fp32': 'f32', 'bf16s': 'bf16'}.get(args.precision, args.precision),
            'data': 'synthetic',
            'config': {'workload': f'MSDN coarse+fine, batch {B} per GPU, 640x480 stored -> 228x304 net -> 55x74 depth, ' + (
                                   'coarse-phase train step (global_step 0): both forwards + both losses, backward of '
                                   'coarse/*, 2x ApplyAdam(beta2=1)' if args.phase == 'coarse' else
                                   f'FINE-phase train step (global_step {gs0}): both forwards + both losses, backward of '
                                   'fine/*, 2x ApplyAdam(beta2=1) -- a profiling line, not the headline'),
                       'per_gpu_batch': B, 'global_batch': B * world,
                       'parallelism': f'dp{world}' + (' (RCCL all-reduce of 283 MB grads/step)' if world > 1 else '')},
            'roofline': roof,
            'step_roofline': step_roofline(args.phase, B, dt / args.steps) if args.precision == 'fp32' else None,
            'igemm_kernels': table,
            'igemm_kernels_note': 'HIP-event durations of every GEMM launch, taken during the warm-up steps; inside the timed '
                                  'region only the roofline kernel is bracketed, in every fourth step (an event pair per '
                                  'launch costs the stream ~15 us)',
        }
        line.update(extra)
        line.update(comm)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(B)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
