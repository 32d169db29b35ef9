"""Driver — the surface of the reference's ``src/ann3depth.py``: same flags (src/ann3depth.py:221-254), same
checkpoint directory convention, same stop conditions, same exit code, same one-line hot loop
``while not should_stop: run(train_op)`` (src/ann3depth.py:126-127).

TensorFlow's session machinery is not reproduced: `Session` below is the minimal stand-in for
`MonitoredTrainingSession` + the reference's hooks (StopAtStepHook, StopAtSignalHook, summary saver,
checkpoint saver/restorer, TraceHook).  The parameter-server path (`--job-name ps`, src/ann3depth.py:62-67) is
replaced by synchronous RCCL data parallelism: launch one process per GPU with
``python -m torch.distributed.run --nproc-per-node N -m ann3depth_amd.ann3depth ...``; the old cluster flags are
accepted and ignored.
"""
import argparse
import json
import logging
import logging.config
import os
import signal
import sys
import time

import torch

from . import _lib, data, dp, models, summary, tfckpt, tracehook


def main(argv=None):
    ini = 'logging.ini' if os.path.exists('logging.ini') else os.path.join(os.path.dirname(__file__), 'logging.ini')
    logging.config.fileConfig(ini, disable_existing_loggers=False)
    logger = logging.getLogger('ann3depth')

    args = parse_args(argv)
    logger.debug(args)
    logger.info(f'This is a {args.job_name} with index {args.task_index}')
    if args.cluster_spec:
        logger.info(f'Ignoring cluster spec {args.cluster_spec}: replicas are RCCL ranks, not TF servers.')
    else:
        logger.info('Using default local cluster spec.')

    if args.job_name == 'ps':
        logger.info('Parameter servers do not exist in this build (gradients are all-reduced over xGMI).')
        return 0
    if args.job_name not in ('worker', 'local'):
        logger.warning(f'No suitable job description found! {args.job_name}')
        return 0

    run_id = args.model + ('' if not args.id else f'_{args.id}')
    ckptdir = str(os.path.join(args.ckptdir, run_id))                    # src/ann3depth.py:73-75
    if args.profiler == 'rocprofv3' and args.trace_every and not tracehook.under_profiler():
        # TraceHook as a profiler capture: nothing above has touched the GPU, so this process may still start the
        # profiled copy of itself as a child; from here on it only waits (ann3depth_amd/tracehook.py)
        out = os.path.join(ckptdir, 'rocprof')
        logger.info(f'Starting the training process under rocprofv3; traces of the traced steps go to {out}.')
        return tracehook.respawn(sys.argv[1:] if argv is None else argv, out)

    rank, local_rank, world = dp.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError('ann3depth_amd needs an MI355X: the training path has no CPU fallback')
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    chief = rank == 0
    logger.info(f'Task: {rank} of {world} -- Chief? {chief}')

    logger.info(f'Checkpoint dir is {ckptdir}.')

    logger.info(f'Loading model {args.model}.')
    model_op = setup_model(args, rank, world)

    size_train = sum(g.count for g in model_op.replica.groups.values()) * 4 / 1024 / 1024
    logger.debug(f'Trainable variables have about {size_train:.1f} MB')

    logger.info('Setting up hooks.')
    stop_at_signal = StopAtSignal()
    if args.job_name != 'local':
        logger.info(f'Starting alarm: {args.timeout} s timeout.')
        signal.alarm(args.timeout)                                       # src/ann3depth.py:108-110

    logger.info('Starting session.')
    with Session(model_op, ckptdir if chief else None, last_step=args.steps, stop_at_signal=stop_at_signal,
                 save_checkpoint_secs=args.ckptfreq, save_summaries_steps=args.sumfreq,
                 trace_every=args.trace_every, logger=logger, world=world,
                 tf_checkpoints=args.tf_checkpoints) as session:
        while not session.should_stop():
            session.run(model_op)
    logger.info('Session stopped.')
    return stop_at_signal.signal_received                                # src/ann3depth.py:129


def setup_model(args, rank=0, world=1):
    """src/ann3depth.py:134-145."""
    model = getattr(models, args.model)
    if world > 1:
        model.reducer = dp.GradReducer()
    if args.beta2 is not None:
        model.beta2 = args.beta2
    model.precision = args.precision
    inputs, targets = data.inputs(args.datadir, args.dataset, args.batchsize, rank=rank, world=world,
                                  seed=args.seed + rank)
    return model(inputs, targets)


class StopAtSignal:
    """tfhelper.StopAtSignalHook (src/tfhelper.py:160-189): remember the signal, stop after the current step."""

    def __init__(self, signals=None):
        self.signal_received = 0
        if signals is None:
            signals = [signal.SIGUSR1, signal.SIGUSR2, signal.SIGALRM, signal.SIGINT, signal.SIGTERM]
        for s in signals:
            signal.signal(s, self._handler)

    def _handler(self, signum, frame):
        self.signal_received = signum


class Session:
    """Stand-in for tf.train.MonitoredTrainingSession as the reference configures it (src/ann3depth.py:113-125):
    restore the newest checkpoint of `checkpoint_dir`, save every `save_checkpoint_secs` and on exit, write loss
    scalars every `save_summaries_steps` steps, log steps/sec, stop at `last_step`, on a signal, or when the input
    pipeline runs dry."""

    def __init__(self, train_op, checkpoint_dir, last_step, stop_at_signal, save_checkpoint_secs, save_summaries_steps,
                 trace_every, logger, world=1, tf_checkpoints=False, keep_checkpoints=5):
        self.keep_checkpoints = keep_checkpoints          # tf.train.Saver(max_to_keep=5), the Saver default
        self.op, self.dir, self.last_step, self.sig = train_op, checkpoint_dir, last_step, stop_at_signal
        self.ckpt_secs, self.sum_steps, self.trace_every = save_checkpoint_secs, save_summaries_steps, trace_every
        self.log, self.world = logger, world
        self.tf_checkpoints = tf_checkpoints
        self.stop = False
        self.trace_next = True               # TraceHook: the first step after every (re)start is traced
        # under `--profiler rocprofv3` (this is then the profiled child): the traced steps are the profiler's regions
        self.roctx = tracehook.Roctx() if tracehook.under_profiler() else None
        self.t_last_ckpt = time.time()
        self.t_last_sum, self.step_last_sum = time.time(), None
        self.summaries = None
        self.events = None

    def __enter__(self):
        rep = self.op.replica
        latest = latest_checkpoint(self.dir) if self.dir else None
        if latest:
            self.log.info(f'Restoring {latest}')
            if tfckpt.is_bundle(latest):                      # a checkpoint written by TensorFlow (or --tf-checkpoints)
                rep.load_tf_variables(tfckpt.read_bundle(latest))
            else:
                rep.load_state_dict(torch.load(latest, map_location=rep.device))
        if self.world > 1:                                    # non-chief replicas take the chief's state
            import torch.distributed as dist
            rep.broadcast_state(dist, 0)
        if self.dir:
            os.makedirs(self.dir, exist_ok=True)
            self.summaries = open(os.path.join(self.dir, 'summaries.jsonl'), 'a')
            self.events = summary.EventFileWriter(self.dir)
        self.step_last_sum = rep.global_step
        if self.world > 1:
            # before the first step: a rank whose shard cannot fill even one batch (or whose reader already failed) must
            # not leave the others waiting in step 1's collectives
            self._decide(self.op)
        return self

    def _decide(self, train_op, want_save=False):
        """The replicas decide TOGETHER (dp.GradReducer.agree_all, one host collective): a signal reaches the ranks at
        different steps, an input shard runs dry on one rank first, the chief's checkpoint timer fires on the chief only —
        and the next step's collectives (and a sharded optimizer state's gather) need everyone.  Agreed values: the
        signal number (exit code of every rank, src/ann3depth.py:129) or 1 when a rank's input has no batch left for
        the next step; whether a rank's reader FAILED (CRC mismatch, malformed record: raised on every rank, never
        mistaken for a clean end of input); whether the chief wants a checkpoint now."""
        rep = train_op.replica
        end = getattr(train_op, 'end', None)
        due = end is not None and end[0] <= train_op.k
        dry = due and isinstance(end[1], data.OutOfRangeError)
        failed = due and not dry
        stop, err, save = rep.reducer.agree_all([self.sig.signal_received or (1 if dry else 0), int(failed),
                                                 int(want_save)])
        if err:
            if failed:
                raise end[1]
            raise RuntimeError('the input pipeline of another replica failed; stopping with it')
        if stop:
            self.stop = True
            if stop > 1 and not self.sig.signal_received:
                self.sig.signal_received = stop
        return bool(save)

    def should_stop(self):
        return self.stop or self.op.replica.global_step >= self.last_step

    def run(self, train_op):
        rep = train_op.replica
        tracing = self.trace_next and self.dir is not None
        marking = self.trace_next and self.roctx is not None
        if tracing:
            _lib.load().a3d_timing_enable(1)
        if marking:
            self.roctx.begin(rep.global_step + 1)
        try:
            out = train_op.run()
        except data.OutOfRangeError as e:
            self.log.info(f'Input pipeline exhausted: {e}')
            self.stop = True
            return None
        finally:
            if marking:
                torch.cuda.synchronize()               # the step's kernels belong inside the region
                self.roctx.end()
            if tracing:
                self._write_trace(rep.global_step)
        step = rep.global_step
        # tfhelper.TraceHook (src/tfhelper.py:192-249): trace the first step and every `trace_every`-th global step
        self.trace_next = bool(self.trace_every) and (step + 1) % self.trace_every == 0
        save_due = bool(self.dir and self.ckpt_secs and time.time() - self.t_last_ckpt >= self.ckpt_secs)
        if self.world > 1:
            save_due = self._decide(train_op, save_due)
        elif self.sig.signal_received:                                   # StopAtSignalHook.after_run
            self.stop = True
        if self.sum_steps and step % self.sum_steps == 0:
            now = time.time()
            rate = (step - self.step_last_sum) / max(now - self.t_last_sum, 1e-9)
            rec = {'global_step': step, **rep.summary_scalars(out),       # the reference's summary tags
                   'global_step/sec': rate, 'images/sec': rate * rep.B * self.world}
            self.log.info('global_step/sec: %.4g  %s', rate, json.dumps(rec))
            if self.summaries:
                self.summaries.write(json.dumps(rec) + '\n')
                self.summaries.flush()
                self.events.add_scalars(step, {k: v for k, v in rec.items() if k != 'global_step'})
                for tag, t, max_outputs in rep.summary_images():
                    self.events.add_images(step, tag, t[:max_outputs].cpu().numpy(), max_outputs)
                self.events.flush()
            self.t_last_sum, self.step_last_sum = now, step
        if save_due:
            self.save()
        return out

    def _write_trace(self, step):
        """Per-launch timings of the implicit-GEMM kernels of one step (hipEvent pairs recorded by the library),
        the stand-in for the reference's RunMetadata FULL_TRACE; whole-process traces come from rocprofv3."""
        lib = _lib.load()
        lib.a3d_timing_enable(0)
        cap = 4096
        arr = (_lib.TimingRecord * cap)()
        n = lib.a3d_timing_collect(arr, cap)
        recs = [{'mode': ('fwd', 'bwd_data', 'bwd_filter')[r.mode], 'tile': f'{r.bm}x{r.bn}', 'waves': r.nwaves,
                 'precision': ('fp32', 'bf16x3', 'bf16')[r.prec], 'kernel': ('igemm', 'igemm_glds', 'conv3_fwd', 'igemm_ring', 'fewch_bwdf', 'igemm2')[r.lds_dma], 'splitk': r.splitk, 'm': r.m, 'n': r.n, 'k': r.k,
                 'ms': round(r.ms, 4), 'tflops': round(r.flops / max(r.ms, 1e-6) / 1e9, 1)} for r in arr[:n]]
        with open(os.path.join(self.dir, f'trace-{step}.json'), 'w') as f:
            json.dump({'global_step': step, 'launches': recs}, f)

    def save(self):
        """Every replica calls this together (a sharded optimizer state is gathered first); the chief writes."""
        rep = self.op.replica
        rep.gather_state()
        if not self.dir:
            return
        torch.cuda.synchronize()
        prefix = f'model.ckpt-{rep.global_step}'
        path = os.path.join(self.dir, prefix + '.pt')
        tmp = path + '.tmp'
        torch.save({k: v.detach().cpu() for k, v in rep.state_dict().items()}, tmp)
        os.replace(tmp, path)
        if self.tf_checkpoints:                               # the same state as a TensorFlow V2 checkpoint
            tfckpt.write_bundle(os.path.join(self.dir, prefix), rep.tf_variables())
        # CheckpointState as tf.train.Saver writes it: with --tf-checkpoints the entries are bundle prefixes, so
        # tf.train.latest_checkpoint finds them (latest_checkpoint() below prefers '<prefix>.pt' when it exists)
        kept = [n for n in self._kept_checkpoints() if n != prefix] + [prefix]
        for old in kept[:-self.keep_checkpoints] if self.keep_checkpoints else []:
            for f in os.listdir(self.dir):
                if f == old + '.pt' or f == old + '.index' or f.startswith(old + '.data-'):
                    os.remove(os.path.join(self.dir, f))
        kept = kept[-self.keep_checkpoints:] if self.keep_checkpoints else kept
        entry = (lambda n: n) if self.tf_checkpoints else (lambda n: n + '.pt')
        tmp = os.path.join(self.dir, 'checkpoint.tmp')
        with open(tmp, 'w') as f:
            f.write(f'model_checkpoint_path: "{entry(prefix)}"\n')
            for n in kept:
                f.write(f'all_model_checkpoint_paths: "{entry(n)}"\n')
        os.replace(tmp, os.path.join(self.dir, 'checkpoint'))
        self.t_last_ckpt = time.time()
        self.log.info(f'Saved checkpoint {path}')

    def _kept_checkpoints(self):
        """Checkpoint prefixes of this directory, oldest first (by global step)."""
        steps = set()
        for f in os.listdir(self.dir):
            if f.startswith('model.ckpt-') and (f.endswith('.pt') or f.endswith('.index')):
                try:
                    steps.add(int(f[len('model.ckpt-'):].rsplit('.', 1)[0]))
                except ValueError:
                    pass
        return [f'model.ckpt-{s_}' for s_ in sorted(steps)]

    def __exit__(self, exc_type, exc, tb):
        self.op.pipeline.close()
        if exc_type is None:
            self.save()                                                  # session close saves a final checkpoint
        if self.summaries:
            self.summaries.close()
        if self.events:
            self.events.close()
        return False


def latest_checkpoint(ckptdir):
    index = os.path.join(ckptdir, 'checkpoint')
    if not os.path.exists(index):
        return None
    with open(index) as f:
        line = f.readline()
    name = line.split('"')[1] if '"' in line else ''
    path = name if os.path.isabs(name) else os.path.join(ckptdir, name)
    if name and not name.endswith('.pt') and os.path.exists(path + '.pt'):
        return path + '.pt'                                           # our own full state beside a bundle of the same step
    if name and (os.path.exists(path) or tfckpt.is_bundle(path)):    # ours (.pt file) or TensorFlow's (bundle prefix)
        return path
    return None


def parse_args(argv=None):
    """The reference's flags verbatim (src/ann3depth.py:221-254), plus --beta2 / --seed / --trace-every / --profiler."""
    parser = argparse.ArgumentParser()
    parser.add_argument('dataset', default='nyu', type=str, help='The dataset to use.')
    parser.add_argument('--model', '-m', default='', type=str, help='Enter a model name.')
    parser.add_argument('--steps', '-s', default=1000000, type=int, help='Total steps')
    parser.add_argument('--batchsize', '-b', default=32, type=int, help='Batchsize')
    parser.add_argument('--ckptdir', '-p', default='checkpoints', help='Checkpoint directory')
    parser.add_argument('--id', default='', type=str, help='Checkpoint path suffix.')
    parser.add_argument('--ckptfreq', '-f', default=900, type=int, help='Create a checkpoint every N seconds.')
    parser.add_argument('--sumfreq', '-r', default=100, type=int, help='Create a summary every N steps.')
    parser.add_argument('--datadir', '-d', default='data', type=str,
                        help='The data directory containing the datasets.')
    parser.add_argument('--timeout', '-k', default=4200, type=int, help='The time after which the process dies.')
    parser.add_argument('--cluster-spec', default='', type=str,
                        help='(ignored) The path to the cluster specification json.')
    parser.add_argument('--job-name', default='local', type=str, help='"worker" or "local"; "ps" exits at once.')
    parser.add_argument('--task-index', default=0, type=int, help='(ignored) rank comes from the launcher.')
    parser.add_argument('--beta2', default=None, type=float,
                        help='NON-REFERENCE: Adam beta2 (the reference hard-codes 1, which freezes the weights).')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x3', 'bf16', 'bf16s'],
                        help='NON-REFERENCE unless fp32: arithmetic of the conv contractions.')
    parser.add_argument('--seed', default=0, type=int, help='Shuffle-queue seed.')
    parser.add_argument('--tf-checkpoints', action='store_true',
                        help='Also write every checkpoint as a TensorFlow V2 bundle (model.ckpt-N.index/.data-*).')
    parser.add_argument('--trace-every', default=5000, type=int,
                        help='TraceHook period (src/ann3depth.py:105): the first step and every N-th global step.')
    parser.add_argument('--profiler', default='launches', choices=['launches', 'rocprofv3'],
                        help='What a traced step records: per-launch timings of the GEMM kernels (trace-N.json), or '
                             'also a rocprofv3 kernel + marker trace (the process is started under the profiler).')
    return parser.parse_args(argv)


if __name__ == '__main__':
    sys.exit(main())
