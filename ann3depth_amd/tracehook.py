"""tfhelper.TraceHook (src/tfhelper.py:192-249) as a real profiler capture.

The reference asks the TensorFlow runtime for a FULL_TRACE of the first step and of every `every_step`-th global step
and writes it next to the checkpoints.  Here the capture is rocprofv3's: with `--profiler rocprofv3` the driver starts
ITSELF once more as a child of `rocprofv3 --kernel-trace --marker-trace --selected-regions`, before anything in the
parent has touched the GPU (a process that has initialised HIP must never be replaced or re-executed on this platform),
waits for it and hands on its exit code.  The child brackets exactly the traced steps with roctxProfilerResume /
roctxProfilerPause and a `global_step N` range, so the kernel trace under <checkpoint dir>/rocprof holds those steps and
nothing else.
"""
import ctypes
import os
import shutil
import signal
import subprocess
import sys

ENV_FLAG = 'A3D_UNDER_ROCPROF'
ROCTX_LIB = 'librocprofiler-sdk-roctx.so'
FORWARDED = (signal.SIGUSR1, signal.SIGUSR2, signal.SIGINT, signal.SIGTERM)


def under_profiler():
    return os.environ.get(ENV_FLAG) == '1'


def profiler_command(argv, outdir, rocprofv3=None):
    """The child's command line: the interpreter comes straight after `--` (no env / shell hop in between)."""
    exe = rocprofv3 or shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    return [exe, '--kernel-trace', '--marker-trace', '--selected-regions', '--output-format', 'csv', '-d', outdir,
            '--', sys.executable, '-m', 'ann3depth_amd.ann3depth', *argv]


def respawn(argv, outdir, popen=subprocess.Popen):
    """Runs this driver under rocprofv3 and returns its exit code.  Signals meant for the training loop
    (StopAtSignalHook's set) are passed on to the child; the alarm is the child's own."""
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ)
    env[ENV_FLAG] = '1'
    proc = popen(profiler_command(list(argv), outdir), env=env)
    old = {s: signal.signal(s, lambda signum, frame: proc.send_signal(signum)) for s in FORWARDED}
    try:
        rc = proc.wait()
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    return rc if rc >= 0 else -rc


class Roctx:
    """roctx ranges and the profiler's pause / resume switch (rocprofiler-sdk's ROCTx library)."""

    def __init__(self, path=None):
        try:
            lib = ctypes.CDLL(path or ROCTX_LIB)
        except OSError:
            lib = ctypes.CDLL(os.path.join('/opt/rocm/lib', ROCTX_LIB))      # fails loudly if the library is absent
        lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        lib.roctxRangePushA.restype = ctypes.c_int
        lib.roctxRangePop.restype = ctypes.c_int
        lib.roctxProfilerPause.argtypes = [ctypes.c_uint64]
        lib.roctxProfilerResume.argtypes = [ctypes.c_uint64]
        self.lib = lib

    def begin(self, step):
        self.lib.roctxProfilerResume(0)
        self.lib.roctxRangePushA(f'global_step {step}'.encode())

    def end(self):
        self.lib.roctxRangePop()
        self.lib.roctxProfilerPause(0)
