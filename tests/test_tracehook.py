"""TraceHook as a profiler capture (ann3depth_amd/tracehook.py; reference: src/tfhelper.py:192-249): the driver starts its
profiled copy as a CHILD before anything touches the GPU, and brackets the traced steps for rocprofv3 --selected-regions."""
import os
import signal
import sys

import pytest
import torch

from ann3depth_amd import ann3depth, tracehook


def test_profiler_command_puts_the_interpreter_right_after_the_separator(tmp_path):
    cmd = tracehook.profiler_command(['--model', 'msdn', 'nyu'], str(tmp_path), rocprofv3='/opt/rocm/bin/rocprofv3')
    sep = cmd.index('--')
    assert cmd[0] == '/opt/rocm/bin/rocprofv3' and cmd[sep + 1] == sys.executable        # no env / shell hop
    assert cmd[sep + 2:] == ['-m', 'ann3depth_amd.ann3depth', '--model', 'msdn', 'nyu']
    head = cmd[:sep]
    assert '--kernel-trace' in head and '--marker-trace' in head and '--selected-regions' in head
    assert '--pmc' not in head and head[head.index('-d') + 1] == str(tmp_path)


class FakeProc:
    def __init__(self, cmd, env):
        self.cmd, self.env, self.signals = cmd, env, []

    def send_signal(self, s):
        self.signals.append(s)

    def wait(self):
        os.kill(os.getpid(), signal.SIGUSR1)        # arrives while the parent waits: must reach the child, not stop us
        return 10


def test_respawn_sets_the_flag_forwards_signals_and_returns_the_exit_code(tmp_path):
    made = []

    def popen(cmd, env):
        made.append(FakeProc(cmd, env))
        return made[0]
    before = signal.getsignal(signal.SIGUSR1)
    rc = tracehook.respawn(['nyu'], str(tmp_path / 'rocprof'), popen=popen)
    assert rc == 10 and made[0].env[tracehook.ENV_FLAG] == '1' and made[0].signals == [signal.SIGUSR1]
    assert os.path.isdir(tmp_path / 'rocprof') and signal.getsignal(signal.SIGUSR1) is before
    assert tracehook.ENV_FLAG not in os.environ                          # the parent's own environment is untouched


def test_driver_respawns_before_any_gpu_call(tmp_path, monkeypatch):
    calls = []
    monkeypatch.setattr(tracehook, 'respawn', lambda argv, out: calls.append((list(argv), out)) or 7)
    monkeypatch.delenv(tracehook.ENV_FLAG, raising=False)
    argv = ['--model', 'msdn', '--ckptdir', str(tmp_path), '--id', 'x', '--profiler', 'rocprofv3', 'nyu']
    assert ann3depth.main(argv) == 7
    assert calls == [(argv, os.path.join(str(tmp_path), 'msdn_x', 'rocprof'))]
    assert not torch.cuda.is_initialized()
    # --trace-every 0 switches the hook off (src/ann3depth.py:105 passes the period through): nothing to respawn for,
    # and the driver goes on to ask for its GPU
    with pytest.raises(Exception):
        if torch.cuda.is_available():
            raise RuntimeError('GPU present: the rest of main() would train')
        ann3depth.main(argv[:-1] + ['--trace-every', '0', 'nyu'])
    assert len(calls) == 1


@pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/' + tracehook.ROCTX_LIB), reason='no ROCTx library')
def test_roctx_brackets_outside_a_profiler_are_harmless():
    r = tracehook.Roctx()
    r.begin(1)
    r.end()
