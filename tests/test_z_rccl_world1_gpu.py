"""The data-parallel schedule on a real RCCL communicator (one rank: the GPU box has one card).  Collected late on
purpose: it runs a child process with its own RCCL group, and a failure there should not hide the suites before it."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_group_runs_the_whole_schedule():
    """hooks -> in-order buckets -> side stream -> RCCL all-reduce(AVG), forced on a 1-rank RCCL group
    (DSSVAE_DP_FORCE=1): same parameters as the plain run, and buckets did overlap with backward."""
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]   # (a free port of its own: the probe's default is a fixed one)
    s.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_world1_probe.py")], capture_output=True,
                         text=True, timeout=170, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    if out.returncode != 0 or not out.stdout.strip().endswith("ok"):
        # the whole story for whoever reads the failure (pytest shortens long assertion messages)
        log_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(log_dir):
            with open(os.path.join(log_dir, "dp_world1_probe_failure.log"), "w") as fh:
                fh.write("returncode %s\n---- stdout\n%s\n---- stderr\n%s\n" % (out.returncode, out.stdout, out.stderr))
        lines = [ln for ln in out.stderr.splitlines() if ln.strip() and not ln.startswith("frame #")]
        raise AssertionError("dp_world1_probe.py: rc %s\nstdout tail: %s\nstderr (no frames): %s" %
                             (out.returncode, out.stdout[-400:], "\n".join(lines[-25:])))
