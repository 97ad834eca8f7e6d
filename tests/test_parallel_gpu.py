"""The data-parallel exchange on the backend the 8-GPU run uses: `nccl` (= RCCL on ROCm).  A one-GPU box cannot hold two
RCCL ranks, so this initialises the backend with world size 1 and puts the REAL 29.5 MB gradient bucket through
ncclAllReduce once ("not even a single-rank nccl init is exercised")."""
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_backend_all_reduces_the_real_bucket_world_size_1():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_rccl_worker.py')], capture_output=True, text=True,
                         timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert 'RCCL_OK' in out.stdout
    print(out.stdout.strip().splitlines()[-1])
