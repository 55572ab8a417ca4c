"""Big-table data parallelism as SURVEY 8(e) specifies (pytest -m gpu): reduce-scatter of the table gradient, every rank steps its
1/N slice (TV + ONE squared norm + clip + AdamW), all-gather of the fp16 copy the forward reads -- project-nerf_amd/sharded.py --
against the replicated optimiser, as two ranks on the box's one GPU (gloo instead of RCCL: everything but the transport)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_sharded_optimizer_equals_the_replicated_optimizer():
    env = dict(os.environ, NERF_SINGLE_DEVICE="1", NERF_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "tests", "dp_sharded_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "SHARDED OPTIMISER OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
