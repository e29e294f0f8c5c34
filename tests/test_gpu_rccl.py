"""-m gpu: the RCCL side of `sharding.py` as far as ONE device lets it run (no multi-GPU box existed in any round).

World 1 on backend "nccl" (= RCCL on ROCm): process-group setup on cuda:0 under the image's dmabuf-IPC setting,
`sharding.warm_up` (the collective every multi-rank entry point starts with, on its NCCL branch), all_reduce /
all_gather_object / barrier on device tensors, and the hand-off message packed and header-checked in device memory.
The point-to-point call itself needs two devices (RCCL refuses two ranks on one: `tools/diag/rccl_probe.py` leg 2,
`profiles/r05r_rccl_probe.log`); its protocol runs over gloo in tests/test_sharding_gloo.py and test_gpu_chained.py.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_collectives_and_handoff_message_on_device():
  sys.path.insert(0, os.path.join(ROOT, 'tools', 'diag'))
  import rccl_probe
  port = rccl_probe.free_port()
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'diag', 'rccl_probe.py'), '--rank', '0', '1', str(port)],
                     capture_output=True, text=True, timeout=300)
  out = p.stdout + p.stderr
  assert p.returncode == 0, out[-3000:]
  assert 'process group up' in out and 'sharding.warm_up' in out
  assert 'hand-off message packed / header-checked in device memory: ok' in out
  assert '[rank 0] done' in out
