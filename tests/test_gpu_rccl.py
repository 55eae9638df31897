"""The RCCL path on the one GPU of the test box: `init_process_group("nccl", world_size=1, device_id=cuda:0)` in a
fresh child process, and the result rows of a launch pushed through the REAL `all_gather_into_tensor` on device
tensors (`force=True` skips the one-rank short cut of pyrayhf_amd.dist).  What N > 1 adds to this - more ranks in the
same collective - is covered by the two-rank gloo tests (tests/test_dist_gloo.py); the first multi-GPU run must not
also be the first time the nccl backend, device_id binding and device-tensor gather execute at all."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from pyrayhf_amd import library, synth, dist as pdist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
alt, den, bmag, bpsi = synth.chapman_profiles(96, 77)
freq = synth.sounder_frequencies(4)[::8]
t = [torch.as_tensor(x, device=dev) for x in (freq, den, bmag, bpsi, alt)]
# equal shards: the result rows of one launch through the collective
vh = library.vertical_forward_operator(*t, "X", 640)
got = pdist.gather_rows(vh, 96, force=True)
assert got.is_cuda and got.data_ptr() != vh.data_ptr(), "the rows did not go through a collective"
assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(vh, nan=-1.0))
assert pdist.gather_rows(vh, 96) is vh                      # without force: the one-rank short cut
# a mixed work list: cut, launched, gathered and put back in global order
segs = [(0, 40, "O", 200), (40, 70, "X", 640), (70, 96, "O", 320)]
rows, local = pdist.shard_segments(segs, 1, 0)
mixed = library.vertical_forward_operator_mixed(t[0], t[1][rows], t[2][rows], t[3][rows], t[4], local)
full = pdist.gather_mixed(mixed, segs, 96, force=True)
want = library.vertical_forward_operator_mixed(*t, segs)
assert torch.equal(torch.nan_to_num(full, nan=-1.0), torch.nan_to_num(want, nan=-1.0))
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", int(torch.isfinite(vh).sum()))
"""


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    return env


def test_result_rows_through_rccl_all_gather_on_one_rank():
    done = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": free_port()}], cwd=ROOT, env=child_env(),
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
    assert "RCCL_ONE_RANK_OK" in done.stdout


def test_bench_force_collective_reports_nccl_and_gather_time():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "2", "--warmup", "1",
           "--profiles", "600", "--no-cpu-baseline", "--no-single-profile", "--no-legs"]
    done = subprocess.run(cmd, cwd=ROOT, env=child_env(), capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
    line = json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["world_size_seen"] == 1 and line["n_gpus"] == 1
    assert line["gather_ms"] is not None and line["gather_ms"] > 0.0
    assert line["kernel_ms_per_rank"]["min"] > 0.0 and len(line["kernel_ms_per_rank"]["ranks"]) == 1
