"""The multi-process legs of the path on a one-GPU box: RCCL itself, and two ranks.

Children are separate python processes (tests/_dist_child.py) started from here: the RCCL process group has to be created
before anything else touches the GPU in its process, which a pytest process that has already run GPU tests cannot offer.
No 8-GPU node is needed: a group of ONE rank on backend "nccl" (= RCCL on ROCm) with DataParallel(always_reduce=True)
launches the same async bucket all-reduces on RCCL's stream, from inside the HIP backward, that N ranks would."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(mode, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_dist_child.py"), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    if any(p.returncode != 0 for p in procs):
        report = "\n".join(f"---- rank {r} of {mode}: exit code {p.returncode}\n{out[-2500:]}" for r, (p, out) in enumerate(zip(procs, outs)))
        raise AssertionError(report)
    return outs


def test_rccl_bucket_allreduce_inside_the_backward_one_rank():
    """backend "nccl" (RCCL), world size 1: small and default UNet through DataParallel -- three backward passes including a
    no_sync() accumulation -- gradients bit-equal to the un-wrapped run, one RCCL all-reduce per bucket per synchronising
    backward (launched == len(buckets)), plus gather_latents / max_over_ranks through RCCL."""
    out = _spawn("nccl1", 1)[0]
    print(out)
    assert "nccl1 small" in out and "nccl1 default" in out


def test_two_ranks_on_one_gpu_average_their_gradients():
    """Two processes sharing the GPU (gloo; RCCL refuses two ranks on one device): the real UNet backward feeds the reducer's
    bucket views on each rank, the collective averages them (device buckets staged through host memory on this backend);
    the result equals the single-process mean of both ranks' batches."""
    outs = _spawn("gloo2", 2)
    print(outs[0])
    assert "gloo2: gradients of 2 ranks" in outs[0]
