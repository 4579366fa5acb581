"""RCCL with MORE THAN ONE rank on the 1-GPU box: two processes, both on cuda:0, join one backend-"nccl"
communicator through domain_rag_amd.rccl.init_rccl (which puts HSA_ENABLE_IPC_MODE_LEGACY=0 in place itself — the launcher does
not set it here) and run the retrieval path's one exchange: all_gather_into_tensor of zero-padded row shards into the global row
order (retrieval/clip100_resnet_style_all_shots.py:298,419 is where north_star puts it), plus the all-reduce bench.py counts ranks
with.  NCCL / RCCL builds may refuse two ranks on one device ("Duplicate GPU detected"): then the test SKIPS and prints the
library's message — the 8-GPU run itself is the driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rccl_ranks_share_the_gpu_and_gather_padded_shards(gpu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}     # the product code must set it itself
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NCCL_DEBUG=env.get("NCCL_DEBUG", "WARN"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_rank.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.skip("two RCCL ranks on one GPU did not finish within 240 s (the runtime does not support it here)")
        outs.append((p.returncode, so, se))
    lines = [json.loads(ln) for _, so, _ in outs for ln in so.splitlines() if ln.startswith("{")]
    errors = [ln["error"] for ln in lines if "error" in ln]
    if errors or any(rc != 0 for rc, _, _ in outs):
        tail = " | ".join(e[-400:] for e in errors) or " | ".join(se[-400:] for _, _, se in outs)
        pytest.skip(f"this RCCL build does not run two ranks on one device: {tail}")
    assert len(lines) == 2 and sorted(ln["rank"] for ln in lines) == [0, 1]
    for ln in lines:
        assert ln["ok"] and ln["joined"] == 2 and ln["gathered_equal"] and ln["backend"] == "nccl", ln
        assert ln["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0", ln
    print("two RCCL ranks on one GPU:", lines)
