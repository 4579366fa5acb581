"""domain_rag_amd.rccl: every rank puts RCCL's platform environment in place itself (it used to be set only
by bench.py's own launcher, so a driver-started torch.distributed.run rank never got it) — without overriding the user's choice."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, env_extra):
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env.update(env_extra, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_importing_the_package_sets_the_ipc_mode_before_any_gpu_call():
    assert _child("import os, domain_rag_amd; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])", {}) == "0"


def test_the_users_value_wins():
    code = "import os, domain_rag_amd; from domain_rag_amd import rccl; print(rccl.prepare_env()['HSA_ENABLE_IPC_MODE_LEGACY'])"
    assert _child(code, {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}) == "1"


def test_bench_sets_it_before_torch_is_imported_and_joins_through_init_rccl():
    """bench.py as a rank of a foreign launcher: the variable is set at the top of the file, ahead of `import torch`, and the
    communicator is opened through rccl.init_rccl (which sets it again and binds the device)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < src.index("import torch  # noqa: E402")
    assert "init_rccl(self.dev)" in src


def test_init_rccl_refuses_cpu_devices():
    code = ("from domain_rag_amd.rccl import init_rccl\n"
            "try:\n    init_rccl('cpu')\n    print('no error')\nexcept RuntimeError as e:\n    print('refused')")
    assert _child(code, {}) == "refused"
