import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


_POISON = {}


def poison_lds():
    """Leave NaN in ~150 KB of every CU's LDS: the weight fragments of an all-NaN actor, staged by the 32x32 on-rows kernel
    (mdr_actor_sample) on a batch that fills the persistent grid.  A kernel launched next that reads LDS it has not written then
    computes with NaN instead of passing by luck (how the bf16x3 window over-read of round 3 was pinned down)."""
    import torch
    from mdr_amd.policy import FusedActor
    if not _POISON:
        nan = float("nan")
        _POISON["actor"] = FusedActor(torch.full((100, 180), nan), torch.full((100,), nan), torch.full((100, 100), nan),
                                      torch.full((100,), nan), torch.full((2, 100), nan), torch.full((2,), nan), layout=0)
        _POISON["junk"] = torch.zeros((256 * 256, 180), device="cuda:0")      # 256 workgroups x 8 waves x 32 agents: one workgroup per CU
    _POISON["actor"].sample(_POISON["junk"], seed=0, step=0)


@pytest.fixture(autouse=True)
def _lds_poisoned_before_every_gpu_test(request):
    """Every GPU test starts on LDS full of NaN (one 1 ms launch): no kernel of this library may depend on what an earlier kernel
    left in LDS."""
    if request.node.get_closest_marker("gpu") is not None and os.environ.get("MDR_TEST_POISON_LDS", "1") != "0":
        import torch
        if torch.cuda.is_available():
            poison_lds()
            if os.environ.get("MDR_TEST_POISON_HBM", "1") != "0":
                # ... and on recycled device memory full of NaN: torch.empty() hands out blocks of the caching allocator, so a buffer
                # this library reads before anything wrote it holds NaN (0xFF bytes: -1 / 255 as integers) instead of old zeros
                junk = torch.full((192 * 1024 * 1024,), float("nan"), device="cuda:0")
                junk.view(torch.int32).fill_(-1)
                del junk
    yield
