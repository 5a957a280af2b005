import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def otw_golden():
    return np.load(os.path.join(GOLDEN, "otw_golden.npz"))


@pytest.fixture(scope="session")
def dtw_golden():
    return np.load(os.path.join(GOLDEN, "dtw_golden.npz"))


@pytest.fixture(scope="session")
def wtw_window_golden():
    return np.load(os.path.join(GOLDEN, "wtw_window_golden.npz"))


@pytest.fixture(scope="session")
def chopin_audio():
    z = np.load(os.path.join(GOLDEN, "chopin_20b_audio.npz"))
    # librosa.load semantics for these PCM16 stereo files: (L + R) / 65536 exactly, float32
    return {k.replace("_lr_sum", ""): (z[k].astype(np.float32) / np.float32(65536.0)) for k in z.files}


@pytest.fixture(scope="session")
def wtw_known_answer():
    return np.loadtxt(os.path.join(GOLDEN, "wtw_test_20b.txt"), dtype=np.int64)


def parse_case(meta):
    cid, variant, c, mrc, mode, euclid = str(meta).split("|")
    return dict(cid=cid, variant=variant, c=int(c), mrc=int(mrc), mode=mode, euclid=bool(int(euclid)),
                group=cid.split("_")[0])
