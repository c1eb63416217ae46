import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cae-lo_amd"), os.path.join(REPO, "oracle"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import caelo  # noqa: E402  (before anything can initialise HIP)
caelo.configure_runtime()  # the test session owns its process: four pipeline streams

GOLDEN = os.path.join(REPO, "tests", "golden")
WEIGHTS = os.path.join(REPO, "weights")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def models(orc):
    return orc.load_models(os.path.join(WEIGHTS, "SphericalRingPCRespondLayer.h5"),
                           os.path.join(WEIGHTS, "EncoderModel4VoxelPatch.h5"))


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from caelo.engine import default_engine
    return default_engine()


@pytest.fixture(scope="session")
def scans():
    from caelo import synth
    cache = {}

    def get(frame, n_beams=64, n_az=2000, quantum=None, scene_kind="boxes"):
        key = (frame, n_beams, n_az, quantum, scene_kind)
        if key not in cache:
            cache[key] = synth.make_scan(frame, n_beams=n_beams, n_az=n_az, quantum=quantum, scene_kind=scene_kind)
        return cache[key]
    return get
