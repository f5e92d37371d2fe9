import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def built_scenes():
    """Host scenes built once per session through librtx_host.so (no GPU involved)."""
    from raytracer_amd import scenes
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = scenes.build(name)
        return cache[name]

    yield get
    for sc in cache.values():
        sc.close()


@pytest.fixture(scope="session")
def device_scenes(built_scenes):
    """rtx_scene handles on cuda:0, one per scene name (GPU tests only)."""
    from raytracer_amd import hip
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = hip.DeviceScene(built_scenes(name).desc, 0)
        return cache[name]

    yield get
    for ds in cache.values():
        ds.close()
