import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a GPU skips the gpu-marked tests instead of failing them; when they were asked
    for explicitly (-m gpu) they run and fail loudly -- the product has no CPU fallback."""
    if "gpu" in (config.getoption("-m") or "") or _hip_device_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


REF_MANIFEST = os.path.join(REPO, "oracle", "_ref.manifest")


def missing_reference_builds():
    """Names listed in oracle/_ref.manifest that are not in oracle/_ref/."""
    if not os.path.exists(REF_MANIFEST):
        return []
    with open(REF_MANIFEST) as fh:
        names = [ln.strip() for ln in fh if ln.strip() and not ln.lstrip().startswith("#")]
    return [n for n in names if not os.path.exists(os.path.join(REPO, "oracle", "_ref", n))]


def _gpu_run(config):
    expr = (config.getoption("-m") or "").strip()
    return ("gpu" in expr and "not gpu" not in expr) or _hip_device_visible()


def pytest_sessionstart(session):
    """The reference builds are expected wherever oracle/_ref.manifest is: a run without one of them stops here (return code 3)
    instead of skipping the tests that compare with it and staying green.  A CPU run in the build container (where
    /root/reference is) builds what is missing first -- `make -C oracle ref`, the same recipe __graft_entry__.build() runs; a
    GPU run never does (/root/reference does not exist on the GPU box: the libraries travel with the snapshot or the run is red)."""
    # (the oracle's own C restatement: built by __graft_entry__.build(); a CPU run of a checkout that has not been built yet builds
    # it here -- two g++ calls -- instead of failing in the first test that loads it)
    if not _gpu_run(session.config) and not all(os.path.exists(os.path.join(REPO, "oracle", n))
                                                 for n in ("libpileup_oracle.so", "libssw_oracle.so")):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "restatement"], check=False)
    if os.environ.get("PEPPER_AMD_ALLOW_MISSING_REF"):
        return
    missing = missing_reference_builds()
    if missing and not _gpu_run(session.config) and os.path.isdir("/root/reference"):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"], check=False)
        missing = missing_reference_builds()
    if missing:
        pytest.exit("oracle/_ref is missing %s (listed in oracle/_ref.manifest): the tests that compare with the reference's own "
                    "code cannot run.  Build them with `make -C oracle ref` where /root/reference exists, or set "
                    "PEPPER_AMD_ALLOW_MISSING_REF=1 to skip those tests knowingly." % ", ".join(missing), returncode=3)


def need_reference_build(what):
    """For a test that compares with oracle/_ref: skip only when skipping was asked for (PEPPER_AMD_ALLOW_MISSING_REF), fail otherwise."""
    if os.environ.get("PEPPER_AMD_ALLOW_MISSING_REF"):
        pytest.skip(what + " not built (PEPPER_AMD_ALLOW_MISSING_REF is set)")
    pytest.fail(what + " not built and oracle/_ref.manifest expects it: run `make -C oracle ref` where /root/reference exists")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
