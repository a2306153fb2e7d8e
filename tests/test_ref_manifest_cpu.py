"""oracle/_ref.manifest: the reference builds the suite expects.  The manifest names exactly what oracle/Makefile's `ref` target
builds; a test run in which one of them is missing stops (return code 3) instead of skipping the comparisons with the reference's
own code -- unless PEPPER_AMD_ALLOW_MISSING_REF says the skips are wanted."""
import os
import re
import shutil
import subprocess
import sys

import conftest

REPO = conftest.REPO


def _manifest():
    with open(conftest.REF_MANIFEST) as fh:
        return [ln.strip() for ln in fh if ln.strip() and not ln.lstrip().startswith("#")]


def test_manifest_lists_what_the_makefile_builds():
    mk = open(os.path.join(REPO, "oracle", "Makefile")).read()
    targets = re.search(r"^ref: (.*)$", mk, re.M).group(1).split()
    assert sorted(os.path.basename(t) for t in targets) == sorted(_manifest())
    assert conftest.missing_reference_builds() == []          # (this run got past pytest_sessionstart, so they are all here)


def test_a_missing_reference_build_stops_a_gpu_run(tmp_path):
    """A copy of the tree's test scaffolding with one library taken away: `-m gpu` exits 3 before collecting anything; with
    PEPPER_AMD_ALLOW_MISSING_REF=1 the run goes on."""
    root = tmp_path / "repo"
    (root / "tests").mkdir(parents=True)
    (root / "oracle" / "_ref").mkdir(parents=True)
    shutil.copy(os.path.join(REPO, "tests", "conftest.py"), root / "tests" / "conftest.py")
    shutil.copy(conftest.REF_MANIFEST, root / "oracle" / "_ref.manifest")
    names = _manifest()
    for n in names[:-1]:
        (root / "oracle" / "_ref" / n).write_bytes(b"")
    (root / "tests" / "test_nothing.py").write_text("import pytest\n\n@pytest.mark.gpu\ndef test_nothing():\n    pass\n")
    env = {k: v for k, v in os.environ.items() if k != "PEPPER_AMD_ALLOW_MISSING_REF"}
    cmd = [sys.executable, "-m", "pytest", str(root / "tests"), "-q", "-m", "gpu", "-p", "no:cacheprovider"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(root))
    assert p.returncode == 3 and names[-1] in (p.stdout + p.stderr), p.stdout + p.stderr
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, PEPPER_AMD_ALLOW_MISSING_REF="1"), cwd=str(root))
    assert p.returncode == 0, p.stdout + p.stderr
    (root / "oracle" / "_ref" / names[-1]).write_bytes(b"")
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(root))
    assert p.returncode == 0, p.stdout + p.stderr
