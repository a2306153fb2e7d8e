"""The one process-wide setting the package makes at import (pepper_amd/__init__.py): GPU_MAX_HW_QUEUES = 16 unless the caller
chose a value or asked for the runtime's default -- 4 serialises the image workers' streams, 32 oversubscribes the device's
resident queues once an inference pass has run (docs/LEDGER_r05.md, "Late finding")."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _queues_after_import(**env):
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "PEPPER_AMD_KEEP_HW_QUEUES")}
    base.update(env)
    out = subprocess.run([sys.executable, "-c", "import os, pepper_amd; print(os.environ.get('GPU_MAX_HW_QUEUES', 'unset'))"],
                         cwd=ROOT, env=base, capture_output=True, text=True, check=True)
    return out.stdout.strip()


def test_hardware_queue_setting():
    assert _queues_after_import() == "16"
    assert _queues_after_import(GPU_MAX_HW_QUEUES="8") == "8"                 # the caller's value wins
    assert _queues_after_import(PEPPER_AMD_KEEP_HW_QUEUES="1") == "unset"     # the runtime's default for an embedding process
