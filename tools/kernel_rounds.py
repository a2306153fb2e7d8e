"""profiles/kernel_rounds.json: for every kernel source under pepper_amd/csrc, the round in which it last changed.

bench.py cites counters from committed profiles (profiles/rNN_*); a profile taken in a round BEFORE the kernel's source last
changed no longer describes the kernel that was timed, and the bench line marks such a leg "stale".  The GPU box has no .git,
so the mapping is computed here (the build container) from the history and committed:

    python tools/kernel_rounds.py            # rewrites profiles/kernel_rounds.json

Rounds are delimited by the driver's "round N: VERDICT ..." commits: a commit after the marker of round N belongs to round N + 1."""
import json
import os
import re
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rounds():
    log = subprocess.run(["git", "-C", REPO, "log", "--format=%H %s"], capture_output=True, text=True, check=True).stdout.splitlines()
    log.reverse()                                        # oldest first
    current, of_commit = 1, {}
    for line in log:
        sha, _, subject = line.partition(" ")
        of_commit[sha] = current
        m = re.match(r"round (\d+): VERDICT", subject)
        if m:
            current = int(m.group(1)) + 1
    out = {}
    src = os.path.join(REPO, "pepper_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if not name.endswith((".hip", ".h", ".cpp")):
            continue
        sha = subprocess.run(["git", "-C", REPO, "log", "-1", "--format=%H", "--", os.path.join("pepper_amd", "csrc", name)],
                             capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", REPO, "status", "--porcelain", "--", os.path.join("pepper_amd", "csrc", name)],
                               capture_output=True, text=True, check=True).stdout.strip()
        out[name] = current if (dirty or not sha) else of_commit.get(sha, current)
    return {"current_round": current, "last_changed_in_round": out}


if __name__ == "__main__":
    table = rounds()
    with open(os.path.join(REPO, "profiles", "kernel_rounds.json"), "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(table))
