"""Which source revision the in-tree library was built from: build() records `git rev-parse HEAD` next to the .so so that
measurements taken on a GPU box (which receives the tree without .git) can name it."""
import json
import os
import subprocess

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "build_info.json")


def record():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=root).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True,
                                    cwd=root).stdout.strip())
    except OSError:
        head, dirty = None, None
    if head:
        with open(PATH, "w") as fh:
            json.dump(dict(git_head=head, dirty=dirty), fh)
    return head


def head():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        h = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=root)
        if h.returncode == 0 and h.stdout.strip():
            return h.stdout.strip()
    except OSError:
        pass
    try:
        with open(PATH) as fh:
            d = json.load(fh)
        return d["git_head"] + ("+dirty" if d.get("dirty") else "")
    except (OSError, ValueError, KeyError):
        return None
