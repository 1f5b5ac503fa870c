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


_HEAD = None


def _gpu_touched():
    try:
        import torch
        return torch.cuda.is_initialized()
    except Exception:       # noqa: BLE001
        return True


def head():
    """Revision the library was built from.  A checkout (.git present: the build container) asks git -- once per process
    and only while the process has not initialised the GPU; everywhere else (the GPU box receives the tree without .git;
    bench.py runs under a profiler's preload there) csrc/build_info.json, written by build(), answers without any child
    process."""
    global _HEAD
    if _HEAD is not None:
        return _HEAD
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.isdir(os.path.join(root, ".git")) and not _gpu_touched():
        try:
            h = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=root)
            if h.returncode == 0 and h.stdout.strip():
                _HEAD = h.stdout.strip()
                return _HEAD
        except OSError:
            pass
    try:
        with open(PATH) as fh:
            d = json.load(fh)
        _HEAD = d["git_head"] + ("+dirty" if d.get("dirty") else "")
    except (OSError, ValueError, KeyError):
        _HEAD = None
    return _HEAD
