"""Decode worker process of `feeder.ProcessDecodePool` (the role of the reference's DataLoader worker processes,
train_tdeed.py:131-139): reads jobs "<id> <shm name> <byte offset> <H> <W> <path>" (or "J <json>" for a run of consecutive
frames) from stdin, decodes the images with Pillow into the shared staging slot as uint8 (3,H,W) RGB (feeder.read_frame's
contract) and answers "<id> ok" / "<id> err <message>".
Imports nothing heavy (no torch): a worker starts in a fraction of a second."""
import sys


def main():
    import numpy as np
    from multiprocessing import shared_memory
    from PIL import Image
    shms = {}
    out = sys.stdout
    for line in sys.stdin:
        line = line.rstrip("\n")
        if not line:
            continue
        if line == "quit":
            break
        if line.startswith("J "):            # several consecutive frames of one slot: {"id", "shm", "off", "H", "W", "paths"}
            import json
            job = json.loads(line[2:])
            jid, name, off, H, W = job["id"], job["shm"], job["off"], job["H"], job["W"]
            paths = job["paths"]
        else:
            jid, name, off, H, W, path = line.split(" ", 5)
            paths = [path]
        try:
            shm = shms.get(name)
            if shm is None:
                shm = shms[name] = shared_memory.SharedMemory(name=name)
                # attaching registers the block with THIS process's resource tracker (Python < 3.13), which would unlink it --
                # and warn about a leak -- when the worker exits; the block belongs to the parent
                try:
                    from multiprocessing import resource_tracker
                    resource_tracker.unregister(shm._name, "shared_memory")
                except Exception:       # noqa: BLE001
                    pass
            H, W, off = int(H), int(W), int(off)
            for i, path in enumerate(paths):
                dst = np.ndarray((3, H, W), dtype=np.uint8, buffer=shm.buf, offset=off + i * 3 * H * W)
                with Image.open(path) as im:
                    a = np.asarray(im.convert("RGB"))
                if a.shape != (H, W, 3):
                    raise ValueError(f"frame {path}: {a.shape} does not fit the slot ({H}, {W}, 3)")
                np.copyto(dst, np.moveaxis(a, 2, 0))
            out.write(f"{jid} ok\n")
        except Exception as e:      # noqa: BLE001  (reported to the parent, which raises)
            out.write(f"{jid} err {type(e).__name__}: {e}\n".replace("\r", " "))
        out.flush()
    for shm in shms.values():
        try:
            shm.close()
        except Exception:           # noqa: BLE001
            pass


if __name__ == "__main__":
    main()
