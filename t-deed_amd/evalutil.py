"""Host-side evaluation helpers around `TDEEDModel.predict`: stitching overlapping clip
predictions back into per-video score tracks, turning the tracks into spotted events, (soft) non-maximum suppression
and the tolerance-based mAP.  These are KB-sized numpy / python computations in the reference as well
(/root/reference/util/eval.py:34-261, 284-349; /root/reference/util/score.py:16-123); they stay on the CPU, restated here
with array operations where the reference loops in python, and are pinned against the reference's own functions by
`tests/golden/eval_utils.npz` (tools/make_goldens.py).

Event lists use the reference's records: {'video', 'events': [{'label', 'frame', 'score'}, ...], 'fps'}."""
from collections import defaultdict

import numpy as np

# util/eval.py:23-32
TOLERANCES = {"default": [1, 2, 4], "soccernet": [3, 6], "soccernetball": [6, 12]}
WINDOWS = {"default": [1, 3], "soccernet": [3, 6], "soccernetball": [6, 12], "tennis": [1, 3], "finegym": [1, 3]}


class ScoreStitcher:
    """Per-video accumulation of clip predictions (util/eval.py:284-349): scores (L, K+1) fp32 summed over the clips that
    cover a frame, support (L,) int32 counting them."""

    def __init__(self, videos, n_cols):
        """videos: iterable of (name, length, fps) like `dataset.videos`."""
        self.tracks = {v: (np.zeros((int(n), n_cols), np.float32), np.zeros(int(n), np.int32)) for v, n, _ in videos}
        self.fps = {v: f for v, _, f in videos}

    @staticmethod
    def _window(start, pred_len, video_len):
        """(offset into the clip, first video frame, number of frames) of a clip that may hang over either end."""
        off = -start if start < 0 else 0
        first = max(start, 0)
        n = min(pred_len - off, video_len - first)
        return off, first, max(n, 0)

    def add(self, video, start, pred_scores):
        """One clip of a dataloader batch: pred_scores (T, K+1).  A frame only counts as covered if the clip predicted
        something there (row sum != 0), util/eval.py:312-313."""
        scores, support = self.tracks[video]
        off, first, n = self._window(int(start), pred_scores.shape[0], scores.shape[0])
        p = pred_scores[off:off + n]
        scores[first:first + n] += p
        support[first:first + n] += (p.sum(axis=1) != 0).astype(np.int32)

    def add_views(self, video, start, pred_scores):
        """All views of one clip at once: pred_scores (V, T, K+1) (dataset-batched branch, util/eval.py:316-346)."""
        scores, support = self.tracks[video]
        off, first, n = self._window(int(start), pred_scores.shape[1], scores.shape[0])
        p = pred_scores[:, off:off + n]
        scores[first:first + n] += p.sum(axis=0)
        support[first:first + n] += p.shape[0]

    def normalised(self):
        """video -> mean score per frame (frames never covered divide by 1), util/eval.py:104-107."""
        out = {}
        for v in sorted(self.tracks):
            s, n = self.tracks[v]
            out[v] = s / np.maximum(n, 1)[:, None].astype(np.float32)
        return out


def stitch_predictions(model, loader, videos, n_cols, augment=False):
    """The prediction loop of `evaluate` (util/eval.py:284-349): run `model.predict` over the clips of `loader` and
    accumulate them per video.  Batches carry 'frame' (B,T,3,H,W), 'video' (names) and 'start' (first frame of each
    clip, may be negative).  augment=False: clips are scored one view each; augment=True: loader batch size 1 and every
    clip is scored twice, plain and horizontally flipped (`augment_inference=True`)."""
    st = ScoreStitcher(videos, n_cols)
    for clip in loader:
        starts = [int(s) for s in np.asarray(clip["start"]).reshape(-1)]
        if not augment:
            _, scores = model.predict(clip["frame"])
            for i in range(len(starts)):
                st.add(clip["video"][i], starts[i], scores[i])
        else:
            for flip in (False, True):
                _, scores = model.predict(clip["frame"], augment_inference=flip)
                st.add_views(clip["video"][0], starts[0], scores)
    return st


def frame_events(norm_scores, classes, fps, high_recall_score_threshold=0.01, labels=None):
    """`process_frame_predictions[_challenge]` (util/eval.py:86-192): per video the arg-max events and the high-recall
    events (every class whose score passes the threshold).  classes: name -> index (1-based, 0 = background).
    labels: optional video -> (L,) int ground truth; then the frame error rate and the foreground F1 counters
    (util/eval.py:34-84) are returned as well.
    Returns (pred_events, pred_events_high_recall, stats | None)."""
    inv = {v: k for k, v in classes.items()}
    cls_idx = np.array(sorted(inv), dtype=np.int64)
    events_all, recall_all = [], []
    n_err = n_tot = 0
    tp, fp, fn = defaultdict(int), defaultdict(int), defaultdict(int)
    for video in sorted(norm_scores):
        s = norm_scores[video]
        pred = s.argmax(axis=1)
        fg = np.nonzero(pred != 0)[0]
        events = [{"label": inv[int(pred[i])], "frame": int(i), "score": float(s[i, pred[i]])} for i in fg]
        ii, jj = np.nonzero(s[:, cls_idx] >= high_recall_score_threshold)       # row-major: frames, then classes ascending
        recall = [{"label": inv[int(cls_idx[j])], "frame": int(i), "score": float(s[i, cls_idx[j]])} for i, j in zip(ii, jj)]
        events_all.append({"video": video, "events": events, "fps": fps[video]})
        recall_all.append({"video": video, "events": recall, "fps": fps[video]})
        if labels is not None:
            true = np.asarray(labels[video])
            n_err += int((true != pred).sum())
            n_tot += true.shape[0]
            p_fg, t_fg = pred != 0, true != 0
            tp[None] += int((p_fg & t_fg).sum())
            fp[None] += int((p_fg & ~t_fg).sum())
            fn[None] += int((~p_fg & t_fg).sum())
            for k in inv:
                tp[k] += int(((pred == k) & (true == k)).sum())
                fp[k] += int(((pred == k) & (true != k)).sum())
                # a foreground frame of class k is missed when the prediction is anything else (background or another class)
                fn[k] += int(((true == k) & (pred != k)).sum())
    stats = None
    if labels is not None:
        def f1(k):
            den = tp[k] + 0.5 * fp[k] + 0.5 * fn[k]
            return tp[k] / (den if den != 0 else 1)
        stats = {"err": n_err / max(n_tot, 1), "f1": {k: f1(k) for k in [None] + sorted(inv)},
                 "tp_fp_fn": {k: (tp[k], fp[k], fn[k]) for k in [None] + sorted(inv)}}
    return events_all, recall_all, stats


def _by_label(events):
    groups = defaultdict(list)            # insertion order = first appearance, like the reference's defaultdict walk
    for e in events:
        groups[e["label"]].append(e)
    return groups


def _class_window(window, i):
    return window[i] if isinstance(window, list) else window


def non_maximum_suppression(pred, window, threshold=0.0):
    """util/eval.py:195-226: per video and label keep the best-scoring event, drop every other event of that label within
    +-window frames of it, repeat; stop at `threshold`.  (The first event found at the winner's frame is the one that
    is removed as "the winner", as in the reference.)"""
    out = []
    for vp in pred:
        kept = []
        for gi, evs in enumerate(_by_label(vp["events"]).values()):
            w = _class_window(window, gi)
            frames = np.array([e["frame"] for e in evs], dtype=np.int64)
            scores = np.array([e["score"] for e in evs], dtype=np.float64)
            alive = np.ones(len(evs), dtype=bool)
            while alive.any():
                cand = np.nonzero(alive)[0]
                best = cand[np.argmax(scores[cand])]                       # first maximum in list order, like max()
                if scores[best] < threshold:
                    break
                kept.append(dict(evs[best]))
                first_same = cand[np.nonzero(frames[cand] == frames[best])[0][0]]
                alive[first_same] = False
                alive &= ~((frames >= frames[best] - w) & (frames <= frames[best] + w))
        kept.sort(key=lambda e: e["frame"])
        nv = dict(vp)
        nv["events"] = kept
        nv["num_events"] = len(kept)
        out.append(nv)
    return out


def soft_non_maximum_suppression(pred, window, threshold=0.01):
    """util/eval.py:228-261: instead of dropping neighbours, scale their score by (distance / window)^2 (the winner's own
    frame gets 0), then remove only the winner."""
    out = []
    for vp in pred:
        kept = []
        for gi, evs in enumerate(_by_label(vp["events"]).values()):
            w = _class_window(window, gi)
            frames = np.array([e["frame"] for e in evs], dtype=np.int64)
            scores = np.array([e["score"] for e in evs], dtype=np.float64)
            alive = np.ones(len(evs), dtype=bool)
            while alive.any():
                cand = np.nonzero(alive)[0]
                best = cand[np.argmax(scores[cand])]
                if scores[best] < threshold:
                    break
                e = dict(evs[best])
                e["score"] = float(scores[best])
                kept.append(e)
                near = alive & (frames >= frames[best] - w) & (frames <= frames[best] + w)
                scores[near] = scores[near] * np.abs(frames[best] - frames[near]) ** 2 / (w ** 2)
                first_same = cand[np.nonzero(frames[cand] == frames[best])[0][0]]
                alive[first_same] = False
        kept.sort(key=lambda e: e["frame"])
        nv = dict(vp)
        nv["events"] = kept
        nv["num_events"] = len(kept)
        out.append(nv)
    return out


def average_precision(pred, truth, tolerance=0):
    """util/score.py:45-96.  pred: [(video, frame, score)] sorted by descending score; truth: video -> [frames].
    Greedy matching: every prediction takes the closest not-yet-recalled ground-truth frame of its video (first one wins
    ties); precision is recorded at every new recall, interpolated (running max from the right) and averaged over the
    number of ground-truth events."""
    total = sum(len(v) for v in truth.values())
    open_gt = {v: list(f) for v, f in truth.items()}
    pc = []
    recalled = 0
    for i, (video, frame, _) in enumerate(pred, 1):
        gts = open_gt.get(video)
        if not gts:
            continue
        d = np.abs(np.asarray(gts) - frame)
        j = int(np.argmin(d))                                   # first closest, like the strict '>' of the reference
        if d[j] <= tolerance:
            hit = gts[j]
            gts[:] = [f for f in gts if f != hit]               # the reference keys recalled events by (video, frame)
            recalled += 1
            pc.append(recalled / i)
    if not pc:
        return 0.0
    interp = np.maximum.accumulate(np.asarray(pc)[::-1])
    return float(interp.sum() / total)


def mean_average_precisions(truth, pred, tolerances=(0, 1, 2, 4)):
    """`compute_mAPs` (util/score.py:99-160) without the printing / plotting: returns (mAP per tolerance, {label: [AP per
    tolerance]}).  truth / pred: lists of the event records described in the module docstring."""
    assert {v["video"] for v in truth} == {v["video"] for v in pred}, "Video set mismatch!"
    by_label = defaultdict(lambda: defaultdict(list))
    for x in truth:
        for e in x["events"]:
            by_label[e["label"]][x["video"]].append(e["frame"])
    flat = defaultdict(list)
    for x in pred:
        for e in x["events"]:
            flat[e["label"]].append((x["video"], e["frame"], e["score"]))
    for lab in flat:
        flat[lab].sort(key=lambda r: r[-1], reverse=True)       # stable, like list.sort in get_predictions
    aps = {lab: [average_precision(flat.get(lab, []), by_label[lab], tol) for tol in tolerances] for lab in sorted(by_label)}
    maps = [float(np.mean([aps[lab][i] for lab in aps])) for i in range(len(tolerances))]
    return maps, aps
