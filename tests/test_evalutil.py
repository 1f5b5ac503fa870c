"""Host-side evaluation helpers against vectors produced by the reference's own
util/eval.py and util/score.py functions (tools/make_goldens.py: eval_utils).  Pure CPU."""
import numpy as np

from helpers import load_golden, act
from tdeed_amd import synth, evalutil as E


def _inputs(meta):
    seed, K1, T = meta["seed"], meta["K1"], meta["T"]
    videos = [tuple(v) for v in meta["videos"]]
    classes = {f"c{k}": k for k in range(1, K1)}
    labels = {v: synth.labels(seed + vi, 1, L, K1 - 1, 1, fg_frac=0.15)[0][0] for vi, (v, L, _) in enumerate(videos)}
    clips = []
    for vi, (v, L, _) in enumerate(videos):
        for ci, start in enumerate(range(-T // 2, L, T // 2)):
            sc = np.abs(act(seed + 10 * vi + ci, f"clip{vi}_{ci}", (T, K1))).astype(np.float32)
            sc[:, 0] *= 2.5
            sc /= sc.sum(axis=1, keepdims=True)
            sc[(ci * 7) % T] = 0.0
            clips.append((v, start, sc))
    return videos, classes, labels, clips


def _pack(evlist, classes):
    return {x["video"]: np.array([[e["frame"], classes[e["label"]], e["score"]] for e in x["events"]], np.float64).reshape(-1, 3)
            for x in evlist}


def _same_events(got, g, tag, videos):
    for v, _, _ in videos:
        want = g[f"{tag}__{v}"]
        assert got[v].shape == want.shape, (tag, v, got[v].shape, want.shape)
        if want.size:
            assert np.array_equal(got[v][:, :2], want[:, :2]), (tag, v)
            assert np.allclose(got[v][:, 2], want[:, 2], rtol=1e-6, atol=1e-9), (tag, v)


def test_stitch_events_nms_map_match_reference():
    meta, g = load_golden("eval_utils")
    videos, classes, labels, clips = _inputs(meta)
    st = E.ScoreStitcher(videos, meta["K1"])
    for v, start, sc in clips:
        st.add(v, start, sc)
    for v, _, _ in videos:
        assert np.allclose(st.tracks[v][0], g[f"scores__{v}"], rtol=1e-6, atol=1e-7)
        assert np.array_equal(st.tracks[v][1], g[f"support__{v}"])
    norm = st.normalised()
    pe, pehr, stats = E.frame_events(norm, classes, st.fps, high_recall_score_threshold=meta["hr_thr"], labels=labels)
    _same_events(_pack(pe, classes), g, "pe", videos)
    _same_events(_pack(pehr, classes), g, "pehr", videos)
    assert abs(stats["err"] - float(g["err"])) < 1e-12
    assert abs(stats["f1"][None] - float(g["f1_any"])) < 1e-12
    assert np.allclose([stats["f1"][k] for k in range(1, meta["K1"])], g["f1_cls"], atol=1e-12)
    assert np.array_equal(np.array([stats["tp_fp_fn"][k] for k in [None] + list(range(1, meta["K1"]))]), g["tpfpfn"])
    nms1 = E.non_maximum_suppression(pehr, window=2, threshold=0.10)
    nms2 = E.non_maximum_suppression(pehr, window=[1, 3, 2, 4], threshold=0.0)
    snms = E.soft_non_maximum_suppression(pehr, window=3, threshold=0.05)
    _same_events(_pack(nms1, classes), g, "nms1", videos)
    _same_events(_pack(nms2, classes), g, "nms2", videos)
    _same_events(_pack(snms, classes), g, "snms", videos)
    truth = [{"video": v, "events": [{"label": f"c{int(k)}", "frame": int(i)} for i, k in enumerate(labels[v]) if k != 0]}
             for v, _, _ in videos]
    assert np.allclose(E.mean_average_precisions(truth, pehr, [0, 1, 2, 4])[0], g["maps_hr"], atol=1e-12)
    assert np.allclose(E.mean_average_precisions(truth, nms1, [1, 2, 4])[0], g["maps_nms"], atol=1e-12)
    assert np.allclose(E.mean_average_precisions(truth, pe, [0, 2])[0], g["maps_pe"], atol=1e-12)


def test_stitcher_views_and_edges():
    videos = [("v", 10, 25.0)]
    st = E.ScoreStitcher(videos, 3)
    p = np.ones((2, 6, 3), np.float32)
    st.add_views("v", -2, p)                 # hangs over the front: frames 0..3 get both views
    st.add_views("v", 7, p)                  # hangs over the end: frames 7..9
    s, n = st.tracks["v"]
    assert n.tolist() == [2, 2, 2, 2, 0, 0, 0, 2, 2, 2] and s[:, 0].tolist() == [2, 2, 2, 2, 0, 0, 0, 2, 2, 2]
    assert np.allclose(st.normalised()["v"][:, 0], [1, 1, 1, 1, 0, 0, 0, 1, 1, 1])
    assert E.average_precision([("v", 3, 0.9), ("v", 8, 0.5)], {"v": [3, 9]}, tolerance=0) == 0.5
    assert E.average_precision([("v", 3, 0.9), ("v", 8, 0.5)], {"v": [3, 9]}, tolerance=1) == 1.0


def test_stitch_predictions_drives_predict_like_evaluate():
    class Fake:
        def __init__(self):
            self.calls = []

        def predict(self, frames, augment_inference=False):
            self.calls.append(augment_inference)
            B, T = frames.shape[:2]
            sc = np.full((B, T, 3), 2.0 if augment_inference else 1.0, np.float32)
            return sc.argmax(-1), sc
    videos = [("a", 12, 25.0), ("b", 9, 25.0)]
    loader = [dict(frame=np.zeros((2, 8, 3, 4, 4), np.uint8), video=["a", "b"], start=np.array([-3, 4]))]
    st = E.stitch_predictions(Fake(), loader, videos, 3)
    assert st.tracks["a"][1].tolist() == [1] * 5 + [0] * 7 and st.tracks["b"][1].tolist() == [0] * 4 + [1] * 5
    m = Fake()
    st = E.stitch_predictions(m, [dict(frame=np.zeros((1, 8, 3, 4, 4), np.uint8), video=["a"], start=np.array([6]))], videos, 3,
                              augment=True)
    assert m.calls == [False, True] and st.tracks["a"][1].tolist() == [0] * 6 + [2] * 6
    assert np.allclose(st.normalised()["a"][6:, 0], 1.5)
