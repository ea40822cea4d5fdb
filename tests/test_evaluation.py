"""Known-answer tests of the evaluator restatements against the reference's only golden DATA:
PSDS_Eval/meta (ground truth, durations, the student's detections at 0.5 and at the 50 PSDS thresholds) with the numbers the
reference publishes for them -- the sed_eval text reports next to the predictions and the printed cell outputs of
PSDS_Eval/PSDS_Evaluation.ipynb.  The fixture tests/golden/psds_eval_meta.npz is built by tests/golden/make_psds_fixture.py.

Time stamps are stored as integer milliseconds.  The published numbers were produced from `pandas.read_csv` of 3-decimal
text with the pandas of 2021, whose default float parser scaled the digit string by repeated division ("7.488" ->
(7488 / 10) / 100 = 7.4879999999999995, one ulp below the correctly rounded double).  One reference/detection pair of the
golden set sits EXACTLY on the 200 ms onset collar (7.688 vs 7.488), so that ulp decides a true positive: `_seconds(ms,
legacy=True)` reproduces the parser and with it every digit of the published reports; with correctly rounded inputs the
Running_water row has 82 instead of 81 true positives (test_collar_boundary_case)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from desed_task_amd.evaluation import evaluation_measures as EM
from desed_task_amd.evaluation.psds import PSDSEval, PSDSEvalError

FIX = os.path.join(os.path.dirname(__file__), "golden", "psds_eval_meta.npz")


def _seconds(ms, legacy):
    ms = ms.astype(np.float64)
    return (ms / 10.0) / 100.0 if legacy else ms / 1000.0


@pytest.fixture(scope="module")
def golden():
    z = np.load(FIX)
    files, labels = z["files"], z["labels"]

    def table(prefix, sl=slice(None), legacy=True):
        t = pd.DataFrame({"filename": files[z[prefix + "_file"][sl]], "onset": _seconds(z[prefix + "_onset_ms"][sl], legacy),
                          "offset": _seconds(z[prefix + "_offset_ms"][sl], legacy),
                          "event_label": labels[np.maximum(z[prefix + "_label"][sl], 0)]})
        if prefix == "gt":
            t.loc[z["gt_label"] < 0, ["onset", "offset", "event_label"]] = np.nan          # clips without events
        return t

    st = z["op_start"]
    ops = {float(z["op_threshold"][i]): table("op", slice(st[i], st[i + 1])) for i in range(len(st) - 1)}
    return dict(gt=table("gt"), durations=pd.DataFrame({"filename": files, "duration": z["durations"]}), p05=table("p05"),
                ops=ops, gt_exact=table("gt", legacy=False), p05_exact=table("p05", legacy=False),
                expected=json.loads(str(z["expected_json"])), labels=list(labels))


def _pct(x, decimals):
    return round(100.0 * x, decimals)


def _check_report(results, expected, class_cols):
    e_over, e_macro = expected["overall"], expected["macro"]
    o, m = results["overall"], results["class_wise_average"]
    assert _pct(o["f_measure"]["f_measure"], 2) == e_over["F-measure (F1)"]
    assert _pct(o["f_measure"]["precision"], 2) == e_over["Precision"]
    assert _pct(o["f_measure"]["recall"], 2) == e_over["Recall"]
    assert round(o["error_rate"]["error_rate"], 2) == e_over["Error rate (ER)"]
    assert round(o["error_rate"]["substitution_rate"], 2) == e_over["Substitution rate"]
    assert round(o["error_rate"]["deletion_rate"], 2) == e_over["Deletion rate"]
    assert round(o["error_rate"]["insertion_rate"], 2) == e_over["Insertion rate"]
    assert _pct(m["f_measure"]["f_measure"], 2) == e_macro["F-measure (F1)"]
    assert _pct(m["f_measure"]["precision"], 2) == e_macro["Precision"]
    assert _pct(m["f_measure"]["recall"], 2) == e_macro["Recall"]
    assert round(m["error_rate"]["error_rate"], 2) == e_macro["Error rate (ER)"]
    assert round(m["error_rate"]["deletion_rate"], 2) == e_macro["Deletion rate"]
    assert round(m["error_rate"]["insertion_rate"], 2) == e_macro["Insertion rate"]
    for prefix, row in expected["classes"].items():          # the report truncates labels to 10 characters
        (label,) = [c for c in results["class_wise"] if c.startswith(prefix)]
        got = class_cols(results["class_wise"][label])
        assert len(got) == len(row)
        for g, want in zip(got, row):
            assert abs(g - want) < 1e-9, (label, got, row)


def test_event_based_report(golden):
    metric = EM.event_based_evaluation_df(golden["gt"], golden["p05"], t_collar=0.2, percentage_of_length=0.2)
    assert metric.evaluated_files == 1168
    assert round(metric.evaluated_length, 2) == 10459.12
    cols = lambda w: [w["count"]["Nref"], w["count"]["Nsys"], _pct(w["f_measure"]["f_measure"], 1),
                      _pct(w["f_measure"]["precision"], 1), _pct(w["f_measure"]["recall"], 1),
                      round(w["error_rate"]["error_rate"], 2), round(w["error_rate"]["deletion_rate"], 2),
                      round(w["error_rate"]["insertion_rate"], 2)]
    _check_report(metric.results(), golden["expected"]["event"], cols)
    assert "Event based metrics" in str(metric)


def test_collar_boundary_case(golden):
    """The one pair on the onset collar: a match with correctly rounded time stamps, not with the legacy-parsed ones."""
    exact = EM.event_based_evaluation_df(golden["gt_exact"], golden["p05_exact"])
    legacy = EM.event_based_evaluation_df(golden["gt"], golden["p05"])
    assert 7.688 - 7.488 <= 0.2 < (7688 / 10) / 100 - (7488 / 10) / 100
    assert exact.class_wise["Running_water"]["Ntp"] == 82 and legacy.class_wise["Running_water"]["Ntp"] == 81
    assert exact.overall["Ntp"] == legacy.overall["Ntp"] + 1
    for c in golden["labels"]:
        if c != "Running_water":
            assert exact.class_wise[c] == legacy.class_wise[c]


def test_segment_based_report(golden):
    metric = EM.segment_based_evaluation_df(golden["gt"], golden["p05"], time_resolution=1.0)
    res = metric.results()
    cols = lambda w: [w["count"]["Nref"], w["count"]["Nsys"], _pct(w["f_measure"]["f_measure"], 1),
                      _pct(w["f_measure"]["precision"], 1), _pct(w["f_measure"]["recall"], 1),
                      round(w["error_rate"]["error_rate"], 2), round(w["error_rate"]["deletion_rate"], 2),
                      round(w["error_rate"]["insertion_rate"], 2), _pct(w["accuracy"]["sensitivity"], 1),
                      _pct(w["accuracy"]["specificity"], 1), _pct(w["accuracy"]["balanced_accuracy"], 1),
                      _pct(w["accuracy"]["accuracy"], 1)]
    _check_report(res, golden["expected"]["segment"], cols)
    e = golden["expected"]["segment"]["overall"]
    acc = res["overall"]["accuracy"]
    assert [_pct(acc[k], 2) for k in ("sensitivity", "specificity", "balanced_accuracy", "accuracy")] == \
        [e["Sensitivity"], e["Specificity"], e["Balanced accuracy"], e["Accuracy"]]
    assert metric.evaluated_files == 1168


def test_notebook_f_scores(golden, tmp_path):
    nb = golden["expected"]["notebook"]
    gt_path = tmp_path / "gt.tsv"
    golden["gt"].to_csv(gt_path, sep="\t", index=False)
    ev_macro, ev_micro, seg_macro, seg_micro = EM.log_sedeval_metrics(golden["p05"], str(gt_path), save_dir=str(tmp_path / "out"))
    assert [_pct(v, 2) for v in (ev_macro, ev_micro, seg_macro, seg_micro)] == \
        [nb["event_macro_f1_pct"], nb["event_micro_f1_pct"], nb["segment_macro_f1_pct"], nb["segment_micro_f1_pct"]]
    assert (tmp_path / "out" / "event_f1.txt").exists() and (tmp_path / "out" / "segment_f1.txt").exists()
    f1 = EM.compute_per_intersection_macro_f1({0.5: golden["p05"]}, golden["gt"], golden["durations"])
    assert _pct(f1, 2) == nb["intersection_f1_pct"]
    assert EM.log_sedeval_metrics(golden["p05"].iloc[:0], str(gt_path)) == (0.0, 0.0, 0.0, 0.0)


def test_psds_scenarios(golden, tmp_path):
    nb = golden["expected"]["notebook"]
    dur_path, gt_path = tmp_path / "dur.tsv", tmp_path / "gt.tsv"
    golden["durations"].to_csv(dur_path, sep="\t", index=False)
    golden["gt"].to_csv(gt_path, sep="\t", index=False)
    psds1 = EM.compute_psds_from_operating_points(golden["ops"], str(gt_path), str(dur_path), dtc_threshold=0.7,
                                                  gtc_threshold=0.7, alpha_ct=0, alpha_st=1, save_dir=str(tmp_path / "s1"))
    psds2 = EM.compute_psds_from_operating_points(golden["ops"], golden["gt"], golden["durations"], dtc_threshold=0.1,
                                                  gtc_threshold=0.1, cttc_threshold=0.3, alpha_ct=0.5, alpha_st=1)
    assert f"{psds1:.3f}" == f"{nb['psds1']:.3f}"
    assert f"{psds2:.3f}" == f"{nb['psds2']:.3f}"
    assert len(os.listdir(tmp_path / "s1" / "predictions_dtc0.7_gtc0.7_cttc0.3")) == 50


def test_psds_properties(golden):
    """Size-independent properties of the score on the golden data."""
    ev = PSDSEval(ground_truth=golden["gt"], metadata=golden["durations"], dtc_threshold=0.5, gtc_threshold=0.5)
    keys = sorted(golden["ops"])
    for k in keys:
        ev.add_operating_point(golden["ops"][k], info={"threshold": k})
    assert ev.num_operating_points() == 50
    with pytest.warns(UserWarning):
        ev.add_operating_point(golden["ops"][keys[3]].sample(frac=1.0, random_state=0))     # same rows, shuffled: a duplicate
    assert ev.num_operating_points() == 50
    base = ev.psds(alpha_ct=0, alpha_st=0, max_efpr=100).value
    assert ev.psds(alpha_ct=0.5, alpha_st=0, max_efpr=100).value <= base          # cross triggers only cost
    assert ev.psds(alpha_ct=0, alpha_st=1, max_efpr=100).value <= base            # instability only costs
    assert 0.0 < base < 1.0
    roc = ev.psds(alpha_ct=0, alpha_st=0, max_efpr=100).plt
    assert (np.diff(roc.yp) >= -1e-12).all() and (np.diff(roc.xp) > 0).all()     # the PSD-ROC is a monotone staircase
    # fewer operating points can only lower the area
    sub = PSDSEval(ground_truth=golden["gt"], metadata=golden["durations"], dtc_threshold=0.5, gtc_threshold=0.5)
    for k in keys[::5]:
        sub.add_operating_point(golden["ops"][k])
    assert sub.psds(max_efpr=100).value <= base + 1e-12
    # the ground truth scored against itself is perfect: every event a true positive, no false positive
    perfect = PSDSEval(ground_truth=golden["gt"], metadata=golden["durations"])
    gt_as_det = golden["gt"].dropna()
    f_avg, per_class = perfect.compute_macro_f_score(gt_as_det)
    assert f_avg == pytest.approx(1.0) and set(per_class) == set(golden["labels"])
    perfect.add_operating_point(gt_as_det)
    assert perfect.psds(max_efpr=100).value == pytest.approx(1.0)
    counts = perfect.operating_points.counts.iloc[0]
    assert counts[:-1, -1].sum() == 0 and np.diag(counts)[:-1].sum() == len(gt_as_det)


def test_psds_small_cases():
    """Hand-computed cases for the three intersection criteria."""
    meta = pd.DataFrame({"filename": ["a.wav", "b.wav"], "duration": [10.0, 10.0]})
    gt = pd.DataFrame({"filename": ["a.wav", "a.wav", "b.wav"], "onset": [0.0, 5.0, np.nan], "offset": [4.0, 9.0, np.nan],
                       "event_label": ["x", "y", np.nan]})
    det = pd.DataFrame({"filename": ["a.wav", "a.wav", "a.wav", "b.wav"], "onset": [0.0, 2.0, 5.0, 1.0],
                        "offset": [1.9, 4.0, 9.0, 2.0], "event_label": ["x", "x", "x", "y"]})
    ev = PSDSEval(ground_truth=gt, metadata=meta, dtc_threshold=0.5, gtc_threshold=0.5, cttc_threshold=0.3)
    assert ev.class_names[:-1] == ["x", "y"]
    counts, tpr, fpr, ctr = ev._evaluate_detections(ev._init_det_table(det))
    # x: both short detections lie inside the x event (relevant), together they cover 3.9 / 4 of it -> 1 TP;
    # the third x detection sits on the y event: cross trigger x->y and a false positive; the y detection in b.wav is a
    # false positive; the y event is missed.
    assert counts.tolist() == [[1, 1, 1], [0, 0, 1], [0, 0, 0]]
    assert tpr.tolist() == [1.0, 0.0]
    assert fpr.tolist() == [3600 / 20.0, 3600 / 20.0]
    assert ctr[0, 1] == 3600 / 4.0 and ctr[1, 0] == 0.0
    # GTC: one detection covering less than half of the event is relevant (DTC) but does not make a true positive
    counts, *_ = ev._evaluate_detections(ev._init_det_table(det.iloc[:1]))
    assert counts.tolist() == [[0, 0, 0], [0, 0, 0], [0, 0, 0]]
    with pytest.warns(UserWarning, match="absent from the ground truth"):
        counts, *_ = ev._evaluate_detections(ev._init_det_table(det.assign(event_label="unknown")))
    assert counts.sum() == 0
    # an operating point without detections (a bare pd.DataFrame(), as the trainer's buffers start out): TPR = 0, no FP
    counts, tpr, fpr, _ = ev._evaluate_detections(ev._init_det_table(pd.DataFrame()))
    assert counts.sum() == 0 and tpr.tolist() == [0.0, 0.0] and fpr.tolist() == [0.0, 0.0]
    ev2 = PSDSEval(ground_truth=gt, metadata=meta)
    ev2.add_operating_point(pd.DataFrame())
    ev2.add_operating_point(det)
    assert 0.0 <= ev2.psds(max_efpr=100).value <= 1.0
    f_avg, _ = ev2.compute_macro_f_score(pd.DataFrame())
    assert np.isnan(f_avg)                                   # no true positive anywhere (mapped to 0 by the caller)
    assert EM.compute_per_intersection_macro_f1({0.5: pd.DataFrame()}, gt, meta) == 0.0
    with pytest.raises(PSDSEvalError):
        PSDSEval(ground_truth=gt, metadata=meta, dtc_threshold=1.5)
    with pytest.raises(PSDSEvalError):
        PSDSEval(ground_truth=gt, metadata=meta).psds()


def test_event_matching_is_optimal():
    """Two reference events, two detections: a greedy pairing finds one match, the maximum matching two."""
    from desed_task_amd.evaluation.sed_eval_metrics import EventBasedMetrics
    ref = [dict(filename="f", onset=1.0, offset=3.0, event_label="x"), dict(filename="f", onset=1.2, offset=3.0, event_label="x")]
    est = [dict(filename="f", onset=1.15, offset=3.0, event_label="x"), dict(filename="f", onset=0.9, offset=3.0, event_label="x")]
    m = EventBasedMetrics(["x"], t_collar=0.2, percentage_of_length=0.2)
    m.evaluate(ref, est)
    assert m.overall["Ntp"] == 2 and m.class_wise["x"]["Ntp"] == 2
    m.evaluate([{"filename": "g"}], [dict(filename="g", onset=0.0, offset=1.0, event_label="x")])
    assert m.overall["Nfp"] == 1 and m.evaluated_files == 2


def _synthetic_scores(n_clips=12, T=40, classes=("a", "b", "c"), levels=8, seed=0):
    """Ground truth with irrational-ish boundaries (no exact ties in the intersection criteria) and posteriors that follow it
    (event evidence + a random walk + some leakage into the next class), quantised to `levels` values."""
    rng = np.random.default_rng(seed)
    hop = 0.064
    ts = np.arange(T + 1) * hop
    mid = (ts[:-1] + ts[1:]) / 2
    scores, gt, dur = {}, {}, {}
    for i in range(n_clips):
        aid = "clip%02d" % i
        dur[aid] = float(ts[-1] - 0.013 * (i % 3))
        ev, act = [], np.zeros((T, len(classes)))
        for k, c in enumerate(classes):
            for _ in range(rng.integers(0, 3)):
                on = rng.uniform(0, ts[-1] - 0.4)
                on, off = round(on, 5) + 1e-4 / 3, round(min(on + rng.uniform(0.15, 1.2), dur[aid]), 5) - 1e-4 / 7
                ev.append((on, off, c))
                inside = (mid >= on - 0.05) & (mid < off + 0.08)
                act[inside, k] += 0.55
                act[inside, (k + 1) % len(classes)] += 0.25          # cross-trigger fodder
        walk = np.cumsum(rng.normal(0, 0.06, (T, len(classes))), 0)
        s = np.clip(0.2 + act + walk + rng.normal(0, 0.05, act.shape), 0.0, 0.999)
        s = np.floor(s * levels) / levels
        scores[aid] = pd.DataFrame(np.concatenate((ts[:-1, None], ts[1:, None], s), 1), columns=["onset", "offset"] + list(classes))
        gt[aid] = ev
    return scores, gt, dur, list(classes), levels


@pytest.mark.parametrize("dtc,gtc,cttc,alpha_ct,alpha_st", [(0.7, 0.7, None, 0.0, 1.0), (0.1, 0.1, 0.3, 0.5, 1.0), (0.5, 0.5, 0.3, 1.0, 0.0)])
def test_psds_from_scores_equals_operating_points(dtc, gtc, cttc, alpha_ct, alpha_st):
    """With scores on a finite grid the all-thresholds PSDS must equal psds_eval-style PSDS with one operating point per distinct
    threshold: ties the sed_scores_eval-role function to the PSDSEval restatement that the reference's golden numbers pin."""
    from desed_task_amd.evaluation.psds_scores import psds_from_scores
    from oracle import sed_oracle as O
    scores, gt, dur, classes, levels = _synthetic_scores()
    got, single, roc, rocs = psds_from_scores(scores, gt, dur, dtc_threshold=dtc, gtc_threshold=gtc, cttc_threshold=cttc,
                                              alpha_ct=alpha_ct, alpha_st=alpha_st, max_efpr=2000.0)
    gt_df = pd.DataFrame([(a + ".wav", o, f, l) for a, ev in gt.items() for o, f, l in ev] +
                         [(a + ".wav", np.nan, np.nan, np.nan) for a, ev in gt.items() if not ev],
                         columns=["filename", "onset", "offset", "event_label"])
    meta = pd.DataFrame({"filename": [a + ".wav" for a in dur], "duration": list(dur.values())})
    ev = PSDSEval(ground_truth=gt_df, metadata=meta, dtc_threshold=dtc, gtc_threshold=gtc, cttc_threshold=0.3 if cttc is None else cttc)
    for j in range(-1, levels + 1):
        th = np.float64((j + 0.5) / levels)                      # between two grid values: enumerates every distinct detection set
        rows = []
        for a, df in scores.items():
            ts = np.concatenate((df.onset.to_numpy(), df.offset.to_numpy()[-1:]))
            for c, on, off in O.decode_events(df[classes].to_numpy(np.float32), np.float32(th)):
                rows.append((a + ".wav", ts[on], ts[off], classes[c]))
        ev.add_operating_point(pd.DataFrame(rows, columns=["filename", "onset", "offset", "event_label"]))
    want = ev.psds(alpha_ct=alpha_ct, alpha_st=alpha_st, max_efpr=2000.0).value
    assert 0.0 < want < 1.0
    assert got == pytest.approx(want, rel=1e-9, abs=1e-12)
    assert set(single) <= set(classes) and len(roc[0]) == len(roc[1])


def test_psds_from_scores_io_and_wrapper(golden, tmp_path):
    """TSV readers + the evaluation_measures wrapper; a perfect score table (1 inside every event, 0 elsewhere) scores 1."""
    from desed_task_amd.evaluation.psds_scores import read_audio_durations, read_ground_truth_events
    gt_path, dur_path = tmp_path / "gt.tsv", tmp_path / "dur.tsv"
    sub = golden["gt"][golden["gt"].filename.isin(sorted(set(golden["gt"].filename))[:40])]
    sub.to_csv(gt_path, sep="\t", index=False)
    golden["durations"][golden["durations"].filename.isin(set(sub.filename))].to_csv(dur_path, sep="\t", index=False)
    gt, dur = read_ground_truth_events(str(gt_path)), read_audio_durations(str(dur_path))
    assert len(gt) == 40 == len(dur) and all(k.endswith(("000", "0")) or True for k in gt)
    assert sum(len(v) for v in gt.values()) == int(sub.event_label.notna().sum())
    labels = golden["labels"]
    hop, scores = 0.016, {}
    for aid, events in gt.items():
        T = int(np.ceil(10.0 / hop))
        ts = np.arange(T + 1) * hop
        s = np.zeros((T, len(labels)))
        for o, f, l in events:
            s[int(np.floor(o / hop)):int(np.ceil(f / hop)), labels.index(l)] = 1.0
        scores[aid] = pd.DataFrame(np.concatenate((ts[:-1, None], ts[1:, None], s), 1), columns=["onset", "offset"] + labels)
    gt_events = {k: v for k, v in gt.items() if v}                                   # the reference drops clips without events
    value = EM.compute_psds_from_scores(scores, gt_events, {k: dur[k] for k in gt_events}, dtc_threshold=0.7, gtc_threshold=0.7,
                                        cttc_threshold=None, alpha_ct=0, alpha_st=1, save_dir=str(tmp_path / "out"))
    assert value == pytest.approx(1.0, abs=1e-12)
    assert len(os.listdir(tmp_path / "out" / "scores")) == len(scores)
    with pytest.raises(ValueError):
        EM.compute_psds_from_scores({}, gt_events, dur)
