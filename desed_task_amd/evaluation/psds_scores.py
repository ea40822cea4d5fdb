"""Threshold-free PSDS from frame-level scores -- the role of `sed_scores_eval.intersection_based.psds` in the reference
(desed_task/evaluation/evaluation_measures.py:258-304; setup.py:16, `sed_scores_eval>=0.0.0`, third party, absent from the
reference tree and from this image): the PSD-ROC is built from EVERY decision threshold instead of a fixed grid of operating
points.

This is an independent exact algorithm for the same definition, not a restatement of that package's code: **parity with
sed_scores_eval itself is unpinned** (the reference holds no golden output of it).  What is pinned: for scores quantised to a
finite set of values the result equals, to rounding, `psds.PSDSEval` fed with one operating point per distinct threshold
(tests/test_evaluation.py::test_psds_from_scores_equals_operating_points), and `psds.PSDSEval` is pinned on the reference's
golden PSDS numbers.  Criteria, rates, effective FPR, staircase and area are shared with / identical to psds.py.

Algorithm.  Per clip and class the detections as a function of the threshold tau are piecewise constant with break points at
the clip's own score values, so every count (true positives, false positives, cross triggers) is a step function of tau per
clip; the data-set level counts are the sums of those step functions, obtained by sorting all break points once and
accumulating the per-clip jumps.  Inside a clip all thresholds are evaluated together: rows = thresholds, columns = frames;
runs of active frames are the detections; their intersections with ground-truth events are differences of per-frame prefix
sums; the ground-truth coverage by relevant detections is one (rows x frames) @ (frames x events) product.
"""
import os
from pathlib import Path

import numpy as np
import pandas as pd

from .psds import PSDSEval


def read_ground_truth_events(path):
    """{audio_id: [(onset, offset, event_label), ...]} from a filename / onset / offset / event_label TSV; clips whose only row
    has no event map to [] (sed_scores_eval.io.read_ground_truth_events, used at sed_trainer.py:503,736)."""
    if isinstance(path, dict):
        return path
    df = pd.read_csv(path, sep="\t")
    out = {}
    for rec in df.to_dict("records"):
        events = out.setdefault(Path(str(rec["filename"])).stem, [])
        if not pd.isna(rec.get("onset")) and not pd.isna(rec.get("event_label")):
            events.append((float(rec["onset"]), float(rec["offset"]), rec["event_label"]))
    return out


def read_audio_durations(path):
    """{audio_id: seconds} from a filename / duration TSV (sed_scores_eval.io.read_audio_durations)."""
    if isinstance(path, dict):
        return path
    df = pd.read_csv(path, sep="\t")
    return {Path(str(f)).stem: float(d) for f, d in zip(df["filename"], df["duration"])}


def _clip_class_steps(s, ts, ov_c, ov_gt, dur_gt, ov_other, ov_world, dtc, gtc, cttc):
    """One clip, one class.  s (T) scores, ts (T+1) frame boundaries, ov_c (T) per-frame overlap with this class's ground truth,
    ov_gt (T, G) per event of this class, dur_gt (G), ov_other (T, K) per other class (K = 0 without cross triggers), ov_world (T).
    -> thresholds u (n), tp / fp (n + 1) and ct (n + 1, K): row 0 = every frame active (tau below all scores), row i + 1 = the
    detections for tau in [u_i, u_{i+1})."""
    T = len(s)
    u = np.unique(s)
    thr = np.concatenate(([-np.inf], u))
    M = s[None, :] > thr[:, None]                                     # (rows, T); last row all False
    pad = np.zeros((len(thr), 1), bool)
    on = M & ~np.concatenate((pad, M[:, :-1]), 1)
    off = M & ~np.concatenate((M[:, 1:], pad), 1)
    r_on, a = np.nonzero(on)                                          # row-major: the k-th onset of a row pairs with its k-th offset
    r_off, b = np.nonzero(off)
    b = b + 1
    K = ov_other.shape[1]
    n_rows = len(thr)
    tp = np.zeros(n_rows)
    fp = np.zeros(n_rows)
    ct = np.zeros((n_rows, K))
    if len(a) == 0:
        return u, tp, fp, ct
    prefix = lambda v: np.concatenate((np.zeros((1,) + v.shape[1:]), np.cumsum(v, 0)))      # noqa: E731
    dur = ts[b] - ts[a]
    p_c = prefix(ov_c)
    same = p_c[b] - p_c[a]
    relevant = (same > 0) & (same >= dtc * dur) if ov_gt.shape[1] else np.zeros(len(a), bool)
    if ov_gt.shape[1]:
        # frames covered by relevant detections, per threshold row, then the coverage of every ground-truth event
        diff = np.zeros((n_rows, T + 1))
        np.add.at(diff, (r_on[relevant], a[relevant]), 1.0)
        np.add.at(diff, (r_on[relevant], b[relevant]), -1.0)
        R = np.cumsum(diff[:, :T], 1)
        cover = R @ ov_gt                                             # (rows, G)
        tp = (cover >= gtc * dur_gt[None, :]).sum(1).astype(np.float64)
    rest = ~relevant
    p_w = prefix(ov_world)
    w = p_w[b] - p_w[a]
    is_fp = rest & ((w >= cttc * dur) if cttc is not None else (w > 0)) & (w > 0)
    fp = np.bincount(r_on[is_fp], minlength=n_rows).astype(np.float64)
    if K:
        p_o = prefix(ov_other)
        x = p_o[b] - p_o[a]                                           # (runs, K)
        hit = rest[:, None] & (x >= cttc * dur[:, None]) & (x > 0)
        for k in range(K):
            ct[:, k] = np.bincount(r_on[hit[:, k]], minlength=n_rows)
    return u, tp, fp, ct


def _frame_overlap(ts, onset, offset):
    return np.maximum(0.0, np.minimum(ts[1:], offset) - np.maximum(ts[:-1], onset))


def psds_from_scores(scores, ground_truth, audio_durations, dtc_threshold=0.5, gtc_threshold=0.5, cttc_threshold=0.3,
                     alpha_ct=0.0, alpha_st=0.0, unit_of_time="hour", max_efpr=100.0):
    """-> (psds, {class: single-class psds}, (efpr axis, effective tpr), {class: (efpr_c, tpr_c)}).
    scores {audio_id: DataFrame(onset, offset, <one column per class>)}, ground_truth {audio_id: [(onset, offset, label)]},
    audio_durations {audio_id: seconds}.  cttc_threshold None: no cross triggers (alpha_ct must be 0), every non-relevant
    detection is a false positive."""
    if cttc_threshold is None and alpha_ct != 0:
        raise ValueError("cross triggers need a cttc_threshold")
    if alpha_st < 0 or not 0 <= alpha_ct <= 1:
        raise ValueError("alpha_st must be >= 0 and alpha_ct in [0, 1]")
    ids = sorted(ground_truth.keys())
    missing = [i for i in ids if i not in scores or i not in audio_durations]
    if missing:
        raise ValueError("scores / durations missing for %d clips, e.g. %s" % (len(missing), missing[:3]))
    if not ids:
        raise ValueError("no clips to evaluate")
    classes = list(scores[ids[0]].columns[2:])
    nc = len(classes)
    cidx = {c: i for i, c in enumerate(classes)}
    nsec = PSDSEval.secs_in_uot[unit_of_time]
    n_gt = np.zeros(nc)
    gt_dur = np.zeros(nc)
    total_dur = 0.0
    use_ct = cttc_threshold is not None and alpha_ct > 0
    # per class: break points and jumps of the per-clip step functions
    thr_l = [[] for _ in range(nc)]
    dtp_l = [[] for _ in range(nc)]
    dfp_l = [[] for _ in range(nc)]
    dct_l = [[] for _ in range(nc)]
    base_tp, base_fp, base_ct = np.zeros(nc), np.zeros(nc), np.zeros((nc, nc))
    for aid in ids:
        df = scores[aid]
        if list(df.columns[2:]) != classes:
            raise ValueError("score tables must share their class columns")
        vals = df.to_numpy(np.float64)
        ts = np.concatenate((vals[:, 0], vals[-1:, 1]))
        S = vals[:, 2:]
        dur_clip = float(audio_durations[aid])
        total_dur += dur_clip
        ov_world = _frame_overlap(ts, 0.0, dur_clip)
        events = [(float(o), float(f), l) for o, f, l in ground_truth[aid]]
        per_class = [[] for _ in range(nc)]
        for o, f, l in events:
            if l not in cidx:
                raise ValueError("ground-truth label %r has no score column" % (l,))
            per_class[cidx[l]].append((o, f))
            n_gt[cidx[l]] += 1
            gt_dur[cidx[l]] += f - o
        ov_gt = [np.stack([_frame_overlap(ts, o, f) for o, f in ev], 1) if ev else np.zeros((len(S), 0)) for ev in per_class]
        ov_cls = np.stack([g.sum(1) for g in ov_gt], 1)              # (T, nc)
        for c in range(nc):
            others = [k for k in range(nc) if k != c] if use_ct else []
            u, tp, fp, ct = _clip_class_steps(S[:, c], ts, ov_cls[:, c], ov_gt[c], np.array([f - o for o, f in per_class[c]]),
                                              ov_cls[:, others], ov_world, dtc_threshold, gtc_threshold, cttc_threshold)
            base_tp[c] += tp[0]; base_fp[c] += fp[0]
            thr_l[c].append(u); dtp_l[c].append(np.diff(tp)); dfp_l[c].append(np.diff(fp))
            if use_ct:
                base_ct[c, others] += ct[0]
                full = np.zeros((len(u), nc))
                full[:, others] = np.diff(ct, axis=0)
                dct_l[c].append(full)
    if total_dur <= 0:
        raise ValueError("the evaluated clips have no duration")
    rocs = {}
    for c in range(nc):
        thr = np.concatenate(thr_l[c])
        order = np.argsort(thr, kind="stable")
        thr = thr[order]
        tp = base_tp[c] + np.cumsum(np.concatenate(dtp_l[c])[order])
        fp = base_fp[c] + np.cumsum(np.concatenate(dfp_l[c])[order])
        last = np.concatenate((thr[1:] != thr[:-1], [True]))         # the operating point after ALL jumps at a score value
        tp = np.concatenate(([base_tp[c]], tp[last]))
        fp = np.concatenate(([base_fp[c]], fp[last]))
        with np.errstate(divide="ignore", invalid="ignore"):
            tpr = tp / n_gt[c]
            efpr = fp * nsec / total_dur
            if use_ct:
                ctc = base_ct[c][None, :] + np.cumsum(np.concatenate(dct_l[c])[order], 0)
                ctc = np.concatenate((base_ct[c][None, :], ctc[last]))
                ctr = ctc * nsec / gt_dur[None, :]
                ctr[:, c] = 0.0
                efpr = efpr + alpha_ct * PSDSEval._mean_ctr(ctr[:, :, None])[:, 0]
        # monotone upper envelope: best TPR reachable at or below each eFPR
        order = np.lexsort((tpr, efpr))
        x, y = efpr[order], np.maximum.accumulate(np.nan_to_num(tpr[order], nan=0.0))
        keep = np.concatenate((x[1:] != x[:-1], [True]))
        x, y = x[keep], y[keep]
        rise = np.concatenate(([True], y[1:] > y[:-1]))
        rocs[classes[c]] = (x[rise], y[rise])
    evaluated = [c for i, c in enumerate(classes) if n_gt[i] > 0]      # like psds_eval, a class is defined by its ground truth
    if not evaluated:
        raise ValueError("the evaluated clips hold no ground-truth event")
    axis = np.unique(np.concatenate([rocs[c][0] for c in evaluated]))
    Y = np.stack([PSDSEval._step_curve(axis, *rocs[c]) for c in evaluated])
    with np.errstate(invalid="ignore"):
        etpr = np.nan_to_num(np.nanmean(Y, 0) - alpha_st * np.nanstd(Y, 0), nan=0.0)
    etpr[etpr < 0] = 0.0
    value = PSDSEval._auc(axis, etpr, max_efpr, True) / max_efpr
    single = {c: PSDSEval._auc(axis, Y[i], max_efpr, True) / max_efpr for i, c in enumerate(evaluated)}
    return value, single, (axis, etpr), rocs


def write_psd_roc(path, psd_roc):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pd.DataFrame({"efpr": psd_roc[0], "etpr": psd_roc[1]}).to_csv(path, sep="\t", index=False)
