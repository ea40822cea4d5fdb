"""Polyphonic Sound Detection Score -- restatement of the `psds_eval` package the reference evaluates with.

The reference calls `psds_eval.PSDSEval` from desed_task/evaluation/evaluation_measures.py:153-255 (and the notebook
PSDS_Eval/PSDS_Evaluation.ipynb); the package itself (setup.py:14, `psds_eval>=0.4.0`) is a third-party dependency that is
absent from the reference tree and from this image.  This module restates its published algorithm (Bilen et al., "A framework
for the robust evaluation of sound event detection", ICASSP 2020) with the class surface the reference uses:
`PSDSEval(ground_truth=, metadata=, dtc_threshold=, gtc_threshold=, cttc_threshold=)`, `.add_operating_point(det, info=)`,
`.psds(alpha_ct=, alpha_st=, max_efpr=) -> PSDS(value, plt, ...)`, `.psd_roc_curves(alpha_ct)`, `.compute_macro_f_score(det)`.

Parity is pinned by the reference's own golden data (tests/golden/psds_eval_meta.npz, built from PSDS_Eval/meta): PSDS
scenario 1 = 0.334, scenario 2 = 0.533 and intersection-F1 = 63.74 % for the student predictions, tests/test_evaluation.py.

Algorithm (per operating point = one table of detections):
  1. every detection is intersected with every ground-truth event of the same file, plus one injected WORLD event per file
     spanning [0, duration];
  2. DTC: a detection is *relevant* when the summed intersection with same-class ground truths is >= dtc * its duration;
  3. GTC: a ground truth is a TRUE POSITIVE when the summed intersection with relevant detections is >= gtc * its duration;
  4. CTTC: a non-relevant detection is a CROSS TRIGGER on class k (or a FALSE POSITIVE for k = WORLD) when its summed
     intersection with class-k ground truths is >= cttc * its duration;
  5. rates: TPR_c = TP_c / N_gt,c;  FPR_c = FP_c / T_dataset;  CTR_c,k = CT_c,k / T_gt,k  (per `duration_unit`);
     eFPR_c = FPR_c + alpha_ct * (1/|C|) * sum_{k != c} CTR_c,k  (the mean runs over all |C| class columns with a zero
     diagonal -- not the paper's |C|-1; the golden PSDS2 discriminates: 0.5327 -> "0.533" this way, 0.5281 with |C|-1);
  6. per-class PSD-ROC = monotone staircase through the (eFPR_c, TPR_c) points of all operating points; PSDS = area under
     mean_c(TPR) - alpha_st * std_c(TPR) up to max_efpr, divided by max_efpr.

Everything is vectorised numpy over (detection, ground-truth) pairs -- no per-file pandas merges: the 50-point PSDS of the
golden validation set (175 k detections) takes ~0.3 s where psds_eval's merge/groupby pipeline needs minutes.
"""
from collections import namedtuple

import numpy as np
import pandas as pd

WORLD = "injected_psds_world_label"

PSDROC = namedtuple("PSDROC", ["xp", "yp", "std", "mean"])
PSDS = namedtuple("PSDS", ["value", "plt", "alpha_st", "alpha_ct", "max_efpr", "duration_unit"])
Thresholds = namedtuple("Thresholds", ["dtc", "gtc", "cttc"])


class PSDSEvalError(ValueError):
    """Raised on malformed inputs (same role as psds_eval.psds.PSDSEvalError, caught at evaluation_measures.py:45)."""


class PSDSEval:
    secs_in_uot = {"minute": 60, "hour": 3600}
    detection_cols = ["filename", "onset", "offset", "event_label"]

    def __init__(self, dtc_threshold=0.5, gtc_threshold=0.5, cttc_threshold=0.3, **kwargs):
        for name, v in (("dtc_threshold", dtc_threshold), ("gtc_threshold", gtc_threshold), ("cttc_threshold", cttc_threshold)):
            if not 0.0 <= v <= 1.0:
                raise PSDSEvalError(f"{name} must be between 0 and 1")
        self.duration_unit = kwargs.get("duration_unit", "hour")
        if self.duration_unit not in self.secs_in_uot:
            raise PSDSEvalError("Invalid duration_unit specified")
        self.nseconds = self.secs_in_uot[self.duration_unit]
        self.threshold = Thresholds(dtc=dtc_threshold, gtc=gtc_threshold, cttc=cttc_threshold)
        self.class_names = []
        self.ground_truth = None
        self.metadata = None
        self._ops = []          # dicts: id, counts, tpr, fpr, ctr, info
        gt, meta = kwargs.get("ground_truth"), kwargs.get("metadata")
        if gt is not None or meta is not None:
            self.set_ground_truth(gt, meta)

    # ------------------------------------------------------------------ tables
    @classmethod
    def _validate(cls, df, name, cols):
        if not isinstance(df, pd.DataFrame):
            raise PSDSEvalError(f"The {name} data must be provided in a pandas.DataFrame")
        missing = [c for c in cols if c not in df.columns]
        if missing:
            raise PSDSEvalError(f"The {name} data columns need to match the following {cols}")

    def set_ground_truth(self, gt_t, meta_t):
        if self.ground_truth is not None or self.metadata is not None:
            raise PSDSEvalError("You cannot set the ground truth more than once per evaluation")
        if gt_t is None or meta_t is None:
            raise PSDSEvalError("The ground truth cannot be set without data")
        self._validate(gt_t, "ground truth", self.detection_cols)
        self._validate(meta_t, "metadata", ["filename", "duration"])
        meta = meta_t.drop_duplicates("filename", keep="first")
        gt = gt_t[self.detection_cols].dropna()
        if (gt.offset < gt.onset).any():
            raise PSDSEvalError("The ground truth dataframe provided has events with offset before onset")
        self._files = {f: i for i, f in enumerate(meta.filename)}
        self._file_dur = meta.duration.to_numpy(np.float64)
        labels = sorted(set(gt.event_label))
        if WORLD in labels:
            raise PSDSEvalError("The ground truth uses the reserved WORLD label")
        self.class_names = labels + [WORLD]
        self._cls = {c: i for i, c in enumerate(self.class_names)}
        nc = len(self.class_names)
        known = gt.filename.isin(self._files)
        if not known.all():
            raise PSDSEvalError("The ground truth contains files that are missing from the metadata")
        # ground-truth table = labelled events in input order, then one WORLD event per file (psds_eval appends them too)
        g_file = np.concatenate([gt.filename.map(self._files).to_numpy(np.int64), np.arange(len(meta), dtype=np.int64)])
        g_on = np.concatenate([gt.onset.to_numpy(np.float64), np.zeros(len(meta))])
        g_off = np.concatenate([gt.offset.to_numpy(np.float64), self._file_dur])
        g_lab = np.concatenate([gt.event_label.map(self._cls).to_numpy(np.int64), np.full(len(meta), nc - 1, np.int64)])
        order = np.argsort(g_file, kind="stable")
        self._g = dict(file=g_file[order], on=g_on[order], off=g_off[order], lab=g_lab[order])
        self._g["dur"] = self._g["off"] - self._g["on"]
        self._g_count = np.bincount(self._g["file"], minlength=len(meta))
        self._g_start = np.cumsum(self._g_count) - self._g_count
        self._n_gt = np.bincount(self._g["lab"], minlength=nc).astype(np.float64)          # events per class (WORLD: files)
        self._gt_dur = np.bincount(self._g["lab"], weights=self._g["dur"], minlength=nc)   # seconds per class (WORLD: dataset)
        self.ground_truth = gt
        self.metadata = meta

    def _init_det_table(self, det_t):
        if isinstance(det_t, pd.DataFrame) and det_t.empty:       # a system that detected nothing is a legal operating point
            det_t = pd.DataFrame({c: [] for c in self.detection_cols})
        self._validate(det_t, "detection", self.detection_cols)
        det = det_t[self.detection_cols].dropna()
        if (det.offset < det.onset).any():
            raise PSDSEvalError("The detection dataframe provided has events with offset before onset")
        known = det.event_label.isin(self.class_names[:-1])
        if not known.all():
            # the score is defined over the ground-truth classes; detections of any other label belong to no evaluated class
            import warnings
            warnings.warn(f"detections of labels absent from the ground truth are ignored: {sorted(set(det.event_label[~known]))}")
            det = det[known]
        det = det[det.filename.isin(self._files)]          # a file without metadata has no WORLD event: never counted
        d_file = det.filename.map(self._files).to_numpy(np.int64)
        order = np.argsort(d_file, kind="stable")
        return dict(file=d_file[order], on=det.onset.to_numpy(np.float64)[order], off=det.offset.to_numpy(np.float64)[order],
                    lab=det.event_label.map(self._cls).to_numpy(np.int64)[order])

    # ------------------------------------------------------------------ one operating point
    def _evaluate_detections(self, d):
        """-> counts (nc, nc) [detected class, ground-truth class; last column = WORLD], tp_ratio, fp_rate, ct_rate."""
        if self.ground_truth is None:
            raise PSDSEvalError("Ground Truth must be provided before adding the first operating point")
        g, nc = self._g, len(self.class_names)
        nd = len(d["file"])
        counts = np.zeros((nc, nc))
        if nd:
            rep = self._g_count[d["file"]]
            pd_ = np.repeat(np.arange(nd), rep)                                  # pair -> detection
            first = np.cumsum(rep) - rep
            pg = self._g_start[d["file"]][pd_] + (np.arange(len(pd_)) - first[pd_])   # pair -> ground truth
            inter = np.minimum(d["off"][pd_], g["off"][pg]) - np.maximum(d["on"][pd_], g["on"][pg])
            keep = inter > 0
            pd_, pg, inter = pd_[keep], pg[keep], inter[keep]
            d_dur = d["off"] - d["on"]
            det_precision = inter / d_dur[pd_]
            gt_coverage = inter / g["dur"][pg]
            same = d["lab"][pd_] == g["lab"][pg]
            # DTC
            dtc_sum = np.bincount(pd_[same], weights=det_precision[same], minlength=nd)
            has_same = np.bincount(pd_[same], minlength=nd) > 0
            relevant = has_same & (dtc_sum >= self.threshold.dtc)
            # GTC
            sel = same & relevant[pd_]
            gtc_sum = np.bincount(pg[sel], weights=gt_coverage[sel], minlength=len(g["file"]))
            gt_hit = (np.bincount(pg[sel], minlength=len(g["file"])) > 0) & (gtc_sum >= self.threshold.gtc)
            tp_per_class = np.bincount(g["lab"][gt_hit], minlength=nc)
            counts[np.arange(nc), np.arange(nc)] = tp_per_class
            # CTTC on the detections that failed the DTC (WORLD column = plain false positives)
            sel = ~same & ~relevant[pd_]
            key = pd_[sel] * nc + g["lab"][pg[sel]]
            ukey, inv = np.unique(key, return_inverse=True)
            ct_sum = np.bincount(inv, weights=det_precision[sel], minlength=len(ukey))
            hit = ukey[ct_sum >= self.threshold.cttc]
            np.add.at(counts, (d["lab"][hit // nc], hit % nc), 1)
        with np.errstate(divide="ignore", invalid="ignore"):
            tp_ratio = np.diag(counts)[:-1] / self._n_gt[:-1]
            fp_rate = counts[:-1, -1] * self.nseconds / self._gt_dur[-1]
            ct_rate = counts[:-1, :-1] * self.nseconds / self._gt_dur[None, :-1]
        ct_rate[np.arange(nc - 1), np.arange(nc - 1)] = 0.0
        return counts, tp_ratio, fp_rate, ct_rate

    @staticmethod
    def _op_id(d):
        import hashlib
        h = hashlib.sha256()
        order = np.lexsort((d["lab"], d["off"], d["on"], d["file"]))
        for k in ("file", "on", "off", "lab"):
            h.update(np.ascontiguousarray(d[k][order]).tobytes())
        return h.hexdigest()

    def add_operating_point(self, detections, info=None):
        d = self._init_det_table(detections)
        op_id = self._op_id(d)
        if any(op["id"] == op_id for op in self._ops):
            import warnings
            warnings.warn("A similar operating point exists, skipping this one")
            return
        counts, tpr, fpr, ctr = self._evaluate_detections(d)
        self._ops.append(dict(id=op_id, counts=counts, tpr=tpr, fpr=fpr, ctr=ctr, info=dict(info or {})))

    def num_operating_points(self):
        return len(self._ops)

    def clear_all_operating_points(self):
        self._ops = []

    @property
    def operating_points(self):
        return pd.DataFrame([{**{k: op[k] for k in ("id", "counts", "tpr", "fpr", "ctr")}, **op["info"]} for op in self._ops])

    # ------------------------------------------------------------------ curves and score
    @staticmethod
    def _step_curve(x, xp, yp):
        """Monotone staircase through (xp, yp) sampled at x: best yp among the points with xp <= x, 0 left of the first."""
        order = np.lexsort((yp, xp))
        xs, ys = xp[order], np.maximum.accumulate(np.nan_to_num(yp[order], nan=0.0))
        idx = np.searchsorted(xs, x, side="right") - 1
        return np.where(idx >= 0, ys[np.maximum(idx, 0)], 0.0)

    def _rates(self):
        if not self._ops:
            raise PSDSEvalError("No operating points have been added")
        tpr = np.stack([op["tpr"] for op in self._ops], 1)       # (classes, ops)
        fpr = np.stack([op["fpr"] for op in self._ops], 1)
        ctr = np.stack([op["ctr"] for op in self._ops], 2)       # (classes, classes, ops)
        return tpr, fpr, ctr

    def _effective_fp_rate(self, alpha_ct):
        if alpha_ct < 0 or alpha_ct > 1:
            raise PSDSEvalError("alpha_ct must be between 0 and 1")
        tpr, fpr, ctr = self._rates()
        return fpr + alpha_ct * self._mean_ctr(ctr)

    @staticmethod
    def _mean_ctr(ctr):
        # a class nobody annotated has zero ground-truth duration: its column is 0/0 and is left out of the mean
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return np.nan_to_num(np.nanmean(np.where(np.isfinite(ctr), ctr, np.nan), axis=1), nan=0.0)

    def psd_roc_curves(self, alpha_ct, linear_interp=False):
        """-> (tpr_vs_fpr, tpr_vs_ctr, tpr_vs_efpr): PSDROC tuples whose yp holds one staircase per class on the axis xp."""
        if linear_interp:
            raise NotImplementedError("only the default staircase PSD-ROC is restated")
        tpr, fpr, ctr = self._rates()
        out = []
        for x in (fpr, self._mean_ctr(ctr), self._effective_fp_rate(alpha_ct)):
            axis = np.unique(x[np.isfinite(x)])
            yp = np.stack([self._step_curve(axis, x[c], tpr[c]) for c in range(tpr.shape[0])])
            out.append(PSDROC(xp=axis, yp=yp, std=np.nanstd(yp, axis=0), mean=np.nanmean(yp, axis=0)))
        return tuple(out)

    @staticmethod
    def _auc(x, y, max_x=None, decreasing_y=False):
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        if max_x is None:
            max_x = x.max()
        if not decreasing_y and (np.diff(y) < 0).any():
            raise PSDSEvalError("non-decreasing property not verified for y")
        if max_x not in x:                                  # close the last step at max_x
            i = int(np.searchsorted(x, max_x))
            x = np.insert(x, i, max_x)
            y = np.insert(y, i, y[i - 1] if i > 0 else 0.0)
        valid = x <= max_x
        return float(np.sum(np.diff(x[valid]) * y[valid][:-1]))

    def psds(self, alpha_ct=0.0, alpha_st=0.0, max_efpr=None, en_interp=False):
        if alpha_st < 0:
            raise PSDSEvalError("alpha_st can't be negative")
        _, _, roc = self.psd_roc_curves(alpha_ct, en_interp)
        if max_efpr is None:
            max_efpr = float(np.max(roc.xp))
        etpr = np.nan_to_num(roc.mean - alpha_st * roc.std, nan=0.0)
        etpr[etpr < 0] = 0.0
        value = self._auc(roc.xp, etpr, max_efpr, alpha_st > 0) / max_efpr
        return PSDS(value=value, plt=PSDROC(xp=roc.xp, yp=etpr, std=roc.std, mean=roc.mean), alpha_st=alpha_st,
                    alpha_ct=alpha_ct, max_efpr=max_efpr, duration_unit=self.duration_unit)

    # ------------------------------------------------------------------ intersection-based F score
    @staticmethod
    def compute_f_score(tp, fp, fn, beta):
        num = (1 + beta ** 2) * tp
        with np.errstate(divide="ignore", invalid="ignore"):
            return num / (num + beta ** 2 * fn + fp)

    def compute_macro_f_score(self, detections, beta=1.0):
        """-> (macro F, {class: F}).  As in psds_eval the ground-truth count is recovered as TP / TPR, so a class without
        any true positive yields NaN and is skipped by the nan-mean (all classes without TP -> NaN, which the reference
        maps to 0 at evaluation_measures.py:192-193)."""
        counts, tpr, _, _ = self._evaluate_detections(self._init_det_table(detections))
        tp = np.diag(counts)[:-1]
        with np.errstate(divide="ignore", invalid="ignore"):
            n_gt = tp / tpr
        f = self.compute_f_score(tp, counts[:-1, -1], n_gt - tp, beta)
        with np.errstate(invalid="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                f_avg = float(np.nanmean(f))
        return f_avg, {c: float(v) for c, v in zip(self.class_names[:-1], f)}
