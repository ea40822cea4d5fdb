"""Event-based and segment-based sound-event-detection metrics -- restatement of the `sed_eval` package.

The reference calls `sed_eval.sound_event.EventBasedMetrics` / `SegmentBasedMetrics` from
desed_task/evaluation/evaluation_measures.py:50-150; the package (setup.py:15, `sed_eval>=0.2.1`) is a third-party dependency
absent from the reference tree and from this image.  This module restates its published algorithm (Mesaros, Heittola,
Virtanen, "Metrics for polyphonic sound event detection", Applied Sciences 6(6), 2016) behind the same class surface:
`Metric(event_label_list=, ...)`, `.evaluate(reference_event_list=, estimated_event_list=)` once per file with lists of
`{"filename", "onset", "offset", "event_label"}` dicts (dicts without a label = "file without events"), `.results()`.

Parity is pinned by the reference's own golden reports (PSDS_Eval/meta/metrics_test/student/{event,segment}_f1.txt via
tests/golden/psds_eval_meta.npz): overall, class-wise-average and all ten per-class rows, tests/test_evaluation.py.

Event-based (onset + offset): a system event matches a reference event of the same label when
|onset_ref - onset_sys| <= t_collar and |offset_ref - offset_sys| <= max(t_collar, percentage_of_length * length_ref);
true positives = size of a MAXIMUM bipartite matching of that relation (sed_eval's `event_matching_type="optimal"`);
substitutions (overall only) = greedy pairing of the left-over events on the time conditions alone.
Segment-based: both event lists are rasterised on a `time_resolution` grid (onset floored, offset ceiled) and compared
cell by cell.
"""
import math

import numpy as np


def _labelled(event_list):
    out = []
    for e in event_list or []:
        label = e.get("event_label")
        if label is None or (isinstance(label, float) and math.isnan(label)):
            continue
        out.append(e)
    return out


def _max_bipartite_matching(adj, n_right):
    """adj[i] = right nodes reachable from left node i -> size of a maximum matching (augmenting paths)."""
    match_r = [-1] * n_right

    def augment(i, seen):
        for j in adj[i]:
            if not seen[j]:
                seen[j] = True
                if match_r[j] < 0 or augment(match_r[j], seen):
                    match_r[j] = i
                    return True
        return False

    size = 0
    for i in range(len(adj)):
        if adj[i] and augment(i, [False] * n_right):
            size += 1
    return size, match_r


def _f_measure(ntp, nref, nsys, empty_system_output_handling=None):
    if nsys > 0:
        precision = ntp / nsys
    else:
        precision = 0.0 if empty_system_output_handling == "zero_score" else float("nan")
    recall = ntp / nref if nref > 0 else float("nan")
    if math.isnan(precision) or math.isnan(recall):
        f = float("nan")
    elif precision + recall == 0:
        f = 0.0
    else:
        f = 2 * precision * recall / (precision + recall)
    return {"f_measure": f, "precision": precision, "recall": recall}


def _nanmean(values):
    v = np.asarray(values, np.float64)
    return float(np.nanmean(v)) if np.isfinite(v).any() else float("nan")


class _Metrics:
    def __init__(self, event_label_list):
        self.event_label_list = list(event_label_list)
        self.evaluated_files = 0
        self.evaluated_length = 0.0

    def results(self):
        return {"overall": self.results_overall_metrics(), "class_wise": self.results_class_wise_metrics(),
                "class_wise_average": self.results_class_wise_average_metrics()}

    def results_class_wise_average_metrics(self):
        cw = self.results_class_wise_metrics()
        out = {}
        for group in next(iter(cw.values())).keys() if cw else []:
            if group == "count":
                continue
            out[group] = {k: _nanmean([cw[c][group][k] for c in cw]) for k in next(iter(cw.values()))[group]}
        return out

    def __str__(self):
        r = self.results()
        lines = [self.title, "=" * 40, f"  Evaluated length                  : {self.evaluated_length:.2f} sec",
                 f"  Evaluated files                   : {self.evaluated_files} ", ""]
        for name, key in (("Overall metrics (micro-average)", "overall"), ("Class-wise average metrics (macro-average)", "class_wise_average")):
            lines += [f"  {name}", "  " + "=" * 38]
            for group, vals in r[key].items():
                lines.append(f"  {group}")
                for k, v in vals.items():
                    lines.append(f"    {k:<32}: {v * 100:.2f} %" if group in ("f_measure", "accuracy") else f"    {k:<32}: {v:.2f} ")
            lines.append("")
        lines += ["  Class-wise metrics", "  " + "=" * 38,
                  "    Event label  | Nref    Nsys  | F        Pre      Rec    | ER       Del      Ins    |"]
        for c, v in r["class_wise"].items():
            lines.append(f"    {c[:12]:<12} | {int(v['count']['Nref']):<7} {int(v['count']['Nsys']):<5} | "
                         f"{v['f_measure']['f_measure'] * 100:<5.1f}%   {v['f_measure']['precision'] * 100:<5.1f}%   "
                         f"{v['f_measure']['recall'] * 100:<5.1f}%  | {v['error_rate']['error_rate']:<8.2f} "
                         f"{v['error_rate']['deletion_rate']:<8.2f} {v['error_rate']['insertion_rate']:<6.2f} |")
        return "\n".join(lines) + "\n"


class EventBasedMetrics(_Metrics):
    title = "Event based metrics (onset-offset)"

    def __init__(self, event_label_list, evaluate_onset=True, evaluate_offset=True, t_collar=0.200, percentage_of_length=0.5,
                 event_matching_type="optimal", empty_system_output_handling=None, **kwargs):
        super().__init__(event_label_list)
        if event_matching_type != "optimal":
            raise NotImplementedError("only sed_eval's default 'optimal' event matching is restated")
        self.evaluate_onset, self.evaluate_offset = evaluate_onset, evaluate_offset
        self.t_collar, self.percentage_of_length = t_collar, percentage_of_length
        self.empty_system_output_handling = empty_system_output_handling
        self.overall = dict(Nref=0.0, Nsys=0.0, Nsubs=0.0, Ntp=0.0, Nfp=0.0, Nfn=0.0)
        self.class_wise = {c: dict(Nref=0.0, Nsys=0.0, Ntp=0.0, Nfp=0.0, Nfn=0.0) for c in self.event_label_list}

    def _time_match(self, ref, est):
        ok = True
        if self.evaluate_onset:
            ok = math.fabs(ref["onset"] - est["onset"]) <= self.t_collar
        if ok and self.evaluate_offset:
            length = ref["offset"] - ref["onset"]
            ok = math.fabs(ref["offset"] - est["offset"]) <= max(self.t_collar, self.percentage_of_length * length)
        return ok

    def evaluate(self, reference_event_list, estimated_event_list):
        ref, est = _labelled(reference_event_list), _labelled(estimated_event_list)
        self.evaluated_files += 1
        self.evaluated_length += max([e["offset"] for e in ref], default=0.0)
        nref, nsys = len(ref), len(est)
        time_ok = [[self._time_match(r, e) for e in est] for r in ref]
        adj = [[i for i in range(nsys) if time_ok[j][i] and ref[j]["event_label"] == est[i]["event_label"]] for j in range(nref)]
        ntp, match_r = _max_bipartite_matching(adj, nsys)
        sys_used = [m >= 0 for m in match_r]
        ref_used = [False] * nref
        for i, j in enumerate(match_r):
            if j >= 0:
                ref_used[j] = True
        nsubs = 0
        for j in range(nref):
            if ref_used[j]:
                continue
            for i in range(nsys):
                if not sys_used[i] and time_ok[j][i]:
                    sys_used[i] = True
                    nsubs += 1
                    break
        o = self.overall
        o["Nref"] += nref; o["Nsys"] += nsys; o["Ntp"] += ntp; o["Nsubs"] += nsubs
        o["Nfp"] += nsys - ntp - nsubs; o["Nfn"] += nref - ntp - nsubs
        for c in self.event_label_list:
            cref = [j for j in range(nref) if ref[j]["event_label"] == c]
            cest = [i for i in range(nsys) if est[i]["event_label"] == c]
            pos = {i: k for k, i in enumerate(cest)}
            ctp, _ = _max_bipartite_matching([[pos[i] for i in adj[j]] for j in cref], len(cest))
            w = self.class_wise[c]
            w["Nref"] += len(cref); w["Nsys"] += len(cest); w["Ntp"] += ctp
            w["Nfp"] += len(cest) - ctp; w["Nfn"] += len(cref) - ctp

    def results_overall_metrics(self):
        o = self.overall
        nref = o["Nref"]
        div = (lambda a: a / nref) if nref > 0 else (lambda a: float("nan"))
        s, d, i = div(o["Nsubs"]), div(o["Nfn"]), div(o["Nfp"])
        return {"f_measure": _f_measure(o["Ntp"], o["Nref"], o["Nsys"], self.empty_system_output_handling),
                "error_rate": {"error_rate": s + d + i, "substitution_rate": s, "deletion_rate": d, "insertion_rate": i}}

    def results_class_wise_metrics(self):
        out = {}
        for c, w in self.class_wise.items():
            nref = w["Nref"]
            d = w["Nfn"] / nref if nref > 0 else float("nan")
            i = w["Nfp"] / nref if nref > 0 else float("nan")
            out[c] = {"count": {"Nref": w["Nref"], "Nsys": w["Nsys"]},
                      "f_measure": _f_measure(w["Ntp"], w["Nref"], w["Nsys"], self.empty_system_output_handling),
                      "error_rate": {"error_rate": d + i, "deletion_rate": d, "insertion_rate": i}}
        return out


class SegmentBasedMetrics(_Metrics):
    title = "Segment based metrics"

    def __init__(self, event_label_list, time_resolution=1.0, **kwargs):
        super().__init__(event_label_list)
        self.time_resolution = time_resolution
        z = lambda: dict(Ntp=0.0, Ntn=0.0, Nfp=0.0, Nfn=0.0, Nref=0.0, Nsys=0.0)
        self.overall = dict(z(), ER=0.0, S=0.0, D=0.0, I=0.0)
        self.class_wise = {c: z() for c in self.event_label_list}

    def _roll(self, events, n_seg):
        idx = {c: i for i, c in enumerate(self.event_label_list)}
        roll = np.zeros((n_seg, len(self.event_label_list)), np.int64)
        for e in events:
            on = int(math.floor(e["onset"] * 1 / self.time_resolution))
            off = int(math.ceil(e["offset"] * 1 / self.time_resolution))
            roll[on:off, idx[e["event_label"]]] = 1
        return roll

    def evaluate(self, reference_event_list, estimated_event_list, evaluated_length_seconds=None):
        ref, est = _labelled(reference_event_list), _labelled(estimated_event_list)
        self.evaluated_files += 1
        n = lambda ev: int(math.ceil(max([e["offset"] for e in ev], default=0.0) * 1 / self.time_resolution))
        n_seg = max(n(ref), n(est))
        if evaluated_length_seconds is not None:
            n_seg = max(n_seg, int(math.ceil(evaluated_length_seconds / self.time_resolution)))
        self.evaluated_length += n_seg * self.time_resolution if evaluated_length_seconds is None else evaluated_length_seconds
        r, s = self._roll(ref, n_seg), self._roll(est, n_seg)
        tp, tn, fp, fn = (r + s > 1), (r + s == 0), (s - r > 0), (r - s > 0)
        o = self.overall
        ntp_seg, nref_seg, nsys_seg = tp.sum(1), r.sum(1), s.sum(1)
        o["Ntp"] += tp.sum(); o["Ntn"] += tn.sum(); o["Nfp"] += fp.sum(); o["Nfn"] += fn.sum()
        o["Nref"] += r.sum(); o["Nsys"] += s.sum()
        o["S"] += (np.minimum(nref_seg, nsys_seg) - ntp_seg).sum()
        o["D"] += np.maximum(0, nref_seg - nsys_seg).sum()
        o["I"] += np.maximum(0, nsys_seg - nref_seg).sum()
        o["ER"] += (np.maximum(nref_seg, nsys_seg) - ntp_seg).sum()
        for k, c in enumerate(self.event_label_list):
            w = self.class_wise[c]
            w["Ntp"] += tp[:, k].sum(); w["Ntn"] += tn[:, k].sum(); w["Nfp"] += fp[:, k].sum(); w["Nfn"] += fn[:, k].sum()
            w["Nref"] += r[:, k].sum(); w["Nsys"] += s[:, k].sum()

    @staticmethod
    def _accuracy(w):
        sens = w["Ntp"] / (w["Ntp"] + w["Nfn"]) if w["Ntp"] + w["Nfn"] > 0 else float("nan")
        spec = w["Ntn"] / (w["Ntn"] + w["Nfp"]) if w["Ntn"] + w["Nfp"] > 0 else float("nan")
        tot = w["Ntp"] + w["Ntn"] + w["Nfp"] + w["Nfn"]
        return {"sensitivity": sens, "specificity": spec, "balanced_accuracy": (sens + spec) / 2,
                "accuracy": (w["Ntp"] + w["Ntn"]) / tot if tot > 0 else float("nan")}

    def results_overall_metrics(self):
        o = self.overall
        nref = o["Nref"]
        div = (lambda a: a / nref) if nref > 0 else (lambda a: float("nan"))
        return {"f_measure": _f_measure(o["Ntp"], o["Nref"], o["Nsys"]),
                "error_rate": {"error_rate": div(o["ER"]), "substitution_rate": div(o["S"]), "deletion_rate": div(o["D"]),
                               "insertion_rate": div(o["I"])},
                "accuracy": self._accuracy(o)}

    def results_class_wise_metrics(self):
        out = {}
        for c, w in self.class_wise.items():
            nref = w["Nref"]
            d = w["Nfn"] / nref if nref > 0 else float("nan")
            i = w["Nfp"] / nref if nref > 0 else float("nan")
            out[c] = {"count": {"Nref": w["Nref"], "Nsys": w["Nsys"]}, "f_measure": _f_measure(w["Ntp"], w["Nref"], w["Nsys"]),
                      "error_rate": {"error_rate": d + i, "deletion_rate": d, "insertion_rate": i},
                      "accuracy": self._accuracy(w)}
        return out
