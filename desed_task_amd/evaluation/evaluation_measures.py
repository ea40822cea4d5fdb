"""Host-side mirror of desed_task/evaluation/evaluation_measures.py: same function names, arguments and
return values, on top of this package's restatements of the two third-party evaluators (`psds.PSDSEval`,
`sed_eval_metrics.{EventBasedMetrics,SegmentBasedMetrics}`) instead of the absent `psds_eval` / `sed_eval` packages.

The threshold-free PSDS (`compute_psds_from_scores`, sed_scores_eval's role) is psds_scores.py's own exact algorithm.

Known answers (tests/test_evaluation.py, from the reference's PSDS_Eval/meta golden data): event-based F1 39.83 % macro /
40.92 % micro, segment-based 69.35 % / 75.47 %, intersection F1 63.74 %, PSDS1 0.334, PSDS2 0.533.
"""
import os

import numpy as np
import pandas as pd

from .psds import PSDSEval, PSDSEvalError
from .sed_eval_metrics import EventBasedMetrics, SegmentBasedMetrics


def get_event_list_current_file(df, fname):
    """Events of one file as a list of dicts; a file whose only row has no label gives [{"filename": fname}]
    (evaluation_measures.py:11-28)."""
    event_file = df[df["filename"] == fname]
    if len(event_file) == 1 and pd.isna(event_file["event_label"].iloc[0]):
        return [{"filename": fname}]
    return event_file.to_dict("records")


def _events_by_file(df):
    """{filename: [event dicts]} in one pass (the reference filters the whole frame once per file)."""
    out = {}
    for rec in df.to_dict("records"):
        out.setdefault(rec["filename"], []).append(rec)
    return out


def psds_results(psds_obj):
    """Prints the three diagnostic PSD scores (evaluation_measures.py:32-47)."""
    try:
        for a_ct, a_st in ((0, 0), (1, 0), (0, 1)):
            score = psds_obj.psds(alpha_ct=a_ct, alpha_st=a_st, max_efpr=100)
            print(f"\nPSD-Score ({a_ct}, {a_st}, 100): {score.value:.5f}")
    except PSDSEvalError:
        print("psds did not work ....")
        raise EnvironmentError


def _classes(reference, estimated):
    classes = []
    classes.extend(reference.event_label.dropna().unique())
    classes.extend(estimated.event_label.dropna().unique())
    return sorted(set(classes))


def _evaluate_files(metric, reference, estimated):
    ref_by_file, est_by_file = _events_by_file(reference), _events_by_file(estimated)
    for fname in reference["filename"].unique():
        metric.evaluate(reference_event_list=ref_by_file.get(fname, []), estimated_event_list=est_by_file.get(fname, []))
    return metric


def event_based_evaluation_df(reference, estimated, t_collar=0.200, percentage_of_length=0.2):
    """Event-based metrics over the files of `reference` (evaluation_measures.py:50-93)."""
    metric = EventBasedMetrics(event_label_list=_classes(reference, estimated), t_collar=t_collar,
                               percentage_of_length=percentage_of_length, empty_system_output_handling="zero_score")
    return _evaluate_files(metric, reference, estimated)


def segment_based_evaluation_df(reference, estimated, time_resolution=1.0):
    """Segment-based metrics over the files of `reference` (evaluation_measures.py:96-132)."""
    metric = SegmentBasedMetrics(event_label_list=_classes(reference, estimated), time_resolution=time_resolution)
    return _evaluate_files(metric, reference, estimated)


def compute_sed_eval_metrics(predictions, groundtruth):
    """(event-based, segment-based) metrics with the task's parameters (evaluation_measures.py:135-150)."""
    return (event_based_evaluation_df(groundtruth, predictions, t_collar=0.200, percentage_of_length=0.2),
            segment_based_evaluation_df(groundtruth, predictions, time_resolution=1.0))


def log_sedeval_metrics(predictions, ground_truth, save_dir=None):
    """(event macro-F1, event micro-F1, segment macro-F1, segment micro-F1); `ground_truth` is a TSV path
    (recipes/dcase2023_task4_baseline/local/utils.py:97-127)."""
    if predictions.empty:
        return 0.0, 0.0, 0.0, 0.0
    gt = pd.read_csv(ground_truth, sep="\t")
    event_res, segment_res = compute_sed_eval_metrics(predictions, gt)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "event_f1.txt"), "w") as f:
            f.write(str(event_res))
        with open(os.path.join(save_dir, "segment_f1.txt"), "w") as f:
            f.write(str(segment_res))
    e, s = event_res.results(), segment_res.results()
    return (e["class_wise_average"]["f_measure"]["f_measure"], e["overall"]["f_measure"]["f_measure"],
            s["class_wise_average"]["f_measure"]["f_measure"], s["overall"]["f_measure"]["f_measure"])


def _read(table):
    return table if isinstance(table, pd.DataFrame) else pd.read_csv(table, sep="\t")


def compute_per_intersection_macro_f1(prediction_dfs, ground_truth_file, durations_file, dtc_threshold=0.5,
                                      gtc_threshold=0.5, cttc_threshold=0.3):
    """Mean over the thresholds of `prediction_dfs` of the intersection-based macro F1 (evaluation_measures.py:153-195).
    The two tables may be TSV paths (as in the reference) or DataFrames."""
    psds = PSDSEval(ground_truth=_read(ground_truth_file), metadata=_read(durations_file), dtc_threshold=dtc_threshold,
                    gtc_threshold=gtc_threshold, cttc_threshold=cttc_threshold)
    psds_macro_f1 = []
    for threshold in prediction_dfs.keys():
        if not prediction_dfs[threshold].empty:
            threshold_f1, _ = psds.compute_macro_f_score(prediction_dfs[threshold])
        else:
            threshold_f1 = 0
        if np.isnan(threshold_f1):
            threshold_f1 = 0.0
        psds_macro_f1.append(threshold_f1)
    return np.mean(psds_macro_f1)


def compute_psds_from_operating_points(prediction_dfs, ground_truth_file, durations_file, dtc_threshold=0.5,
                                       gtc_threshold=0.5, cttc_threshold=0.3, alpha_ct=0, alpha_st=0, max_efpr=100,
                                       save_dir=None):
    """PSDS of the operating points `prediction_dfs` ({threshold: detections}) (evaluation_measures.py:198-255).  With
    `save_dir` the per-threshold predictions are written as in the reference; the PSD-ROC is stored as a TSV (efpr, etpr)
    instead of a matplotlib figure."""
    psds_eval = PSDSEval(ground_truth=_read(ground_truth_file), metadata=_read(durations_file), dtc_threshold=dtc_threshold,
                         gtc_threshold=gtc_threshold, cttc_threshold=cttc_threshold)
    for i, k in enumerate(prediction_dfs.keys()):
        psds_eval.add_operating_point(prediction_dfs[k], info={"name": f"Op {i + 1:02d}", "threshold": k})
    psds_score = psds_eval.psds(alpha_ct=alpha_ct, alpha_st=alpha_st, max_efpr=max_efpr)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        pred_dir = os.path.join(save_dir, f"predictions_dtc{dtc_threshold}_gtc{gtc_threshold}_cttc{cttc_threshold}")
        os.makedirs(pred_dir, exist_ok=True)
        for k in prediction_dfs.keys():
            prediction_dfs[k].to_csv(os.path.join(pred_dir, f"predictions_th_{k:.2f}.tsv"), sep="\t", index=False)
        filename = (f"PSDS_dtc{dtc_threshold}_gtc{gtc_threshold}_cttc{cttc_threshold}"
                    f"_ct{alpha_ct}_st{alpha_st}_max{max_efpr}_psds_eval.tsv")
        pd.DataFrame({"efpr": psds_score.plt.xp, "etpr": psds_score.plt.yp}).to_csv(os.path.join(save_dir, filename), sep="\t",
                                                                                    index=False)
    return psds_score.value


def compute_psds_from_scores(scores, ground_truth_file, durations_file, dtc_threshold=0.5, gtc_threshold=0.5,
                             cttc_threshold=0.3, alpha_ct=0, alpha_st=0, max_efpr=100, num_jobs=4, save_dir=None):
    """Threshold-free PSDS of the score tables `scores` ({audio_id: DataFrame}) (evaluation_measures.py:258-304).  The ground
    truth / durations may be the dicts of `read_ground_truth_events` / `read_audio_durations` (as the reference passes them)
    or TSV paths.  Computed by psds_scores.psds_from_scores -- an exact all-thresholds PSD-ROC with psds.py's criteria, NOT a
    restatement of sed_scores_eval's code (parity with that package unpinned, see psds_scores.py).  `num_jobs` is accepted and
    unused; with `save_dir` the score tables and the PSD-ROC are written as TSV files."""
    from .psds_scores import psds_from_scores, read_audio_durations, read_ground_truth_events, write_psd_roc
    psds, _, psd_roc, _ = psds_from_scores(scores, read_ground_truth_events(ground_truth_file), read_audio_durations(durations_file),
                                           dtc_threshold=dtc_threshold, gtc_threshold=gtc_threshold, cttc_threshold=cttc_threshold,
                                           alpha_ct=alpha_ct, alpha_st=alpha_st, max_efpr=max_efpr)
    if save_dir is not None:
        from ..postprocess import write_sed_scores
        write_sed_scores(scores, os.path.join(save_dir, "scores"))
        write_psd_roc(os.path.join(save_dir, f"PSDS_dtc{dtc_threshold}_gtc{gtc_threshold}_cttc{cttc_threshold}"
                                             f"_ct{alpha_ct}_st{alpha_st}_max{max_efpr}_sed_scores_eval.tsv"), psd_roc)
    return psds
