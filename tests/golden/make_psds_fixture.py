"""Builds tests/golden/psds_eval_meta.npz from the reference's only golden evaluation DATA.

Run in the build container (needs /root/reference):  python tests/golden/make_psds_fixture.py

Inputs (data files of the reference, re-encoded losslessly -- every time stamp in them has at most three decimals, so it is
stored as an integer number of milliseconds and `ms / 1000.0` reproduces the parsed double exactly; asserted below):
  PSDS_Eval/meta/validation.tsv, validation_durations.tsv                      ground truth + clip durations
  PSDS_Eval/meta/metrics_test/student/predictions0.5.csv                       detections at threshold 0.5
  PSDS_Eval/meta/metrics_test/student/predictions_operating_points/*.tsv       detections at the 50 PSDS thresholds
Expected outputs (the numbers the reference publishes for those inputs):
  PSDS_Eval/meta/metrics_test/student/{event,segment}_f1.txt                   sed_eval reports (overall, macro, per class)
  PSDS_Eval/PSDS_Evaluation.ipynb cell outputs                                 PSDS1 0.334, PSDS2 0.533, intersection F1 63.74 %
"""
import glob
import json
import os
import re

import numpy as np
import pandas as pd

REF = "/root/reference/PSDS_Eval"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "psds_eval_meta.npz")


def to_ms(col):
    ms = np.round(col.to_numpy(np.float64) * 1000.0).astype(np.int32)
    assert np.array_equal(ms / 1000.0, col.to_numpy(np.float64)), "a time stamp is not an exact millisecond count"
    return ms


def parse_report(path):
    """sed_eval text report -> {"overall": {...}, "macro": {...}, "classes": {label_prefix: [numbers...]}}."""
    txt = open(path).read()
    sec = {"overall": {}, "macro": {}, "classes": {}}
    cur = None
    for line in txt.splitlines():
        if "Overall metrics" in line:
            cur = "overall"
        elif "Class-wise average" in line:
            cur = "macro"
        elif "Class-wise metrics" in line:
            cur = "classes"
        elif cur in ("overall", "macro"):
            m = re.match(r"\s+([A-Za-z][A-Za-z0-9 ()\-]*?)\s+:\s+([0-9.]+)", line)
            if m:
                sec[cur][m.group(1).strip()] = float(m.group(2))
        elif cur == "classes" and "|" in line and not line.strip().startswith(("Event label", "---")):
            cells = [c.strip() for c in line.split("|")]
            nums = [float(x.rstrip("%")) for c in cells[1:] for x in c.split()]
            sec["classes"][cells[0].rstrip(".")] = nums
    return sec


def main():
    meta = os.path.join(REF, "meta")
    dur = pd.read_csv(os.path.join(meta, "validation_durations.tsv"), sep="\t")
    files = list(dur.filename)
    fidx = {f: i for i, f in enumerate(files)}
    gt = pd.read_csv(os.path.join(meta, "validation.tsv"), sep="\t")
    labels = sorted(gt.event_label.dropna().unique())
    lidx = {c: i for i, c in enumerate(labels)}
    gt_nan = gt.onset.isna().to_numpy()
    out = {
        "files": np.array(files),
        "labels": np.array(labels),
        "durations": dur.duration.to_numpy(np.float64),
        # ground truth in file order, rows without an event keep label -1 (clips with no event)
        "gt_file": gt.filename.map(fidx).to_numpy(np.int16),
        "gt_label": np.where(gt_nan, -1, gt.event_label.map(lidx).fillna(-1)).astype(np.int8),
        "gt_onset_ms": np.where(gt_nan, -1, to_ms(gt.onset.fillna(0))).astype(np.int32),
        "gt_offset_ms": np.where(gt_nan, -1, to_ms(gt.offset.fillna(0))).astype(np.int32),
    }
    p05 = pd.read_csv(os.path.join(meta, "metrics_test/student/predictions0.5.csv"), index_col=0)
    out.update(p05_file=p05.filename.map(fidx).to_numpy(np.int16), p05_label=p05.event_label.map(lidx).to_numpy(np.int8),
               p05_onset_ms=to_ms(p05.onset), p05_offset_ms=to_ms(p05.offset))
    ths, f_, l_, on_, off_, start = [], [], [], [], [], [0]
    for path in sorted(glob.glob(os.path.join(meta, "metrics_test/student/predictions_operating_points/*.tsv"))):
        ths.append(float(re.search(r"th_([0-9.]+)\.tsv", path).group(1)))
        p = pd.read_csv(path, sep="\t")
        f_.append(p.filename.map(fidx).to_numpy(np.int16)); l_.append(p.event_label.map(lidx).to_numpy(np.int8))
        on_.append(to_ms(p.onset)); off_.append(to_ms(p.offset)); start.append(start[-1] + len(p))
    out.update(op_threshold=np.array(ths), op_start=np.array(start, np.int64), op_file=np.concatenate(f_),
               op_label=np.concatenate(l_), op_onset_ms=np.concatenate(on_), op_offset_ms=np.concatenate(off_))
    expected = {
        "event": parse_report(os.path.join(meta, "metrics_test/student/event_f1.txt")),
        "segment": parse_report(os.path.join(meta, "metrics_test/student/segment_f1.txt")),
        # PSDS_Evaluation.ipynb, printed cell outputs (cells 23, 25, 32, 43)
        "notebook": {"event_macro_f1_pct": 39.83, "event_micro_f1_pct": 40.92, "segment_macro_f1_pct": 69.35,
                     "segment_micro_f1_pct": 75.47, "intersection_f1_pct": 63.74, "psds1": 0.334, "psds2": 0.533},
    }
    out["expected_json"] = np.array(json.dumps(expected))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes;", start[-1], "detections in", len(ths), "operating points")
    print(json.dumps(expected["event"]["classes"], indent=0)[:400])


if __name__ == "__main__":
    main()
