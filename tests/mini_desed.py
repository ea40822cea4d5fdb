"""A miniature DESED on disk for the command-line tests: file lists, annotations, durations and a recipe config derived from the
reference's confs/default.yaml, plus a directory of stand-ins for the third-party packages this image lacks (torchaudio -- whose
`load` answers with a deterministic synthetic waveform per file name --, h5py, dcase_util).  Test infrastructure: nothing here
computes anything the product ships."""
import os

import numpy as np
import pandas as pd
import yaml

STUBS = {
    "torchaudio/__init__.py": '''
import hashlib, os
import numpy as np
import torch
FS, CLIP = 16000, 1
def load(path):
    seed = int(hashlib.md5(os.path.basename(path).encode()).hexdigest()[:8], 16)
    g = torch.Generator().manual_seed(seed)
    n = FS * CLIP + (seed % 3 - 1) * 800
    t = torch.arange(n) / FS
    return (0.1 * torch.randn(1, n, generator=g) + 0.3 * torch.sin(2 * np.pi * (300 + seed % 2000) * t)[None]).float(), FS
''',
    "torchaudio/transforms.py": "",
    "h5py.py": "",
    "dcase_util/__init__.py": "",
    "dcase_util/data.py": "class DecisionEncoder:\n    def __init__(self, *a, **k): pass\n",
}


def write_stubs(root):
    for rel, text in STUBS.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    return root


def build(tmp, recipe, n_epochs=2, batch_size=(1, 1, 2), clip=1):
    """-> path of the config file.  6 synthetic + 10 weak + 8 unlabelled training clips, 4 + 4 validation / test clips."""
    from local.classes_dict import classes_labels
    classes = list(classes_labels.keys())
    data = os.path.join(tmp, "data")
    rng = np.random.RandomState(0)

    def folder(name, n):
        d = os.path.join(data, name)
        os.makedirs(d, exist_ok=True)
        files = ["%s_%02d.wav" % (name, i) for i in range(n)]
        for f in files:
            open(os.path.join(d, f), "wb").close()
        return d, files

    def strong_tsv(path, files):
        rows = []
        for f in files:
            for _ in range(2):
                on = float(rng.uniform(0, 0.5))
                rows.append((f, round(on, 3), round(on + float(rng.uniform(0.1, 0.4)), 3), classes[rng.randint(10)]))
        pd.DataFrame(rows, columns=["filename", "onset", "offset", "event_label"]).to_csv(path, sep="\t", index=False)

    def durations(path, files):
        pd.DataFrame([(f, float(clip)) for f in files], columns=["filename", "duration"]).to_csv(path, sep="\t", index=False)

    cfg = yaml.safe_load(open(os.path.join(recipe, "confs", "default.yaml")))
    d = cfg["data"]
    d["synth_folder"], fs_ = folder("synth", 6)
    d["synth_tsv"] = os.path.join(data, "synth.tsv")
    strong_tsv(d["synth_tsv"], fs_)
    d["weak_folder"], fw = folder("weak", 10)
    d["weak_tsv"] = os.path.join(data, "weak.tsv")
    pd.DataFrame([(f, ",".join(sorted({classes[rng.randint(10)], classes[rng.randint(10)]}))) for f in fw],
                 columns=["filename", "event_labels"]).to_csv(d["weak_tsv"], sep="\t", index=False)
    d["unlabeled_folder"], _ = folder("unlabeled", 8)
    d["synth_val_folder"], fv = folder("synth_val", 4)
    d["synth_val_tsv"] = os.path.join(data, "synth_val.tsv")
    strong_tsv(d["synth_val_tsv"], fv)
    d["synth_val_dur"] = os.path.join(data, "synth_val_dur.tsv")
    durations(d["synth_val_dur"], fv)
    d["test_folder"], ft = folder("test", 4)
    d["test_tsv"] = os.path.join(data, "test.tsv")
    strong_tsv(d["test_tsv"], ft)
    d["test_dur"] = os.path.join(data, "test_dur.tsv")
    durations(d["test_dur"], ft)
    d["audio_max_len"] = clip
    cfg["training"].update(batch_size=list(batch_size), batch_size_val=2, num_workers=0, n_epochs_warmup=1, n_epochs=n_epochs)
    cfg["scaler"]["savepath"] = os.path.join(tmp, "scaler.ckpt")
    path = os.path.join(tmp, "conf.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path
