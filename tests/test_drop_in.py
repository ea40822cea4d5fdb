"""Zero-edit drop-in: the reference's own `train_sed.py` import block, executed verbatim against the alias layer
(desed_task_amd/drop_in in front of the reference on sys.path), binds the MI355X classes for the hot path and the reference's
own modules for everything else.  Needs /root/reference (build container only; the reference never travels)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
RECIPE = os.path.join(REF, "recipes", "dcase2023_task4_baseline")

SCRIPT = r'''
import os, sys, types
import torch

ROOT, REF, RECIPE = sys.argv[1:4]
# sys.path as `cd recipes/dcase2023_task4_baseline; PYTHONPATH=<repo>/desed_task_amd/drop_in:<repo> python train_sed.py` sets it
# up with the reference pip-installed (`pip install -e .` puts its root on the path, behind PYTHONPATH)
sys.path[:0] = [RECIPE, os.path.join(ROOT, "desed_task_amd", "drop_in"), ROOT, REF]


class _Any:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return None
    def __getattr__(self, k): return _Any()


def _missing(k):
    if k.startswith("__"):
        raise AttributeError(k)         # (inspect.getmodule walks sys.modules reading __file__)
    return _Any()


def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); m.__getattr__ = _missing; sys.modules[name] = m; return m


# third-party packages of the reference that this image lacks (inert stubs: only the import block is executed)
for n in ("pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.loggers", "torchaudio", "torchaudio.transforms",
          "h5py", "soundfile", "librosa", "dcase_util", "dcase_util.data", "sed_scores_eval", "sed_scores_eval.base_modules",
          "sed_scores_eval.base_modules.scores", "thop", "psds_eval", "sed_eval", "codecarbon", "torchmetrics",
          "torchmetrics.classification", "torchmetrics.classification.f_beta", "desed"):
    stub(n)
sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module

src = open(os.path.join(RECIPE, "train_sed.py")).read().splitlines()
block = "\n".join(src[:22])                       # the import block, train_sed.py:1-22, verbatim
assert "from desed_task.nnet.CRNN import CRNN" in block and "from local.sed_trainer import SEDTask4" in block
ns = {}
exec(compile(block, "train_sed.py[1:22]", "exec"), ns)

import importlib
A = importlib.import_module("desed_task_amd.nnet.CRNN")        # (the package re-exports the class under the module's name)
import desed_task_amd.sed_trainer as T
import desed_task_amd.utils.schedulers as S
assert ns["CRNN"] is A.CRNN, ns["CRNN"]
assert ns["SEDTask4"] is T.SEDTask4
assert ns["ExponentialWarmup"] is S.ExponentialWarmup
# everything the alias layer does not replace is still the reference's own code
import inspect
for name, where in (("ConcatDatasetBatchSampler", "desed_task/dataio/sampler.py"), ("StronglyAnnotatedSet", "desed_task/dataio/datasets.py"),
                    ("ManyHotEncoder", "desed_task/utils/encoder.py"), ("resample_folder", "local/resample_folder.py"),
                    ("generate_tsv_wav_durations", "local/utils.py")):
    f = inspect.getsourcefile(ns[name])
    assert f.startswith(REF) and f.endswith(where), (name, f)
assert isinstance(ns["classes_labels"], dict) and len(ns["classes_labels"]) == 10
import desed_task.utils.scaler, desed_task.nnet.CNN, local.sed_trainer_pretrained
from desed_task_amd.utils.scaler import TorchScaler
assert desed_task.utils.scaler.TorchScaler is TorchScaler
print("DROP_IN_OK")
'''


WIRING = r'''
"""train_sed.py:53-306 -- `single_run()` itself, executed verbatim against the alias layer: the reference's own ManyHotEncoder,
StronglyAnnotatedSet / WeakSet / UnlabeledSet, ConcatDatasetBatchSampler, `CRNN(**config["net"])`, `torch.optim.Adam`,
`ExponentialWarmup`, `SEDTask4(...)`, then `trainer.fit` / `trainer.test` -- with tests/lightning_order.Trainer in place of
pl.Trainer (Lightning is not installed), inert stubs for the other absent packages, `torchaudio.load` answering with a deterministic
synthetic waveform per file name, 1-second clips and the kernels on the CPU emulator."""
import hashlib, os, sys, types
import numpy as np
import pandas as pd
import torch
import yaml

ROOT, REF, RECIPE, TMP = sys.argv[1:5]
sys.path[:0] = [RECIPE, os.path.join(ROOT, "desed_task_amd", "drop_in"), ROOT, REF]
os.environ["SED_WHOLE_STEP"] = "1"          # the whole-step mode on the CPU device too (launcher.StepDriver behind training_step)
from tests.emu_support import bind_emulator
bind_emulator()


class _Any:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return None
    def __getattr__(self, k): return _Any()


def _missing(k):
    if k.startswith("__"):
        raise AttributeError(k)         # (inspect.getmodule walks sys.modules reading __file__)
    return _Any()


def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); m.__getattr__ = _missing; sys.modules[name] = m; return m


for n in ("pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.loggers", "torchaudio", "torchaudio.transforms",
          "h5py", "soundfile", "librosa", "dcase_util", "dcase_util.data", "sed_scores_eval", "sed_scores_eval.base_modules",
          "sed_scores_eval.base_modules.scores", "thop", "psds_eval", "sed_eval", "codecarbon", "torchmetrics",
          "torchmetrics.classification", "torchmetrics.classification.f_beta", "desed"):
    stub(n)

FS, CLIP = 16000, 1


def fake_load(path):
    seed = int(hashlib.md5(os.path.basename(path).encode()).hexdigest()[:8], 16)
    g = torch.Generator().manual_seed(seed)
    n = FS * CLIP + (seed % 3 - 1) * 800                    # a little shorter / exact / longer than the clip: pad_audio's three branches
    t = torch.arange(n) / FS
    return (0.1 * torch.randn(1, n, generator=g) + 0.3 * torch.sin(2 * np.pi * (300 + seed % 2000) * t)[None]).float(), FS


sys.modules["torchaudio"].load = fake_load
from desed_task_amd._lightning_standin import LightningModule
from tests.lightning_order import Trainer as _Loop
pl = sys.modules["pytorch_lightning"]
pl.LightningModule = LightningModule
pl.seed_everything = lambda seed, workers=False: (__import__("random").seed(seed), np.random.seed(seed), torch.manual_seed(seed))
seen = {}


class Logger:
    def __init__(self, save_dir, name):
        self.log_dir = os.path.join(save_dir, name, "version_0"); os.makedirs(self.log_dir, exist_ok=True)
    def log_hyperparams(self, *a, **k): pass
    def log_metrics(self, metrics, *a, **k): seen.setdefault("metrics", {}).update(metrics)


sys.modules["pytorch_lightning.loggers"].TensorBoardLogger = Logger


class Trainer(_Loop):
    """+ what single_run() reads back: the checkpoint callback's best path (written at the end of fit in Lightning's layout), test()."""
    def __init__(self, **kw):
        seen["trainer_kwargs"] = kw
        super().__init__(**kw)
        self.logger = kw["logger"]
        self.checkpoint_callback = types.SimpleNamespace(best_model_path=None)

    def fit(self, model, ckpt_path=None):
        model.logger = self.logger
        seen["model"] = model
        super().fit(model, ckpt_path)
        from desed_task_amd.launcher import checkpoint_dict
        path = os.path.join(self.logger.log_dir, "last.ckpt")
        torch.save(checkpoint_dict(model, epoch=self.max_epochs), path)
        self.checkpoint_callback.best_model_path = path
        seen["weights_after_fit"] = model.sed_student.arena.flat.detach().clone()

    def test(self, model):
        model.eval()
        seen["weights_at_test"] = model.sed_student.arena.flat.detach().clone()
        loader = model.test_dataloader()
        with torch.no_grad():
            for i, batch in enumerate(loader):
                if i >= 2:
                    break
                model.test_step(batch, i)
        seen["test_steps"] = i + 1


pl.Trainer = Trainer

# ---- a miniature DESED: file lists, annotations, durations (the audio itself is synthesised by fake_load) --------------------------
from local.classes_dict import classes_labels
classes = list(classes_labels.keys())
data = os.path.join(TMP, "data")
rng = np.random.RandomState(0)


def folder(name, n):
    d = os.path.join(data, name); os.makedirs(d, exist_ok=True)
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
    pd.DataFrame([(f, float(CLIP)) for f in files], columns=["filename", "duration"]).to_csv(path, sep="\t", index=False)


cfg = yaml.safe_load(open(os.path.join(RECIPE, "confs", "default.yaml")))
d = cfg["data"]
d["synth_folder"], fs_ = folder("synth", 6); d["synth_tsv"] = os.path.join(data, "synth.tsv"); strong_tsv(d["synth_tsv"], fs_)
d["weak_folder"], fw = folder("weak", 10); d["weak_tsv"] = os.path.join(data, "weak.tsv")
pd.DataFrame([(f, ",".join(sorted({classes[rng.randint(10)], classes[rng.randint(10)]}))) for f in fw],
             columns=["filename", "event_labels"]).to_csv(d["weak_tsv"], sep="\t", index=False)
d["unlabeled_folder"], _ = folder("unlabeled", 8)
d["synth_val_folder"], fv = folder("synth_val", 4); d["synth_val_tsv"] = os.path.join(data, "synth_val.tsv"); strong_tsv(d["synth_val_tsv"], fv)
d["synth_val_dur"] = os.path.join(data, "synth_val_dur.tsv"); durations(d["synth_val_dur"], fv)
d["test_folder"], ft = folder("test", 4); d["test_tsv"] = os.path.join(data, "test.tsv"); strong_tsv(d["test_tsv"], ft)
d["test_dur"] = os.path.join(data, "test_dur.tsv"); durations(d["test_dur"], ft)
d["audio_max_len"] = CLIP
cfg["training"].update(batch_size=[1, 1, 2], batch_size_val=2, num_workers=0, n_epochs_warmup=1)
cfg["scaler"]["savepath"] = os.path.join(TMP, "scaler.ckpt")

src = open(os.path.join(RECIPE, "train_sed.py")).read()
ns = {"__name__": "train_sed"}
exec(compile(src, os.path.join(RECIPE, "train_sed.py"), "exec"), ns)           # the whole file, verbatim (prepare_run is not called)
import desed_task_amd.sed_trainer as T
from desed_task_amd.arena import FusedAdam
from desed_task_amd.lookahead import LookaheadLoader
assert ns["SEDTask4"] is T.SEDTask4 and ns["pl"].Trainer is Trainer

ns["single_run"](cfg, os.path.join(TMP, "exp", "wiring"), "0", fast_dev_run=True)

model, kw = seen["model"], seen["trainer_kwargs"]
assert isinstance(model, T.SEDTask4) and type(model.sed_student).__module__.startswith("desed_task_amd.")
assert type(model.encoder).__module__ == "desed_task.utils.encoder" and "desed_task_amd" not in sys.modules[type(model.encoder).__module__].__file__
assert type(model.train_sampler).__name__ == "ConcatDatasetBatchSampler" and isinstance(model.train_loader, LookaheadLoader)
assert isinstance(model.opt, FusedAdam) and "weight_decay" in model.opt.param_groups[0]          # torch.optim.Adam, adopted in place
assert model.scheduler["scheduler"].optimizer is model.opt
assert kw["max_epochs"] == 3 and kw["limit_train_batches"] == 2 and kw["accumulate_grad_batches"] == 1 and kw["gradient_clip_val"] == 0.0
# 3 epochs x 2 batches went through the whole-step driver: Adam and the schedule advanced once per batch
assert model._driver is not None and model.scheduler["scheduler"].step_num == 1 + 6, model.scheduler["scheduler"].step_num
assert int(model.opt.state_dict()["state"][0]["step"]) == 6
for k in ("train/student/loss_strong", "train/student/loss_weak", "train/student/tot_self_loss", "train/weight", "train/lr", "train/step"):
    assert k in model.logged, k
assert np.isfinite(float(model.logged["train/student/loss_strong"]))
# validation ran every epoch on the device scoring path with the reference's datasets; its objective was logged
assert "val/obj_metric" in model.logged and "val/synth/student/psds1_sed_scores_eval" in model.logged
# checkpoint round trip: single_run() reloaded `state_dict` from the file the trainer wrote, then tested
assert os.path.exists(os.path.join(TMP, "exp", "wiring", "version_0", "last.ckpt"))
assert torch.equal(seen["weights_after_fit"], seen["weights_at_test"]) and seen["test_steps"] == 2
print("SINGLE_RUN_OK", float(model.logged["train/student/loss_strong"]), float(model.logged["val/obj_metric"]))
'''


@pytest.mark.skipif(not os.path.isdir(RECIPE), reason="needs the reference checkout (build container only)")
def test_single_run_wiring(tmp_path):
    """`single_run()` (train_sed.py:53-306) executed, not just its imports."""
    script = tmp_path / "run_single_run.py"
    script.write_text(WIRING)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(script), ROOT, REF, RECIPE, str(tmp_path)], capture_output=True, text=True, timeout=1500,
                       env=env, cwd=str(tmp_path))
    assert "SINGLE_RUN_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.skipif(not os.path.isdir(RECIPE), reason="needs the reference checkout (build container only)")
def test_reference_import_block_binds_hip_classes(tmp_path):
    script = tmp_path / "run_import_block.py"
    script.write_text(textwrap.dedent(SCRIPT))
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(script), ROOT, REF, RECIPE], capture_output=True, text=True, timeout=300, env=env,
                       cwd=str(tmp_path))
    assert "DROP_IN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_alias_modules_are_reexports_only():
    """The alias layer carries no logic: every module is a docstring + imports (and the extend_path line)."""
    n = 0
    for base in (os.path.join(ROOT, "desed_task_amd", "drop_in"), os.path.join(ROOT, "desed_task_amd", "drop_in_2024")):
      for d, _, files in os.walk(base):
        for f in files:
            if f.endswith(".py"):
                n += 1
                import ast
                tree = ast.parse(open(os.path.join(d, f)).read())
                for node in tree.body:
                    ok = isinstance(node, (ast.Import, ast.ImportFrom, ast.Try)) or (isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant)) \
                        or (isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "__path__")
                    assert ok, (f, ast.dump(node)[:80])
    assert n >= 12
