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


def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); m.__getattr__ = lambda k: _Any(); sys.modules[name] = m; return m


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
