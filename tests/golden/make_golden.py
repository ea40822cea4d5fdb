#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz by importing the REFERENCE itself.

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden.py

The reference's Python never travels to the GPU box; only the arrays written here do.
Third-party packages the reference imports but this image lacks are replaced by inert stubs in
sys.modules.  The only stubs that carry arithmetic are the three torchaudio classes
(MelSpectrogram / AmplitudeToDB / TimeMasking): torchaudio is not vendored by the reference, so
parity at that boundary is *unpinned* by the reference.  For them the fixture G1 is
produced by an independent float64 numpy implementation (np.fft.rfft) instead.

Inputs are closed-form (oracle.sed_oracle.lcg_fill / synth_audio), so fixtures stay small.
"""
import math
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
RECIPE = os.path.join(REF, "recipes", "dcase2023_task4_baseline")

from oracle import sed_oracle as O  # noqa: E402  (only for closed-form inputs + torchaudio stand-ins)


# ---------------------------------------------------------------------------------------------
# stubs
# ---------------------------------------------------------------------------------------------
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, k):
        return _Any()


class MelSpectrogram(torch.nn.Module):          # stand-in for torchaudio.transforms.MelSpectrogram
    def __init__(self, sample_rate, n_fft, win_length, hop_length, f_min, f_max, n_mels, window_fn, wkwargs, power):
        super().__init__()
        assert power == 1 and window_fn is torch.hamming_window and wkwargs == {"periodic": False}
        self.feats = dict(n_mels=n_mels, n_window=n_fft, hop_length=hop_length, sample_rate=sample_rate,
                          f_min=f_min, f_max=f_max)

    def forward(self, x):
        return O.mel_spectrogram(x, self.feats)


class AmplitudeToDB(torch.nn.Module):           # stand-in for torchaudio.transforms.AmplitudeToDB
    def __init__(self, stype="power", top_db=None):
        super().__init__()
        self.multiplier = 10.0 if stype == "power" else 20.0
        self.amin = 1e-10
        self.db_multiplier = math.log10(max(self.amin, 1.0))

    def forward(self, x):
        return self.multiplier * torch.log10(torch.clamp(x, min=self.amin)) - self.multiplier * self.db_multiplier


class TimeMasking(torch.nn.Module):             # stand-in (per-clip masks, torchaudio >= 2.1)
    def __init__(self, time_mask_param, iid_masks=False, p=1.0):
        super().__init__()
        self.mask_param, self.p = time_mask_param, p

    def forward(self, spec):
        B, _, L = spec.shape
        mp = O.specaug_mask_param(self.mask_param, self.p, L)
        if mp < 1:
            return spec
        s, e = O.specaug_bounds(torch.rand(B), torch.rand(B), mp, L)
        idx = torch.arange(L).view(1, 1, L)
        return spec.masked_fill((idx >= s.view(B, 1, 1)) & (idx < e.view(B, 1, 1)), 0.0)


class LightningModule(torch.nn.Module):         # 10-line shim: hparams dict + log
    def __init__(self):
        super().__init__()
        self.hparams = {}
        self.logged = {}

    def log(self, name, value, **kw):
        self.logged[name] = float(value)


def install_stubs():
    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms", MelSpectrogram=MelSpectrogram, AmplitudeToDB=AmplitudeToDB,
                          TimeMasking=TimeMasking)
    ta.load = _Any()
    _stub("pytorch_lightning", LightningModule=LightningModule)
    tm = _stub("torchmetrics")
    tm.classification = _stub("torchmetrics.classification")
    tm.classification.f_beta = _stub("torchmetrics.classification.f_beta", MultilabelF1Score=_Any)
    _stub("codecarbon", OfflineEmissionsTracker=_Any)
    sse = _stub("sed_scores_eval")
    sse.base_modules = _stub("sed_scores_eval.base_modules")
    sse.base_modules.scores = _stub("sed_scores_eval.base_modules.scores", create_score_dataframe=_Any)
    du = _stub("dcase_util")
    du.data = _stub("dcase_util.data", DecisionEncoder=_Any)
    for n in ("h5py", "psds_eval", "sed_eval", "soundfile", "thop", "scipy.ndimage.filters"):
        _stub(n)
    sys.modules["psds_eval"].PSDSEval = _Any
    sys.modules["psds_eval"].plot_psd_roc = _Any
    sys.modules["thop"].profile = _Any
    sys.modules["thop"].clever_format = _Any
    sys.path.insert(0, REF)
    sys.path.insert(0, RECIPE)


# ---------------------------------------------------------------------------------------------
def np_mel_float64(audio, n_fft=2048, hop=256, n_mels=128, sr=16000, fmin=0.0, fmax=8000.0):
    """Independent float64 implementation of a1 (reflect pad, np.hamming, rfft, HTK triangles)."""
    a = np.pad(audio.astype(np.float64), ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    win = np.hamming(n_fft)
    n_frames = 1 + (a.shape[1] - n_fft) // hop
    frames = np.stack([a[:, i * hop:i * hop + n_fft] for i in range(n_frames)], 1) * win
    mag = np.abs(np.fft.rfft(frames, axis=-1))                                  # (B, T, 1025)
    freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mpts = np.linspace(mel(fmin), mel(fmax), n_mels + 2)
    fpts = 700.0 * (10 ** (mpts / 2595.0) - 1.0)
    fd = np.diff(fpts)
    sl = fpts[None, :] - freqs[:, None]
    fb = np.maximum(0, np.minimum(-sl[:, :-2] / fd[:-1], sl[:, 2:] / fd[1:]))   # (1025, 128)
    return np.transpose(mag @ fb, (0, 2, 1)), fb                               # (B, 128, T)


def main():
    install_stubs()
    from desed_task.nnet.CRNN import CRNN
    from desed_task.utils.scaler import TorchScaler
    from desed_task.utils.schedulers import ExponentialWarmup
    from desed_task.data_augm import mixup
    from local.sed_trainer import SEDTask4
    import yaml

    torch.set_num_threads(8)
    out = {}
    with open(os.path.join(RECIPE, "confs", "default.yaml")) as f:
        config = yaml.safe_load(f)

    # ---- G1: mel + log + scaler from the independent float64 implementation ---------------
    audio = O.synth_audio(2, 160000, seed=1234)
    mel64, fb64 = np_mel_float64(audio.numpy())
    logm = np.clip(20 * np.log10(np.maximum(mel64, 1e-5)), -50, 80)
    mn, mx = logm.min((1, 2), keepdims=True), logm.max((1, 2), keepdims=True)
    scaled = (logm - mn) / (mx - mn + 1e-8) * 2 - 1
    out["g1_mel_lin"] = mel64[:, :, ::25].astype(np.float32)                  # strided sample of frames
    out["g1_scaled"] = scaled[:, :, ::25].astype(np.float32)
    out["g1_minmax"] = np.stack([mn.ravel(), mx.ravel()], 1).astype(np.float32)
    out["g1_fb_nnz"] = np.array([(fb64.astype(np.float32) != 0).sum()], dtype=np.int64)
    out["g1_fb_colsum"] = fb64.sum(0).astype(np.float32)

    # ---- G2: mixup with recorded draws (reference function, seeded) -----------------------
    x = O.lcg_fill((6, 8, 20), 5, 1.0, 1.5)
    y = (O.lcg_fill((6, 10, 7), 6, 0.5, 0.5) < 0.3).float()
    np.random.seed(3); torch.manual_seed(3)
    c = np.random.beta(0.2, 0.2); perm = torch.randperm(6)
    np.random.seed(3); torch.manual_seed(3)
    mx_, my_ = mixup(x, y, mixup_label_type="soft")
    out.update(g2_c=np.array([c]), g2_perm=perm.numpy(), g2_x=mx_.numpy(), g2_y=my_.numpy())

    # ---- reference CRNN with LCG weights ---------------------------------------------------
    def ref_crnn(sd):
        net = CRNN(**config["net"])
        missing = net.load_state_dict(sd, strict=True)
        return net

    sd = O.make_state_dict(seed=7)
    B, T = 3, 160                                                              # T -> 40 output frames
    xin = O.lcg_fill((B, 128, T), 21, 1.0)                                     # already-scaled features in [-1,1]

    # G3/G4/G5 eval mode
    net = ref_crnn(sd); net.eval()
    taps = {}
    hooks = []
    for i in range(7):
        hooks.append(net.cnn.cnn._modules[f"pooling{i}"].register_forward_hook(
            lambda m, a, o, i=i: taps.__setitem__(f"block{i}", o.detach())))
    hooks.append(net.rnn.register_forward_hook(lambda m, a, o: taps.__setitem__("gru", o.detach())))
    with torch.no_grad():
        strong, weak = net(xin)
    for i in range(7):
        out[f"g3_eval_block{i}"] = O.tap_sample(taps[f"block{i}"]).numpy().copy()
        out[f"g3_eval_block{i}_sum"] = np.array([taps[f"block{i}"].double().sum().item()])
    out["g4_eval_gru"] = taps["gru"].numpy()
    out["g5_eval_strong"], out["g5_eval_weak"] = strong.numpy(), weak.numpy()

    # G3/G5/G7 train mode, dropout=0, specaug off; backward of a simple loss
    cfg = dict(config["net"]); cfg["dropout"] = 0.0
    net = CRNN(**cfg, specaugm_t_p=0.0, specaugm_f_p=0.0); net.load_state_dict(sd); net.train()
    taps.clear()
    for i in range(7):
        net.cnn.cnn._modules[f"pooling{i}"].register_forward_hook(
            lambda m, a, o, i=i: taps.__setitem__(f"block{i}", o.detach()))
    strong, weak = net(xin)
    tgt_s = (O.lcg_fill(tuple(strong.shape), 31, 0.5, 0.5) < 0.2).float()
    tgt_w = (O.lcg_fill(tuple(weak.shape), 32, 0.5, 0.5) < 0.3).float()
    loss = torch.nn.functional.binary_cross_entropy(strong, tgt_s) + torch.nn.functional.binary_cross_entropy(weak, tgt_w)
    loss.backward()
    for i in range(7):
        out[f"g3_train_block{i}"] = O.tap_sample(taps[f"block{i}"]).numpy().copy()
        out[f"g3_train_block{i}_sum"] = np.array([taps[f"block{i}"].double().sum().item()])
        out[f"g3_train_rm{i}"] = net.cnn.cnn._modules[f"batchnorm{i}"].running_mean.numpy().copy()
        out[f"g3_train_rv{i}"] = net.cnn.cnn._modules[f"batchnorm{i}"].running_var.numpy().copy()
    out["g5_train_strong"], out["g5_train_weak"] = strong.detach().numpy(), weak.detach().numpy()
    out["g7_loss"] = np.array([loss.item()])
    names = [n for n, _ in net.named_parameters()]
    out["g7_param_names"] = np.array(names)
    out["g7_grad_norms"] = np.array([p.grad.norm().item() for _, p in net.named_parameters()])
    for n, p in net.named_parameters():
        if n in ("cnn.cnn.conv0.weight", "cnn.cnn.conv3.bias", "cnn.cnn.batchnorm2.weight", "cnn.cnn.glu4.linear.weight",
                 "rnn.rnn.weight_hh_l0", "rnn.rnn.weight_ih_l1_reverse", "rnn.rnn.bias_hh_l1", "dense.weight",
                 "dense_softmax.bias", "cnn.cnn.conv6.weight"):
            out["g7_grad__" + n] = p.grad.numpy().reshape(-1)[:512].copy()

    # ---- G6/G8: SEDTask4.training_step x3 with the Lightning hook order ---------------------
    config["training"]["batch_size"] = [2, 2, 4]
    cfg = dict(config["net"]); cfg["dropout"] = 0.0
    student = CRNN(**cfg, specaugm_t_p=0.0, specaugm_f_p=0.0); student.load_state_dict(sd)

    class Enc:
        labels = list(range(10))
    opt = torch.optim.Adam(student.parameters(), config["opt"]["lr"], betas=(0.9, 0.999))
    sched = {"scheduler": ExponentialWarmup(opt, config["opt"]["lr"], 100), "interval": "step"}
    task = SEDTask4(config, Enc(), student, opt=opt, scheduler=sched)
    task.train()
    audio8 = O.synth_audio(8, 16000 * 2 + 1024, seed=77)                       # 2.064 s clips -> 130 frames -> 32 out
    n_out = (1 + audio8.shape[1] // 256) // 4
    labels8 = O.synth_labels((2, 2, 4), 10, n_out, seed=5)
    logs_all = []
    draws = []
    for step in range(3):
        # force mixup ON and record its draws: random.random() < 0.5
        random.seed(4)                                                         # random.random() -> 0.236 < 0.5
        assert random.random() < 0.5
        random.seed(4); np.random.seed(100 + step); torch.manual_seed(100 + step)
        st = (np.random.get_state(), torch.get_rng_state())
        cw = np.random.beta(0.2, 0.2); pw = torch.randperm(2); cs = np.random.beta(0.2, 0.2); ps = torch.randperm(2)
        draws.append((cw, pw.numpy(), cs, ps.numpy()))
        np.random.set_state(st[0]); torch.set_rng_state(st[1])
        batch = (audio8.clone(), labels8.clone(), None, None)
        task.logged = {}
        loss = task.training_step(batch, step)
        task.on_before_zero_grad()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched["scheduler"].step()
        logs_all.append(dict(task.logged, loss=loss.item()))
        if step == 0:
            out["g8_teacher_after_ema1__dense.weight"] = task.sed_teacher.dense.weight.detach().numpy().copy()
    keys = sorted(logs_all[0].keys())
    out["g6_keys"] = np.array(keys)
    out["g6_values"] = np.array([[l[k] for k in keys] for l in logs_all], dtype=np.float64)
    out["g6_mix_c"] = np.array([[d[0], d[2]] for d in draws])
    out["g6_mix_perm_weak"] = np.stack([d[1] for d in draws])
    out["g6_mix_perm_strong"] = np.stack([d[3] for d in draws])
    for n in ("cnn.cnn.conv0.weight", "cnn.cnn.glu3.linear.bias", "rnn.rnn.weight_hh_l1_reverse", "dense.bias"):
        out["g6_student_after3__" + n] = dict(student.named_parameters())[n].detach().numpy().reshape(-1)[:256].copy()
        out["g6_teacher_after3__" + n] = dict(task.sed_teacher.named_parameters())[n].detach().numpy().reshape(-1)[:256].copy()
    out["g6_teacher_rm0_after3"] = task.sed_teacher.cnn.cnn.batchnorm0.running_mean.numpy().copy()

    # ---- G8: update_ema at step_num = 1, 2, 1000 --------------------------------------------
    a = CRNN(**config["net"]); b = CRNN(**config["net"])
    a.load_state_dict(O.make_state_dict(seed=7)); b.load_state_dict(O.make_state_dict(seed=8))
    for gs in (1, 2, 1000):
        task.update_ema(0.999, gs, a, b)
        out[f"g8_ema_step{gs}__dense.weight"] = b.dense.weight.detach().numpy().copy()
        out[f"g8_ema_step{gs}__conv1.bias"] = b.cnn.cnn.conv1.bias.detach().numpy().copy()

    # ---- G9: ExponentialWarmup -------------------------------------------------------------
    dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = ExponentialWarmup(dummy, 1e-3, 5900)
    vals = []
    for s in (1, 100, 5900, 23600):
        sch.step_num = s
        vals.append([s, sch._get_scaling_factor(), sch._get_lr()])
    out["g9_warmup"] = np.array(vals)

    # ---- G10: TorchScaler (reference) on a random tensor ------------------------------------
    sc = TorchScaler("instance", "minmax", [1, 2])
    xs = O.lcg_fill((3, 16, 9), 55, 30.0, -10.0)
    out["g10_scaler"] = sc(xs).numpy()
    # ManyHotEncoder.n_frames arithmetic (desed_task/utils/encoder.py:39-40)
    out["g10_n_frames"] = np.array([int(int((10 * 16000 / 256)) / 4)])

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    tot = sum(v.nbytes for v in out.values())
    print(f"wrote {len(out)} arrays, {tot/1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
