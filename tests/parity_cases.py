"""Parity cases shared by the CPU-emulator tests (small shapes) and the GPU tests (`-m gpu`, through the real
C-ABI library).  Every case compares the HIP path against the CPU oracle on the same seeded inputs."""
import math

import numpy as np
import pytest
import torch

from oracle import sed_oracle as O
from desed_task_amd import _lib
from desed_task_amd import features as Fh


def to(dev, *ts):
    r = [t.to(dev) for t in ts]
    return r[0] if len(r) == 1 else r


def case_mfma_selftest(dev):
    lib = _lib.get()
    g = torch.Generator().manual_seed(3)
    for shape, K in ((32, 8), (32, 64), (16, 12), (16, 64)):
        A = torch.randn(shape, K, generator=g)
        B = torch.randn(K, shape, generator=g)       # asymmetric B: catches transposed C maps
        C = torch.zeros(shape, shape)
        Ad, Bd, Cd = to(dev, A, B, C)
        lib.call("sed_selftest_mfma", Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), K, shape, _lib.stream_ptr(Ad))
        ref = (A.double() @ B.double()).float()
        err = (Cd.cpu() - ref).abs().max().item()
        assert err < 1e-4, (shape, K, err)
    # bf16 32x32x16 map (split-bf16 paths): operands rounded to bf16 on both sides
    for K in (16, 64):
        A = torch.randn(32, K, generator=g)
        B = torch.randn(K, 32, generator=g)
        C = torch.zeros(32, 32)
        Ad, Bd, Cd = to(dev, A, B, C)
        lib.call("sed_selftest_mfma", Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), K, 3216, _lib.stream_ptr(Ad))
        ref = (A.bfloat16().double() @ B.bfloat16().double()).float()
        err = (Cd.cpu() - ref).abs().max().item()
        assert err < 1e-4, ("bf16", K, err)
    # bf16 16x16x32 map (128-channel GLU backward)
    for K in (32, 96):
        A = torch.randn(16, K, generator=g)
        B = torch.randn(K, 16, generator=g)
        C = torch.zeros(16, 16)
        Ad, Bd, Cd = to(dev, A, B, C)
        lib.call("sed_selftest_mfma", Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), K, 1632, _lib.stream_ptr(Ad))
        ref = (A.bfloat16().double() @ B.bfloat16().double()).float()
        err = (Cd.cpu() - ref).abs().max().item()
        assert err < 1e-4, ("bf16 16x16x32", K, err)


def make_mel():
    return Fh.MelSpectrogram(16000, 2048, 2048, 256, 0, 8000, 128, torch.hamming_window, {"periodic": False}, 1)


def case_mel(dev, batch=2, n_samples=256 * 12):
    audio = O.synth_audio(batch, n_samples, seed=1234)
    ref = O.mel_spectrogram(audio)
    mel = make_mel()
    got = mel(to(dev, audio))
    assert tuple(got.shape) == tuple(ref.shape)
    err = (got.cpu() - ref).abs().max().item()
    assert err < 2e-4 * ref.abs().max().item(), err
    a = O.scale_minmax(O.take_log(ref))
    b = Fh.minmax_scale(got, apply_log=True).cpu()
    assert (a - b).abs().max().item() < 1e-3          # north_star tolerance, scaler domain
    c = Fh.minmax_scale(Fh.take_log(got)).cpu()
    assert (a - c).abs().max().item() < 1e-3
    fused = mel.frames_major(to(dev, audio), apply_log=True).transpose(1, 2).cpu()
    assert (fused - O.take_log(ref)).abs().max().item() < 2e-2   # dB domain near the 1e-5 floor
    return got


def case_mel_walk_batch_independent(dev, batch=48, n_samples=160000, probe=(0, 7, 8, 23, 47)):
    """Size-independent property of the XCD-aware frame walk at the BASELINE batch: every frame of every clip is produced exactly once
    and does not depend on which workgroup / XCD / segment it fell to -- the batched launch (one segment per clip when 8 | B) equals,
    bit for bit, the launches of single clips (eight segments per clip) and of a 12-clip batch (two segments per clip)."""
    g = torch.Generator().manual_seed(5)
    audio = to(dev, 0.1 * torch.randn(batch, n_samples, generator=g))
    mel = make_mel()
    out = torch.full((batch, mel.n_mels, 1 + n_samples // 256), float("nan"), device=audio.device)
    whole = mel(audio)
    assert tuple(whole.shape) == tuple(out.shape) and bool(torch.isfinite(whole).all())
    for i in probe:
        if i < batch:
            assert torch.equal(mel(audio[i:i + 1])[0], whole[i]), i
    if batch >= 12:
        assert torch.equal(mel(audio[:12]), whole[:12])
    return whole


def case_mel_in_graph_beside_tails(dev, replays=300, beside="tails", launch=None, after_replay=None):
    """The mel kernel as a hipGraph node on a side stream beside the student's and the teacher's BiGRU + head tails (the "tails" fork of
    the pipelined step; beside = "gemm": beside the BiGRU's split-bf16 input projection alone), replayed `replays` times on changing
    waveforms: every output bit-equal to the solo launch.  (Round 5: a form of the wave-per-frame kernel in which a wave transformed two
    or more frames per launch had a few mirror-paired bins of a wave's later frames wrong in 0.2 - 7 % of such replays, never eager;
    the shipped kernel gives a wave exactly one frame.  tools/mel_graph_race.py is the same loop with a library argument for variant
    builds; tools/mel_repro/race.py runs it on the diagnostics-only multi-frame reproducer: launch(audio, out) replaces the mel call,
    after_replay(rep, i, n_bad_elements) is called after every replay.)"""
    task = build_task(dev, (1, 1, 2), O.make_state_dict(seed=7), dropout=0.5, specaug=True, rampup=5)
    mel = task.mel_spec
    B, N = 4, 16000 + 1024
    g = torch.Generator().manual_seed(1)
    audios = [to(dev, 0.1 * torch.randn(B, N, generator=g)) for _ in range(8)]
    if launch is None:
        def launch(audio_, out_):
            mel.frames_major(audio_, out=out_)
    refs = []
    for a in audios:
        r = torch.empty(B, 1 + N // mel.hop_length, mel.n_mels, device=a.device)
        launch(a, r)
        refs.append(r)
    static_audio = audios[0].clone()
    out = torch.empty_like(refs[0])
    x = task.scaled_logmel(mel(audios[0]))
    with torch.no_grad():
        h = task.sed_student.forward_cnn(x)
    torch.cuda.synchronize()
    main, s_mel, s_t = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.randn(1024, 1024, device=out.device)
    # direct launches of the two kernels of a BiGRU layer (diagnostics: beside = "gemm" / "gru")
    Bq, Tq, Hq = h.shape[0], h.shape[1], 128
    rnn0 = task.sed_student.rnn.rnn
    gi_buf = torch.zeros(Bq, Tq, 2, 3 * Hq, device=out.device)
    go_buf = torch.empty(Bq, Tq, 2 * Hq, device=out.device)

    def gemm_only(stream):
        lib = _lib.get()
        lib.call("sed_gemm_pair_bf16x3", h.data_ptr(), h.data_ptr(), rnn0.weight_ih_l0.data_ptr(), rnn0.weight_ih_l0_reverse.data_ptr(),
                 rnn0.bias_ih_l0.data_ptr(), rnn0.bias_ih_l0_reverse.data_ptr(), gi_buf.data_ptr(), gi_buf.data_ptr() + 3 * Hq * 4,
                 Bq * Tq, 3 * Hq, Hq, Hq, Hq, 6 * Hq, 0, 1, 1, 0, stream.cuda_stream)

    # beside = "storm": the split-bf16 GEMM at the production size of the BiGRU input projection (M = 48 x 156 rows: 708 workgroups), a
    # dozen launches queued on the co-runner's stream BEFORE the mel node, so that MFMA waves are resident on every CU from the first to
    # the last mel wave -- with the two small GEMMs of "gemm" the co-runner only arrives a few microseconds into the mel kernel, which
    # is why round 5 saw the fault only in waves that lived long enough to transform a second frame (profiles/r06_mel_mechanism.md)
    Ms = 48 * 156
    h_big = torch.randn(Ms, Hq, device=out.device) if beside == "storm" else None
    gi_big = torch.zeros(Ms, 2, 3 * Hq, device=out.device) if beside == "storm" else None

    def gemm_storm(stream, n=12):
        lib = _lib.get()
        for _ in range(n):
            lib.call("sed_gemm_pair_bf16x3", h_big.data_ptr(), h_big.data_ptr(), rnn0.weight_ih_l0.data_ptr(), rnn0.weight_ih_l0_reverse.data_ptr(),
                     rnn0.bias_ih_l0.data_ptr(), rnn0.bias_ih_l0_reverse.data_ptr(), gi_big.data_ptr(), gi_big.data_ptr() + 3 * Hq * 4,
                     Ms, 3 * Hq, Hq, Hq, Hq, 6 * Hq, 0, 1, 1, 0, stream.cuda_stream)

    def gru_only(stream):
        lib = _lib.get()
        lib.call("sed_gru_fwd", gi_buf.data_ptr(), rnn0.weight_hh_l0.data_ptr(), rnn0.weight_hh_l0_reverse.data_ptr(),
                 rnn0.bias_hh_l0.data_ptr(), rnn0.bias_hh_l0_reverse.data_ptr(), go_buf.data_ptr(), None, Bq, Tq, Hq, stream.cuda_stream)

    def body():
        with torch.no_grad():
            hs = task.sed_student.forward_cnn(x)
            cur = torch.cuda.current_stream()
            if beside == "storm":
                s_t.wait_stream(cur)
                with torch.cuda.stream(s_t):
                    gemm_storm(s_t)
            s_mel.wait_stream(cur)
            with torch.cuda.stream(s_mel):
                launch(static_audio, out)
            s_t.wait_stream(cur)
            with torch.cuda.stream(s_t):
                if beside == "tails":
                    task.sed_teacher.forward_tail(h)
                elif beside == "matmul":
                    big @ big
                elif beside == "rnn":
                    task.sed_teacher.rnn(h, arena=task.sed_teacher.arena)
                elif beside == "cnn":
                    task.sed_teacher.forward_cnn(x)
                elif beside == "gemm":
                    gemm_only(s_t); gemm_only(s_t)
                elif beside == "gru":
                    gru_only(s_t); gru_only(s_t)
            if beside == "tails":
                task.sed_student.forward_tail(hs)
            elif beside == "matmul":
                big @ big
            elif beside == "rnn":
                task.sed_student.rnn(hs, arena=task.sed_student.arena)
            elif beside == "cnn":
                task.sed_student.forward_cnn(x)
            cur.wait_stream(s_t)
            cur.wait_stream(s_mel)

    with torch.cuda.stream(main):
        body(); body()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=main, capture_error_mode="thread_local"):
        body()
    bad = []
    for rep in range(replays):
        i = rep % 8
        with torch.cuda.stream(main):
            static_audio.copy_(audios[i], non_blocking=True)
            out.fill_(-777.0)
            graph.replay()
        torch.cuda.synchronize()
        ne = out != refs[i]
        nbad = int(ne.sum())
        if nbad:
            bad.append((rep, nbad, sorted({(int(b), int(t)) for b, t, m in ne.nonzero().tolist()})[:4]))
        if after_replay is not None:
            after_replay(rep, i, nbad)
    assert not bad, (len(bad), bad[:5])
    return replays


def case_logscale_generic(dev):
    x = O.lcg_fill((3, 16, 9), 55, 30.0, -10.0)
    got, mm = Fh.minmax_scale(to(dev, x), return_minmax=True)
    np.testing.assert_allclose(got.cpu().numpy(), O.scale_minmax(x).numpy(), atol=1e-6)
    np.testing.assert_allclose(mm[:, 0].cpu().numpy(), x.amin((1, 2)).numpy())
    np.testing.assert_allclose(mm[:, 1].cpu().numpy(), x.amax((1, 2)).numpy())


def case_mixup_specaug(dev):
    x = O.lcg_fill((6, 8, 20), 5, 1.0, 1.5)
    y = (O.lcg_fill((6, 10, 7), 6, 0.5, 0.5) < 0.3).float()
    perm = torch.tensor([3, 0, 5, 1, 2, 4])
    c = 0.3721
    rx, ry = O.mixup_apply(x, y, c, perm)
    gx = Fh.mixup_(to(dev, x.clone()), perm, c, mode=0)
    gy = Fh.mixup_(to(dev, y.clone()), perm, c, mode=1)
    np.testing.assert_allclose(gx.cpu().numpy(), rx.numpy(), atol=1e-6)
    np.testing.assert_allclose(gy.cpu().numpy(), ry.numpy(), atol=1e-6)
    xin = O.lcg_fill((3, 16, 30), 9, 1.0)
    bounds = torch.tensor([[2, 5, 10, 13], [0, 0, 29, 30], [15, 16, 0, 4]], dtype=torch.int32)
    ref = O.specaug_apply(xin, (bounds[:, 0].long(), bounds[:, 1].long()), (bounds[:, 2].long(), bounds[:, 3].long()))
    got = Fh.specaug_apply(to(dev, xin), to(dev, bounds))
    assert torch.equal(got.cpu(), ref)
    # the mask draws: one kernel on two torch.rand calls == the reference's tensor arithmetic on the same draws, bit for bit
    device = torch.device(dev)
    for (B, n_freq, n_time, f_l, f_p, t_l, t_p, iid) in ((48, 128, 626, 10, 0.2, 5, 0.2, True), (5, 128, 626, 10, 0.2, 5, 0.2, False),
                                                          (7, 64, 40, 10, 0.2, 5, 0.0, True), (3, 16, 9, 0, 0.2, 3, 0.5, True)):
        torch.manual_seed(123)
        got_b = Fh.specaug_bounds(B, n_freq, n_time, f_l, f_p, t_l, t_p, device, iid_masks=iid).cpu()
        torch.manual_seed(123)
        want = torch.zeros(B, 4, dtype=torch.int32)
        n = B if iid else 1
        for col, (cap, p, axis_len) in enumerate(((f_l, f_p, n_freq), (t_l, t_p, n_time))):
            mask_param = min(cap, int(axis_len * p))
            if mask_param < 1:
                continue
            u = torch.rand(2, n, device=device).cpu()
            value = u[0] * mask_param                                  # torchaudio mask_along_axis_iid arithmetic
            start = (u[1] * (axis_len - value)).long()
            want[:, 2 * col] = start.to(torch.int32)
            want[:, 2 * col + 1] = (start + value.long()).to(torch.int32)
        assert torch.equal(got_b, want), (B, n_freq, n_time, iid)
    # the seeded form (what the training step uses): uniforms from the kernels' counter-based generator, same mask arithmetic
    for (B, n_freq, n_time, f_l, f_p, t_l, t_p, iid) in ((48, 128, 626, 10, 0.2, 5, 0.2, True), (5, 128, 626, 10, 0.2, 5, 0.2, False),
                                                          (7, 64, 40, 10, 0.2, 5, 0.0, True)):
        seed = 987654321 + B
        got_b = Fh.specaug_bounds(B, n_freq, n_time, f_l, f_p, t_l, t_p, device, iid_masks=iid, seed=seed).cpu()
        n = B if iid else 1
        u = np_hash_uniform(4 * n, seed).reshape(n, 4)
        want = torch.zeros(B, 4, dtype=torch.int32)
        for col, (cap, p, axis_len) in enumerate(((f_l, f_p, n_freq), (t_l, t_p, n_time))):
            mask_param = min(cap, int(axis_len * p))
            if mask_param < 1:
                continue
            value = u[:, 2 * col] * mask_param
            start = (u[:, 2 * col + 1] * (axis_len - value)).long()
            want[:, 2 * col] = start.to(torch.int32)
            want[:, 2 * col + 1] = (start + value.long()).to(torch.int32)
        assert torch.equal(got_b, want), ("seeded", B, n_freq, n_time, iid)
        if f_p > 0:
            assert (got_b[:, 1] - got_b[:, 0]).max().item() <= min(f_l, int(n_freq * f_p)) and got_b[:, 1].max().item() <= n_freq
    # batched in-place mixup (one launch, no scratch copies) == the one-group kernel, incl. the no-op sentinel and hard labels
    xs = [O.lcg_fill((6, 8, 20), 5, 1.0, 1.5), (O.lcg_fill((6, 10, 7), 6, 0.5, 0.5) < 0.3).float(), O.lcg_fill((12, 5, 9), 8, 1.0),
          (O.lcg_fill((12, 10), 9, 0.5, 0.5) < 0.3).float()]
    perms = [perm, perm, torch.tensor([11, 3, 0, 7, 1, 10, 2, 9, 8, 4, 6, 5]), torch.tensor([11, 3, 0, 7, 1, 10, 2, 9, 8, 4, 6, 5])]
    cs, modes = [0.3721, 0.3721, 0.91, 0.91], [0, 1, 0, 2]
    dev_x = [to(dev, t.clone()) for t in xs]
    Fh.mixup_multi_([(dx, pm, cc, md, None, None) for dx, pm, cc, md in zip(dev_x, perms, cs, modes)])
    for t, dx, pm, cc, md in zip(xs, dev_x, perms, cs, modes):
        want = Fh.mixup_(to(dev, t.clone()), pm, cc, mode=md)
        assert torch.equal(dx.cpu(), want.cpu()), md
    # a frame-major (B, F, T) view, as the trainer passes the features
    base = O.lcg_fill((6, 20, 8), 15, 1.0)
    v1, v2 = to(dev, base.clone()).transpose(1, 2), to(dev, base.clone()).transpose(1, 2)
    Fh.mixup_multi_([(v1, perm, 0.25, 0, None, None)])
    Fh.mixup_(v2, perm, 0.25, mode=0)
    assert torch.equal(v1.cpu(), v2.cpu())
    # weak labels of the weakly annotated clips
    lab = (O.lcg_fill((5, 10, 17), 3, 0.5, 0.5) < 0.1).float()
    lab[1] = 0
    assert torch.equal(Fh.weak_labels(to(dev, lab)).cpu(), (lab.sum(-1) > 0).float())


# ------------------------------------------------------------------------------------------------
# CNN block (K6)
# ------------------------------------------------------------------------------------------------
def np_keep_mask(shape, seed, p):
    """Host replica of sed_keep() (csrc/sed_common.h) over a (B,T,F,C) channels-last element index -> float 0/1 mask.
    64-bit integer tensor arithmetic with explicit 32-bit wrap-around (torch: multi-threaded; the 60 M-element masks of the
    full-size cases take a second instead of several)."""
    n = int(np.prod(shape))
    thr = int(round(p * (1 << 24))) if p > 0 else 0
    M = 0xFFFFFFFF
    sm = int(seed) & M                                   # the seed goes through the murmur3 finaliser (scalar side of sed_hash)
    sm ^= sm >> 16; sm = (sm * 0x85EBCA6B) & M
    sm ^= sm >> 13; sm = (sm * 0xC2B2AE35) & M
    sm ^= sm >> 16
    x = torch.arange(n, dtype=torch.int64)
    x = (x * 0x9E3779B1 + sm) & M
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & M
    return ((x >> 8) >= thr).to(torch.float32).reshape(shape)


def np_hash_uniform(n, seed):
    """Host replica of the kernels' counter-based uniforms: (sed_hash(i, seed) >> 8) / 2^24 for i < n, as float32."""
    M = 0xFFFFFFFF
    sm = int(seed) & M
    sm ^= sm >> 16; sm = (sm * 0x85EBCA6B) & M
    sm ^= sm >> 13; sm = (sm * 0xC2B2AE35) & M
    sm ^= sm >> 16
    x = torch.arange(n, dtype=torch.int64)
    x = (x * 0x9E3779B1 + sm) & M
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & M
    return (x >> 8).to(torch.float32) * np.float32(1.0 / 16777216.0)


def case_cnn_prologue(dev):
    """sed_cnn_prologue_bf16 (one launch) == its three parts (weight packs, seeded SpecAugment bands, copy of the input), bit for bit --
    every combination of the optional parts, a copy whose length is no multiple of four floats, one draw for the whole batch."""
    from desed_task_amd import ops
    shapes = [(32, 16), (64, 32), (128, 64), (128, 128)]
    ws = [to(dev, O.lcg_fill((co, ci, 3, 3), 31 + i, 0.2)) for i, (co, ci) in enumerate(shapes)]
    for need_dgrad in (True, False):
        want = ops.pack_conv_weights(ws, need_dgrad, "bf16x3")
        for (B, n_freq, n_time, iid, n_copy) in ((6, 128, 40, True, 6 * 40 * 128), (5, 64, 33, False, 5 * 33 * 64 + 3), (3, 16, 9, True, 0)):
            seed = 424242 + B
            req = Fh.specaug_request(B, n_freq, n_time, 10, 0.2, 5, 0.2, iid, seed)
            want_b = Fh.specaug_bounds(B, n_freq, n_time, 10, 0.2, 5, 0.2, torch.device(dev), iid_masks=iid, seed=seed)
            src = to(dev, O.lcg_fill((max(n_copy, 1),), 77, 1.0))[:n_copy]
            for with_bounds in (True, False):
                for with_copy in (True, False):
                    if not (with_bounds or with_copy):
                        continue
                    pro = {}
                    if with_bounds:
                        got_b = torch.full((B, 4), -7, dtype=torch.int32, device=dev)
                        pro["bounds"] = dict(req, out=got_b)
                    if with_copy:
                        dst = torch.full((n_copy + 5,), float("nan"), device=dev)
                        pro["copy"] = (src, dst[:n_copy])
                    got = ops.pack_conv_weights(ws, need_dgrad, "bf16x3", prologue=pro)
                    for (gf, gd), (wf, wd) in zip(got, want):
                        assert torch.equal(gf.cpu().view(torch.int32), wf.cpu().view(torch.int32))
                        assert (gd is None) == (wd is None)
                        if gd is not None:
                            assert torch.equal(gd.cpu().view(torch.int32), wd.cpu().view(torch.int32))
                    if with_bounds:
                        assert torch.equal(got_b.cpu(), want_b.cpu()), (B, iid)
                    if with_copy:
                        assert torch.equal(dst[:n_copy].cpu(), src.cpu()) and bool(torch.isnan(dst[n_copy:]).all()), n_copy
    assert Fh.specaug_request(4, 8, 4, 10, 0.0, 5, 0.2, True, 1) is None           # neither mask can be longer than 0: no draw, no seed


def case_backward_entries_whole_and_split(dev):
    """The two backward entry points whose optimizer-only half can run on its own (sed_head_bwd + sed_head_bwd_reduce, sed_gru_bwd +
    sed_gru_bias_reduce): the one-call form a C caller binds == the split form the Python op layer launches, bit for bit."""
    from desed_task_amd import _lib
    lib = _lib.get()
    f32 = dict(device=dev, dtype=torch.float32)
    st = None if dev == "cpu" else torch.cuda.current_stream().cuda_stream
    # ---- head ----
    B, T, D, NC = 3, 70, 256, 10
    x = to(dev, O.lcg_fill((B, T, D), 3, 1.0)); w1 = to(dev, O.lcg_fill((NC, D), 4, 0.05)); w2 = to(dev, O.lcg_fill((NC, D), 5, 0.05))
    b1 = to(dev, O.lcg_fill((NC,), 6, 0.1)); b2 = to(dev, O.lcg_fill((NC,), 7, 0.1))
    strong, psoft = torch.empty(B, T, NC, **f32), torch.empty(B, T, NC, **f32)
    weak, den = torch.empty(B, NC, **f32), torch.empty(B, NC, **f32)
    seed, thr24, dscale = 1234, 1 << 23, 2.0
    lib.call("sed_head_fwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), strong.data_ptr(), psoft.data_ptr(),
             weak.data_ptr(), den.data_ptr(), B, T, D, NC, seed, thr24, dscale, None, None, None, st)
    ds = to(dev, O.lcg_fill((B, T, NC), 8, 1.0)); dw = to(dev, O.lcg_fill((B, NC), 9, 1.0))
    n_scr = int(lib.value("sed_head_bwd_scratch_floats", B, T, D, NC))
    outs = []
    for split in (False, True):
        dx = torch.full((B, T, D), float("nan"), **f32)
        g = [torch.full((NC, D), float("nan"), **f32), torch.full((NC, D), float("nan"), **f32), torch.full((NC,), float("nan"), **f32),
             torch.full((NC,), float("nan"), **f32)]
        scr = torch.full((n_scr,), float("nan"), **f32)
        ptrs = [None] * 4 if split else [t.data_ptr() for t in g]
        lib.call("sed_head_bwd", x.data_ptr(), w1.data_ptr(), w2.data_ptr(), strong.data_ptr(), psoft.data_ptr(), weak.data_ptr(),
                 den.data_ptr(), ds.data_ptr(), dw.data_ptr(), dx.data_ptr(), ptrs[0], ptrs[1], ptrs[2], ptrs[3], B, T, D, NC, seed, thr24,
                 dscale, None, None, None, scr.data_ptr(), st)
        if split:
            assert all(bool(torch.isnan(t).all()) for t in g)          # untouched until the second half runs
            lib.call("sed_head_bwd_reduce", scr.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), B, T, D, NC, st)
        outs.append([dx.cpu()] + [t.cpu() for t in g])
    for a, b_ in zip(*outs):
        assert not bool(torch.isnan(a).any()) and torch.equal(a, b_)
    # ---- losses: sed_mt_loss == sed_mt_loss_records + sed_mt_loss_finish ----
    B, T, NC, ns, nw = 6, 11, 10, 2, 2
    ss = torch.sigmoid(to(dev, O.lcg_fill((B, T, NC), 21, 2.0))); st_ = torch.sigmoid(to(dev, O.lcg_fill((B, T, NC), 22, 2.0)))
    ws = torch.sigmoid(to(dev, O.lcg_fill((B, NC), 23, 2.0))); wt = torch.sigmoid(to(dev, O.lcg_fill((B, NC), 24, 2.0)))
    lab = to(dev, (O.lcg_fill((B, NC, T), 25, 0.5, 0.5) < 0.3).float()); labw = to(dev, (O.lcg_fill((nw, NC), 26, 0.5, 0.5) < 0.3).float())
    outs = []
    for split in (False, True):
        sc = torch.full((16,), float("nan"), **f32)
        gs, gw = torch.full((B, T, NC), float("nan"), **f32), torch.full((B, NC), float("nan"), **f32)
        work = torch.zeros(8 * B + 1, **f32)
        lib.call("sed_mt_loss_records" if split else "sed_mt_loss", ss.data_ptr(), ws.data_ptr(), st_.data_ptr(), wt.data_ptr(), lab.data_ptr(),
                 labw.data_ptr(), sc.data_ptr(), gs.data_ptr(), gw.data_ptr(), B, T, NC, ns, nw, 1.7, None, 0, 0, None, work.data_ptr(), st)
        if split:
            assert bool(torch.isnan(sc).all())                   # the first half leaves the scalars alone
            lib.call("sed_mt_loss_finish", work.data_ptr(), sc.data_ptr(), B, st)
        outs.append([sc[:9].cpu(), gs.cpu(), gw.cpu()])
    for a, b_ in zip(*outs):
        assert not bool(torch.isnan(a).any()) and torch.equal(a, b_)
    # ---- BiGRU recurrence ----
    B, T, H = 3, 9, 128
    gi = to(dev, O.lcg_fill((B, T, 2, 3 * H), 11, 0.5))
    whh = [to(dev, O.lcg_fill((3 * H, H), 12 + d, 0.08)) for d in range(2)]
    bhh = [to(dev, O.lcg_fill((3 * H,), 14 + d, 0.1)) for d in range(2)]
    out, saved = torch.empty(B, T, 2 * H, **f32), torch.empty(B, T, 2, 4, H, **f32)
    lib.call("sed_gru_fwd", gi.data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), bhh[0].data_ptr(), bhh[1].data_ptr(), out.data_ptr(),
             saved.data_ptr(), B, T, H, st)
    dout = to(dev, O.lcg_fill((B, T, 2 * H), 16, 1.0))
    outs = []
    for split in (False, True):
        dgi, dgh = torch.full((B, T, 2, 3 * H), float("nan"), **f32), torch.full((B, T, 2, 3 * H), float("nan"), **f32)
        hprev = torch.full((B, T, 2, H), float("nan"), **f32)
        db = [torch.full((3 * H,), float("nan"), **f32) for _ in range(4)]
        scr = torch.full((2 * B * 6 * H,), float("nan"), **f32)
        ptrs = [None] * 4 if split else [t.data_ptr() for t in db]
        lib.call("sed_gru_bwd", dout.data_ptr(), out.data_ptr(), saved.data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dgi.data_ptr(),
                 dgh.data_ptr(), hprev.data_ptr(), ptrs[0], ptrs[1], ptrs[2], ptrs[3], B, T, H, scr.data_ptr(), st)
        if split:
            assert all(bool(torch.isnan(t).all()) for t in db)
            lib.call("sed_gru_bias_reduce", scr.data_ptr(), db[0].data_ptr(), db[1].data_ptr(), db[2].data_ptr(), db[3].data_ptr(), B, H, st)
        outs.append([dgi.cpu(), dgh.cpu(), hprev.cpu()] + [t.cpu() for t in db])
    for a, b_ in zip(*outs):
        assert not bool(torch.isnan(a).any()) and torch.equal(a, b_)


def case_cnn_block(dev, layer, B, T, F, training=True, dropout_p=0.5, seed=1234, tol=2e-5, precision="f32", block0_fused=None):
    import torch.nn.functional as TF
    from desed_task_amd.ops import ConvBlockFn, pack_conv_weights
    filt = (1,) + O.NB_FILTERS
    CIN, COUT = filt[layer], filt[layer + 1]
    PT, PF = O.POOLING[layer]
    k = 100 * layer
    x = O.lcg_fill((B, T, F, CIN), k + 1, 1.0)
    w = O.lcg_fill((COUT, CIN, 3, 3), k + 2, 1.0 / np.sqrt(9 * CIN))
    bias = O.lcg_fill((COUT,), k + 3, 0.3)
    gam = O.lcg_fill((COUT,), k + 4, 0.25, 1.0)
    bet = O.lcg_fill((COUT,), k + 5, 0.2)
    wg = O.lcg_fill((COUT, COUT), k + 6, 1.0 / np.sqrt(COUT))
    bg = O.lcg_fill((COUT,), k + 7, 0.2)
    rm = O.lcg_fill((COUT,), k + 8, 0.2)
    rv = O.lcg_fill((COUT,), k + 9, 0.3, 1.0)
    gout = O.lcg_fill((B, T // PT, F // PF, COUT), k + 10, 1.0)
    bounds = torch.tensor([[1, 3, 2, 4]] * B, dtype=torch.int32) if layer == 0 else None

    # ---- oracle (NCHW, torch autograd on CPU) ----
    params = [t.clone().requires_grad_(True) for t in (w, bias, gam, bet, wg, bg)]
    xo = x.clone()
    if layer == 0:
        xo = O.specaug_apply(xo[..., 0].transpose(1, 2), (bounds[:, 0].long(), bounds[:, 1].long()),
                             (bounds[:, 2].long(), bounds[:, 3].long())).transpose(1, 2).unsqueeze(-1)
    xo = xo.permute(0, 3, 1, 2).contiguous().requires_grad_(layer > 0)
    rm_o, rv_o = rm.clone(), rv.clone()
    h = TF.conv2d(xo, params[0], params[1], padding=1)
    h = TF.batch_norm(h, rm_o, rv_o, params[2], params[3], training=training, momentum=O.BN_MOMENTUM, eps=O.BN_EPS)
    lin = TF.linear(h.permute(0, 2, 3, 1), params[4], params[5]).permute(0, 3, 1, 2)
    h = lin * torch.sigmoid(h)
    if dropout_p > 0:
        keep = np_keep_mask((B, T, F, COUT), seed, dropout_p).permute(0, 3, 1, 2)
        h = h * keep / (1 - dropout_p)
    ref = TF.avg_pool2d(h, (PT, PF))
    ref.backward(gout.permute(0, 3, 1, 2))

    # ---- HIP ----
    xd = to(dev, x[..., 0].contiguous() if layer == 0 else x).requires_grad_(layer > 0)
    pd = [to(dev, t).requires_grad_(True) for t in (w, bias, gam, bet, wg, bg)]
    rm_d, rv_d = to(dev, rm.clone()), to(dev, rv.clone())
    cfg = dict(pool=(PT, PF), bn_training=training, dropout_p=dropout_p, apply_dropout=dropout_p > 0, seed=seed,
               bounds=to(dev, bounds) if bounds is not None else None, update_running=True, conv_precision=precision)
    if block0_fused is not None:            # first block: fused (pre-BN tensor never in HBM) vs the unfused kernels
        cfg["block0_fused"] = block0_fused
    if layer > 0 and precision != "f32":
        cfg["packed"] = pack_conv_weights([pd[0].detach()], True, precision)[0]
    out = ConvBlockFn.apply(xd, *pd, rm_d, rv_d, cfg)
    out.backward(to(dev, gout))

    def cmp(name, a, b, scale=None):
        a, b = a.detach().cpu(), b.detach().cpu()
        s = scale if scale is not None else max(1.0, b.abs().max().item())
        err = (a - b).abs().max().item() / s
        assert err < tol, "%s layer %d: rel err %.3e (max ref %.3e)" % (name, layer, err, b.abs().max().item())

    cmp("out", out, ref.permute(0, 2, 3, 1))
    cmp("running_mean", rm_d, rm_o)
    cmp("running_var", rv_d, rv_o)
    names = ("conv_w", "conv_b", "bn_w", "bn_b", "glu_w", "glu_b")
    for nm, a, b in zip(names, pd, params):
        if nm == "conv_b" and training:
            assert a.grad.abs().max().item() < 1e-3 * max(1.0, pd[0].grad.abs().max().item())   # analytically zero
            continue
        cmp("d_" + nm, a.grad, b.grad)
    if layer > 0:
        cmp("dx", xd.grad, xo.grad.permute(0, 2, 3, 1))


# ------------------------------------------------------------------------------------------------
# BiGRU (K7)
# ------------------------------------------------------------------------------------------------
def case_bigru(dev, B=2, T=7, I=128, tol=2e-5, H=128):
    from desed_task_amd.ops import BiGRULayerFn
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
    shapes = {"weight_ih": (3 * H, I), "weight_hh": (3 * H, H), "bias_ih": (3 * H,), "bias_hh": (3 * H,)}
    ws = []
    k = 500 + I
    for sfx in ("", "_reverse"):
        for nm in names:
            k += 1
            ws.append(O.lcg_fill(shapes[nm], k, 1.0 / np.sqrt(H)))
    x = O.lcg_fill((B, T, I), 41, 1.0)
    gout = O.lcg_fill((B, T, 2 * H), 42, 1.0)
    # oracle
    wo = [w.clone().requires_grad_(True) for w in ws]
    xo = x.clone().requires_grad_(True)
    ref, _ = torch._VF.gru(xo, torch.zeros(2, B, H), wo, True, 1, 0.0, False, True, True)
    ref.backward(gout)
    # HIP
    wd = [to(dev, w).requires_grad_(True) for w in ws]
    xd = to(dev, x).requires_grad_(True)
    out = BiGRULayerFn.apply(xd, *wd)
    out.backward(to(dev, gout))

    def cmp(name, a, b):
        a, b = a.detach().cpu(), b.detach().cpu()
        err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
        assert err < tol, "%s: rel err %.3e" % (name, err)

    cmp("out", out, ref)
    cmp("dx", xd.grad, xo.grad)
    for i, (a, b) in enumerate(zip(wd, wo)):
        cmp("dw%d" % i, a.grad, b.grad)


def case_gemm(dev, entry="sed_gemm"):
    lib = _lib.get()
    g = torch.Generator().manual_seed(5)
    shapes = ((130, 70, 45, 0, 1, 1), (130, 200, 64, 0, 0, 1), (96, 40, 300, 1, 0, 4), (33, 384, 128, 0, 1, 1))
    if entry != "sed_gemm":         # 16-byte-friendly shapes that stay on the split-bf16 kernels, all three layouts, ragged tiles
        shapes = ((132, 72, 48, 0, 1, 1), (132, 200, 64, 0, 0, 1), (96, 40, 300, 1, 0, 4), (36, 384, 128, 0, 1, 1),
                  (384, 128, 520, 1, 0, 3), (260, 256, 384, 0, 0, 1)) + shapes[:1]
    for (M, N, K, ta, tb, split) in shapes:
        A = torch.randn((K, M) if ta else (M, K), generator=g)
        Bm = torch.randn((N, K) if tb else (K, N), generator=g)
        bias = torch.randn(N, generator=g)
        ref = (A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double() + bias.double()
        Ad, Bd, bd = to(dev, A, Bm, bias)
        C = torch.zeros(M, N, device=Ad.device)
        lib.call(entry, Ad.data_ptr(), Bd.data_ptr(), bd.data_ptr(), C.data_ptr(), M, N, K, A.shape[1], Bm.shape[1], N, ta, tb,
                 split, 0, _lib.stream_ptr(Ad))
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 1e-4 * max(1.0, ref.abs().max().item()), (M, N, K, ta, tb, err)
    # K-concatenated B operand (dX of a bidirectional GRU layer): C = A . [B0 ; B1]
    kcat = "sed_gemm_kcat" if entry == "sed_gemm" else "sed_gemm_kcat_bf16x3"
    for (M, N, K, ks) in ((132, 128, 768, 384), (70 * 4, 256, 96, 64)):
        A = torch.randn(M, K, generator=g)
        B0, B1 = torch.randn(ks, N, generator=g), torch.randn(K - ks, N, generator=g)
        ref = A.double() @ torch.cat([B0, B1]).double()
        Ad, B0d, B1d = to(dev, A, B0, B1)
        C = torch.full((M, N), 7.0, device=Ad.device)             # overwritten, not accumulated
        lib.call(kcat, Ad.data_ptr(), B0d.data_ptr(), B1d.data_ptr(), C.data_ptr(), M, N, K, ks, K, N, N, _lib.stream_ptr(Ad))
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 1e-4 * max(1.0, ref.abs().max().item()), (kcat, M, N, K, err)


def case_linear_n96_tile(dev, shapes=((300, 192, 64, 0), (513, 96, 96, 1), (130, 768, 32, 0))):
    """sed_linear_bf16x3 through the 128 x 96 tile of the split-bf16 GEMM (picked on its own for BEATs' N = 768 layers at M = 23 808,
    where 128 x 128 tiles leave the second round of resident workgroups half empty; forced here with the tuning key) vs float64."""
    lib = _lib.get()
    g = torch.Generator().manual_seed(12)
    _lib.set_tuning("gemm_ntn", 3)
    try:
        for (M, N, K, act) in shapes:
            A, W, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
            ref = A.double() @ W.double().t() + bias.double()
            if act:
                ref = torch.nn.functional.gelu(ref)
            Ad, Wd, bd = to(dev, A, W, bias)
            C = torch.full((M, N), 7.0, device=Ad.device)
            lib.call("sed_linear_bf16x3", Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), C.data_ptr(), M, N, K, act, _lib.stream_ptr(Ad))
            err = (C.cpu().double() - ref).abs().max().item()
            assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, act, err)
    finally:
        _lib.set_tuning("gemm_ntn", 0)


def case_linear_packed(dev, shapes=((300, 128, 64, 0), (513, 256, 96, 1), (256, 128, 32, 1))):
    """sed_pack_weights_bf16x3 + sed_linear_packed_bf16x3 (the BEATs encoder's large Linear layers: frozen weight split into bf16
    hi / lo planes once, 256 x 128 tiles, A fragments straight from HBM, optional exact-GELU epilogue) vs float64: ragged M (rows
    past M are computed and dropped), several K tiles, more tiles than one XCD's share."""
    lib = _lib.get()
    g = torch.Generator().manual_seed(11)
    for (M, N, K, act) in shapes:
        A, W, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
        ref = A.double() @ W.double().t() + bias.double()
        if act:
            ref = torch.nn.functional.gelu(ref)
        Ad, Wd, bd = to(dev, A, W, bias)
        Wp = torch.empty(2 * N * K, dtype=torch.int16, device=Ad.device)
        lib.call("sed_pack_weights_bf16x3", Wd.data_ptr(), Wp.data_ptr(), N, K, _lib.stream_ptr(Ad))
        hi = Wp[:N * K].view(torch.bfloat16).float().cpu().view(N, K)
        lo = Wp[N * K:].view(torch.bfloat16).float().cpu().view(N, K)
        assert (hi + lo - W).abs().max().item() <= 2.0 ** -16 * W.abs().max().item()
        C = torch.full((M, N), 7.0, device=Ad.device)
        lib.call("sed_linear_packed_bf16x3", Ad.data_ptr(), Wp.data_ptr(), bd.data_ptr(), C.data_ptr(), M, N, K, act, _lib.stream_ptr(Ad))
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, act, err)
        # no bias; shapes the kernel does not take are refused, not mangled
        lib.call("sed_linear_packed_bf16x3", Ad.data_ptr(), Wp.data_ptr(), None, C.data_ptr(), M, N, K, 0, _lib.stream_ptr(Ad))
        err = (C.cpu().double() - A.double() @ W.double().t()).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, "nobias", err)
    try:
        lib.call("sed_linear_packed_bf16x3", Ad.data_ptr(), Wp.data_ptr(), None, C.data_ptr(), M, 100, K, 0, _lib.stream_ptr(Ad))
        raise AssertionError("N % 128 != 0 must be refused")
    except RuntimeError:
        pass


def tile_image(X, bk=16):
    """Independent host restatement of the image sed_split_tiles_bf16x3 documents in include/sed_hip.h (X: (R, K) float32 on the CPU ->
    int16 tensor of 2 * ceil(R / 256) * 256 * K bf16 bit patterns): block (r // 256, k // bk) = [hi | lo][256][bk], octet o of row r at
    slot o ^ ((r >> 3) & 1); rows >= R zero."""
    R, K = X.shape
    P = (R + 255) // 256
    Xp = torch.zeros(P * 256, K)
    Xp[:R] = X
    hi = Xp.to(torch.bfloat16)
    lo = (Xp - hi.float()).to(torch.bfloat16)
    out = []
    rows = torch.arange(256)
    for plane in (hi, lo):
        v = plane.view(torch.int16).view(P, 256, K // bk, bk // 8, 8)             # (panel, row, ktile, octet, 8)
        sw = ((rows >> 3) & 1).view(1, 256, 1, 1, 1)
        octs = torch.arange(bk // 8).view(1, 1, 1, bk // 8, 1)
        src = (octs ^ sw).expand(P, 256, K // bk, bk // 8, 8)                       # slot s holds octet s ^ sw
        out.append(torch.gather(v, 3, src.contiguous()))
    img = torch.stack(out, 0)                                                       # (plane, panel, row, ktile, slot, 8)
    return img.permute(1, 3, 0, 2, 4, 5).contiguous().view(-1)                      # (panel, ktile, plane, row, slot, 8)


def tile_unimage(img, R, K, bk=16):
    """Inverse of tile_image: int16 image -> (R, K) float32 = hi + lo."""
    P = (R + 255) // 256
    v = img.view(P, K // bk, 2, 256, bk // 8, 8)                                    # (panel, ktile, plane, row, slot, 8)
    rows = torch.arange(256)
    sw = ((rows >> 3) & 1).view(1, 1, 1, 256, 1, 1)
    octs = torch.arange(bk // 8).view(1, 1, 1, 1, bk // 8, 1)
    src = (octs ^ sw).expand(P, K // bk, 2, 256, bk // 8, 8)                       # octet o sits in slot o ^ sw
    planes = torch.gather(v, 4, src.contiguous()).view(torch.bfloat16).float()      # (panel, ktile, plane, row, octet, 8)
    full = planes.permute(2, 0, 3, 1, 4, 5).contiguous().view(2, P * 256, K)
    return (full[0] + full[1])[:R]


def case_linear_tiles(dev, shapes=((300, 256, 64, 0), (513, 512, 96, 1), (256, 256, 16, 1), (700, 768, 160, 0), (2100, 256, 48, 0)), form=0):
    """sed_split_tiles_bf16x3 + sed_linear_tiles_bf16x3 (round 6: both operands as K-tiled bf16 hi / lo images, four LDS stages filled by
    LDS-DMA three K tiles ahead, two wave groups one barrier apart): the image bit for bit against the host restatement above, the
    product vs float64 -- ragged M (zero rows in the image, never stored), one to ten K tiles (fewer than the pipeline's depth
    included), several N tiles and more row panels than XCDs, GELU epilogue, no bias."""
    lib = _lib.get()
    g = torch.Generator().manual_seed(12)
    FORM = form          # sed_set_tuning("linear_tiles"): 0 = the shipped eight-wave form, 5 = the loader-wave form
    _lib.set_tuning("linear_tiles", FORM)
    for (M, N, K, act) in shapes:
        A, W, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
        ref = A.double() @ W.double().t() + bias.double()
        if act:
            ref = torch.nn.functional.gelu(ref)
        Ad, Wd, bd = to(dev, A, W, bias)
        st = _lib.stream_ptr(Ad)
        At = torch.full((2 * ((M + 255) // 256) * 256 * K,), 77, dtype=torch.int16, device=Ad.device)
        Wt = torch.full((2 * N * K,), 77, dtype=torch.int16, device=Ad.device)
        lib.call("sed_split_tiles_bf16x3", Ad.data_ptr(), At.data_ptr(), M, K, st)
        lib.call("sed_split_tiles_bf16x3", Wd.data_ptr(), Wt.data_ptr(), N, K, st)
        assert torch.equal(At.cpu(), tile_image(A)), (M, K, "A image")
        assert torch.equal(Wt.cpu(), tile_image(W)), (N, K, "W image")
        C = torch.full((M, N), 7.0, device=Ad.device)
        lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), bd.data_ptr(), C.data_ptr(), M, N, K, act, st)
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, act, err)
        lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), None, C.data_ptr(), M, N, K, 0, st)
        err = (C.cpu().double() - A.double() @ W.double().t()).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, "nobias", err)
        # the product written as the next Linear's image (fc1 -> fc2): hi + lo of every element within 2^-16 relative of the fp32 result
        Ct = torch.full((2 * ((M + 255) // 256) * 256 * N,), 77, dtype=torch.int16, device=Ad.device)
        lib.call("sed_linear_tiles_out_bf16x3", At.data_ptr(), Wt.data_ptr(), bd.data_ptr(), Ct.data_ptr(), M, N, K, act, st)
        got = tile_unimage(Ct.cpu(), M, N)
        err = (got.double() - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, act, "image out", err)
        # two K halves as two partial sums (sed_linear_tiles_split2_bf16x3): C2[0] + C2[1] = the product, the bias in the first
        if (K // 16) % 2 == 0 and not act:
            C2 = torch.full((2, M, N), 7.0, device=Ad.device)
            lib.call("sed_linear_tiles_split2_bf16x3", At.data_ptr(), Wt.data_ptr(), bd.data_ptr(), C2.data_ptr(), M, N, K, st)
            err = ((C2[0] + C2[1]).cpu().double() - ref).abs().max().item()
            assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, "split2", err)
            half = A[:, :K // 2].double() @ W[:, :K // 2].double().t() + bias.double()
            assert (C2[0].cpu().double() - half).abs().max().item() < 3e-5 * max(1.0, half.abs().max().item()), (M, N, K, "split2 first half")
        # a few workgroups only: every workgroup walks several tiles (the persistent loop's tile hand-over and its DMA cursor)
        _lib.set_tuning("linear_tiles", 16 if FORM == 0 else 16 + 1)     # (odd grid requests: the loader-wave form, even: the eight-wave form)
        try:
            C.fill_(7.0)
            lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), bd.data_ptr(), C.data_ptr(), M, N, K, act, st)
        finally:
            _lib.set_tuning("linear_tiles", FORM)
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (M, N, K, act, "16 workgroups", err)
    try:
        lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), None, C.data_ptr(), M, 128, K, 0, st)
        raise AssertionError("N % 256 != 0 must be refused")
    except RuntimeError:
        pass
    finally:
        _lib.set_tuning("linear_tiles", 0)


def case_layernorm_tiles(dev, shapes=((300, 256), (513, 768), (70, 1024))):
    """sed_layernorm_tiles: y = LayerNorm(x (+ x2) + alpha * res) as fp32 AND as the K-tiled bf16 hi / lo image (rows < M), against
    torch.nn.functional.layer_norm in float64 and against the host restatement of the image (hi + lo of y within 2^-16 relative)."""
    lib = _lib.get()
    g = torch.Generator().manual_seed(14)
    for (M, D) in shapes:
        x, x2, res = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
        gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
        xd, x2d, rd, gd, bd = to(dev, x, x2, res, gamma, beta)
        st = _lib.stream_ptr(xd)
        for (use2, user, alpha) in ((False, False, 1.0), (True, True, 1.7), (False, True, 0.5)):
            pre = x.double() + (x2.double() if use2 else 0) + (alpha * res.double() if user else 0)
            ref = torch.nn.functional.layer_norm(pre, (D,), gamma.double(), beta.double(), 1e-5)
            y = torch.full((M, D), 7.0, device=xd.device)
            yt = torch.zeros(2 * ((M + 255) // 256) * 256 * D, dtype=torch.int16, device=xd.device)
            lib.call("sed_layernorm_tiles", xd.data_ptr(), x2d.data_ptr() if use2 else None, rd.data_ptr() if user else None, float(alpha), gd.data_ptr(),
                     bd.data_ptr(), y.data_ptr(), yt.data_ptr(), M, D, 1e-5, st)
            tol = 2e-5 * max(1.0, ref.abs().max().item())
            assert (y.cpu().double() - ref).abs().max().item() < tol, (M, D, use2, user)
            got = tile_unimage(yt.cpu(), M, D)
            assert (got.double() - y.cpu().double()).abs().max().item() <= 2.0 ** -15 * max(1.0, ref.abs().max().item()), (M, D, "image")


def case_linear_tiles_race_screen(dev, reps=600):
    """The LDS-DMA Linear orders its stages by counted vmcnt waits and raw barriers only (sed_gemm_bf16.hip): a hazard there would show as a
    rare wrong tile that comes and goes with timing.  `reps` launches of the two production shapes with the deepest pipelines (QKV: K = 768,
    837 tiles; fc2: K = 3072) must return the bits of the first launch every time -- alone, and while a second stream keeps the CUs busy with
    the generic split-bf16 GEMM (other workgroups' LDS and VMEM traffic beside the persistent ones)."""
    lib = _lib.get()
    g = torch.Generator().manual_seed(13)
    M = 23808
    side = torch.cuda.Stream()
    Xs = torch.randn(4096, 768, generator=g).to(dev); Ws = torch.randn(768, 768, generator=g).to(dev); Ys = torch.empty(4096, 768, device=dev)
    for (N, K, act) in ((2304, 768, 0), (768, 3072, 1)):
        A, W, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
        Ad, Wd, bd = to(dev, A, W, bias)
        st = _lib.stream_ptr(Ad)
        At = torch.empty(2 * ((M + 255) // 256) * 256 * K, dtype=torch.int16, device=Ad.device)
        Wt = torch.empty(2 * N * K, dtype=torch.int16, device=Ad.device)
        lib.call("sed_split_tiles_bf16x3", Ad.data_ptr(), At.data_ptr(), M, K, st)
        lib.call("sed_split_tiles_bf16x3", Wd.data_ptr(), Wt.data_ptr(), N, K, st)
        C0 = torch.empty(M, N, device=Ad.device)
        lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), bd.data_ptr(), C0.data_ptr(), M, N, K, act, st)
        ref = A[:512].double() @ W.double().t() + bias.double()
        if act:
            ref = torch.nn.functional.gelu(ref)
        assert (C0[:512].cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
        C = torch.empty_like(C0)
        bad = 0
        for rep in range(reps):
            if rep >= reps // 2:            # second half: a co-runner on another stream
                with torch.cuda.stream(side):
                    for _ in range(2):
                        lib.call("sed_linear_bf16x3", Xs.data_ptr(), Ws.data_ptr(), None, Ys.data_ptr(), 4096, 768, 768, 0, side.cuda_stream)
            lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), bd.data_ptr(), C.data_ptr(), M, N, K, act, st)
            bad += int(not torch.equal(C, C0))
        torch.cuda.synchronize()
        assert bad == 0, (N, K, "launches that differ from the first", bad, "of", reps)


# ------------------------------------------------------------------------------------------------
# whole mean-teacher step (a16): SEDTask4.training_step + EMA + backward + Adam + scheduler
# ------------------------------------------------------------------------------------------------
def recipe_config(batch_sizes=(12, 12, 24)):
    """The keys of recipes/dcase2023_task4_baseline/confs/default.yaml the step reads."""
    return {
        "training": {"batch_size": list(batch_sizes), "const_max": 2, "num_workers": 0, "ema_factor": 0.999,
                     "self_sup_loss": "mse", "mixup": "soft", "n_epochs_warmup": 50},
        "scaler": {"statistic": "instance", "normtype": "minmax", "dims": [1, 2], "savepath": None},
        "opt": {"lr": 0.001},
        "feats": {"n_mels": 128, "n_filters": 2048, "hop_length": 256, "n_window": 2048, "sample_rate": 16000, "f_min": 0, "f_max": 8000},
        "net": {"dropout": 0.5, "rnn_layers": 2, "n_in_channel": 1, "nclass": 10, "attention": True, "n_RNN_cell": 128,
                "activation": "glu", "rnn_type": "BGRU", "kernel_size": [3] * 7, "padding": [1] * 7, "stride": [1] * 7,
                "nb_filters": [16, 32, 64, 128, 128, 128, 128],
                "pooling": [[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]], "dropout_recurrent": 0, "use_embeddings": False},
    }


def build_task(dev, batch_sizes, sd, dropout=None, specaug=True, rampup=100, lr=1e-3, pretrained=False, torch_adam=False, whole_step=False,
               train_data=None):
    """torch_adam: the optimizer is built as train_sed.py:199-201 builds it (torch.optim.Adam; SEDTask4 adopts it)."""
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.arena import FusedAdam
    from desed_task_amd.utils.schedulers import ExponentialWarmup
    config = recipe_config(batch_sizes)
    if pretrained:          # recipes/dcase2023_task4_baseline/confs/pretrained.yaml: embeddings delivered with the batch
        from desed_task_amd.sed_trainer_pretrained import SEDTask4
        config["net"] = pretrained_net_config()
        config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
    else:
        from desed_task_amd.sed_trainer import SEDTask4
    net_cfg = dict(config["net"])
    if dropout is not None:
        net_cfg["dropout"] = dropout
    extra = {} if specaug else {"specaugm_t_p": 0.0, "specaugm_f_p": 0.0}
    student = CRNN(**net_cfg, **extra)
    if sd is not None:
        student.load_state_dict({k: v.clone() for k, v in sd.items()})
    student = student.to(dev) if dev != "cpu" else student
    if torch_adam:
        opt = torch.optim.Adam(student.parameters(), lr, betas=(0.9, 0.999))
    else:
        opt = FusedAdam(student.parameters(), lr=lr, betas=(0.9, 0.999), arena=student.arena)
    sched = {"scheduler": ExponentialWarmup(opt, lr, rampup), "interval": "step"}

    class Enc:
        labels = list(range(10))
    task = SEDTask4(config, Enc(), student, opt=opt, scheduler=sched, train_data=train_data)
    assert isinstance(opt, FusedAdam)
    # most cases call training_step / the drivers themselves; the whole-step surface has its own cases (case_lightning_surface)
    task.whole_step = whole_step
    task.train()
    if dev != "cpu":
        task.to(dev)
        opt.arena = task.sed_student.arena      # .to() rebuilt the arenas
    return task


def case_training_step(dev, small=False, golden=None):
    import random
    from desed_task_amd.launcher import StepDriver
    if small:
        bs, n_samp, steps = (1, 1, 2), 16000 + 1024, 2
    else:
        bs, n_samp, steps = (2, 2, 4), 16000 * 2 + 1024, 3          # == tests/golden/make_golden.py G6
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    audio = O.synth_audio(B, n_samp, seed=77)
    n_out = (1 + n_samp // 256) // 4
    labels = O.synth_labels(bs, 10, n_out, seed=5)
    task = build_task(dev, bs, sd, dropout=0.0, specaug=False, rampup=100)
    driver = StepDriver(task, world_size=1)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100)
    for step in range(steps):
        random.seed(4); np.random.seed(100 + step); torch.manual_seed(100 + step)
        assert random.random() < 0.5
        cw = np.random.beta(0.2, 0.2); pw = torch.randperm(bs[1]); cs = np.random.beta(0.2, 0.2); ps = torch.randperm(bs[0])
        mix = dict(c_weak=cw, perm_weak=pw, c_strong=cs, perm_strong=ps)
        random.seed(4); np.random.seed(100 + step); torch.manual_seed(100 + step)
        loss = driver.run_step((to(dev, audio.clone()), to(dev, labels.clone()), None, None), step)
        tot, logs = orc.training_step(audio, labels, mix=mix)
        ref_grads = orc.optimizer_step(tot)
        hip_params = dict(task.sed_student.named_parameters())
        for k in O.PARAM_KEYS:
            if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
                continue                                   # analytically zero, see below
            g, r = hip_params[k].grad.detach().cpu(), ref_grads[k]
            # step 0 starts from identical parameters: rounding-order tolerance.  Later steps inherit the
            # Adam sign-of-near-zero-gradient divergence of a few parameter elements (see the comment below).
            rel = 1e-4 if step == 0 else 3e-2
            assert (g - r).abs().max().item() <= rel * r.abs().max().item() + 5e-8, "step %d grad %s" % (step, k)
            # ... and the bulk of every tensor agrees far tighter than its worst element: median elementwise error
            emax, emed = grad_error_stats(g, r)
            if DIAG is not None:
                DIAG.append(("grad", step, k, emax, emed))
            assert emed <= (1e-5 if step == 0 else 2e-3), "step %d grad %s: median error %.3e" % (step, k, emed)
        got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
        got["loss"] = float(loss.detach().cpu())
        logs["loss"] = tot.item()
        for k in sorted(logs):
            a, b = got[k], logs[k]
            assert abs(a - b) <= 2e-5 + 2e-4 * abs(b), "step %d %s: hip %.8g oracle %.8g" % (step, k, a, b)
        if golden is not None and not small:
            keys = list(golden["g6_keys"])
            ref = golden["g6_values"][step]
            for k, b in zip(keys, ref):
                a = got[k]
                assert abs(a - b) <= 2e-5 + 5e-4 * abs(b), "step %d %s: hip %.8g reference %.8g" % (step, k, a, b)
        # frame-level posteriors: the north-star acceptance tensor (1e-3 abs)
        s_s, w_s, s_t, w_t = [t.detach().cpu() for t in task.last_outputs]
        assert (s_s - orc.last["strong_s"]).abs().max().item() < 1e-3
        assert (w_s - orc.last["weak_s"]).abs().max().item() < 1e-3
        assert (s_t - orc.last["strong_t"]).abs().max().item() < 1e-3
        assert (w_t - orc.last["weak_t"]).abs().max().item() < 1e-3
    st = dict(task.sed_student.named_parameters())
    tt = dict(task.sed_teacher.named_parameters())
    # Parameters after `steps` Adam updates.  Adam divides by sqrt(v): elements whose gradient is ~0 get an update whose
    # SIGN is decided by float rounding, so parameters are compared as update vectors (relative L2), and the gradients
    # themselves elementwise (above, every step).
    for k in O.PARAM_KEYS:
        if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
            # d(loss)/d(conv bias) is analytically 0 under train-mode BatchNorm (the bias cancels in the mean
            # subtraction).  The reference's autograd returns ~1e-9 rounding noise which Adam's normalisation turns
            # into a random walk; the HIP path returns an exact 0.  The parameter has no effect on any output.
            continue
        if ref_grads[k].abs().max().item() < 1e-4:
            continue        # gradient at rounding-noise level relative to Adam's normalisation (e.g. dense_softmax.bias)
        for mine, theirs, what in ((st[k].detach().cpu(), orc.student[k].detach(), "student"),
                                   (tt[k].detach().cpu(), orc.teacher[k], "teacher")):
            upd = (theirs - sd[k]).norm().item()
            err = (mine - theirs).norm().item()
            assert err <= 0.15 * upd + 1e-6, "%s %s: |err| %.3e vs |update| %.3e" % (what, k, err, upd)
            # the L2 bound above is dominated by the few sign-flipped elements; the typical element agrees much tighter
            med_err = (mine - theirs).abs().median().item()
            med_upd = (theirs - sd[k]).abs().median().item()
            if DIAG is not None:
                DIAG.append(("param", what, k, err / max(upd, 1e-30), med_err / max(med_upd, 1e-30)))
            assert med_err <= 0.01 * med_upd + 1e-8, "%s %s: median |err| %.3e vs median |update| %.3e" % (what, k, med_err, med_upd)
    # teacher BN buffers (updated under no_grad in train mode, never EMA'd).  running_var, not running_mean: the mean
    # contains the conv bias, whose reference value random-walks (see above).
    for i in (0, 3, 6):
        a = getattr(task.sed_teacher.cnn.cnn, "batchnorm%d" % i).running_var.cpu()
        b = orc.teacher["cnn.cnn.batchnorm%d.running_var" % i]
        assert (a - b).abs().max().item() < 2e-3 * b.abs().max().item(), "teacher running_var %d" % i
    if golden is not None and not small:
        for n in ("cnn.cnn.conv0.weight", "cnn.cnn.glu3.linear.bias", "rnn.rnn.weight_hh_l1_reverse", "dense.bias"):
            ref = golden["g6_student_after3__" + n]
            init = sd[n].numpy().reshape(-1)[:256]
            mine = st[n].detach().cpu().numpy().reshape(-1)[:256]
            assert np.linalg.norm(mine - ref) <= 0.15 * np.linalg.norm(ref - init) + 1e-6, n
    return task


# ------------------------------------------------------------------------------------------------
# drop-in CRNN module vs the REFERENCE's own outputs (golden G5: eval and train-mode posteriors, G7: gradients)
# ------------------------------------------------------------------------------------------------
def case_crnn_vs_reference_golden(dev, golden):
    from desed_task_amd.nnet.CRNN import CRNN
    sd = O.make_state_dict(seed=7)
    xin = O.lcg_fill((3, 128, 160), 21, 1.0)
    cfg = dict(recipe_config()["net"])
    # eval mode (inference path: running statistics, no dropout / SpecAugment)
    net = CRNN(**cfg)
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.to(dev) if dev != "cpu" else net
    net.eval()
    with torch.no_grad():
        strong, weak = net(to(dev, xin))
    assert tuple(strong.shape) == (3, 10, 40) and tuple(weak.shape) == (3, 10)
    assert np.abs(strong.cpu().numpy() - golden["g5_eval_strong"]).max() < 2e-5
    assert np.abs(weak.cpu().numpy() - golden["g5_eval_weak"]).max() < 2e-5
    # train mode, dropout 0, SpecAugment off: posteriors, BN running stats and gradients recorded from the reference
    cfg["dropout"] = 0.0
    net = CRNN(**cfg, specaugm_t_p=0.0, specaugm_f_p=0.0)
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.to(dev) if dev != "cpu" else net
    net.train()
    strong, weak = net(to(dev, xin))
    assert np.abs(strong.detach().cpu().numpy() - golden["g5_train_strong"]).max() < 2e-5
    assert np.abs(weak.detach().cpu().numpy() - golden["g5_train_weak"]).max() < 2e-5
    for i in range(7):
        bn = getattr(net.cnn.cnn, "batchnorm%d" % i)
        assert np.abs(bn.running_mean.cpu().numpy() - golden["g3_train_rm%d" % i]).max() < 1e-5
        assert np.abs(bn.running_var.cpu().numpy() - golden["g3_train_rv%d" % i]).max() < 1e-5
    tgt_s = to(dev, (O.lcg_fill(tuple(strong.shape), 31, 0.5, 0.5) < 0.2).float())
    tgt_w = to(dev, (O.lcg_fill(tuple(weak.shape), 32, 0.5, 0.5) < 0.3).float())
    loss = torch.nn.functional.binary_cross_entropy(strong, tgt_s) + torch.nn.functional.binary_cross_entropy(weak, tgt_w)
    assert abs(loss.item() - float(golden["g7_loss"][0])) < 2e-6
    loss.backward()
    names = list(golden["g7_param_names"])
    norms = golden["g7_grad_norms"]
    params = dict(net.named_parameters())
    for n, ref in zip(names, norms):
        if n.startswith("cnn.cnn.conv") and n.endswith(".bias"):
            continue                                  # analytically zero (see case_training_step)
        g = params[n].grad
        assert abs(g.norm().item() - ref) <= 2e-3 * ref + 1e-7, n
        key = "g7_grad__" + n
        if key in golden.files:
            got = g.detach().cpu().numpy().reshape(-1)[:512]
            assert np.abs(got - golden[key]).max() <= 1e-4 * np.abs(golden[key]).max() + 1e-7, n


def case_edge_shapes(dev):
    """Ragged / degenerate inputs: odd frame counts (AvgPool floor drops the last frame), a single clip,
    clips shorter than one pooling window, and an empty batch."""
    from desed_task_amd.nnet.CRNN import CRNN
    sd = O.make_state_dict(seed=7)
    net = CRNN(**recipe_config()["net"])
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.to(dev) if dev != "cpu" else net
    net.eval()
    for B, T in ((1, 37), (2, 9), (1, 4)):
        x = O.lcg_fill((B, 128, T), 50 + T, 1.0)
        with torch.no_grad():
            strong, weak = net(to(dev, x))
            ref_s, ref_w = O.crnn_forward(sd, x, training=False)
        assert tuple(strong.shape) == (B, 10, T // 4)
        assert (strong.cpu() - ref_s).abs().max().item() < 2e-5 and (weak.cpu() - ref_w).abs().max().item() < 2e-5
    mel = make_mel()
    empty = mel(to(dev, torch.zeros(0, 4096)))
    assert tuple(empty.shape) == (0, 128, 17)
    assert tuple(Fh.minmax_scale(empty, apply_log=True).shape) == (0, 128, 17)


def case_edge_round4_entries(dev):
    """Degenerate arguments of the entry points added in round 4: empty batches give zeroed (or untouched) outputs and SED_OK, bad
    arguments the documented error -- never a launch on a null pointer."""
    from desed_task_amd import _lib
    lib = _lib.get()
    f32 = dict(device=dev, dtype=torch.float32)
    st = None if dev == "cpu" else torch.cuda.current_stream().cuda_stream
    NC, D, H = 10, 256, 128
    g = [torch.full((NC, D), 3.0, **f32), torch.full((NC, D), 3.0, **f32), torch.full((NC,), 3.0, **f32), torch.full((NC,), 3.0, **f32)]
    lib.call("sed_head_bwd_reduce", None, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), 0, 156, D, NC, st)
    assert all(float(t.abs().max()) == 0.0 for t in g)                      # empty batch: zero gradients
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_head_bwd_reduce", None, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), 2, 156, D, NC, st)
    with pytest.raises(RuntimeError, match="unsupported"):
        lib.call("sed_head_bwd_reduce", None, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), 2, 156, 100, NC, st)
    db = [torch.full((3 * H,), 3.0, **f32) for _ in range(4)]
    lib.call("sed_gru_bias_reduce", None, db[0].data_ptr(), db[1].data_ptr(), db[2].data_ptr(), db[3].data_ptr(), 0, H, st)
    assert all(float(t.abs().max()) == 0.0 for t in db)
    lib.call("sed_gru_bias_reduce", None, None, None, None, None, 4, H, st)                     # nothing asked for: nothing done
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_gru_bias_reduce", None, db[0].data_ptr(), None, db[2].data_ptr(), db[3].data_ptr(), 4, H, st)      # half a pair
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_gru_bias_reduce", None, db[0].data_ptr(), db[1].data_ptr(), db[2].data_ptr(), db[3].data_ptr(), 4, H, st)   # no records
    # prologue: nothing to do at all is not an error; a misaligned copy and a draw for another batch size are
    lib.call("sed_cnn_prologue_bf16", 0, None, None, None, None, None, None, 0, 1, 0, 1, 0, 1, 0, None, None, None, 0, st)
    src = torch.zeros(64, **f32); dst = torch.zeros(64, **f32)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_cnn_prologue_bf16", 0, None, None, None, None, None, None, 0, 1, 0, 1, 0, 1, 0, None, src.data_ptr() + 4, dst.data_ptr(), 8, st)
    b4 = torch.zeros(4, 4, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_cnn_prologue_bf16", 0, None, None, None, None, None, b4.data_ptr(), 4, 3, 5, 128, 2, 40, 1, None, None, None, 0, st)
    # split-K dX: empty problem fine, missing scratch / a K split point off the tile grid are not
    a = torch.zeros(8, 64, **f32); b0 = torch.zeros(32, 16, **f32); c = torch.zeros(8, 16, **f32)
    lib.call("sed_gemm_kcat_splitk_bf16x3", a.data_ptr(), b0.data_ptr(), b0.data_ptr(), c.data_ptr(), 0, 16, 64, 32, 64, 16, 16, 2, None, st)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_gemm_kcat_splitk_bf16x3", a.data_ptr(), b0.data_ptr(), b0.data_ptr(), c.data_ptr(), 8, 16, 64, 32, 64, 16, 16, 2, None, st)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_gemm_kcat_splitk_bf16x3", a.data_ptr(), b0.data_ptr(), b0.data_ptr(), c.data_ptr(), 8, 16, 64, 24, 64, 16, 16, 2, c.data_ptr(), st)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.call("sed_mt_loss_finish", None, c.data_ptr(), 4, st)


def case_dyn_args_step(dev, graph=False, steps=4, n_samp=16000 + 1024, seed0=0):
    """Step-varying arguments through device memory (desed_task_amd/graph.py) == the by-value eager path.

    Two identical tasks, identical host RNG streams: one runs the plain StepDriver, the other runs every step under a
    DynArgs context (CPU / emulator: same launches, arguments read from memory) or, with graph=True, through
    GraphedStepDriver (GPU: eager warm-up steps, one capture, then replays).  Dropout, mixup (both outcomes of the coin
    flip), the rampup weight, the EMA factor and Adam's bias corrections all change from step to step."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd.launcher import StepDriver
    bs = (1, 1, 2)
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    audio = O.synth_audio(B, n_samp, seed=77)
    n_out = (1 + n_samp // 256) // 4
    labels = O.synth_labels(bs, 10, n_out, seed=5)

    def seed_all(step):
        random.seed(40 + step); np.random.seed(100 + step); torch.manual_seed(100 + step)
        if dev != "cpu":
            torch.cuda.manual_seed(100 + step)
        from desed_task_amd import ops as _ops
        _ops.reseed_dropout()       # the private dropout-seed stream restarts only when the torch seed CHANGES: whatever test ran
                                    # before may have left it on this very seed

    results = []
    for mode in ("eager", "dyn"):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        if mode == "eager":
            driver = StepDriver(task, world_size=1)
        elif graph:
            driver = G.GraphedStepDriver(task, world_size=1, warmup=1)
        else:
            driver = StepDriver(task, world_size=1, ema_side_stream=False)
            dyn = G.DynArgs(dev)
        flips = []
        for step in range(steps):
            seed_all(seed0 + step)
            flips.append(random.random() < 0.5)
            seed_all(seed0 + step)
            batch = (to(dev, audio.clone()), to(dev, labels.clone()), None, None)
            if mode == "dyn" and not graph:
                with G.dyn_step(dyn):
                    loss = driver.run_step(batch, step)
            else:
                loss = driver.run_step(batch, step)
        if dev != "cpu":
            torch.cuda.synchronize()
        results.append((float(loss.detach()), {k: v.detach().cpu().clone() for k, v in task.sed_student.state_dict().items()},
                        {k: v.detach().cpu().clone() for k, v in task.sed_teacher.state_dict().items()},
                        task.scheduler["scheduler"].step_num))
        assert any(flips) and not all(flips), "the seeds should exercise both outcomes of the mixup coin flip"
    (l0, s0, t0, n0), (l1, s1, t1, n1) = results
    assert n0 == n1 == steps + 1                    # ExponentialWarmup.step_num starts at 1
    # The two paths are arithmetically identical, and since round 3 no kernel of the step adds floats with atomics: identical bits on
    # the (in-order) emulator AND on the GPU.  (Until then the GPU comparison allowed Adam's +-lr flips of near-zero gradients.)
    strict = True
    assert abs(l0 - l1) <= (2e-5 if strict else 2e-3) * max(1.0, abs(l0)), (l0, l1)
    lr = 1e-3
    for name, a, b in (("student", s0, s1), ("teacher", t0, t1)):
        for k in a:
            if a[k].dtype.is_floating_point:
                d = (a[k] - b[k]).abs()
                if strict:
                    assert d.max().item() <= 2e-5, (name, k, d.max().item())
                else:
                    assert d.max().item() <= 2.5 * lr * steps, (name, k, d.max().item())
                    assert (d > 5e-5).float().mean().item() <= 0.05, (name, k, (d > 5e-5).float().mean().item())
            else:
                assert torch.equal(a[k], b[k]), (name, k)


def case_prefetch_equals_unpipelined(dev, point="tails", graph=False, steps=5, n_samp=16000 + 1024, protocol=True):
    """Software-pipelined mel front-end (SEDTask4.launch_prefetch: the mel kernel of batch k + 1 on a side stream under step k)
    == the unpipelined order, on a sequence of DIFFERENT batches (an off-by-one in the hand-over would mix clips up).  The last
    step announces no successor; every step's loss and the final weights are compared.  graph=True: through
    GraphedStepDriver (one eager step, the capture, replays), where the next batch's waveforms travel through a static buffer."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd.launcher import StepDriver
    bs = (1, 1, 2)
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    batches = [(to(dev, O.synth_audio(B, n_samp, seed=300 + 11 * i)), to(dev, O.synth_labels(bs, 10, n_out, seed=20 + i))) for i in range(steps)]

    def seed_all(step):
        random.seed(40 + step); np.random.seed(100 + step); torch.manual_seed(100 + step)
        if dev != "cpu":
            torch.cuda.manual_seed(100 + step)
        from desed_task_amd import ops as _ops
        _ops.reseed_dropout()       # the private dropout-seed stream restarts only when the torch seed CHANGES: whatever test ran
                                    # before may have left it on this very seed

    results = []
    for mode in ("plain", "pipelined"):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        pf = point if mode == "pipelined" else None
        driver = G.GraphedStepDriver(task, world_size=1, warmup=1, prefetch=pf) if graph else StepDriver(task, world_size=1, prefetch=pf)
        losses = []
        next_labels = [b[1].clone() for b in batches]
        for step in range(steps):
            # per-step reseeding is fine while only the mel kernel moves; with the whole front half one step early the draws of
            # step k + 1 (mixup: global generators; teacher CNN masks: its private stream) are made DURING step k, so the
            # generators are seeded once and must simply be consumed in the same order
            if step == 0 or point != "teacher":
                seed_all(step)
            a, l = batches[step]
            batch = (a, l.clone(), None, None)
            if mode == "pipelined":
                # (the announced labels are mixed in place one step early under prefetch "teacher": a private copy per use)
                nxt = (batches[step + 1][0], next_labels[step + 1], None, None) if step + 1 < steps else None
                if step > 0 and point == "teacher":
                    batch = (a, next_labels[step], None, None)
                loss = driver.run_step(batch, step, next_batch=nxt)
            else:
                loss = driver.run_step(batch, step)
            losses.append(float(loss.detach()))
        if dev != "cpu":
            torch.cuda.synchronize()
        if mode == "pipelined":
            assert task._feat_buf is not None and (point != "teacher" or task._pro is not None)
            if graph:
                assert driver.next_audio_buffer() is not None
        results.append((losses, task.sed_student.arena.flat.detach().cpu().clone(), task.sed_teacher.arena.flat.detach().cpu().clone()))
    (l0, s0, t0), (l1, s1, t1) = results
    strict = True           # no float atomics in the step (round 3): the pipelined order reproduces the plain one bit for bit on the GPU too
    for a, b in zip(l0, l1):
        assert abs(a - b) <= (0.0 if strict else 2e-3) * max(1.0, abs(a)), (l0, l1)
    if strict:
        assert torch.equal(s0, s1) and torch.equal(t0, t1)
    else:
        for a, b in ((s0, s1), (t0, t1)):
            d = (a - b).abs()
            assert d.max().item() <= 2.5 * 1e-3 * steps and (d > 5e-5).float().mean().item() <= 0.05, (d.max().item(), (d > 5e-5).float().mean().item())
    if not protocol:
        return
    # protocol: a step that is handed another batch than the announced one must fail loudly
    task = build_task(dev, bs, sd, dropout=0.0, specaug=False, rampup=5)
    driver = StepDriver(task, world_size=1, prefetch=point)
    seed_all(0)
    driver.run_step((batches[0][0], batches[0][1].clone(), None, None), 0, next_batch=(batches[1][0], batches[1][1].clone(), None, None))
    try:
        driver.run_step((batches[0][0], batches[0][1].clone(), None, None), 1)        # announced: batches[1]
        raise AssertionError("a batch other than the announced one must be refused")
    except RuntimeError:
        pass


def case_pipelined_epoch_boundary(dev, point="teacher", graph=True, epochs=3, per_epoch=3, n_samp=16000 + 1024, reset_after=None):
    """The pipelined front end across EPOCH BOUNDARIES: the last batch of an epoch announces no successor, the first
    batch of the next epoch was announced by nobody.  A captured step has both halves of the pipeline baked in, so the graphed
    driver must run the unannounced-successor step eagerly and re-prime the front half inline afterwards -- the whole sequence
    must equal the UNPIPELINED StepDriver bit for bit (weights of student and teacher, BatchNorm statistics of both, every loss).
    reset_after = k: `task.reset_pipeline()` after step k (weights were loaded in between: launcher.load_checkpoint) although step
    k + 1 had been announced -- its front half must then be recomputed inline from UNMIXED labels; that sequence is compared with
    the eager pipelined StepDriver doing the same (the recomputed front half draws again, so the plain order is no reference)."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver
    bs = (1, 1, 2)
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    steps = epochs * per_epoch
    batches = [(to(dev, O.synth_audio(B, n_samp, seed=500 + 7 * i)), to(dev, O.synth_labels(bs, 10, n_out, seed=60 + i))) for i in range(steps)]
    originals = [b[1].clone() for b in batches]

    def bn_state(task):
        out = []
        for model in (task.sed_student, task.sed_teacher):
            for i in range(7):
                bn = getattr(model.cnn.cnn, "batchnorm%d" % i)
                out += [bn.running_mean.detach().cpu().clone(), bn.running_var.detach().cpu().clone()]
        return torch.cat(out)

    results = []
    for mode in ("reference", "pipelined"):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        if mode == "reference":
            driver = StepDriver(task, world_size=1, prefetch=None if reset_after is None else point)
        else:
            driver = G.GraphedStepDriver(task, world_size=1, warmup=1, prefetch=point) if graph else StepDriver(task, world_size=1, prefetch=point)
        pipelined = mode == "pipelined" or reset_after is not None
        random.seed(41); np.random.seed(101); torch.manual_seed(101)
        if dev != "cpu":
            torch.cuda.manual_seed(101)
        _ops.reseed_dropout()
        losses = []
        for step in range(steps):
            a, l = batches[step]
            last_of_epoch = (step + 1) % per_epoch == 0
            nxt = None
            if pipelined and not last_of_epoch:
                nxt = (batches[step + 1][0], batches[step + 1][1], None, None)
            loss = driver.run_step((a, l.clone(), None, None), step, next_batch=nxt) if pipelined else driver.run_step((a, l.clone(), None, None), step)
            losses.append(float(loss.detach()))
            if reset_after is not None and step == reset_after:
                assert nxt is not None, "reset_after must not be the last step of an epoch"
                task.reset_pipeline()
        if dev != "cpu":
            torch.cuda.synchronize()
        if mode == "pipelined" and graph:
            want_fallbacks = epochs - (1 if per_epoch <= 2 else 0)          # (an epoch end inside the warm-up / capture steps is eager anyway)
            assert driver.eager_fallbacks >= epochs - 1 and driver.eager_fallbacks <= epochs, driver.eager_fallbacks
            assert driver.reprimes == (epochs - 1) + (1 if reset_after is not None else 0), driver.reprimes
            del want_fallbacks
        # the announced label tensors are only READ by the pipelined front half (mixed in the hand-over buffer)
        for b, o in zip(batches, originals):
            assert torch.equal(b[1], o), "an announced label tensor was modified in place"
        results.append((losses, task.sed_student.arena.flat.detach().cpu().clone(), task.sed_teacher.arena.flat.detach().cpu().clone(), bn_state(task)))
    (l0, s0, t0, b0), (l1, s1, t1, b1) = results
    assert l0 == l1, (l0, l1)
    assert torch.equal(s0, s1) and torch.equal(t0, t1) and torch.equal(b0, b1)
    return l0


def case_lightning_surface(dev, epochs=3, per_epoch=3, n_samp=16000 + 1024, limit_train_batches=None, warmup=1, pretrained=False,
                           recipe2024=False):
    """The whole-step mode of SEDTask4 behind Lightning 1.9's hook order (tests/lightning_order.Trainer: training_step ->
    on_before_zero_grad -> optimizer_zero_grad -> backward -> optimizer.step -> lr_scheduler_step, the optimizer a plain
    torch.optim.Adam as train_sed.py:199-201 builds it, the batches from `train_dataloader()`) must equal, BIT FOR BIT,
      (a) the step driver driven by hand with explicit next_batch announcements -- graph.GraphedStepDriver on the GPU (eager warm-up,
          capture, replays, the eager fall-back at every epoch end, the inline front half after it), launcher.StepDriver on the
          emulator -- and
      (b) the same Trainer loop with the hooks doing their work one by one (whole_step = False: the unpipelined eager order);
    compared: every step's loss, the student and teacher weights, the BatchNorm statistics of both, Adam's moments and step count,
    the scheduler's step_num, and the logged keys of the last step.  limit_train_batches: the trainer stops an epoch early
    (train_sed.py:256, fast_dev_run) -- the step before the cut must not announce a batch nobody trains on."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver
    from desed_task_amd.lookahead import BatchList
    from tests.lightning_order import Trainer
    bs = (2, 1, 1, 2, 2) if recipe2024 else (1, 1, 2)
    B = sum(bs)
    sd = O.make_state_dict(seed=7, **({"embedding_size": 768} if pretrained else {}))
    n_out = (1 + n_samp // 256) // 4
    audios = [to(dev, O.synth_audio(B, n_samp, seed=700 + 7 * i)) for i in range(per_epoch)]
    if recipe2024:      # recipes/dcase2024_task4_baseline: 27 classes, five data sets, embeddings + valid_class_mask in the batch
        ns = bs[0] + bs[1] + bs[2]
        labelss = []
        for i in range(per_epoch):
            lab = (O.lcg_fill((B, 27, n_out), 50 + i, 0.5, 0.5) < 0.1).float()
            lab[ns:ns + bs[3], :, 1:] = 0.0
            lab[ns + bs[3]:] = 0.0
            labelss.append(to(dev, lab))
        valid = torch.zeros(B, 27, dtype=torch.bool)
        valid[:bs[0], 10:] = True
        valid[bs[0]:, :10] = True
        valid = to(dev, valid)
    else:
        labelss = [to(dev, O.synth_labels(bs, 10, n_out, seed=80 + i)) for i in range(per_epoch)]
    embs = ([to(dev, torch.randn(B, 768, 31, generator=torch.Generator().manual_seed(5 + i))) for i in range(per_epoch)]
            if (pretrained or recipe2024) else None)
    used = per_epoch if limit_train_batches is None else limit_train_batches

    class Clips(BatchList):             # a loader hands out fresh tensors every time: the step mixes labels (2024: embeddings too) in place
        def __getitem__(self, i):
            if recipe2024:
                return (audios[i], labelss[i].clone(), [1.0] * B, embs[i].clone(), valid)
            return (audios[i], labelss[i].clone(), [1.0] * B) + ((embs[i],) if pretrained else ())

    def state(task):
        out = [task.sed_student.arena.flat.detach().cpu().clone(), task.sed_teacher.arena.flat.detach().cpu().clone()]
        for model in (task.sed_student, task.sed_teacher):
            for i in range(7):
                bn = getattr(model.cnn.cnn, "batchnorm%d" % i)
                out += [bn.running_mean.detach().cpu().clone(), bn.running_var.detach().cpu().clone()]
        osd = task.opt.state_dict()
        assert set(osd["param_groups"][0]) >= {"lr", "betas", "eps", "params"}
        out += [torch.cat([osd["state"][i]["exp_avg"].reshape(-1).cpu() for i in sorted(osd["state"])]),
                torch.cat([osd["state"][i]["exp_avg_sq"].reshape(-1).cpu() for i in sorted(osd["state"])]),
                torch.tensor([float(osd["state"][0]["step"]), float(task.scheduler["scheduler"].step_num)])]
        return out

    def seed():
        random.seed(41); np.random.seed(101); torch.manual_seed(101)
        if dev != "cpu":
            torch.cuda.manual_seed(101)
        _ops.reseed_dropout()

    results = {}
    for mode in ("driver", "whole", "hooks"):
        if recipe2024:
            task = build_task_2024(dev, bs, 27, torch_adam=mode != "driver", whole_step=mode == "whole", train_data=Clips([None] * per_epoch))
        else:
            task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5, torch_adam=mode != "driver", whole_step=mode == "whole",
                              train_data=Clips([None] * per_epoch), pretrained=pretrained)
        task.whole_step_warmup = warmup
        seed()
        losses = []
        if mode == "driver":
            driver = (G.GraphedStepDriver(task, world_size=1, warmup=warmup, prefetch="teacher") if dev != "cpu"
                      else StepDriver(task, world_size=1, prefetch="teacher"))
            data = Clips([None] * per_epoch)
            for epoch in range(epochs):
                # (through a DataLoader like the trainer's: creating its iterator draws the epoch's base seed from torch's global
                #  generator -- the reference's own loop does the same --, which moves the mixup permutations that follow)
                epoch_batches = list(torch.utils.data.DataLoader(data, batch_size=None))[:used]
                for i in range(used):
                    nxt = epoch_batches[i + 1] if i + 1 < used else None
                    losses.append(float(driver.run_step(epoch_batches[i], i, next_batch=nxt).detach()))
        else:
            tr = Trainer(max_epochs=epochs, limit_train_batches=1.0 if limit_train_batches is None else limit_train_batches)
            tr.fit(task)
            losses = [float(l) for l in tr.losses]
            assert tr.global_step == epochs * used
            if mode == "whole":
                drv = task._driver
                assert drv is not None and not task._served and task.opt.served is False
                if dev != "cpu":
                    assert drv.graph is not None and drv.eager_fallbacks >= epochs - 1 and drv.reprimes >= epochs - 1
            else:
                assert task._driver is None
        if dev != "cpu":
            torch.cuda.synchronize()
        logged = {k: float(v) for k, v in task.logged.items()}
        assert len(logged) == (9 if recipe2024 else 11) and logged["train/step"] == epochs * used, logged     # (logged BEFORE the scheduler's step)
        results[mode] = (losses, state(task), logged)
    ref = results["driver"]
    for mode in ("whole", "hooks"):
        got = results[mode]
        assert got[0] == ref[0], (mode, got[0], ref[0])
        for a, b in zip(got[1], ref[1]):
            assert torch.equal(a, b), mode
        assert got[2] == ref[2], (mode, got[2], ref[2])
    return ref[0]


def case_lightning_surface_abandoned_epoch(dev, n_samp=2048 + 1024, per_epoch=3, warmup=1):
    """ADVICE r05 (medium, both): (1) an epoch that is ABANDONED while a successor is announced -- the loop leaves epoch 1 after its
    second batch without the module knowing (`abandon`), then a new iter(loader) starts epoch 2 -- must not make the next step consume
    the stale prefetched front half: whole-step mode == the hooks one by one, bit for bit (losses, weights, BatchNorm statistics,
    Adam state); (2) training_step called twice with NO optimizer.step() in between (a hand-written loop) applies two Adam updates."""
    import random
    from desed_task_amd import ops as _ops
    from desed_task_amd.lookahead import BatchList
    from tests.lightning_order import Trainer
    bs = (1, 1, 2)
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    audios = [to(dev, O.synth_audio(B, n_samp, seed=900 + 7 * i)) for i in range(per_epoch)]
    labelss = [to(dev, O.synth_labels(bs, 10, n_out, seed=60 + i)) for i in range(per_epoch)]

    class Clips(BatchList):
        def __getitem__(self, i):
            return (audios[i], labelss[i].clone(), [1.0] * B)

    def seed():
        random.seed(43); np.random.seed(103); torch.manual_seed(103)
        if dev != "cpu":
            torch.cuda.manual_seed(103)
        _ops.reseed_dropout()

    def state(task):
        out = [task.sed_student.arena.flat.detach().cpu().clone(), task.sed_teacher.arena.flat.detach().cpu().clone()]
        for model in (task.sed_student, task.sed_teacher):
            for i in range(7):
                bn = getattr(model.cnn.cnn, "batchnorm%d" % i)
                out += [bn.running_mean.detach().cpu().clone(), bn.running_var.detach().cpu().clone()]
        osd = task.opt.state_dict()
        out += [torch.cat([osd["state"][i]["exp_avg"].reshape(-1).cpu() for i in sorted(osd["state"])]),
                torch.tensor([float(osd["state"][0]["step"]), float(task.scheduler["scheduler"].step_num)])]
        return out

    # Reference: the step driver by hand with the SAME announcements (the look-ahead draws the announced batch's mixup on the host one
    # step early, so the hooks-one-by-one order cannot be the reference once an announced batch is dropped) and an explicit
    # reset_pipeline() where the epoch is abandoned.
    from desed_task_amd import graph as G
    from desed_task_amd.launcher import StepDriver
    results = {}
    for mode in ("whole", "driver"):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5, torch_adam=mode == "whole", whole_step=mode == "whole",
                          train_data=Clips([None] * per_epoch))
        task.whole_step_warmup = warmup
        seed()
        if mode == "whole":
            tr = Trainer(max_epochs=4, abandon={1: 2}).fit(task)
            assert tr.global_step == 3 * per_epoch + 2
            losses = [float(l) for l in tr.losses]
        else:
            driver = (G.GraphedStepDriver(task, world_size=1, warmup=warmup, prefetch="teacher") if dev != "cpu"
                      else StepDriver(task, world_size=1, prefetch="teacher"))
            data, losses = Clips([None] * per_epoch), []
            for epoch in range(4):
                epoch_batches = list(torch.utils.data.DataLoader(data, batch_size=None))
                used = 2 if epoch == 1 else per_epoch
                for i in range(used):
                    nxt = epoch_batches[i + 1] if i + 1 < per_epoch else None      # (epoch 1: batch 1 announces batch 2, which never comes)
                    losses.append(float(driver.run_step(epoch_batches[i], i, next_batch=nxt).detach()))
                if epoch == 1:
                    task.reset_pipeline()
        if dev != "cpu":
            torch.cuda.synchronize()
        results[mode] = (losses, state(task))
    assert results["whole"][0] == results["driver"][0], (results["whole"][0], results["driver"][0])
    for a_, b_ in zip(results["whole"][1], results["driver"][1]):
        assert torch.equal(a_, b_)

    # (2) two training_step calls in a row, nobody calls optimizer.step(): each is a whole optimisation step
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5, torch_adam=True, whole_step=True, train_data=Clips([None] * per_epoch))
    task.whole_step_warmup = warmup
    seed()
    task.train()
    flats = [task.sed_student.arena.flat.detach().cpu().clone()]
    step0 = task.scheduler["scheduler"].step_num
    for i in range(3):
        task.training_step((audios[i], labelss[i].clone(), [1.0] * B), i)
        if dev != "cpu":
            torch.cuda.synchronize()
        flats.append(task.sed_student.arena.flat.detach().cpu().clone())
    osd = task.opt.state_dict()
    assert float(osd["state"][0]["step"]) == 3.0 and task.scheduler["scheduler"].step_num == step0 + 3
    for i in range(3):
        assert not torch.equal(flats[i], flats[i + 1]), "step %d left the weights where they were" % (i + 1)


def _soak_state(task):
    from desed_task_amd.launcher import bn_buffers
    osd = task.opt.state_dict()
    return [task.sed_student.arena.flat.detach().clone(), task.sed_teacher.arena.flat.detach().clone(),
            torch.cat([b.detach().reshape(-1) for b in bn_buffers(task)]),
            torch.cat([osd["state"][i]["exp_avg"].reshape(-1) for i in sorted(osd["state"])]),
            torch.cat([osd["state"][i]["exp_avg_sq"].reshape(-1) for i in sorted(osd["state"])])]


def case_corruption_soak(dev, steps=2000, K=8, bs=(12, 12, 24), n_samp=160000, marks=(1, 500, 2000), lr=2e-4):
    """the B = 48 pipelined step captured once and replayed over the SAME sequence of `steps` batches (K different
    batches in rotation) twice from the same state, plus once eagerly: student and teacher weights, all BatchNorm statistics and both Adam
    moments must be BIT-equal between the two replayed runs and equal to the eager run at every mark.  A transient fault anywhere in
    the ~120 kernels of the step -- like the packed-op_sel hazard of round 6, ~1e-5 per instruction -- shows up as a difference that
    the later steps carry along."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    g = torch.Generator().manual_seed(17)
    audios = [to(dev, 0.1 * torch.randn(B, n_samp, generator=g)) for _ in range(K)]
    labelss = [to(dev, O.synth_labels(bs, 10, n_out, seed=30 + i)) for i in range(K)]
    marks = tuple(m for m in marks if m <= steps)

    def run(kind):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=50, lr=lr)
        random.seed(47); np.random.seed(107); torch.manual_seed(107)
        if dev != "cpu":
            torch.cuda.manual_seed(107)
        _ops.reseed_dropout()
        driver = (G.GraphedStepDriver(task, world_size=1, warmup=1, prefetch="teacher") if (kind == "graph" and dev != "cpu")
                  else StepDriver(task, world_size=1, prefetch="teacher"))
        snaps, cur = {}, labelss[0].clone()
        for i in range(steps):
            nl = labelss[(i + 1) % K].clone() if i + 1 < steps else None
            nxt = (audios[(i + 1) % K], nl, None, None) if nl is not None else None
            loss = driver.run_step((audios[i % K], cur, None, None), i, next_batch=nxt)
            cur = nl
            if i + 1 in marks:
                if dev != "cpu":
                    torch.cuda.synchronize()
                snaps[i + 1] = _soak_state(task) + [loss.detach().clone().reshape(1)]
        if kind == "graph" and steps > 3 and dev != "cpu":
            assert driver.graph is not None and driver.eager_fallbacks == 1       # (only the last step -- no successor -- ran eagerly)
        return snaps

    first, second, eager = run("graph"), run("graph"), run("eager")
    names = ("student", "teacher", "BatchNorm buffers", "Adam exp_avg", "Adam exp_avg_sq", "loss")
    for m in marks:
        assert all(bool(torch.isfinite(t).all()) for t in first[m]), "non-finite state at step %d" % m
        for name, a_, b_, c_ in zip(names, first[m], second[m], eager[m]):
            assert torch.equal(a_, b_), "step %d: %s differs between two replayed runs (%d elements, max %.3e)" % (
                m, name, int((a_ != b_).sum()), float((a_ - b_).abs().max()))
            assert torch.equal(a_, c_), "step %d: %s differs between the replayed and the eager run (%d elements, max %.3e)" % (
                m, name, int((a_ != c_).sum()), float((a_ - c_).abs().max()))
    return {m: float(first[m][-1]) for m in marks}


def case_step_beside_gemm_storm(dev, reps=60, bs=(3, 3, 6), n_samp=160000, storm=260):
    """The canary of the same item: two full training steps (every kernel of the step: the mel kernel, block 0, the GLU blocks, the
    BiGRU recurrences, the heads, the losses, their backward twins, Adam, the EMA) launched while a storm of split-bf16 GEMMs
    (v_mfma_f32_32x32x16_bf16 waves on every CU -- the co-runner of round 6's hazard) runs on another stream, `reps` times: weights of
    both models, BatchNorm statistics and Adam moments after the two steps equal the solo run's BIT for BIT.  (The instruction-level
    probe showed the hazard in eager launches on two streams as readily as in a replayed graph.)"""
    import random
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    g = torch.Generator().manual_seed(19)
    audios = [to(dev, 0.1 * torch.randn(B, n_samp, generator=g)) for _ in range(2)]
    labelss = [to(dev, O.synth_labels(bs, 10, n_out, seed=70 + i)) for i in range(2)]
    H, Ms = 128, 48 * 156
    h_big = to(dev, torch.randn(Ms, H))
    gi = torch.zeros(Ms, 2, 3 * H, device=h_big.device)
    w0, w1, b0 = to(dev, 0.1 * torch.randn(3 * H, H)), to(dev, 0.1 * torch.randn(3 * H, H)), torch.zeros(3 * H, device=h_big.device)
    side = torch.cuda.Stream() if dev != "cpu" else None

    def two_steps(with_storm):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = StepDriver(task, world_size=1)
        random.seed(49); np.random.seed(109); torch.manual_seed(109)
        if dev != "cpu":
            torch.cuda.manual_seed(109)
        _ops.reseed_dropout()
        if with_storm and side is not None:
            side.wait_stream(torch.cuda.current_stream())
            lib = _lib.get()
            for _ in range(storm):
                lib.call("sed_gemm_pair_bf16x3", h_big.data_ptr(), h_big.data_ptr(), w0.data_ptr(), w1.data_ptr(), b0.data_ptr(), b0.data_ptr(),
                         gi.data_ptr(), gi.data_ptr() + 3 * H * 4, Ms, 3 * H, H, H, H, 6 * H, 0, 1, 1, 0, side.cuda_stream)
        for i in range(2):
            driver.run_step((audios[i], labelss[i].clone(), None, None), i)
        if dev != "cpu":
            torch.cuda.synchronize()
        return _soak_state(task)

    ref = two_steps(False)
    again = two_steps(False)
    for a_, b_ in zip(ref, again):
        assert torch.equal(a_, b_)
    names = ("student", "teacher", "BatchNorm buffers", "Adam exp_avg", "Adam exp_avg_sq")
    for rep in range(reps):
        got = two_steps(True)
        for name, a_, b_ in zip(names, ref, got):
            assert torch.equal(a_, b_), "repetition %d: %s differs from the solo run beside the GEMM storm (%d elements, max %.3e)" % (
                rep, name, int((a_ != b_).sum()), float((a_ - b_).abs().max()))


def case_step_bit_reproducible(dev, steps=3, n_samp=16000 + 1024):
    """Since round 3 no kernel of the default training step adds floats with atomics (loss sums, head and BiGRU bias gradients
    moved to per-workgroup records summed in a fixed order): the same seeded steps give the SAME BITS -- run twice eagerly, and once
    more through the hipGraph driver (step-varying arguments from device memory, replayed launches)."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd import ops
    from desed_task_amd.launcher import StepDriver
    bs = (1, 1, 2)
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    audio = to(dev, O.synth_audio(B, n_samp, seed=77))
    labels = to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5))
    finals = []
    for mode in ("eager", "eager", "graph"):
        task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = StepDriver(task, world_size=1) if mode == "eager" else G.GraphedStepDriver(task, world_size=1, warmup=1)
        for step in range(steps):
            random.seed(40 + step); np.random.seed(100 + step); torch.manual_seed(100 + step)
            ops.reseed_dropout()
            loss = driver.run_step((audio, labels.clone(), None, None), step)
        torch.cuda.synchronize()
        finals.append((float(loss.detach()), task.sed_student.arena.flat.detach().cpu().clone(), task.sed_teacher.arena.flat.detach().cpu().clone(),
                       task.sed_student.arena.flat_grad.detach().cpu().clone()))
    for name, other in (("second eager run", finals[1]), ("hipGraph replay", finals[2])):
        assert finals[0][0] == other[0], (name, finals[0][0], other[0])
        for what, a, b_ in zip(("student", "teacher", "gradient"), finals[0][1:], other[1:]):
            assert torch.equal(a, b_), "%s: %s differs (max %.3e, %d elements)" % (name, what, (a - b_).abs().max().item(), int((a != b_).sum()))


def case_step_ignores_uninitialised_memory(dev, n_samp=4096 + 1024, steps=2):
    """Every scratch / output buffer of the step comes from torch.empty[_like]: with those poisoned (NaN, then 3e30) the seeded
    steps must give the very same bits as without -- no kernel result may depend on memory it did not write (a masked lane that
    multiplies garbage by zero would turn the NaN run into NaNs; a stale value would change the 3e30 run)."""
    import random
    from desed_task_amd import ops
    from desed_task_amd.launcher import StepDriver
    orig_empty, orig_empty_like = torch.empty, torch.empty_like

    def run(poison):
        if poison is not None:
            def e(*a, **k):
                t = orig_empty(*a, **k)
                return t.fill_(poison) if t.dtype.is_floating_point else t
            def el(x, *a, **k):
                t = orig_empty_like(x, *a, **k)
                return t.fill_(poison) if t.dtype.is_floating_point else t
            torch.empty, torch.empty_like = e, el
        try:
            bs = (1, 1, 2)
            task = build_task(dev, bs, O.make_state_dict(seed=7), dropout=0.5, specaug=True, rampup=5)
            d = StepDriver(task, world_size=1)
            audio = to(dev, O.synth_audio(4, n_samp, seed=100))
            labels = to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5))
            for step in range(steps):
                random.seed(40 + step); np.random.seed(100 + step); torch.manual_seed(100 + step)
                ops.reseed_dropout()
                loss = d.run_step((audio, labels.clone(), None, None), step)
            if dev != "cpu":
                torch.cuda.synchronize()
            return (float(loss.detach()), task.sed_student.arena.flat.detach().cpu().clone(), task.sed_student.arena.flat_grad.detach().cpu().clone(),
                    task.sed_teacher.arena.flat.detach().cpu().clone())
        finally:
            torch.empty, torch.empty_like = orig_empty, orig_empty_like

    base = run(None)
    for poison in (float("nan"), 3e30):
        got = run(poison)
        assert got[0] == base[0], (poison, got[0], base[0])
        for name, a, b_ in zip(("student", "gradient", "teacher"), base[1:], got[1:]):
            assert torch.equal(a, b_), "poison %r: %s differs (%d NaN)" % (poison, name, int(torch.isnan(b_).sum()))


def case_bn_fold_equals_separate_pass(dev, n_samp=8192 + 1024):
    """BatchNorm backward folded into the data-gradient convolution's operand staging (sed_conv3x3_bf16x3_bnbwd, blocks 1-6) vs the
    separate in-place pass (sed_bn_bwd_apply) it replaces: one training step's gradients, every one of them -- identical bits on
    the in-order emulator (the fold evaluates the same expression operation for operation and nothing else changes), atomics-level
    agreement on the GPU."""
    import random
    from desed_task_amd import ops
    lib = _lib_get()
    bs = (1, 1, 1)
    sd = O.make_state_dict(seed=7)
    audio = to(dev, O.synth_audio(3, n_samp, seed=100))
    labels = O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5)
    grads, calls = [], []
    prev = ops.BN_BWD_FOLD
    orig = lib.call
    try:
        for fold in (True, False):
            ops.BN_BWD_FOLD = fold
            seen = {}
            lib.call = lambda name, *a, _s=seen: (_s.__setitem__(name, _s.get(name, 0) + 1), orig(name, *a))[1]
            task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=100)
            random.seed(4); np.random.seed(7); torch.manual_seed(5)
            ops.reseed_dropout()
            loss = task.training_step((audio, to(dev, labels.clone()), None, None), 0)
            loss.backward()
            grads.append(task.sed_student.arena.gather_grads().detach().cpu().clone())
            calls.append(seen)
    finally:
        ops.BN_BWD_FOLD = prev
        lib.call = orig
    assert calls[0].get("sed_conv3x3_bf16x3_bnbwd") == 6 and "sed_bn_bwd_apply" not in calls[0], calls[0]
    assert calls[1].get("sed_bn_bwd_apply") == 6 and "sed_conv3x3_bf16x3_bnbwd" not in calls[1], calls[1]
    if dev == "cpu":
        assert torch.equal(grads[0], grads[1])
    else:
        d = (grads[0] - grads[1]).abs()
        assert d.max().item() <= 1e-5 * max(1.0, grads[1].abs().max().item()), d.max().item()


def _lib_get():
    from desed_task_amd import _lib
    return _lib.get()


# ------------------------------------------------------------------------------------------------
# full-size cases (BASELINE.json configs): 10 s clips, production batch shapes
# ------------------------------------------------------------------------------------------------
def case_full_size_step_vs_oracle(dev, bs=(4, 4, 8)):
    """One mean-teacher step on full 10 s clips at the C1 batch (16 clips = [4,4,8]; the reference's CPU-runnable
    configuration) against the oracle trainer: logged scalars, the 1e-3 posterior criterion and all gradients."""
    import random
    from desed_task_amd.launcher import StepDriver
    torch.set_num_threads(min(32, torch.get_num_threads()))
    B, n_samp = sum(bs), 160000
    sd = O.make_state_dict(seed=11)
    audio = O.synth_audio(B, n_samp, seed=3)
    n_out = (1 + n_samp // 256) // 4
    assert n_out == 156
    labels = O.synth_labels(bs, 10, n_out, seed=9)
    task = build_task(dev, bs, sd, dropout=0.0, specaug=False, rampup=100)
    driver = StepDriver(task, world_size=1)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100)
    random.seed(4); np.random.seed(7); torch.manual_seed(7)
    assert random.random() < 0.5
    cw = np.random.beta(0.2, 0.2); pw = torch.randperm(bs[1]); cs = np.random.beta(0.2, 0.2); ps = torch.randperm(bs[0])
    random.seed(4); np.random.seed(7); torch.manual_seed(7)
    loss = driver.run_step((to(dev, audio.clone()), to(dev, labels.clone()), None, None), 0)
    tot, logs = orc.training_step(audio, labels, mix=dict(c_weak=cw, perm_weak=pw, c_strong=cs, perm_strong=ps))
    ref_grads = orc.optimizer_step(tot)
    got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
    got["loss"] = float(loss.detach().cpu()); logs["loss"] = tot.item()
    for k in sorted(logs):
        assert abs(got[k] - logs[k]) <= 2e-5 + 2e-4 * abs(logs[k]), "%s: hip %.8g oracle %.8g" % (k, got[k], logs[k])
    s_s, w_s, s_t, w_t = [t.detach().cpu() for t in task.last_outputs]
    assert tuple(s_s.shape) == (B, 10, 156) and tuple(w_s.shape) == (B, 10)
    for a, b in ((s_s, "strong_s"), (w_s, "weak_s"), (s_t, "strong_t"), (w_t, "weak_t")):
        assert (a - orc.last[b]).abs().max().item() < 1e-3, b
    params = dict(task.sed_student.named_parameters())
    for k in O.PARAM_KEYS:
        if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
            continue
        g, r = params[k].grad.detach().cpu(), ref_grads[k]
        assert (g - r).abs().max().item() <= 2e-4 * r.abs().max().item() + 5e-8, k


def case_full_size_properties(dev, B=48):
    """Size-independent properties at the bench configuration (B = 48 = 12/12/24 clips of 10 s), no oracle run needed:
      P1 eval-mode forward is clip-wise: a 48-clip batch == three 16-clip batches (BN on running stats);
      P2 weak posteriors are convex combinations over time of the strong ones: min_t strong <= weak <= max_t strong;
      P3 first step without dropout/SpecAugment: teacher == student, so both consistency losses are exactly 0 and the
         EMA at alpha = 1 - 1/2 leaves the teacher on the segment between old and new student parameters;
      P4 mixup with the gate off leaves features/labels untouched; the scaler output spans exactly [-1, 1] per clip;
      P5 the gradient of the flat arena is what Adam consumes: one step moves every parameter by at most lr (|m/sqrt(v)| <= 1
         at t = 1 up to eps) and by exactly lr*sign(g) where |g| is not tiny."""
    import random
    from desed_task_amd.launcher import StepDriver
    from desed_task_amd.nnet.CRNN import CRNN
    bs = (B // 4, B // 4, B // 2)
    sd = O.make_state_dict(seed=13)
    audio = to(dev, O.synth_audio(B, 160000, seed=5))
    labels = to(dev, O.synth_labels(bs, 10, 156, seed=6))
    # ---- P1, P2: eval-mode CRNN on real features ----
    task = build_task(dev, bs, sd, dropout=0.0, specaug=False, rampup=100)
    with torch.no_grad():
        feats = task.scaled_logmel(task.mel_spec(audio))
    assert tuple(feats.shape) == (B, 128, 626)
    fmin, fmax = feats.amin(dim=(1, 2)), feats.amax(dim=(1, 2))
    assert (fmin + 1).abs().max().item() < 1e-6 and (fmax - 1).abs().max().item() < 1e-5           # P4 (scaler)
    net = task.sed_student
    net.eval()
    with torch.no_grad():
        strong, weak = net(feats)
        parts = [net(feats[i:i + B // 3].contiguous()) for i in range(0, B, B // 3)]
    assert tuple(strong.shape) == (B, 10, 156)
    assert (strong - torch.cat([p[0] for p in parts])).abs().max().item() < 2e-6                    # P1
    assert (weak - torch.cat([p[1] for p in parts])).abs().max().item() < 2e-6
    assert bool(((weak <= strong.amax(dim=2) + 1e-6) & (weak >= strong.amin(dim=2) - 1e-6)).all())  # P2
    # ---- P3, P4, P5: one training step at full size ----
    task = build_task(dev, bs, sd, dropout=0.0, specaug=False, rampup=100, lr=1e-3)
    before = task.sed_student.arena.flat.detach().clone()
    t_before = task.sed_teacher.arena.flat.detach().clone()
    driver = StepDriver(task, world_size=1)
    random.seed(2)
    assert random.random() >= 0.5                       # gate off -> no mixup this step
    random.seed(2)
    lab_in = labels.clone()
    loss = driver.run_step((audio, lab_in, None, None), 0)
    if dev != "cpu":
        torch.cuda.synchronize()
    assert torch.equal(lab_in, labels)                                                               # P4 (labels untouched)
    assert float(task.logged["train/student/strong_self_sup_loss"]) == 0.0                           # P3
    assert float(task.logged["train/student/weak_self_sup_loss"]) == 0.0
    s_s, w_s, s_t, w_t = task.last_outputs
    assert torch.equal(s_s.detach(), s_t) and torch.equal(w_s.detach(), w_t)
    assert torch.isfinite(loss).item()
    after = task.sed_student.arena.flat.detach()
    grad = task.sed_student.arena.flat_grad.detach()
    step = (after - before).abs()
    assert step.max().item() <= 1e-3 * 1.0001                                                        # P5
    big = grad.abs() > 1e-4                             # eps / |g| <= 1e-4: the step is lr * sign(g) to 1e-7
    assert big.float().mean().item() > 0.02
    assert ((after - before)[big] + 1e-3 * torch.sign(grad[big])).abs().max().item() < 5e-7
    t_after = task.sed_teacher.arena.flat.detach()
    assert (t_after - (0.5 * t_before + 0.5 * before)).abs().max().item() < 1e-7                     # P3 (EMA, alpha = 1/2)


# ------------------------------------------------------------------------------------------------
# K13: inference post-processing (median filter + thresholds + event regions), bit-exact vs scipy / the oracle
# ------------------------------------------------------------------------------------------------
class _Encoder:
    """The two members of desed_task.utils.encoder.ManyHotEncoder that batched_decode_preds touches
    (encoder.py:26-40, :76-78), 2023 recipe values."""

    def __init__(self, labels, audio_len=10, frame_hop=256, net_pooling=4, fs=16000):
        self.labels, self.audio_len, self.frame_hop, self.net_pooling, self.fs = list(labels), audio_len, frame_hop, net_pooling, fs

    def _frame_to_time(self, frame):
        frame = frame * self.net_pooling / (self.fs / self.frame_hop)
        return np.clip(frame, a_min=0, a_max=self.audio_len)


def case_postprocess(dev):
    from desed_task_amd import postprocess as PP
    g = torch.Generator().manual_seed(11)
    # ---- median filter: every window length incl. even ones, short clips (T < win: multiple reflections), ties ----
    for (B, T, NC, win) in ((3, 156, 10, 7), (2, 5, 10, 7), (1, 1, 3, 7), (2, 40, 27, 3), (2, 33, 4, 4), (1, 20, 2, 15), (2, 9, 10, 1)):
        x = torch.rand(B, T, NC, generator=g)
        x[:, ::3] = (x[:, ::3] * 4).round() / 4            # ties
        y = PP.median_filter_scores(to(dev, x), win).cpu().numpy()
        for b in range(B):
            ref = O.median_filter_scores(x[b].numpy(), win)
            assert np.array_equal(y[b], ref), ("median", B, T, NC, win)
    # ---- thresholds -> regions: saturated, empty, alternating and random columns; padded clips ----
    B, T, NC = 4, 156, 10
    x = torch.rand(B, T, NC, generator=g)
    x[0, :, 0] = 1.0; x[0, :, 1] = 0.0; x[0, ::2, 2] = 1.0; x[0, 1::2, 2] = 0.0; x[1, 100:, 3] = 0.9; x[1, :7, 4] = 0.95
    thresholds = [0.1, 0.5, 0.5000001, 0.9]
    for true_len in (None, [156, 100, 1, 0]):
        counts, events = PP.threshold_events(to(dev, x), thresholds, true_len)
        for k, th in enumerate(thresholds):
            for b in range(B):
                n = T if true_len is None else true_len[b]
                ref = O.decode_events(x[b, :n].numpy(), np.float32(th))
                got = [(c, int(events[k, b, c, e, 0]), int(events[k, b, c, e, 1])) for c in range(NC) for e in range(counts[k, b, c])]
                assert got == ref, ("events", th, b, true_len)
    # ---- the reference-shaped entry point against the reference's own loop (utils.py:16-73) restated with the oracle ----
    enc = _Encoder(["c%d" % i for i in range(NC)])
    strong = x.transpose(1, 2)                                   # (B, NC, T) as the CRNN returns it
    files = ["/data/synth/clip_%d.wav" % i for i in range(B)]
    for pad in (None, torch.tensor([1.0, 0.75, 0.5, 1.0])):
        raw, post, dfs = PP.batched_decode_preds(to(dev, strong), files, enc, thresholds=[0.5, 0.7], median_filter=7, pad_indx=pad)
        for j in range(B):
            c_scores = strong[j].transpose(0, 1).numpy()
            n = T if pad is None else int(T * pad[j].item())
            c_scores = c_scores[:n]
            filt = O.median_filter_scores(c_scores, 7)
            aid = "clip_%d" % j
            assert np.array_equal(raw[aid].values[:, 2:].astype(np.float32), c_scores), ("raw", j)
            assert np.array_equal(post[aid].values[:, 2:].astype(np.float32), filt), ("post", j)
            assert list(post[aid].columns) == ["onset", "offset"] + enc.labels
            for th in (0.5, 0.7):
                ref = [(enc.labels[c], float(enc._frame_to_time(on)), float(enc._frame_to_time(off)))
                       for c, on, off in O.decode_events(filt, np.float32(th))]
                d = dfs[th][dfs[th]["filename"] == aid + ".wav"]
                got = [(r.event_label, float(r.onset), float(r.offset)) for r in d.itertuples()]
                assert got == ref, ("decode", j, th)


def case_validation_step(dev):
    """SEDTask4.validation_step: eval-mode student/teacher forward + batched decoding into the reference's
    buffers, against the oracle CRNN forward + the reference loop restated with scipy."""
    import pandas as pd  # noqa: F401
    bs, n_samp = (2, 2, 4), 16000 * 2 + 1024
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    audio = O.synth_audio(B, n_samp, seed=21)
    n_out = (1 + n_samp // 256) // 4
    labels = O.synth_labels(bs, 10, n_out, seed=5)
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=100)
    task.hparams["data"] = {"weak_folder": "/d/weak", "synth_val_folder": "/d/synth_val"}
    task.hparams["training"].update(val_thresholds=[0.3, 0.5], median_window=7)
    task.encoder = _Encoder(["c%d" % i for i in range(10)], audio_len=n_samp / 16000.0)
    task.eval()                                                  # Lightning puts the module in eval mode for validation
    files = ["/d/synth_val/s%d.wav" % i for i in range(3)] + ["/d/weak/w%d.wav" % i for i in range(3)] + ["/d/other/u%d.wav" % i for i in range(2)]
    task.validation_step((to(dev, audio), to(dev, labels), None, files, None), 0)
    # oracle: eval-mode forward of the same weights (teacher == student at construction)
    feats = O.scale_minmax(O.take_log(O.mel_spectrogram(audio)))
    strong, weak = O.crnn_forward(sd, feats, training=False)
    lw = (labels[3:6].sum(-1) >= 1).float()
    ref_w = torch.nn.functional.binary_cross_entropy(weak[3:6], lw).item()
    ref_s = torch.nn.functional.binary_cross_entropy(strong[:3], labels[:3]).item()
    for who in ("student", "teacher"):
        assert abs(float(task.logged["val/weak/%s/loss_weak" % who]) - ref_w) < 2e-5 * max(1.0, ref_w)
        assert abs(float(task.logged["val/synth/%s/loss_strong" % who]) - ref_s) < 2e-5 * max(1.0, ref_s)
    enc = task.encoder
    for j in range(3):
        filt = O.median_filter_scores(strong[j].transpose(0, 1).numpy(), 7)
        post = task.val_scores_postprocessed_buffer_student_synth["s%d" % j].values[:, 2:].astype(np.float32)
        assert np.abs(post - filt).max() < 2e-5                  # posteriors: fp32 rounding vs the oracle forward
    for th in (0.3, 0.5):
        for buf in (task.val_buffer_student_synth, task.val_buffer_teacher_synth):
            df = buf[th]
            assert list(df.columns) == ["event_label", "onset", "offset", "filename"]
            assert set(df["filename"]) <= {"s0.wav", "s1.wav", "s2.wav"}
            # regions recomputed from the task's OWN filtered scores must match the decoded events exactly
            for j in range(3):
                own = task.val_scores_postprocessed_buffer_student_synth["s%d" % j].values[:, 2:].astype(np.float32)
                if buf is task.val_buffer_teacher_synth:
                    own = task.val_scores_postprocessed_buffer_teacher_synth["s%d" % j].values[:, 2:].astype(np.float32)
                ref = [(enc.labels[c], float(enc._frame_to_time(on)), float(enc._frame_to_time(off)))
                       for c, on, off in O.decode_events(own, np.float32(th))]
                d = df[df["filename"] == "s%d.wav" % j]
                assert [(r.event_label, float(r.onset), float(r.offset)) for r in d.itertuples()] == ref
    f1 = float(task.get_weak_student_f1_seg_macro.compute())
    assert 0.0 <= f1 <= 1.0
    # ---- validation_epoch_end: ground truth = the student's own 0.5 decoding -> perfect synth metrics ----
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        d05 = task.val_buffer_student_synth[0.5]
        assert len(d05) > 0
        tsv, dur = os.path.join(tmp, "gt.tsv"), os.path.join(tmp, "dur.tsv")
        d05[["filename", "onset", "offset", "event_label"]].to_csv(tsv, sep="\t", index=False)
        pd.DataFrame({"filename": ["s%d.wav" % j for j in range(3)], "duration": n_samp / 16000.0}).to_csv(dur, sep="\t", index=False)
        task.hparams["data"].update(synth_val_tsv=tsv, synth_val_dur=dur)
        import copy
        saved = copy.deepcopy((task.val_buffer_student_synth, task.val_buffer_teacher_synth,
                               task.val_scores_postprocessed_buffer_student_synth, task.val_scores_postprocessed_buffer_teacher_synth,
                               task.get_weak_student_f1_seg_macro, task.get_weak_teacher_f1_seg_macro))
        # default objective (None): weak F1 + threshold-free PSDS1 of the student's score tables
        obj = task.validation_epoch_end([])
        psds1 = float(task.logged["val/synth/student/psds1_sed_scores_eval"])
        assert 0.0 < psds1 <= 1.0 and abs(float(obj) - (f1 + psds1)) < 1e-6
        (task.val_buffer_student_synth, task.val_buffer_teacher_synth, task.val_scores_postprocessed_buffer_student_synth,
         task.val_scores_postprocessed_buffer_teacher_synth, task.get_weak_student_f1_seg_macro, task.get_weak_teacher_f1_seg_macro) = saved
        task.hparams["training"]["obj_metric_synth_type"] = "event"
        obj = task.validation_epoch_end([])
        assert abs(float(obj) - (f1 + 1.0)) < 1e-6
        assert float(task.logged["val/synth/student/event_f1_macro"]) == 1.0
        # intersection F1 averages over the val thresholds (0.3 scores worse against the 0.5 ground truth)
        assert 0.0 < float(task.logged["val/synth/student/intersection_f1_macro"]) <= 1.0
        assert all(len(v) == 0 for v in task.val_buffer_student_synth.values()) and task.get_weak_student_f1_seg_macro.tp is None


def case_test_epoch(dev, out_dir):
    """SEDTask4.test_step x2 + on_test_epoch_end (SURVEY 8f ranks 1 + 2 end to end): device scoring -> decoded operating points
    -> PSDS / event / intersection metrics.  The ground truth is the oracle decoding (numpy region search) of the task's own
    post-processed student scores at 0.5, so the student's 0.5 operating point must score a perfect event-based and
    intersection-based F1 through the whole chain, and the PSD-ROC must reach TPR 1."""
    import os
    import pandas as pd
    from desed_task_amd.evaluation.psds import PSDSEval
    bs, n_samp = (1, 1, 2), 16000 + 1024
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=100)
    task.hparams["training"].update(n_test_thresholds=10, median_window=7)
    task.hparams["log_dir"] = str(out_dir)
    task.encoder = _Encoder(["c%d" % i for i in range(10)], audio_len=n_samp / 16000.0)
    task.eval()
    files = []
    for step in range(2):
        audio = O.synth_audio(B, n_samp, seed=31 + step)
        labels = O.synth_labels(bs, 10, n_out, seed=5 + step)
        names = ["/d/test/t%d_%d.wav" % (step, i) for i in range(B)]
        files += names
        task.test_step((to(dev, audio), to(dev, labels), None, names, None), step)
    assert np.allclose(sorted(task.test_psds_buffer_student), np.arange(0.05, 1, 0.1)) and len(task.test_psds_buffer_student) == 10
    assert len(task.test_scores_raw_buffer_student) == 2 * B and "test/student/loss_strong" in task.logged
    enc = task.encoder
    rows = []
    for f in files:
        aid = os.path.basename(f)[:-4]
        own = task.test_scores_postprocessed_buffer_student[aid].values[:, 2:].astype(np.float32)
        ev = O.decode_events(own, np.float32(0.5))
        for c, on, off in ev:
            rows.append((aid + ".wav", float(enc._frame_to_time(on)), float(enc._frame_to_time(off)), enc.labels[c]))
        if not ev:
            rows.append((aid + ".wav", np.nan, np.nan, np.nan))
    gt = pd.DataFrame(rows, columns=["filename", "onset", "offset", "event_label"])
    assert gt.event_label.notna().sum() > 0, "the random-init posteriors must cross 0.5 somewhere for this case to bite"
    tsv, dur = os.path.join(str(out_dir), "gt.tsv"), os.path.join(str(out_dir), "dur.tsv")
    gt.to_csv(tsv, sep="\t", index=False)
    pd.DataFrame({"filename": [os.path.basename(f) for f in files], "duration": n_samp / 16000.0}).to_csv(dur, sep="\t", index=False)
    task.hparams["data"] = {"test_tsv": tsv, "test_dur": dur}
    res = task.on_test_epoch_end()
    for who in ("student", "teacher"):                           # the teacher is a copy of the student at construction
        assert res["test/%s/event_f1_macro" % who] == 1.0
        assert res["test/%s/intersection_f1_macro" % who] == 1.0
        for k in ("psds1_psds_eval", "psds2_psds_eval", "psds1_sed_scores_eval", "psds2_sed_scores_eval"):
            assert 0.0 < res["test/%s/%s" % (who, k)] <= 1.0
        assert os.path.exists(os.path.join(str(out_dir), "metrics_test", who, "event_f1.txt"))
        assert len(os.listdir(os.path.join(str(out_dir), "metrics_test", who, "scenario1", "predictions_dtc0.7_gtc0.7_cttc0.3"))) == 10
    assert abs(float(res["hp_metric"]) - max(res["test/student/psds1_psds_eval"], res["test/student/psds2_psds_eval"])) < 1e-6
    # the 0.5 operating point alone is a perfect detector: its PSD-ROC is the unit step
    ev = PSDSEval(ground_truth=gt, metadata=pd.read_csv(dur, sep="\t"), dtc_threshold=0.7, gtc_threshold=0.7)
    ev.add_operating_point(task.decoded_student_05_buffer)
    assert ev.psds(alpha_st=1, max_efpr=100).value == 1.0
    # evaluation mode: only the score tables are written
    task.evaluation = True
    task._exp_dir = os.path.join(str(out_dir), "eval")
    assert task.on_test_epoch_end() == {}
    assert len(os.listdir(os.path.join(task._exp_dir, "metrics_test", "student_scores", "postprocessed"))) == 2 * B


# ------------------------------------------------------------------------------------------------
# embedding fusion: K14 + cat_tf, CRNN(use_embeddings=True, aggregation_type="pool1d")
# ------------------------------------------------------------------------------------------------
def case_embcat_op(dev):
    """EmbCatFn forward / backward against torch ops on the same dropout mask: ragged pooling windows (496 -> 156, 51 -> 16),
    up-sampling (Te < T: repeated windows), E not a multiple of the 32-channel tile, dropout on and off."""
    from desed_task_amd import ops
    for (B, T, Te, C, E, p, seed) in ((2, 16, 51, 128, 768, 0.5, 77), (1, 156, 496, 128, 64, 0.5, 5), (3, 16, 10, 32, 40, 0.0, 0),
                                       (2, 7, 7, 128, 33, 0.25, 123456)):
        x = O.lcg_fill((B, T, C), 1 + T, 1.0)
        emb = O.lcg_fill((B, E, Te), 2 + Te, 1.0)
        w = O.lcg_fill((C, C + E), 3, 1.0 / math.sqrt(C + E))
        b = O.lcg_fill((C,), 4, 0.1)
        gy = O.lcg_fill((B, T, C), 5, 1.0)
        # reference
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        z = torch.cat((xr, torch.nn.functional.adaptive_avg_pool1d(emb, T).transpose(1, 2)), -1)
        if p > 0:
            z = z * np_keep_mask((B, T, C + E), seed, p) / (1.0 - p)
        yr = torch.nn.functional.linear(z, wr, br)
        yr.backward(gy)
        # HIP path
        xd, wd, bd = to(dev, x).requires_grad_(True), to(dev, w).requires_grad_(True), to(dev, b).requires_grad_(True)
        cfg = dict(dropout_p=p, apply_dropout=p > 0, seed=seed)
        y = ops.EmbCatFn.apply(xd, to(dev, emb), wd, bd, cfg)
        y.backward(to(dev, gy))
        tol = lambda ref: 3e-5 * max(1.0, float(ref.detach().abs().max()))      # noqa: E731
        assert float((y.detach().cpu() - yr.detach()).abs().max()) < tol(yr), ("y", B, T, Te, C, E)
        assert float((xd.grad.cpu() - xr.grad).abs().max()) < tol(xr.grad), ("dx", B, T, Te, C, E)
        assert float((wd.grad.cpu() - wr.grad).abs().max()) < tol(wr.grad), ("dw", B, T, Te, C, E)
        assert float((bd.grad.cpu() - br.grad).abs().max()) < tol(br.grad), ("db", B, T, Te, C, E)
    # frozen-CNN variant: no input gradient requested
    y = ops.EmbCatFn.apply(to(dev, x), to(dev, emb), wd, bd, cfg)
    wd.grad = None
    y.sum().backward()
    assert wd.grad is not None


def pretrained_net_config():
    """`net:` of recipes/dcase2023_task4_baseline/confs/pretrained.yaml (BEATs embeddings, 768 x 496 per clip)."""
    cfg = dict(recipe_config()["net"])
    cfg.update(use_embeddings=True, embedding_size=768, embedding_type="frame", aggregation_type="pool1d")
    return cfg


def case_embedding_crnn_vs_reference_golden(dev, golden):
    """CRNN(**pretrained.yaml net) on the HIP kernels against the reference module's recorded outputs
    (tests/golden/golden_emb.npz): eval and train-mode posteriors, loss, all parameter gradients incl. cat_tf."""
    from desed_task_amd.nnet.CRNN import CRNN
    sd = O.make_state_dict(seed=7, embedding_size=768)
    xin = O.lcg_fill((2, 128, 64), 41, 0.5, 0.5)
    emb = O.lcg_fill((2, 768, 51), 42, 1.0)
    cfg = pretrained_net_config()
    net = CRNN(**cfg)
    assert [n for n, _ in net.named_parameters()] == list(golden["param_names"])
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.to(dev) if dev != "cpu" else net
    assert net.arena.is_intact()
    net.eval()
    with torch.no_grad():
        strong, weak = net(to(dev, xin), embeddings=to(dev, emb))
    assert np.abs(strong.cpu().numpy() - golden["eval_strong"]).max() < 2e-5
    assert np.abs(weak.cpu().numpy() - golden["eval_weak"]).max() < 2e-5
    try:
        net(to(dev, xin))
        raise AssertionError("a use_embeddings CRNN must refuse a call without embeddings")
    except ValueError:
        pass
    cfg["dropout"] = 0.0
    net = CRNN(**cfg, specaugm_t_p=0.0, specaugm_f_p=0.0)
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.to(dev) if dev != "cpu" else net
    net.train()
    strong, weak = net(to(dev, xin), embeddings=to(dev, emb))
    assert np.abs(strong.detach().cpu().numpy() - golden["train_strong"]).max() < 2e-5
    assert np.abs(weak.detach().cpu().numpy() - golden["train_weak"]).max() < 2e-5
    tgt_s = to(dev, (O.lcg_fill(tuple(strong.shape), 31, 0.5, 0.5) < 0.2).float())
    tgt_w = to(dev, (O.lcg_fill(tuple(weak.shape), 32, 0.5, 0.5) < 0.3).float())
    loss = torch.nn.functional.binary_cross_entropy(strong, tgt_s) + torch.nn.functional.binary_cross_entropy(weak, tgt_w)
    assert abs(loss.item() - float(golden["loss"][0])) < 2e-6
    loss.backward()
    params = dict(net.named_parameters())
    for n, ref in zip(list(golden["param_names"]), golden["grad_norms"]):
        if n.startswith("cnn.cnn.conv") and n.endswith(".bias"):
            continue                                  # analytically zero (see case_training_step)
        assert abs(params[n].grad.norm().item() - ref) <= 2e-3 * ref + 1e-7, n
    for n, sl in (("cat_tf.bias", lambda g: g), ("cat_tf.weight", lambda g: g[::8, ::7]),
                  ("cnn.cnn.conv6.weight", lambda g: g.reshape(-1)[:512])):
        got, ref = sl(params[n].grad.detach().cpu().numpy()), golden["grad__" + n]
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, n


def case_pretrained_training_step(dev):
    """sed_trainer_pretrained.SEDTask4 (embeddings in the batch, CRNN with cat_tf) x2 steps through the StepDriver against
    the oracle trainer on the same embeddings: losses, posteriors (1e-3), every gradient incl. cat_tf, EMA'd teacher."""
    import random
    from desed_task_amd.launcher import StepDriver
    bs, n_samp, steps = (1, 1, 2), 16000 + 1024, 2
    B = sum(bs)
    sd = O.make_state_dict(seed=7, embedding_size=768)
    audio = O.synth_audio(B, n_samp, seed=77)
    n_out = (1 + n_samp // 256) // 4
    labels = O.synth_labels(bs, 10, n_out, seed=5)
    emb = O.lcg_fill((B, 768, 53), 9, 1.0)                       # BEATs frame rate: ~3.2 embedding frames per CRNN frame
    task = build_task(dev, bs, sd, dropout=0.0, specaug=False, rampup=100, pretrained=True)
    assert "cat_tf.weight" in dict(task.sed_teacher.named_parameters())
    driver = StepDriver(task, world_size=1)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100)
    keys = O.param_keys(sd)
    for step in range(steps):
        random.seed(4); np.random.seed(100 + step); torch.manual_seed(100 + step)
        assert random.random() < 0.5
        cw = np.random.beta(0.2, 0.2); pw = torch.randperm(bs[1]); cs = np.random.beta(0.2, 0.2); ps = torch.randperm(bs[0])
        mix = dict(c_weak=cw, perm_weak=pw, c_strong=cs, perm_strong=ps)
        random.seed(4); np.random.seed(100 + step); torch.manual_seed(100 + step)
        loss = driver.run_step((to(dev, audio.clone()), to(dev, labels.clone()), None, to(dev, emb)), step)
        tot, logs = orc.training_step(audio, labels, mix=mix, embeddings=emb)
        ref_grads = orc.optimizer_step(tot)
        hip_params = dict(task.sed_student.named_parameters())
        for k in keys:
            if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
                continue
            g, r = hip_params[k].grad.detach().cpu(), ref_grads[k]
            rel = 1e-4 if step == 0 else 3e-2
            assert (g - r).abs().max().item() <= rel * r.abs().max().item() + 5e-8, "step %d grad %s" % (step, k)
        got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
        got["loss"] = float(loss.detach().cpu())
        logs["loss"] = tot.item()
        for k in sorted(logs):
            assert abs(got[k] - logs[k]) <= 2e-5 + 2e-4 * abs(logs[k]), "step %d %s: hip %.8g oracle %.8g" % (step, k, got[k], logs[k])
        s_s, w_s, s_t, w_t = [t.detach().cpu() for t in task.last_outputs]
        assert (s_s - orc.last["strong_s"]).abs().max().item() < 1e-3 and (s_t - orc.last["strong_t"]).abs().max().item() < 1e-3
        assert (w_s - orc.last["weak_s"]).abs().max().item() < 1e-3 and (w_t - orc.last["weak_t"]).abs().max().item() < 1e-3
    for k in ("cat_tf.weight", "cat_tf.bias", "dense.weight"):
        for mine, theirs in ((dict(task.sed_student.named_parameters())[k].detach().cpu(), orc.student[k].detach()),
                             (dict(task.sed_teacher.named_parameters())[k].detach().cpu(), orc.teacher[k])):
            upd = (theirs - sd[k]).norm().item()
            assert (mine - theirs).norm().item() <= 0.15 * upd + 1e-6, k
    # a batch without embeddings is refused; e2e is refused at construction
    try:
        task.training_step((to(dev, audio), to(dev, labels), None), 0)
        raise AssertionError("missing embeddings must raise")
    except ValueError:
        pass
    from desed_task_amd.sed_trainer_pretrained import SEDTask4
    try:
        SEDTask4({"pretrained": {"e2e": True}}, None, None)
        raise AssertionError("e2e must be refused")
    except NotImplementedError:
        pass


def case_embcat_full_size(dev):
    """K14 at the pretrained recipe's full size (B 48, 156 frames, BEATs 768 x 496): against torch ops on the same device with
    dropout off; with dropout on, the kept elements are the scaled undropped values and the keep rate is 1 - p."""
    from desed_task_amd import ops
    B, T, Te, C, E, p = 48, 156, 496, 128, 768, 0.5
    g = torch.Generator().manual_seed(3)
    x = to(dev, torch.randn(B, T, C, generator=g))
    emb = to(dev, torch.randn(B, E, Te, generator=g))
    w = to(dev, torch.randn(C, C + E, generator=g) / math.sqrt(C + E)).requires_grad_(True)
    b = to(dev, torch.zeros(C)).requires_grad_(True)
    z_ref = torch.cat((x, torch.nn.functional.adaptive_avg_pool1d(emb, T).transpose(1, 2)), -1)
    y = ops.EmbCatFn.apply(x, emb, w, b, dict(apply_dropout=False))
    y_ref = torch.nn.functional.linear(z_ref, w, b)
    assert float((y - y_ref).abs().max()) < 3e-5 * float(y_ref.abs().max())
    lib = _lib.get()
    z = torch.empty(B, T, C + E, device=x.device)
    thr24, dscale = ops.dropout_params(p)
    lib.call("sed_embcat_fwd", x.data_ptr(), emb.data_ptr(), z.data_ptr(), B, T, Te, C, E, 99, thr24, dscale, None, None, 0, _lib.stream_ptr(x))
    kept = z != 0
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 2e-3, rate
    assert float((z[kept] - z_ref[kept] * dscale).abs().max()) < 1e-5 * float(z_ref.abs().max()) * dscale


def case_dataset_scaler(dev, tmp_dir):
    """SEDTask4._init_scaler with statistic "dataset" (sed_trainer.py:218-250): fitted on the log-mels of the training loader,
    saved, reloaded; against the oracle's mel + log on the same clips."""
    import os
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer import SEDTask4
    config = recipe_config((1, 1, 2))
    path = os.path.join(str(tmp_dir), "scaler.ckpt")
    config["scaler"] = {"statistic": "dataset", "normtype": "standard", "dims": [0, 2], "savepath": path}
    audio = O.synth_audio(6, 8192 + 1024, seed=3)

    class Clips(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            return audio[i], torch.zeros(10, 9), 1.0

    class Enc:
        labels = list(range(10))
    sampler = [[0, 1], [2, 3], [4, 5]]
    task = SEDTask4(config, Enc(), CRNN(**config["net"]), train_data=Clips(), train_sampler=sampler)
    logm = O.take_log(O.mel_spectrogram(audio))
    want_mean = torch.stack([logm[2 * i:2 * i + 2].mean((0, 2), keepdim=True).mean(0).unsqueeze(0) for i in range(3)]).mean(0)
    assert tuple(task.scaler.mean.shape) == (1, 128, 1)
    assert float((task.scaler.mean.cpu() - want_mean).abs().max()) < 2e-3
    assert os.path.exists(path)
    x = to(dev, logm[:2].clone())
    y = task.scaler.to(x.device)(x)
    ref = (logm[:2] - task.scaler.mean.cpu()) / (torch.sqrt(task.scaler.mean_squared.cpu() - task.scaler.mean.cpu() ** 2) + 1e-8)
    assert float((y.cpu() - ref).abs().max()) < 1e-4
    task2 = SEDTask4(config, Enc(), CRNN(**config["net"]))               # no loader needed: loaded from savepath
    assert torch.equal(task2.scaler.mean.cpu(), task.scaler.mean.cpu())


def case_mt_loss(dev):
    """MeanTeacherLossFn (K9) against torch losses, both `self_sup_loss` modes (sed_trainer.py:97-103, :309-342): the six
    scalars, the total and its gradient w.r.t. the student's strong / weak posteriors."""
    from desed_task_amd import ops
    B, T, NC, ns, nw = 7, 13, 10, 2, 3
    g = torch.Generator().manual_seed(5)
    strong_s = torch.rand(B, T, NC, generator=g) * 0.96 + 0.02
    weak_s = torch.rand(B, NC, generator=g) * 0.96 + 0.02
    strong_t = torch.rand(B, T, NC, generator=g) * 0.96 + 0.02
    weak_t = torch.rand(B, NC, generator=g) * 0.96 + 0.02
    labels = (torch.rand(B, NC, T, generator=g) < 0.2).float()
    labels_weak = (torch.rand(nw, NC, generator=g) < 0.3).float()
    weight = 1.37
    bce = torch.nn.functional.binary_cross_entropy
    for mode in ("mse", "bce"):
        selfsup = torch.nn.functional.mse_loss if mode == "mse" else bce
        ss, ws = strong_s.detach().clone().requires_grad_(True), weak_s.detach().clone().requires_grad_(True)
        ref = [bce(ss[:ns], labels[:ns].transpose(1, 2)), bce(ws[ns:ns + nw], labels_weak), bce(strong_t[:ns], labels[:ns].transpose(1, 2)),
               bce(weak_t[ns:ns + nw], labels_weak), selfsup(ss, strong_t), selfsup(ws, weak_t)]
        tot = ref[0] + ref[1] + weight * (ref[4] + ref[5])
        tot.backward()
        sd, wd = to(dev, strong_s.detach().clone()).requires_grad_(True), to(dev, weak_s.detach().clone()).requires_grad_(True)
        scalars, total = ops.MeanTeacherLossFn.apply(sd, wd, to(dev, strong_t), to(dev, weak_t), to(dev, labels), to(dev, labels_weak),
                                                     ns, nw, weight, mode == "bce")
        assert total.dim() == 0 and total.requires_grad and not scalars.requires_grad
        total.backward()
        got = scalars.detach().cpu()
        for k in range(6):
            assert abs(float(got[k]) - float(ref[k])) < 2e-6 * max(1.0, abs(float(ref[k]))), (mode, k)
        assert abs(float(got[6]) - weight * float(ref[4] + ref[5])) < 5e-6 * max(1.0, abs(float(tot)))
        assert abs(float(got[7]) - float(tot)) < 5e-6 * max(1.0, abs(float(tot))) and float(total) == float(got[7])
        assert float((sd.grad.cpu() - ss.grad).abs().max()) < 1e-6 * max(1.0, float(ss.grad.abs().max())), mode
        assert float((wd.grad.cpu() - ws.grad).abs().max()) < 1e-6 * max(1.0, float(ws.grad.abs().max())), mode


# ------------------------------------------------------------------------------------------------
# the BENCHMARKED configuration: dropout + SpecAugment + mixup ON, compared as a whole with the oracle
# ------------------------------------------------------------------------------------------------
class StochasticRecorder:
    """Records what the HIP path drew in one training step so the oracle can be run on identical draws: the dropout seed of
    every call site (7 CNN blocks + post-GRU head [+ embcat] per model) and the SpecAugment bounds per model.  The draws
    themselves are untouched -- `ops.new_seed` and the CNN's prologue still produce them; the recorder only listens and
    tags each draw with the model (student / teacher) whose forward asked for it (the two tails run in swapped order when they
    are overlapped on two HIP streams)."""

    def __init__(self, task):
        import importlib
        cnn_mod = importlib.import_module("desed_task_amd.nnet.CNN")
        crnn_mod = importlib.import_module("desed_task_amd.nnet.CRNN")      # (the package re-exports the class under that name)
        self.rec = {"student": {"seeds": [], "objs": [], "bounds": None, "bounds_all": []},
                    "teacher": {"seeds": [], "objs": [], "bounds": None, "bounds_all": []}}
        self._cur = [None]
        self._mods = (cnn_mod, crnn_mod)
        self._orig_seed = cnn_mod.new_seed
        self._orig_bounds = Fh.specaug_bounds
        rec, cur = self.rec, self._cur

        def new_seed(generator=None, _o=self._orig_seed):
            s = _o(generator)
            rec[cur[0]]["seeds"].append(int(s))
            rec[cur[0]]["objs"].append(s)           # graph.DynSeed under a hipGraph step: the value lives in DynArgs
            return s

        def specaug_bounds(*a, _o=self._orig_bounds, **k):
            b = _o(*a, **k)
            rec[cur[0]]["bounds"] = b
            rec[cur[0]]["bounds_all"].append(b)
            return b

        cnn_mod.new_seed = new_seed
        crnn_mod.new_seed = new_seed
        Fh.specaug_bounds = specaug_bounds
        self._models = []
        for name, model in (("student", task.sed_student), ("teacher", task.sed_teacher)):
            for meth in ("forward_cnn", "forward_tail"):
                orig = getattr(model, meth)

                def wrapped(*a, _o=orig, _n=name, _m=model, _meth=meth, **k):
                    prev, cur[0] = cur[0], _n
                    try:
                        out = _o(*a, **k)
                        if _meth == "forward_cnn" and _m.training:
                            # the bands are drawn inside the CNN's one-launch prologue (no features.specaug_bounds call to listen to)
                            b = _m.cnn.last_bounds
                            if b is not None:
                                rec[_n]["bounds"] = b
                                rec[_n]["bounds_all"].append(b)
                        return out
                    finally:
                        cur[0] = prev
                object.__setattr__(model, meth, wrapped)
                self._models.append((model, meth))

    def reset(self):
        for v in self.rec.values():
            v["seeds"], v["objs"], v["bounds"], v["bounds_all"] = [], [], None, []

    def close(self):
        for m in self._mods:
            m.new_seed = self._orig_seed
        Fh.specaug_bounds = self._orig_bounds
        for model, meth in self._models:
            object.__delattr__(model, meth)

    def seed_values(self, who, dyn=None):
        """The seeds `who` drew, in call order: recorded ints (eager) or, under a captured step, the CURRENT contents of the DynArgs
        host mirror for the recorded call sites (= the draws of the last step run)."""
        r = self.rec[who]
        return list(r["seeds"]) if dyn is None else [dyn.seed_value(o) for o in r["objs"]]

    @staticmethod
    def draws_from(seeds8, bounds, B, n_frames, p=0.5):
        """(aug, drop_masks) in the oracle's conventions from explicit values: the 7 CNN seeds + the head's, and a (B, 4) bounds
        tensor -- for steps whose draws were made at different times (pipelined front half: the teacher's CNN one step early)."""
        b = bounds.cpu().long()
        aug = dict(f=(b[:, 0], b[:, 1]), t=(b[:, 2], b[:, 3]))
        assert len(seeds8) == 8
        masks = []
        T, Fq = n_frames, 128
        for i, co in enumerate(O.NB_FILTERS):
            masks.append(np_keep_mask((B, T, Fq, co), seeds8[i], p).permute(0, 3, 1, 2))
            T, Fq = T // O.POOLING[i][0], Fq // O.POOLING[i][1]
        masks.append(np_keep_mask((B, T, 256), seeds8[7], p))
        return aug, masks

    def oracle_draws(self, who, B, n_frames, p=0.5, embedding_size=None, dyn=None):
        """-> (aug, drop_masks) in the oracle's conventions (masks NCHW for the CNN blocks, (B,T',256) for the head).
        dyn: the graph.DynArgs of a captured step -- the seeds of the LAST step run (capture or replay) are then read from its host
        mirror (the call sites recorded during the capture keep their slots), the bounds from the graph's static tensor."""
        r = self.rec[who]
        b = r["bounds"].cpu().long()
        aug = dict(f=(b[:, 0], b[:, 1]), t=(b[:, 2], b[:, 3]))
        seeds = list(r["seeds"]) if dyn is None else [dyn.seed_value(o) for o in r["objs"]]
        assert len(seeds) == (8 if embedding_size is None else 9), (who, len(seeds))
        masks = []
        T, Fq = n_frames, 128
        for i, co in enumerate(O.NB_FILTERS):
            masks.append(np_keep_mask((B, T, Fq, co), seeds[i], p).permute(0, 3, 1, 2))
            T, Fq = T // O.POOLING[i][0], Fq // O.POOLING[i][1]
        if embedding_size is None:
            masks.append(np_keep_mask((B, T, 256), seeds[7], p))
        else:           # forward_tail order: embcat seed first, then the head's
            masks.append(np_keep_mask((B, T, 256), seeds[8], p))
            masks.append(np_keep_mask((B, T, 128 + embedding_size), seeds[7], p))
        return aug, masks


def _mixup_draws(bs, seeds):
    """The draws SEDTask4.training_step will make after seeding (coin, beta, randperm weak, beta, randperm strong)."""
    import random
    random.seed(seeds[0]); np.random.seed(seeds[1]); torch.manual_seed(seeds[2])
    mix = None
    if 0.5 > random.random():
        cw = np.random.beta(0.2, 0.2); pw = torch.randperm(bs[1]); cs = np.random.beta(0.2, 0.2); ps = torch.randperm(bs[0])
        mix = dict(c_weak=cw, perm_weak=pw, c_strong=cs, perm_strong=ps)
    random.seed(seeds[0]); np.random.seed(seeds[1]); torch.manual_seed(seeds[2])
    return mix


def _mixup_draws_n(bs, seeds, n):
    """The draws of the next n front halves after seeding ONCE (pipelined front end: step k + 1's draws are made during step k)."""
    import random
    random.seed(seeds[0]); np.random.seed(seeds[1]); torch.manual_seed(seeds[2])
    out = []
    for _ in range(n):
        mix = None
        if 0.5 > random.random():
            cw = np.random.beta(0.2, 0.2); pw = torch.randperm(bs[1]); cs = np.random.beta(0.2, 0.2); ps = torch.randperm(bs[0])
            mix = dict(c_weak=cw, perm_weak=pw, c_strong=cs, perm_strong=ps)
        out.append(mix)
    random.seed(seeds[0]); np.random.seed(seeds[1]); torch.manual_seed(seeds[2])
    return out


DIAG = None      # a list here collects the error statistics of case_training_step (diagnostics)
STATS = None     # a list here collects (name, max, median) of every step-0 gradient comparison (diagnostics)


def grad_error_stats(g, r):
    """(max, median) over the elements of max(|g - r| - 5e-8, 0) / max |r| for one gradient tensor."""
    d = ((g - r).abs().reshape(-1) - 5e-8).clamp(min=0)        # 5e-8 absolute: gradients that are themselves rounding residue
    s = max(r.abs().max().item(), 1e-30)
    return d.max().item() / s, d.median().item() / s


def case_stochastic_training_step(dev, bs=(2, 2, 4), n_samp=16000 * 2 + 1024, steps=2, grads=True):
    """`steps` full mean-teacher steps with dropout (all 8 sites per model, the head's included), SpecAugment and mixup ON --
    the configuration bench.py times -- against OracleTrainer.training_step(..., aug_s, aug_t, drop_s, drop_t) on the draws the
    HIP path made (reference: local/sed_trainer.py:304-342, desed_task/nnet/CRNN.py:207-219,303-306, CNN.py:90-91).
    Done = posteriors <= 1e-3 abs, logged scalars <= 2e-4 rel, step-0 gradients <= 1e-4 of the per-tensor maximum."""
    from desed_task_amd.launcher import StepDriver
    torch.set_num_threads(min(32, torch.get_num_threads()))
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    audio = O.synth_audio(B, n_samp, seed=78)
    n_frames = 1 + n_samp // 256
    n_out = n_frames // 4
    labels = O.synth_labels(bs, 10, n_out, seed=5)
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=100)
    driver = StepDriver(task, world_size=1)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100)
    rec = StochasticRecorder(task)
    worst = {"post": 0.0, "scalar": 0.0, "grad_max": 0.0, "grad_med": 0.0}
    try:
        mixed = []
        for step in range(steps):
            mix = _mixup_draws(bs, (4 + step, 100 + step, 100 + step))       # seeds 4 -> mixup on, 5 -> off
            mixed.append(mix is not None)
            rec.reset()
            loss = driver.run_step((to(dev, audio.clone()), to(dev, labels.clone()), None, None), step)
            aug_s, drop_s = rec.oracle_draws("student", B, n_frames)
            aug_t, drop_t = rec.oracle_draws("teacher", B, n_frames)
            assert rec.rec["student"]["seeds"] != rec.rec["teacher"]["seeds"]
            tot, logs = orc.training_step(audio, labels, mix=mix, aug_s=aug_s, aug_t=aug_t, drop_s=drop_s, drop_t=drop_t)
            ref_grads = orc.optimizer_step(tot) if grads else None
            got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
            got["loss"] = float(loss.detach().cpu()); logs["loss"] = tot.item()
            for k in sorted(logs):
                a, b = got[k], logs[k]
                worst["scalar"] = max(worst["scalar"], abs(a - b) / max(abs(b), 1e-1))
                assert abs(a - b) <= 2e-5 + 2e-4 * abs(b), "step %d %s: hip %.8g oracle %.8g" % (step, k, a, b)
            for a, name in zip([t.detach().cpu() for t in task.last_outputs], ("strong_s", "weak_s", "strong_t", "weak_t")):
                err = (a - orc.last[name]).abs().max().item()
                worst["post"] = max(worst["post"], err)
                assert err < 1e-3, "step %d %s: %.3e" % (step, name, err)
            if grads:
                hip_params = dict(task.sed_student.named_parameters())
                for k in O.PARAM_KEYS:
                    if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
                        continue                                   # analytically zero (see case_training_step)
                    emax, emed = grad_error_stats(hip_params[k].grad.detach().cpu(), ref_grads[k])
                    if step == 0:
                        worst["grad_max"], worst["grad_med"] = max(worst["grad_max"], emax), max(worst["grad_med"], emed)
                        if STATS is not None:
                            STATS.append((k, emax, emed))
                        assert emax <= 1e-4 and emed <= 1e-5, "step 0 grad %s: max %.3e median %.3e" % (k, emax, emed)
                    else:
                        # later steps inherit Adam's sign flips of near-zero gradient elements (a few parameter elements move by
                        # +-lr instead of -+lr): the bulk must still agree -- median error -- while single elements may not
                        assert emed <= 2e-3 and emax <= 6e-2, "step %d grad %s: max %.3e median %.3e" % (step, k, emax, emed)
        assert any(mixed) or steps < 2
    finally:
        rec.close()
    return worst


def case_long_horizon_training(dev, bs=(2, 2, 4), n_samp=16000 + 1024, steps=300, window=50, graph_from=None):
    """Training DYNAMICS, not single steps (VERDICT r03 missing #2: the reference's acceptance is a 200-epoch PSDS that needs the
    DESED audio no box has): `steps` consecutive optimiser steps of the full mean-teacher recipe -- a FRESH synthetic batch every
    step, dropout on all 8 sites per model, SpecAugment, mixup, lr warm-up, consistency ramp-up, EMA teacher, Adam -- on the HIP
    path and, on the draws the HIP path made, on OracleTrainer (reference order: local/sed_trainer.py:269-365, train_sed.py:188-202).
    Both sides carry their OWN weights forward: nothing is resynchronised, so any systematic difference in a gradient, in Adam,
    in the schedule or in the EMA compounds over the run.
    Done = total loss within 1e-3 relative on each of the first 50 steps, the loss-curve mean over every window of `window` steps
    within 1 %, the last window's consistency loss within 2 %; the final student / teacher weight distances are returned (Adam
    turns the sign of a rounding-level gradient element into +-lr, so single weights drift apart while the curves coincide).
    graph_from = k: steps >= k run through GraphedStepDriver replays (the benchmarked launch path) instead of eager launches."""
    from desed_task_amd.launcher import StepDriver
    from desed_task_amd.graph import GraphedStepDriver
    torch.set_num_threads(min(32, torch.get_num_threads()))
    B = sum(bs)
    n_frames = 1 + n_samp // 256
    n_out = n_frames // 4
    sd = O.make_state_dict(seed=7)
    rampup = max(20, steps // 3)
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=rampup)
    driver = StepDriver(task, world_size=1) if graph_from is None else GraphedStepDriver(task, world_size=1, warmup=graph_from)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=rampup)
    rec = StochasticRecorder(task)
    hip, ref, hip_self, ref_self, n_mixed = [], [], [], [], 0
    try:
        for step in range(steps):
            audio = O.synth_audio(B, n_samp, seed=1000 + step)
            labels = O.synth_labels(bs, 10, n_out, seed=2000 + step)
            mix = _mixup_draws(bs, (4 + step, 100 + step, 100 + step))
            n_mixed += mix is not None
            replay = graph_from is not None and step > graph_from
            if not replay:
                rec.reset()
            loss = driver.run_step((to(dev, audio), to(dev, labels.clone()), None, None), step)
            dyn = driver.dyn if (graph_from is not None and step >= graph_from) else None
            aug_s, drop_s = rec.oracle_draws("student", B, n_frames, dyn=dyn)
            aug_t, drop_t = rec.oracle_draws("teacher", B, n_frames, dyn=dyn)
            tot, logs = orc.training_step(audio, labels, mix=mix, aug_s=aug_s, aug_t=aug_t, drop_s=drop_s, drop_t=drop_t)
            orc.optimizer_step(tot)
            hip.append(float(loss.detach().cpu())); ref.append(tot.item())
            hip_self.append(float(task.logged["train/student/tot_self_loss"])); ref_self.append(float(logs["train/student/tot_self_loss"]))
            if step < 50:
                assert abs(hip[-1] - ref[-1]) <= 1e-3 * abs(ref[-1]), "step %d: loss hip %.7g oracle %.7g" % (step, hip[-1], ref[-1])
    finally:
        rec.close()
    hip_t, ref_t = torch.tensor(hip, dtype=torch.float64), torch.tensor(ref, dtype=torch.float64)
    out = {"steps": steps, "mixed_steps": int(n_mixed), "first_loss": ref[0], "last_loss": ref[-1], "windows": [],
           "max_rel_first50": float(((hip_t - ref_t).abs() / ref_t.abs())[:50].max())}
    for w0 in range(0, steps - window + 1, window):
        a, b = hip_t[w0:w0 + window].mean().item(), ref_t[w0:w0 + window].mean().item()
        out["windows"].append((w0, round(a, 6), round(b, 6), abs(a - b) / abs(b)))
        assert abs(a - b) <= 1e-2 * abs(b), "loss-curve mean over steps [%d, %d): hip %.6g oracle %.6g" % (w0, w0 + window, a, b)
    a, b = sum(hip_self[-window:]) / window, sum(ref_self[-window:]) / window
    out["self_loss_last_window"] = (a, b)
    assert abs(a - b) <= 2e-2 * abs(b) + 1e-6, "consistency loss, last window: hip %.6g oracle %.6g" % (a, b)
    assert ref[-1] < 0.9 * ref[0] or steps < 100, "the run should actually train (loss %.4g -> %.4g)" % (ref[0], ref[-1])
    for who, model, refsd in (("student", task.sed_student, orc.student), ("teacher", task.sed_teacher, orc.teacher)):
        num = den = 0.0
        worst = 0.0
        for k, p_ in model.named_parameters():
            d = (p_.detach().cpu().double() - refsd[k].detach().double())
            num += float((d * d).sum()); den += float((refsd[k].detach().double() ** 2).sum())
            worst = max(worst, float(d.abs().max()))
        out[who + "_rel_l2"] = (num / den) ** 0.5
        out[who + "_max_abs"] = worst
    # the weights moved a lot further from their start than the two runs moved apart
    moved = 0.0
    for k, p_ in task.sed_student.named_parameters():
        moved += float(((p_.detach().cpu().double() - sd[k].double()) ** 2).sum())
    out["student_moved_l2"] = moved ** 0.5
    return out


def case_head_dropout(dev, B=3, T=39, p=0.5, seed=4242, D=256, NC=10):
    """HeadFn (post-GRU Dropout(0.5) + dense + dense_softmax + class-softmax attention pooling, CRNN.py:152-178,:304) forward
    and backward against torch ops on the same keep mask."""
    from desed_task_amd.ops import HeadFn
    x = O.lcg_fill((B, T, D), 61, 1.0)
    w1 = O.lcg_fill((NC, D), 62, 1.0 / 16); b1 = O.lcg_fill((NC,), 63, 0.1)
    w2 = O.lcg_fill((NC, D), 64, 1.0 / 16); b2 = O.lcg_fill((NC,), 65, 0.1)
    gs = O.lcg_fill((B, T, NC), 66, 1.0); gw = O.lcg_fill((B, NC), 67, 1.0)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    h = ref_in[0] * np_keep_mask((B, T, D), seed, p) / (1.0 - p)
    strong = torch.sigmoid(torch.nn.functional.linear(h, ref_in[1], ref_in[2]))
    sof = torch.softmax(torch.nn.functional.linear(h, ref_in[3], ref_in[4]), dim=-1).clamp(min=1e-7, max=1)
    weak = (strong * sof).sum(1) / sof.sum(1)
    ((strong * gs).sum() + (weak * gw).sum()).backward()
    hip_in = [to(dev, t).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    s_h, w_h = HeadFn.apply(*hip_in, dict(dropout_p=p, apply_dropout=p > 0, seed=seed))
    ((s_h * to(dev, gs)).sum() + (w_h * to(dev, gw)).sum()).backward()
    assert (s_h.detach().cpu() - strong.detach()).abs().max().item() < 2e-6
    assert (w_h.detach().cpu() - weak.detach()).abs().max().item() < 2e-6
    for nm, a, b in zip(("dx", "dW1", "db1", "dW2", "db2"), hip_in, ref_in):
        emax, _ = grad_error_stats(a.grad.detach().cpu(), b.grad)
        # (db2, the class-softmax bias, is a sum of terms that cancel over the classes: atomics-order rounding shows there first)
        assert emax < 5e-5, "%s: %.3e" % (nm, emax)
    kept = (hip_in[0].grad.detach().cpu() != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.05 if p > 0 else kept > 0.99


def case_b48_forward_vs_oracle(dev, bs=(12, 12, 24), single_bf16=False):
    """BASELINE config C2 itself -- 48 clips (12/12/24) of 10 s, dropout + SpecAugment + mixup ON, student != teacher weights --
    compared with the oracle, forward pass: student and teacher posteriors (1e-3 abs), the six loss scalars, all 7 + 7
    BatchNorm running statistics (batch-size dependent reductions) and the per-clip min/max of the scaler.

    single_bf16=True: the same forward with the numerics of a ONE-product bf16 convolution (what a plain `bf16` line would compute):
    the 3 x 3 weights of blocks 1-6 and the activations entering those convolutions are rounded to bf16 first, so the lo planes of the
    split operands are zero and hi*hi + hi*lo + lo*hi collapses to the single product hi*hi (fp32 accumulation, everything else as
    shipped).  Nothing is asserted but finiteness: the posterior errors against the fp32 oracle are RETURNED -- the number DESIGN.md
    quotes for why the contractions are issued as three MFMAs per product."""
    torch.set_num_threads(min(64, torch.get_num_threads()))
    B, n_samp = sum(bs), 160000
    n_frames = 1 + n_samp // 256
    sd, sd_t = O.make_state_dict(seed=13), O.make_state_dict(seed=14)
    audio = O.synth_audio(B, n_samp, seed=5)
    labels = O.synth_labels(bs, 10, 156, seed=6)
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=100)
    task.sed_teacher.load_state_dict({k: v.clone() for k, v in sd_t.items()})
    assert task.sed_teacher.arena.is_intact()
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100, teacher_sd=sd_t)
    rec = StochasticRecorder(task)
    import importlib
    cnn_mod = importlib.import_module("desed_task_amd.nnet.CNN")
    orig_block = cnn_mod.ConvBlockFn
    if single_bf16:
        with torch.no_grad():
            for model in (task.sed_student, task.sed_teacher):
                for i in range(1, 7):
                    w = getattr(model.cnn.cnn, "conv%d" % i).weight
                    w.copy_(w.to(torch.bfloat16).float())

        class _RoundedInput:
            @staticmethod
            def apply(x, *a):
                return orig_block.apply(x.to(torch.bfloat16).float() if x.dim() == 4 else x, *a)
        cnn_mod.ConvBlockFn = _RoundedInput
    try:
        mix = _mixup_draws(bs, (4, 100, 100))
        assert mix is not None
        audio_d = to(dev, audio)
        with torch.no_grad():
            _, mm = Fh.minmax_scale(task.mel_spec(audio_d), apply_log=True, return_minmax=True)
            loss = task.training_step((audio_d, to(dev, labels.clone()), None, None), 0)
        aug_s, drop_s = rec.oracle_draws("student", B, n_frames)
        aug_t, drop_t = rec.oracle_draws("teacher", B, n_frames)
    finally:
        rec.close()
        cnn_mod.ConvBlockFn = orig_block
    with torch.no_grad():
        logm = O.take_log(O.mel_spectrogram(audio))
        tot, logs = orc.training_step(audio, labels, mix=mix, aug_s=aug_s, aug_t=aug_t, drop_s=drop_s, drop_t=drop_t)
    mm = mm.cpu()
    assert (mm[:, 0] - logm.amin((1, 2))).abs().max().item() < 2e-3 and (mm[:, 1] - logm.amax((1, 2))).abs().max().item() < 2e-3   # dB
    out = {}
    for a, name in zip([t.detach().cpu() for t in task.last_outputs], ("strong_s", "weak_s", "strong_t", "weak_t")):
        out[name] = (a - orc.last[name]).abs().max().item()
        assert single_bf16 or out[name] < 1e-3, "%s: %.3e" % (name, out[name])
        assert math.isfinite(out[name])
    got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
    got["loss"] = float(loss.detach().cpu()); logs["loss"] = tot.item()
    if single_bf16:
        out["loss_rel"] = abs(got["loss"] - logs["loss"]) / abs(logs["loss"])
        return out
    for k in sorted(logs):
        assert abs(got[k] - logs[k]) <= 2e-5 + 2e-4 * abs(logs[k]), "%s: hip %.8g oracle %.8g" % (k, got[k], logs[k])
    for who, model, ref in (("student", task.sed_student, orc.student), ("teacher", task.sed_teacher, orc.teacher)):
        for i in range(7):
            bn = getattr(model.cnn.cnn, "batchnorm%d" % i)
            for nm, a in (("running_mean", bn.running_mean), ("running_var", bn.running_var)):
                b = ref["cnn.cnn.batchnorm%d.%s" % (i, nm)].detach()
                err = (a.cpu() - b).abs().max().item()
                assert err <= 2e-5 * max(1.0, b.abs().max().item()), "%s bn%d %s: %.3e" % (who, i, nm, err)
    return out


def case_b48_graph_step_vs_oracle(dev, bs=(12, 12, 24), n_samp=160000, warmup=1, replays=1, tol_scale=1.0, prefetch=None, surface="driver"):
    """The benchmarked configuration THROUGH THE BENCHMARKED LAUNCH PATH, with the backward pass: B = 48 (12/12/24) clips of 10 s,
    dropout + SpecAugment + mixup on, student != teacher, run by graph.GraphedStepDriver -- `warmup` eager steps, the capture step,
    `replays` replayed steps -- and EVERY step compared with OracleTrainer on the draws the HIP path made: logged scalars <= 2e-4
    rel, posteriors <= 1e-3 abs, all student gradients <= 1e-4 (max) / 1e-5 (median) of the per-tensor maximum
    (local/sed_trainer.py:269-365).  The launch geometry of the weight-gradient / split-K / partial-sum kernels depends on the
    batch, so only this size exercises what bench.py times.  The student is put back on its initial weights after every step (on
    both sides): Adam's +-lr sign flips of near-zero gradient elements would otherwise widen the later steps' tolerances; Adam's
    moments, the schedule and the EMA teacher keep evolving, so step-varying arguments still change from replay to replay.

    prefetch="teacher" (bench.py's default front end): a DIFFERENT batch every step; step k's front half (mel, mixup, log / min-max)
    and the teacher's CNN forward ran under step k - 1's backward, so the draws that belong to step k's oracle step were made at
    three different times -- mixup and the teacher's CNN seeds / SpecAugment bounds during step k - 1, the student's and the
    teacher head's during step k -- and are carried over accordingly.  Runs unchanged under SED_DDP_REHEARSE=1 with a one-rank
    process group (graph up to the end of backward + RCCL all-reduce + eager Adam: the N > 1 structure).

    surface="lightning" (with prefetch="teacher"): the same steps, but nobody calls the driver -- tests/lightning_order.Trainer runs
    Lightning 1.9's hook order over `train_dataloader()` with a torch.optim.Adam, and SEDTask4's whole-step mode does the rest
    (its own GraphedStepDriver, the next batch from the look-ahead loader).  The epoch's last batch has no successor: one more step,
    run eagerly by the driver's fall-back, compared like the others."""
    from desed_task_amd.graph import GraphedStepDriver
    from desed_task_amd.lookahead import BatchList
    torch.set_num_threads(min(64, torch.get_num_threads()))
    B = sum(bs)
    n_frames = 1 + n_samp // 256
    lightning = surface == "lightning"
    n_steps = warmup + 1 + replays + (1 if lightning else 0)
    pipelined = prefetch is not None
    assert prefetch in (None, "teacher") and (pipelined or not lightning)
    sd, sd_t = O.make_state_dict(seed=13), O.make_state_dict(seed=14)
    # (driven by hand, the last step announces one more batch so that it is a replay too)
    n_batches = (n_steps if lightning else n_steps + 1) if pipelined else 1
    audios = [O.synth_audio(B, n_samp, seed=5 + 31 * i) for i in range(n_batches)]
    labelss = [O.synth_labels(bs, 10, n_frames // 4, seed=6 + i) for i in range(n_batches)]
    audio_d = [to(dev, a) for a in audios]
    loader_batches = [(audio_d[i], to(dev, labelss[i].clone()), [1.0] * B) for i in range(n_batches)] if lightning else None
    task = build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=100, torch_adam=lightning, whole_step=lightning,
                      train_data=BatchList(loader_batches) if lightning else None)
    task.whole_step_warmup = warmup
    task.sed_teacher.load_state_dict({k: v.clone() for k, v in sd_t.items()})
    driver = None if lightning else GraphedStepDriver(task, world_size=1, warmup=warmup, prefetch=prefetch)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100, teacher_sd=sd_t)
    flat0 = task.sed_student.arena.flat.detach().clone()
    rec = StochasticRecorder(task)
    worst = {"post": 0.0, "scalar": 0.0, "grad_max": 0.0, "grad_med": 0.0, "modes": [],
             "exchange": bool(driver.eager.exchange) if driver is not None else False}
    mixes, t_cnn = {}, {}           # pipelined: per step, the mixup draws / (teacher CNN seeds, bounds) made one step early
    ctx = {}

    def mode_of(step):
        if step < warmup:
            return "eager"
        return "capture" if step == warmup else ("replay" if step < warmup + 1 + replays else "eager-last")

    def before(step):
        seeds = (4 + step, 100 + step, 100 + step)                       # seeds 4 -> mixup on, 5 -> off, 6 -> on
        if not pipelined:
            ctx["mix"] = _mixup_draws(bs, seeds)
        elif step == 0:
            mixes[0], mixes[1] = _mixup_draws_n(bs, seeds, 2)            # inline front half of step 0, then step 1's prefetch
        else:
            mixes[step + 1] = _mixup_draws_n(bs, seeds, 1)[0]
        if mode_of(step) != "replay":
            rec.reset()                         # a replay re-runs no Python: the capture's call sites stay valid

    def after(step, loss):
            driver = ctx["driver"]
            mode = mode_of(step)
            bi = step if pipelined else 0
            audio, labels = audios[bi], labelss[bi]
            mix = ctx.get("mix")
            torch.cuda.synchronize()
            assert (driver.graph is not None) == (mode != "eager")
            dyn = driver.dyn if mode in ("capture", "replay") else None
            aug_s, drop_s = rec.oracle_draws("student", B, n_frames, dyn=dyn)
            if not pipelined:
                aug_t, drop_t = rec.oracle_draws("teacher", B, n_frames, dyn=dyn)
            else:
                mix = mixes[step]
                tv = rec.seed_values("teacher", dyn=dyn)
                tb = rec.rec["teacher"]["bounds_all"]
                if step == 0:                       # inline: [7 CNN seeds (step 0), head (0)] then the prefetch's [7 CNN seeds (step 1)]
                    assert len(tv) == 15 and len(tb) == 2, (len(tv), len(tb))
                    t_cnn[0] = (tv[0:7], tb[0].cpu().clone())
                    head, t_cnn[1] = tv[7], (tv[8:15], tb[1].cpu().clone())
                elif mode == "eager-last":          # no successor: the eager fall-back, [head (k)] only
                    assert len(tv) == 1 and len(tb) == 0, (len(tv), len(tb))
                    head = tv[0]
                else:                               # [head (k)] then the prefetch's [7 CNN seeds (k + 1)]; bounds: the graph's static tensor
                    assert len(tv) == 8 and len(tb) == 1, (len(tv), len(tb))
                    head, t_cnn[step + 1] = tv[0], (tv[1:8], tb[0].cpu().clone())
                aug_t, drop_t = StochasticRecorder.draws_from(list(t_cnn[step][0]) + [head], t_cnn[step][1], B, n_frames)
                assert driver.eager_fallbacks == (1 if mode == "eager-last" else 0) and driver.reprimes == 0
            tot, logs = orc.training_step(audio, labels, mix=mix, aug_s=aug_s, aug_t=aug_t, drop_s=drop_s, drop_t=drop_t)
            ref_grads = orc.optimizer_step(tot)
            got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
            got["loss"] = float(loss.detach().cpu()); logs["loss"] = tot.item()
            for k in sorted(logs):
                a, b = got[k], logs[k]
                worst["scalar"] = max(worst["scalar"], abs(a - b) / max(abs(b), 1e-1))
                assert abs(a - b) <= tol_scale * (2e-5 + 2e-4 * abs(b)), "step %d (%s) %s: hip %.8g oracle %.8g" % (step, mode, k, a, b)
            for a, name in zip([t.detach().cpu() for t in task.last_outputs], ("strong_s", "weak_s", "strong_t", "weak_t")):
                err = (a - orc.last[name]).abs().max().item()
                worst["post"] = max(worst["post"], err)
                assert err < 1e-3, "step %d (%s) %s: %.3e" % (step, mode, name, err)
            hip_params = dict(task.sed_student.named_parameters())
            n_checked = 0
            for k in O.PARAM_KEYS:
                if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
                    continue                                   # analytically zero (see case_training_step)
                emax, emed = grad_error_stats(hip_params[k].grad.detach().cpu(), ref_grads[k])
                worst["grad_max"], worst["grad_med"] = max(worst["grad_max"], emax), max(worst["grad_med"], emed)
                if STATS is not None:
                    STATS.append((mode, k, emax, emed))
                assert emax <= tol_scale * 1e-4 and emed <= tol_scale * 1e-5, \
                    "step %d (%s) grad %s: max %.3e median %.3e" % (step, mode, k, emax, emed)
                n_checked += 1
            assert n_checked == len(O.PARAM_KEYS) - 7
            # the teacher after the step's EMA (alpha = 1 - 1/(step_num + 1): 1/2, 2/3, 3/4 ...) -- read back from the graph's arena
            t_err = max((dict(task.sed_teacher.named_parameters())[k].detach().cpu() - orc.teacher[k]).abs().max().item() for k in O.PARAM_KEYS)
            assert t_err <= 2e-6, "step %d (%s) teacher after EMA: %.3e" % (step, mode, t_err)
            # Adam moved the weights, none by more than lr * (1 - b1) / sqrt(1 - b2) (its worst case with a gradient history); then
            # both students go back to their initial weights
            moved = (task.sed_student.arena.flat.detach() - flat0).abs().max().item()
            assert 0 < moved <= 1e-3 * 3.17, (step, moved)       # (the first step runs at the optimizer's own lr)
            task.sed_student.arena.flat.copy_(flat0)
            with torch.no_grad():
                for k in orc.keys:
                    orc.student[k].copy_(sd[k])
            worst["modes"].append(mode)

    try:
        if lightning:
            from tests.lightning_order import Trainer
            task.on_train_batch_start = lambda batch, i: before(i)

            def batch_end(out, batch, i):
                ctx["driver"] = task._driver
                after(i, out["loss"])
            task.on_train_batch_end = batch_end
            Trainer(max_epochs=1).fit(task)
            assert worst["modes"][-1] == "eager-last" and type(task.opt).__name__ == "FusedAdam"
        else:
            ctx["driver"] = driver
            for step in range(n_steps):
                before(step)
                bi = step if pipelined else 0
                if pipelined:       # (protocol: the batch IS the tensor announced one step earlier)
                    batch = (audio_d[bi], to(dev, labelss[bi].clone()), None, None)
                    nxt = (audio_d[bi + 1], to(dev, labelss[bi + 1].clone()), None, None)
                    loss = driver.run_step(batch, step, next_batch=nxt)
                else:
                    loss = driver.run_step((audio_d[bi].clone(), to(dev, labelss[bi].clone()), None, None), step)
                after(step, loss)
        assert worst["modes"].count("replay") == replays and "capture" in worst["modes"]
        if pipelined:
            assert any(m is not None for m in mixes.values()) and any(m is None for m in list(mixes.values())[:n_steps])
    finally:
        rec.close()
    return worst


# ------------------------------------------------------------------------------------------------
# rest of the embedding-fusion surface: classes_mask / pad_mask, dropstep_recurrent, "interpolate"
# ------------------------------------------------------------------------------------------------
def net_config_2024():
    """`net:` of recipes/dcase2024_task4_baseline/confs/pretrained.yaml (n_RNN_cell = 192, 27 classes)."""
    cfg = dict(recipe_config()["net"])
    cfg.update(dropout=0.2, rnn_layers=1, nclass=27, n_RNN_cell=192, dropstep_recurrent=0.3, dropstep_recurrent_len=16,
               use_embeddings=True, embedding_size=768, embedding_type="frame", aggregation_type="pool1d",
               specaugm_t_p=0.0, specaugm_t_l=5, specaugm_f_p=0.0, specaugm_f_l=10)
    return cfg


def golden_emb2_inputs():
    xin = O.lcg_fill((3, 128, 64), 41, 0.5, 0.5)
    emb = O.lcg_fill((3, 768, 51), 42, 1.0)
    cm = torch.zeros(3, 27, dtype=torch.bool)
    cm[0, :10] = True; cm[1, 10:] = True; cm[2] = True
    pad = torch.zeros(3, 1, 16, dtype=torch.bool)
    pad[0, 0, 12:] = True; pad[1, 0, 15:] = True
    return xin, emb, cm, pad


def case_crnn_masks_vs_reference_golden(dev, golden):
    """CRNN with the 2024 recipe's options on the HIP kernels against the reference module's recorded outputs
    (tests/golden/golden_emb2.npz): classes_mask + pad_mask inside the head kernels, dropstep_recurrent on the recorded spans
    (with embeddings: two spans fused into the embcat kernel; without: the time-mask + dropout kernel), gradients of every
    parameter; eval-mode posteriors of aggregation_type "interpolate"."""
    from desed_task_amd.nnet.CRNN import CRNN
    xin, emb, cm, pad = golden_emb2_inputs()
    for tag, use_emb in (("a", True), ("b", False)):
        cfg = dict(net_config_2024(), dropout=0.0, use_embeddings=use_emb)
        sd = O.make_state_dict(seed=7, nclass=27, embedding_size=768 if use_emb else None, hidden=192)
        net = CRNN(**cfg)
        assert [n for n, _ in net.named_parameters()] == list(golden[tag + "_param_names"])
        net.load_state_dict({k: v.clone() for k, v in sd.items()})
        net = net.to(dev) if dev != "cpu" else net
        net.train()
        spans = [torch.from_numpy(s.astype(np.int32)) for s in golden[tag + "_dropstep"]]
        order = iter(spans)
        net._dropstep_bounds = lambda B, n_time, device, _o=order: next(_o).to(device).contiguous()      # the reference's own draws
        strong, weak = net(to(dev, xin), pad_mask=to(dev, pad), embeddings=to(dev, emb) if use_emb else None, classes_mask=to(dev, cm))
        assert np.abs(strong.detach().cpu().numpy() - golden[tag + "_strong"]).max() < 2e-5, tag
        assert np.abs(weak.detach().cpu().numpy() - golden[tag + "_weak"]).max() < 2e-5, tag
        assert float(strong.detach()[0, 10:].abs().max()) == 0.0 and float(weak.detach()[1, :10].abs().max()) == 0.0
        tgt_s = to(dev, (O.lcg_fill(tuple(strong.shape), 31, 0.5, 0.5) < 0.2).float())
        tgt_w = to(dev, (O.lcg_fill(tuple(weak.shape), 32, 0.5, 0.5) < 0.3).float())
        loss = torch.nn.functional.binary_cross_entropy(strong, tgt_s) + torch.nn.functional.binary_cross_entropy(weak, tgt_w)
        assert abs(loss.item() - float(golden[tag + "_loss"][0])) < 2e-5 * float(golden[tag + "_loss"][0]), tag
        loss.backward()
        params = dict(net.named_parameters())
        for n, ref in zip(list(golden[tag + "_param_names"]), golden[tag + "_grad_norms"]):
            if n.startswith("cnn.cnn.conv") and n.endswith(".bias"):
                continue
            assert abs(params[n].grad.norm().item() - ref) <= 2e-3 * ref + 1e-7, (tag, n)
        got = params["dense_softmax.weight"].grad.detach().cpu().numpy()[:, ::8]
        ref = golden[tag + "_grad__dense_softmax.weight"]
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, tag
        got = params["cnn.cnn.conv6.weight"].grad.detach().cpu().numpy().reshape(-1)[:512]
        ref = golden[tag + "_grad__cnn.cnn.conv6.weight"]
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, tag
    cfg = dict(net_config_2024(), dropout=0.0, aggregation_type="interpolate")
    sd = O.make_state_dict(seed=7, nclass=27, embedding_size=768, hidden=192)
    net = CRNN(**cfg)
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    net = net.to(dev) if dev != "cpu" else net
    net.eval()
    with torch.no_grad():
        strong, weak = net(to(dev, xin), embeddings=to(dev, emb))
    assert np.abs(strong.cpu().numpy() - golden["c_strong"]).max() < 2e-5
    assert np.abs(weak.cpu().numpy() - golden["c_weak"]).max() < 2e-5
    for agg in ("frame", "global"):
        try:
            CRNN(**dict(cfg, aggregation_type=agg))
            raise AssertionError("aggregation_type %s must be refused" % agg)
        except NotImplementedError:
            pass


def case_dropstep_draws_and_dropout(dev):
    """dropstep_recurrent with its own draws and the dropout of the same call site on: the time spans follow torchaudio's mask
    arithmetic on torch.rand draws (as SpecAugment's), DropStepFn / EmbCatFn equal torch ops on the same spans and keep mask."""
    from desed_task_amd import ops
    B, T, C, E, Te, p = 3, 16, 128, 40, 51, 0.5
    x = O.lcg_fill((B, T, C), 3, 1.0); emb = O.lcg_fill((B, E, Te), 4, 1.0)
    bx = torch.tensor([[2, 5], [0, 0], [14, 16]], dtype=torch.int32)
    be = torch.tensor([[0, 4], [7, 9], [3, 3]], dtype=torch.int32)
    gy = O.lcg_fill((B, T, C), 5, 1.0)
    # without embeddings
    xr = x.clone().requires_grad_(True)
    ref = O.time_mask(xr, (bx[:, 0].long(), bx[:, 1].long())) * np_keep_mask((B, T, C), 77, p) / (1 - p)
    ref.backward(gy)
    xd = to(dev, x).clone().requires_grad_(True)
    y = ops.DropStepFn.apply(xd, to(dev, bx), dict(dropout_p=p, apply_dropout=True, seed=77))
    y.backward(to(dev, gy))
    assert torch.equal(y.detach().cpu(), ref.detach()) and torch.equal(xd.grad.cpu(), xr.grad)
    # with embeddings (both spans, pool1d and interpolate)
    w = O.lcg_fill((C, C + E), 6, 0.1); bias = O.lcg_fill((C,), 7, 0.1)
    for mode in (0, 1):
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if mode == 0:
            re = torch.nn.functional.adaptive_avg_pool1d(emb, T).transpose(1, 2)
        else:
            re = torch.nn.functional.interpolate(emb.unsqueeze(1), size=(E, T), mode="nearest-exact").squeeze(1).transpose(1, 2)
        z = torch.cat((O.time_mask(xr, (bx[:, 0].long(), bx[:, 1].long())), O.time_mask(re, (be[:, 0].long(), be[:, 1].long()))), -1)
        z = z * np_keep_mask((B, T, C + E), 99, p) / (1 - p)
        yr = torch.nn.functional.linear(z, wr, bias)
        yr.backward(gy)
        xd, wd = to(dev, x).clone().requires_grad_(True), to(dev, w).clone().requires_grad_(True)
        cfg = dict(dropout_p=p, apply_dropout=True, seed=99, tmask=to(dev, torch.cat((bx, be), 1).contiguous()), mode=mode)
        yd = ops.EmbCatFn.apply(xd, to(dev, emb), wd, to(dev, bias).requires_grad_(True), cfg)
        yd.backward(to(dev, gy))
        tol = lambda r: 3e-5 * max(1.0, float(r.detach().abs().max()))      # noqa: E731
        assert float((yd.detach().cpu() - yr.detach()).abs().max()) < tol(yr), mode
        assert float((xd.grad.cpu() - xr.grad).abs().max()) < tol(xr.grad), mode
        assert float((wd.grad.cpu() - wr.grad).abs().max()) < tol(wr.grad), mode
    # the CRNN's own draws: reproducible from torch's seed, within the cap min(len, int(T * p))
    from desed_task_amd.nnet.CRNN import CRNN
    net = CRNN(**dict(net_config_2024(), use_embeddings=False))
    net = net.to(dev) if dev != "cpu" else net
    device = torch.device(dev)
    torch.manual_seed(5)
    if dev != "cpu":
        torch.cuda.manual_seed(5)
    b1 = net._dropstep_bounds(48, 156, device).cpu()
    assert tuple(b1.shape) == (48, 2) and int((b1[:, 1] - b1[:, 0]).max()) <= 16 and int(b1.min()) >= 0 and int(b1[:, 1].max()) <= 156
    assert int((b1[:, 1] - b1[:, 0]).max()) > 0


def inputs_2024(bs=(2, 1, 1, 2, 2), nclass=27):
    """== tests/golden/make_golden_2024.py::inputs()"""
    B = sum(bs)
    n_samp = 16000 * 2 + 1024
    audio = O.synth_audio(B, n_samp, seed=77)
    n_out = (1 + n_samp // 256) // 4
    labels = (O.lcg_fill((B, nclass, n_out), 5, 0.5, 0.5) < 0.1).float()
    ns = bs[0] + bs[1] + bs[2]
    labels[ns:ns + bs[3], :, 1:] = 0.0
    labels[ns + bs[3]:] = 0.0
    emb = O.lcg_fill((B, 768, 53), 9, 1.0)
    valid = torch.zeros(B, nclass, dtype=torch.bool)
    valid[:bs[0], 10:] = True
    valid[bs[0]:, :10] = True
    valid[bs[0] + 1] = True
    return audio, labels, emb, valid


def case_training_step_2024(dev, golden):
    """desed_task_amd.sed_trainer_pretrained_2024.SEDTask4 (the 2024 recipe's 5-data-set step with class masks, mixup of features
    AND embeddings per data set, consistency losses without MAESTRO) x2 steps through the StepDriver against (i) the scalars the
    reference's own training_step logged (tests/golden/golden_2024.npz) and (ii) the oracle on the same draws: posteriors,
    every gradient."""
    import random
    from desed_task_amd.arena import FusedAdam
    from desed_task_amd.launcher import StepDriver
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer_pretrained_2024 import SEDTask4
    from desed_task_amd.utils.schedulers import ExponentialWarmup
    bs, nclass = (2, 1, 1, 2, 2), 27
    audio, labels, emb, valid = inputs_2024(bs, nclass)
    config = recipe_config(bs)
    config["training"].update(mixup_prob=0.5, epoch_decay=100)
    config["net"] = dict(net_config_2024(), dropout=0.0, dropstep_recurrent=0.0)
    config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
    sd = O.make_state_dict(seed=7, nclass=nclass, embedding_size=768, hidden=192)
    student = CRNN(**config["net"])
    student.load_state_dict({k: v.clone() for k, v in sd.items()})
    student = student.to(dev) if dev != "cpu" else student
    opt = FusedAdam(student.parameters(), lr=1e-3, betas=(0.9, 0.999), arena=student)
    sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 100), "interval": "step"}

    class Enc:
        labels = list(range(nclass))
    task = SEDTask4(config, Enc(), student, None, opt=opt, scheduler=sched)
    task.train()
    if dev != "cpu":
        task.to(dev)
    driver = StepDriver(task, world_size=1)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100)
    keys = O.param_keys(sd)
    gkeys = list(golden["keys"])
    for step in range(2):
        random.seed(4 + step); np.random.seed(100 + step); torch.manual_seed(100 + step)
        gate = 0.5 > random.random()
        assert gate == (step == 0)
        mix = None
        if gate:
            mix = []
            for n in (bs[3], bs[1] + bs[2], bs[0]):
                for _ in range(2):
                    mix.append((np.random.beta(0.2, 0.2), torch.randperm(n)))
            np.testing.assert_allclose([c for c, _ in mix], golden["mix_c"], rtol=0, atol=0)         # the reference drew the same
        random.seed(4 + step); np.random.seed(100 + step); torch.manual_seed(100 + step)
        loss = driver.run_step((to(dev, audio.clone()), to(dev, labels.clone()), None, to(dev, emb.clone()), to(dev, valid.clone())), step)
        tot, logs = orc.training_step_2024(audio, labels, emb, valid, mix=mix)
        ref_grads = orc.optimizer_step(tot)
        got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
        got["loss"] = float(loss.detach().cpu()); logs["loss"] = tot.item()
        for k in sorted(logs):
            assert abs(got[k] - logs[k]) <= 2e-5 + 2e-4 * abs(logs[k]), "step %d %s: hip %.8g oracle %.8g" % (step, k, got[k], logs[k])
        for k, b in zip(gkeys, golden["values"][step]):
            assert abs(got[k] - b) <= 2e-5 + 5e-4 * abs(b), "step %d %s: hip %.8g reference %.8g" % (step, k, got[k], b)
        for a, name in zip([t.detach().cpu() for t in task.last_outputs], ("strong_s", "weak_s", "strong_t", "weak_t")):
            assert (a - orc.last[name]).abs().max().item() < 1e-3, (step, name)
        hip_params = dict(task.sed_student.named_parameters())
        for k in keys:
            if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
                continue
            emax, emed = grad_error_stats(hip_params[k].grad.detach().cpu(), ref_grads[k])
            assert (emax <= 1e-4 and emed <= 1e-5) if step == 0 else (emax <= 6e-2 and emed <= 2e-3), "step %d grad %s: %.2e / %.2e" % (step, k, emax, emed)
    st = dict(task.sed_student.named_parameters())
    for n in ("cat_tf.weight", "cnn.cnn.conv0.weight"):
        ref = golden["student_after2__" + n]
        init = sd[n].numpy().reshape(-1)[:256]
        mine = st[n].detach().cpu().numpy().reshape(-1)[:256]
        assert np.linalg.norm(mine - ref) <= 0.15 * np.linalg.norm(ref - init) + 1e-6, n
    # recurrent widths outside the two recipes' are refused, loudly
    try:
        CRNN(**dict(config["net"], n_RNN_cell=256))
        raise AssertionError("n_RNN_cell = 256 must be refused")
    except NotImplementedError:
        pass


# ------------------------------------------------------------------------------------------------
# frozen BEATs extractor
# ------------------------------------------------------------------------------------------------
def build_task_2024(dev, bs=(2, 1, 1, 2, 2), nclass=27, dropout=0.5, dropstep=0.3, seed=7, torch_adam=False, whole_step=False,
                    train_data=None):
    """The 2024 recipe's task (27 classes, n_RNN_cell 192, frozen 768-d embeddings, dropstep_recurrent) on closed-form weights."""
    from desed_task_amd.arena import FusedAdam
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer_pretrained_2024 import SEDTask4
    from desed_task_amd.utils.schedulers import ExponentialWarmup
    config = recipe_config(bs)
    config["training"].update(mixup_prob=0.5, epoch_decay=100)
    config["net"] = dict(net_config_2024(), dropout=dropout, dropstep_recurrent=dropstep)
    if dropout > 0:
        config["net"].update(specaugm_t_p=0.2, specaugm_f_p=0.2)
    config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
    sd = O.make_state_dict(seed=seed, nclass=nclass, embedding_size=768, hidden=192)
    student = CRNN(**config["net"])
    student.load_state_dict({k: v.clone() for k, v in sd.items()})
    student = student.to(dev) if dev != "cpu" else student
    if torch_adam:
        opt = torch.optim.Adam(student.parameters(), 1e-3, betas=(0.9, 0.999))
    else:
        opt = FusedAdam(student.parameters(), lr=1e-3, betas=(0.9, 0.999), arena=student)
    sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 5), "interval": "step"}

    class Enc:
        labels = list(range(nclass))
    task = SEDTask4(config, Enc(), student, None, opt=opt, scheduler=sched, train_data=train_data)
    task.whole_step = whole_step
    task.train()
    if dev != "cpu":
        task.to(dev)
    return task


def case_prefetch_2024_equals_unpipelined(dev, graph=False, steps=5, n_samp=16000 + 1024, te=31):
    """The 2024 five-data-set step with its front half (mel, per-data-set mixup of features AND embeddings, weak labels, log /
    min-max) and the teacher's CNN forward pipelined under the previous step's backward == the unpipelined order, bit for bit, over
    different batches, with dropout + SpecAugment + dropstep + mixup on; the last step announces no successor.  The announced labels
    and embeddings are only read (mixed in the hand-over buffers).  graph=True: GraphedStepDriver (eager, capture, replays, eager
    fallback at the end)."""
    import random
    from desed_task_amd import graph as G
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver
    bs, nclass = (2, 1, 1, 2, 2), 27
    B = sum(bs)
    n_out = (1 + n_samp // 256) // 4
    ns = bs[0] + bs[1] + bs[2]
    batches = []
    for i in range(steps):
        audio = O.synth_audio(B, n_samp, seed=700 + 13 * i)
        labels = (O.lcg_fill((B, nclass, n_out), 50 + i, 0.5, 0.5) < 0.1).float()
        labels[ns:ns + bs[3], :, 1:] = 0.0
        labels[ns + bs[3]:] = 0.0
        emb = O.lcg_fill((B, 768, te), 90 + i, 1.0)
        valid = torch.zeros(B, nclass, dtype=torch.bool)
        valid[:bs[0], 10:] = True
        valid[bs[0]:, :10] = True
        batches.append(tuple(to(dev, t) for t in (audio, labels, emb, valid)))
    originals = [(b[1].clone(), b[2].clone()) for b in batches]
    results = []
    for mode in ("plain", "pipelined"):
        task = build_task_2024(dev, bs, nclass)
        pf = "teacher" if mode == "pipelined" else None
        driver = G.GraphedStepDriver(task, world_size=1, warmup=1, prefetch=pf) if graph else StepDriver(task, world_size=1, prefetch=pf)
        random.seed(41); np.random.seed(101); torch.manual_seed(101)
        if dev != "cpu":
            torch.cuda.manual_seed(101)
        _ops.reseed_dropout()
        losses = []
        for step in range(steps):
            a, l, e, v = batches[step]
            if mode == "pipelined":
                batch = (a, l, None, e, v)                                   # announced tensors are only read: no private copies
                nxt = (batches[step + 1][0], batches[step + 1][1], None, batches[step + 1][2], batches[step + 1][3]) if step + 1 < steps else None
                loss = driver.run_step(batch, step, next_batch=nxt)
            else:
                loss = driver.run_step((a, l.clone(), None, e.clone(), v), step)      # (the plain step mixes its batch in place)
            losses.append(float(loss.detach()))
        if dev != "cpu":
            torch.cuda.synchronize()
        if mode == "pipelined":
            for b, (lo, eo) in zip(batches[1:], originals[1:]):
                assert torch.equal(b[1], lo) and torch.equal(b[2], eo), "an announced tensor was modified in place"
            assert task._pro is not None and "embeddings" in task._pro
            if graph:
                assert driver.next_extra_buffers().keys() == {"embeddings"} and driver.eager_fallbacks == 1
        results.append((losses, task.sed_student.arena.flat.detach().cpu().clone(), task.sed_teacher.arena.flat.detach().cpu().clone()))
    (l0, s0, t0), (l1, s1, t1) = results
    assert l0 == l1, (l0, l1)
    assert torch.equal(s0, s1) and torch.equal(t0, t1)
    return l0


def np_kaldi_fbank64(wave, n_mels=128):
    """Independent float64 numpy implementation of Kaldi's fbank with the options BEATs uses (cross-check of the unpinned front-end)."""
    x = wave.astype(np.float64) * 32768.0
    m = 1 + (len(x) - 400) // 160
    fr = np.stack([x[i * 160:i * 160 + 400] for i in range(m)])
    fr = fr - fr.mean(1, keepdims=True)
    fr = fr - 0.97 * np.concatenate([fr[:, :1], fr[:, :-1]], 1)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 399)) ** 0.85
    spec = np.abs(np.fft.rfft(fr * win, 512, axis=1)) ** 2
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)      # noqa: E731
    lo, hi = mel(20.0), mel(8000.0)
    d = (hi - lo) / (n_mels + 1)
    melf = mel(31.25 * np.arange(257))
    bank = np.zeros((n_mels, 257))
    for b in range(n_mels):
        l, c, r = lo + b * d, lo + (b + 1) * d, lo + (b + 2) * d
        bank[b] = np.maximum(0.0, np.minimum((melf - l) / (c - l), (r - melf) / (r - c)))
    bank[:, 256] = 0.0
    return np.log(np.maximum(spec @ bank.T, np.finfo(np.float32).eps))


def case_beats_fbank(dev):
    from oracle import beats_oracle as BO
    from desed_task_amd.beats import KaldiFbank
    audio = O.synth_audio(2, 400 + 160 * 37 + 55, seed=21)               # ragged tail: the last partial frame is dropped
    fb = KaldiFbank(128)
    got = fb(to(dev, audio)).cpu()                                      # mean 0, std 0.5 -> raw log energies
    ref = torch.stack([BO.kaldi_fbank(w * 2 ** 15) for w in audio])
    assert tuple(got.shape) == tuple(ref.shape) == (2, 38, 128)
    assert (got - ref).abs().max().item() < 2e-3                        # log domain; fp32 FFT order differs
    ref64 = np.stack([np_kaldi_fbank64(w.numpy()) for w in audio])
    assert np.abs(got.numpy() - ref64).max() < 2e-3 and np.abs(ref.numpy() - ref64).max() < 2e-3
    norm = fb(to(dev, audio), 15.41663, 6.55582).cpu()
    assert (norm - (ref - 15.41663) / (2 * 6.55582)).abs().max().item() < 2e-4


def case_beats_vs_reference_golden(dev, golden):
    """desed_task_amd.beats.BEATs (2 encoder layers of the iter3 configuration, closed-form weights) on the HIP kernels against the
    output of the reference module itself (tests/golden/golden_beats.npz), plus the BEATsModel wrapper's global / frame outputs."""
    from oracle import beats_oracle as BO
    from desed_task_amd.beats import BEATs, BEATsConfig, BEATsModel
    cfg = dict(BO.BEATS_ITER3_CFG, encoder_layers=int(golden["cfg_layers"][0]))
    sd = BO.make_beats_state_dict(cfg, seed=3)
    model = BEATs(BEATsConfig(cfg))
    assert set(model.state_dict().keys()) == set(golden["state_dict_keys"]) == set(sd.keys())
    model.load_state_dict(sd)
    model = model.to(dev) if dev != "cpu" else model
    model.eval()
    audio = O.synth_audio(2, 400 + 255 * 160, seed=21)
    feats, pm = model.extract_features(to(dev, audio))
    assert pm is None and tuple(feats.shape) == (2, 128, 768)
    ref = golden["features"]
    err = np.abs(feats.cpu().numpy() - ref).max()
    assert err < 2e-4 * max(1.0, np.abs(ref).max()), err
    fb = model.preprocess(to(dev, audio)).cpu().numpy()[:, ::7, ::5]
    assert np.abs(fb - golden["fbank"]).max() < 2e-4
    wrapped = BEATsModel(checkpoint={"cfg": cfg, "model": sd})
    wrapped = wrapped.to(dev) if dev != "cpu" else wrapped
    out = wrapped(to(dev, audio))
    assert tuple(out["frame"].shape) == (2, 768, 128) and tuple(out["global"].shape) == (2, 768)
    assert np.abs(out["frame"].cpu().numpy() - ref.transpose(0, 2, 1)).max() < 2e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(out["global"].cpu().numpy() - ref.mean(1)).max() < 1e-4
    try:
        model.extract_features(to(dev, audio), padding_mask=torch.zeros(2, 41200, dtype=torch.bool))
        raise AssertionError("padding masks must be refused")
    except NotImplementedError:
        pass


def case_beats12_vs_reference_golden(dev, golden):
    """The extractor at its real depth and length against the reference module's own output (tests/golden/golden_beats12.npz):
    all 12 layers (deep-norm alpha = 24 ** 0.25, backbone.py:216-220,268,280), one 10 s clip = 496 tokens (ragged last attention
    tile of 64; bucket table beyond the exact range, backbone.py:400-445; gate, :662-682)."""
    from oracle import beats_oracle as BO
    from desed_task_amd.beats import BEATs, BEATsConfig
    cfg = dict(BO.BEATS_ITER3_CFG)
    assert cfg["encoder_layers"] == int(golden["cfg_layers"][0]) == 12
    sd = BO.make_beats_state_dict(cfg, seed=5)
    model = BEATs(BEATsConfig(cfg))
    model.load_state_dict(sd)
    model = model.to(dev) if dev != "cpu" else model
    model.eval()
    audio = O.synth_audio(1, 160000, seed=23)
    feats, _ = model.extract_features(to(dev, audio))
    assert tuple(feats.shape) == (1, 496, 768)
    f = feats.cpu().numpy()
    scale = max(1.0, float(golden["abs_max"][0]))
    e1 = np.abs(f[:, :, ::4] - golden["features_ch4"]).max()
    e2 = np.abs(f[:, ::8, :] - golden["features_tok8"]).max()
    assert max(e1, e2) < 3e-4 * scale, (e1, e2)
    return max(e1, e2)


def case_beats_chain_2024(dev, bs=(12, 6, 6, 12, 24), n_samp=160000, layers=12, nclass=27):
    """BASELINE config 4 as ONE chain: waveforms -> frozen BEATs extractor (`BEATsModel(...)(audio)["frame"]`, the producer of the
    recipe's embeddings, recipes/dcase2023_task4_baseline/extract_embeddings.py:46-51,62-66) -> (B, 768, tokens) -> the 2024
    recipe's five-data-set training step (recipes/dcase2024_task4_baseline/local/sed_trainer_pretrained.py:318-430) on the SAME
    waveforms, against oracle(BEATs) -> oracle(step): embeddings, logged scalars, posteriors and every gradient.  (The reference's
    own `pretrained.e2e` branch raises, :305-316 -- the chain in the recipe runs through the hdf5 file; here the tensor is handed
    over in HBM.)  Mixup on (features AND embeddings), class masks on, dropout / dropstep off (those draws are covered by
    case_stochastic_training_step / case_dropstep_draws_and_dropout)."""
    import random
    from oracle import beats_oracle as BO
    from desed_task_amd.arena import FusedAdam
    from desed_task_amd.beats import BEATsModel
    from desed_task_amd.launcher import StepDriver
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer_pretrained_2024 import SEDTask4
    from desed_task_amd.utils.schedulers import ExponentialWarmup
    torch.set_num_threads(min(64, max(8, torch.get_num_threads())))
    B = sum(bs)
    audio = O.synth_audio(B, n_samp, seed=31)
    n_out = (1 + n_samp // 256) // 4
    labels = (O.lcg_fill((B, nclass, n_out), 5, 0.5, 0.5) < 0.1).float()
    ns = bs[0] + bs[1] + bs[2]
    labels[ns:ns + bs[3], :, 1:] = 0.0
    labels[ns + bs[3]:] = 0.0
    valid = torch.zeros(B, nclass, dtype=torch.bool)
    valid[:bs[0], 10:] = True
    valid[bs[0]:, :10] = True
    # ---- stage 1: the extractor ----------------------------------------------------------------------------------------
    bcfg = dict(BO.BEATS_ITER3_CFG, encoder_layers=layers)
    bsd = BO.make_beats_state_dict(bcfg, seed=5)
    extractor = BEATsModel(checkpoint={"cfg": bcfg, "model": bsd})
    extractor = extractor.to(dev) if dev != "cpu" else extractor
    audio_d = to(dev, audio)
    emb_d = extractor(audio_d)["frame"]
    n_tok = ((1 + (n_samp - 400) // 160) // 16) * 8
    assert tuple(emb_d.shape) == (B, 768, n_tok)
    with torch.no_grad():
        emb_o = torch.cat([BO.beats_embeddings(bsd, bcfg, audio[i:i + 6])["frame"] for i in range(0, B, 6)])
    # Embeddings: 5e-4 of the feature scale on 99.9 % of the elements.  The maximum gets a wider bound on purpose: a fp32 filterbank
    # is ill-conditioned in the rare (frame, mel bin) whose energy is ~1e-14 of the frame's peak (an FFT bin the clip's components
    # happen to cancel in): there the fp32 FFT rounding IS the value -- the oracle's rfft (pocketfft) and the kernel's radix-2 FFT
    # are 5e-4 / 2e-3 off the float64 filterbank in the same bins ("BEATs front-end conditioning") -- and one such
    # bin moves one token's embedding (measured on the MI355X: 99.9 % quantile 2.4e-4, maximum 1.7e-3 of 60 x 768 x 496 values of magnitude 5.8).
    d_emb = (emb_d.cpu() - emb_o).abs()
    scale = max(1.0, emb_o.abs().max().item())
    e_emb, q_emb = d_emb.max().item(), torch.quantile(d_emb.flatten()[::7], 0.999).item()
    assert q_emb < 5e-4 * scale and e_emb < 1e-2 * scale, (q_emb, e_emb)
    # ---- stage 2: the 2024 step on the extractor's output ------------------------------------------------------------
    config = recipe_config(bs)
    config["training"].update(mixup_prob=0.5, epoch_decay=100)
    config["net"] = dict(net_config_2024(), dropout=0.0, dropstep_recurrent=0.0)
    config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
    sd = O.make_state_dict(seed=7, nclass=nclass, embedding_size=768, hidden=192)
    student = CRNN(**config["net"])
    student.load_state_dict({k: v.clone() for k, v in sd.items()})
    student = student.to(dev) if dev != "cpu" else student
    opt = FusedAdam(student.parameters(), lr=1e-3, betas=(0.9, 0.999), arena=student)
    sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 100), "interval": "step"}

    class Enc:
        labels = list(range(nclass))
    task = SEDTask4(config, Enc(), student, None, opt=opt, scheduler=sched)
    task.train()
    if dev != "cpu":
        task.to(dev)
    driver = StepDriver(task, world_size=1)
    orc = O.OracleTrainer(sd, batch_sizes=bs, lr=1e-3, rampup_len=100)
    random.seed(4); np.random.seed(100); torch.manual_seed(100)
    assert 0.5 > random.random()
    mix = []
    for n in (bs[3], bs[1] + bs[2], bs[0]):
        for _ in range(2):
            mix.append((np.random.beta(0.2, 0.2), torch.randperm(n)))
    random.seed(4); np.random.seed(100); torch.manual_seed(100)
    loss = driver.run_step((audio_d, to(dev, labels.clone()), None, emb_d.clone(), to(dev, valid.clone())), 0)
    tot, logs = orc.training_step_2024(audio, labels, emb_o, valid, mix=mix)
    ref_grads = orc.optimizer_step(tot)
    got = {k: (float(v) if not torch.is_tensor(v) else float(v.detach().cpu())) for k, v in task.logged.items()}
    got["loss"] = float(loss.detach().cpu()); logs["loss"] = tot.item()
    worst = {"emb_max": e_emb, "emb_q999": q_emb, "post": 0.0, "grad_max": 0.0, "grad_med": 0.0}
    for k in sorted(logs):
        assert abs(got[k] - logs[k]) <= 2e-5 + 2e-4 * abs(logs[k]), "%s: hip %.8g oracle %.8g" % (k, got[k], logs[k])
    for a, name in zip([t.detach().cpu() for t in task.last_outputs], ("strong_s", "weak_s", "strong_t", "weak_t")):
        err = (a - orc.last[name]).abs().max().item()
        worst["post"] = max(worst["post"], err)
        assert err < 1e-3, (name, err)
    hip_params = dict(task.sed_student.named_parameters())
    for k in O.param_keys(sd):
        if k.startswith("cnn.cnn.conv") and k.endswith(".bias"):
            continue
        emax, emed = grad_error_stats(hip_params[k].grad.detach().cpu(), ref_grads[k])
        worst["grad_max"], worst["grad_med"] = max(worst["grad_max"], emax), max(worst["grad_med"], emed)
        assert emax <= 2e-4 and emed <= 2e-5, "grad %s: %.2e / %.2e" % (k, emax, emed)       # (the two sides' embeddings differ by ~5e-5)
    return worst


def case_attention_relpos(dev, B=2, T=100, H=2, gated=True, bias=True, variant=0):
    """sed_attention_relpos (BEATs K-B5; variant 0 = matrix-core kernel, 1 = vector-pipe kernel) against a float64 restatement of
    backbone.py:529-531,640-682: scores = q k^T / 8 + gate * bias[s - t], softmax over the keys, times v.  Ragged last tiles."""
    from desed_task_amd import _lib
    torch.manual_seed(3)
    hd, D = 64, 64 * H
    qkv = torch.randn(B * T, 3 * D) * 0.7
    relb = torch.randn(H, 2 * T - 1) * 0.5 if bias else None
    gw, gb, ga = (torch.randn(8, hd) * 0.2, torch.randn(8) * 0.1, torch.randn(H) * 0.5 + 1.0) if gated else (None, None, None)
    q = qkv[:, :D].view(B, T, H, hd).permute(0, 2, 1, 3).double()
    k = qkv[:, D:2 * D].view(B, T, H, hd).permute(0, 2, 1, 3).double()
    v = qkv[:, 2 * D:].view(B, T, H, hd).permute(0, 2, 1, 3).double()
    sc = q @ k.transpose(-1, -2) / 8.0
    if bias:
        idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1             # [t][s] -> s - t + T - 1
        bm = relb.double()[:, idx]                                                      # (H, T, T)
        gate = torch.ones(B, H, T, dtype=torch.float64)
        if gated:
            proj = q @ gw.double().t() + gb.double()                                    # (B,H,T,8)
            g_a, g_b = torch.sigmoid(proj[..., :4].sum(-1)), torch.sigmoid(proj[..., 4:].sum(-1))
            gate = g_a * (g_b * ga.double()[None, :, None] - 1.0) + 2.0
        sc = sc + gate[..., None] * bm[None]
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    out = torch.empty(B * T, D, device=dev)
    qd = to(dev, qkv)
    dv = [to(dev, t) if t is not None else None for t in (relb, gw, gb, ga)]
    ptr = lambda t: t.data_ptr() if t is not None else None          # noqa: E731
    _lib.set_tuning("attn_valu", variant)
    try:
        _lib.get().call("sed_attention_relpos", qd.data_ptr(), ptr(dv[0]), ptr(dv[1]), ptr(dv[2]), ptr(dv[3]), out.data_ptr(), B, T, H, hd,
                        _lib.stream_ptr(out))
    finally:
        _lib.set_tuning("attn_valu", 0)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), (variant, err)


def case_posconv(dev, B=2, T=100, groups=2, K=128):
    """BEATs position convolution (K-B4): the split-bf16 MFMA entry and the f32 vector-pipe entry against a float64 restatement of
    x + GELU(SamePad(grouped Conv1d(k, padding k/2))) (backbone.py:30-43,118-120).  Ragged last token tile, halo at both ends."""
    from desed_task_amd import _lib
    torch.manual_seed(5)
    CG = 48
    D = CG * groups
    x = torch.randn(B, T, D) * 0.8
    w = torch.randn(D, CG, K) / np.sqrt(CG * K) * 3.0                   # Conv1d weight (out, in / groups, k)
    bias = torch.randn(D) * 0.1
    conv = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), bias.double(), padding=K // 2, groups=groups)[:, :, :T]
    ref = x.double() + torch.nn.functional.gelu(conv.transpose(1, 2))
    wt = w.view(groups, CG, CG, K).permute(0, 3, 1, 2).contiguous()        # (groups, K, co, ci)
    w_hi = wt.to(torch.bfloat16)
    wsplit = torch.stack((w_hi, (wt - w_hi.float()).to(torch.bfloat16))).contiguous().view(torch.int16)
    xd, wtd, wsd, bd = to(dev, x), to(dev, wt), to(dev, wsplit), to(dev, bias)
    lib = _lib.get()
    for entry, wbuf, tol in (("sed_posconv_bf16x3", wsd, 2e-5), ("sed_posconv", wtd, 5e-6)):
        y = torch.empty_like(xd)
        lib.call(entry, xd.data_ptr(), wbuf.data_ptr(), bd.data_ptr(), y.data_ptr(), B, T, D, K, groups, _lib.stream_ptr(xd))
        err = (y.cpu().double() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item()), (entry, err)
