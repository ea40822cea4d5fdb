"""CPU (fiber-emulator) run of the whole mean-teacher step: drop-in CRNN + SEDTask4 + StepDriver against the
oracle trainer on identical mixup draws (dropout / SpecAugment off).  Also state-dict compatibility."""
import os

import numpy as np
import pytest
import torch

from oracle import sed_oracle as O
from tests import parity_cases as P
from tests.emu_support import emu, emu_sequential  # noqa: F401


def test_state_dict_layout(emu):
    from desed_task_amd.nnet.CRNN import CRNN
    net = CRNN(**P.recipe_config()["net"])
    sd = net.state_dict()
    shapes = O.crnn_param_shapes()
    assert [n for n, _ in net.named_parameters()] == O.PARAM_KEYS            # reference parameters() order
    assert sum(p.numel() for p in net.parameters()) == 1112420
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    buffers = [k for k in sd if k not in shapes]
    assert len(buffers) == 21 and all("batchnorm" in k for k in buffers)
    net.load_state_dict(O.make_state_dict(seed=7), strict=True)
    assert net.arena.is_intact()
    assert net.eval() is None                                                # reference quirk Q5


def test_training_steps_match_oracle(emu):
    P.case_training_step("cpu", small=True)


def test_crnn_matches_reference_golden(emu):
    import os
    import numpy as np
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))
    P.case_crnn_vs_reference_golden("cpu", G)


def test_edge_shapes(emu):
    P.case_edge_shapes("cpu")


def test_dyn_args_step_equals_eager(emu):
    """graph.DynArgs: dropout seeds, mixup c/perm, loss weight, EMA factor and Adam factors read from memory.  Bit-exact even with the
    emulator's workgroups on a thread pool: since round 3 no kernel of the step adds floats with atomics."""
    P.case_dyn_args_step("cpu", graph=False, steps=2, n_samp=8000 + 1024, seed0=1)       # seeds 41, 42: mixup on, then off


def test_prefetched_front_end_equals_unpipelined(emu):
    """Software-pipelined mel front-end == the unpipelined order, bit for bit, over a sequence of different batches (the
    "backward" fork point and the hipGraph form run on the GPU: tests/test_gpu_parity.py)."""
    P.case_prefetch_equals_unpipelined("cpu", point="tails", steps=2, n_samp=2048 + 1024)


def test_prefetched_teacher_forward_equals_unpipelined(emu):
    """The whole front half of step k + 1 and the teacher's CNN forward under step k's backward == the unpipelined order, bit for
    bit (the teacher's CNN draws its seeds from its own private stream, so running it early changes no mask)."""
    P.case_prefetch_equals_unpipelined("cpu", point="teacher", steps=3, n_samp=2048 + 1024, protocol=False)


def test_parked_side_launches_keep_gradients_in_the_arena(emu):
    """The PARKING logic of the launches that leave the critical chain (ops.defer_off_chain: the head's weight-gradient sums, the BiGRU
    side sections; ops.AFTER_FORWARD: the loss sums) on the emulator, where parked launches go out at the flush points in place
    (ops.SIDE_ON_CPU): the step must equal the unparked one bit for bit, every .grad must still be the arena's view -- autograd only
    adopts a returned gradient nobody else references; round 4 shipped a parked launch that held the gradient tensors, for an hour --
    and nothing may stay parked.  (The stream side of it runs on the GPU: test_side_stream_backward_leaves_every_gradient_in_the_arena.)"""
    import random
    import numpy as np
    import torch
    from desed_task_amd import ops
    from desed_task_amd.launcher import StepDriver
    O = P.O
    bs, n_samp = (1, 1, 2), 22 * 256
    out = []
    for hook in (False, True):
        ops.SIDE_ON_CPU = hook
        try:
            task = P.build_task("cpu", bs, O.make_state_dict(seed=7), dropout=0.5, specaug=True, rampup=5)
            d = StepDriver(task, world_size=1)
            assert d.gru_dw_side is hook
            audio = O.synth_audio(sum(bs), n_samp, seed=100)
            labels = O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5)
            random.seed(40); np.random.seed(100); torch.manual_seed(100)
            ops.reseed_dropout()
            losses = []
            for step in range(2):
                losses.append(float(d.run_step((audio, labels.clone(), None, None), step).detach()))
                assert task.sed_student.arena.grads_are_flat(), (hook, step)
                assert not ops._deferred and ops.AFTER_FORWARD is None
            out.append((losses, task.sed_student.arena.flat.detach().clone(), task.sed_teacher.arena.flat.detach().clone()))
        finally:
            ops.SIDE_ON_CPU = False
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


def test_step_ignores_uninitialised_memory(emu):
    """Poisoned torch.empty buffers (NaN / 3e30) change no bit of two seeded training steps."""
    P.case_step_ignores_uninitialised_memory("cpu", n_samp=22 * 256)       # 23 frames -> 11 at block 1: an odd count in BOTH pooling blocks


def test_bn_backward_fold_equals_separate_pass(emu):
    """The BatchNorm backward inside the data-gradient convolution == the separate pass, bit for bit."""
    P.case_bn_fold_equals_separate_pass("cpu", n_samp=4096 + 1024)


def test_validation_step(emu):
    """SEDTask4.validation_step on the emulator: eval forward + batched decoding vs the oracle."""
    P.case_validation_step("cpu")


def test_test_epoch(emu, tmp_path):
    """test_step + on_test_epoch_end (SURVEY 8f ranks 1 + 2): device scoring -> operating points -> PSDS / F1 metrics."""
    P.case_test_epoch("cpu", tmp_path)


def test_embcat_op(emu):
    """K14: pooling + concat + dropout kernel and the cat_tf GEMMs vs torch ops."""
    P.case_embcat_op("cpu")


def test_embedding_crnn_matches_reference_golden(emu):
    import os
    import numpy as np
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_emb.npz"))
    P.case_embedding_crnn_vs_reference_golden("cpu", G)


def test_pretrained_training_step(emu):
    """sed_trainer_pretrained.SEDTask4: the mean-teacher step with frozen embeddings in the batch vs the oracle trainer."""
    P.case_pretrained_training_step("cpu")


def test_stochastic_step_matches_oracle(emu):
    """The benchmarked configuration (dropout on all 8 sites per model incl. the head, SpecAugment, mixup) as a whole against
    the oracle on the recorded draws: 2 steps, posteriors / scalars / every gradient."""
    P.case_stochastic_training_step("cpu", bs=(1, 1, 2), n_samp=16000 + 1024, steps=2)


def test_head_dropout(emu):
    P.case_head_dropout("cpu")
    P.case_head_dropout("cpu", B=2, T=5, p=0.25, seed=7)
    P.case_head_dropout("cpu", B=2, T=70, p=0.5, seed=9, D=384, NC=27)        # the 2024 recipe's head: 2 x 192 features, 27 classes


def test_crnn_masks_dropstep_interpolate_vs_reference_golden(emu):
    """classes_mask / pad_mask, dropstep_recurrent and aggregation_type "interpolate" against the reference module's outputs."""
    import os
    import numpy as np
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_emb2.npz"))
    P.case_crnn_masks_vs_reference_golden("cpu", G)


def test_dropstep_ops(emu):
    P.case_dropstep_draws_and_dropout("cpu")


def test_training_step_2024_vs_reference_golden(emu):
    """The 2024 recipe's 5-data-set step against the scalars its own reference trainer logged, and against the oracle."""
    import os
    import numpy as np
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_2024.npz"))
    P.case_training_step_2024("cpu", G)


def test_beats_fbank_and_extractor_vs_reference_golden(emu):
    """Kaldi fbank kernel vs the oracle and an independent float64 implementation; the BEATs encoder kernels
    (patch embedding, position convolution, gated relative-position attention, deep-norm layers) vs the reference module's output."""
    import os
    import numpy as np
    P.case_beats_fbank("cpu")
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_beats.npz"))
    P.case_beats_vs_reference_golden("cpu", G)


def test_beats_derived_weights_are_dropped_by_a_parents_load_state_dict():
    """loading through a PARENT module (BEATsModel / a LightningModule) reaches BEATs via `_load_from_state_dict`, not via its
    own load_state_dict: the fused q / k / v copy, the position-convolution planes and the packed weight images must not survive it; the
    per-weight caches are keyed by the tensor's version as well as its pointer (an in-place copy keeps the pointer)."""
    from desed_task_amd.beats import BEATs, BEATsConfig
    cfg = dict(input_patch_size=16, embed_dim=64, conv_bias=False, encoder_layers=1, encoder_embed_dim=128, encoder_ffn_embed_dim=128,
               encoder_attention_heads=2, activation_fn="gelu", layer_norm_first=False, deep_norm=True, conv_pos=16, conv_pos_groups=4,
               relative_position_embedding=True, num_buckets=32, max_distance=80, gru_rel_pos=True, dropout=0.0, attention_dropout=0.0,
               encoder_layerdrop=0.0)
    m = BEATs(BEATsConfig(cfg))
    m._packed = {"stale": 1}

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.child = m

    p = Parent()
    p.load_state_dict(p.state_dict())
    assert m._packed is None
    w = m.encoder.layers[0].fc1.weight
    v0 = w._version
    with torch.no_grad():
        w.copy_(torch.zeros_like(w))
    assert w._version != v0


def test_beats_12_layers_vs_reference_golden(emu):
    """12 layers x 496 tokens (one 10 s clip) against the reference module's own output."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_beats12.npz"))
    P.case_beats12_vs_reference_golden("cpu", G)


def test_beats_into_2024_step_chain(emu):
    """Extractor -> embeddings -> the 2024 step as one chain (toy size: 5 clips of 2 s, 2 layers)."""
    P.case_beats_chain_2024("cpu", bs=(1, 1, 1, 1, 1), n_samp=16000 * 2 + 1024, layers=2)


def test_attention_relpos_kernels(emu):
    """Both attention kernels (matrix-core default, vector-pipe) vs a float64 restatement: ragged tiles, gate / bias on and off."""
    for variant in (0, 1):
        P.case_attention_relpos("cpu", B=1, T=100, H=2, gated=True, bias=True, variant=variant)
    P.case_attention_relpos("cpu", B=2, T=70, H=1, gated=False, bias=True)
    P.case_attention_relpos("cpu", B=1, T=33, H=1, gated=False, bias=False)


def test_posconv_kernels(emu):
    P.case_posconv("cpu", B=1, T=70, groups=1, K=16)
    P.case_posconv("cpu", B=1, T=33, groups=2, K=8)


def test_pipelined_epoch_boundaries(emu):
    """Eager pipelined driver across epoch boundaries (no successor announced at an epoch's end) == the unpipelined order.  (The
    hipGraph forms and the reset_pipeline() case -- graphed against eager pipelined -- run on the GPU: without a graph both sides of
    the reset comparison would be the same driver.)"""
    P.case_pipelined_epoch_boundary("cpu", point="teacher", graph=False, epochs=2, per_epoch=2, n_samp=2048 + 1024)


def test_prefetched_2024_step_equals_unpipelined(emu):
    """The 2024 five-data-set step, front half + teacher CNN forward one step early == the unpipelined order, bit for bit."""
    P.case_prefetch_2024_equals_unpipelined("cpu", graph=False, steps=3, n_samp=2048 + 1024, te=9)


def test_lightning_hook_order_whole_step_equals_driver(emu):
    """SEDTask4's whole-step mode driven by Lightning 1.9's hook order (tests/lightning_order.py: torch.optim.Adam, batches from
    train_dataloader()'s look-ahead loader) == the pipelined step driver driven by hand == the hooks run one by one, bit for bit,
    across epoch ends.  (The hipGraph form of the same comparison runs on the GPU.)"""
    P.case_lightning_surface("cpu", epochs=2, per_epoch=3, n_samp=2048 + 1024)


def test_lightning_abandoned_epoch_and_back_to_back_steps(emu):
    """a stale announced successor is dropped, not consumed; training_step twice without optimizer.step() updates twice."""
    P.case_lightning_surface_abandoned_epoch("cpu")


def test_lightning_hook_order_limit_train_batches(emu):
    """`limit_train_batches` (train_sed.py:256): the batch before the cut announces no successor."""
    P.case_lightning_surface("cpu", epochs=2, per_epoch=3, n_samp=2048 + 1024, limit_train_batches=2)


def test_lightning_hook_order_2024_recipe(emu):
    """The 2024 five-data-set trainer (27 classes, embeddings + valid_class_mask in the batch, labels AND embeddings mixed in the
    hand-over buffers) through the same three routes: whole-step behind Lightning's hook order == the driver by hand == hook by hook."""
    P.case_lightning_surface("cpu", epochs=2, per_epoch=2, n_samp=2048 + 1024, recipe2024=True)
