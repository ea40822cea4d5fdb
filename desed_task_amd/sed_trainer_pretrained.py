"""Drop-in for the pretrained-embeddings trainer, recipes/dcase2023_task4_baseline/local/sed_trainer_pretrained.py
(SURVEY 8f rank 3): the same mean-teacher step with frozen per-clip embeddings (BEATs: 768 x 496) handed to the student and
the teacher (`detect(features, model, embeddings)`, :276-280), fused into the CRNN by K14 + `cat_tf`
(nnet/CRNN.py, `use_embeddings=True, aggregation_type="pool1d"`).

Built: `pretrained.e2e: False` -- embeddings pre-computed and delivered with the batch `(audio, labels, padded_indxs,
embeddings)` (:291-292).  Not built: `e2e: True` (running the BEATs / AST extractor inside the step is SURVEY 8f rank 4) --
raises NotImplementedError.  Everything else (losses, EMA, scheduler, logged keys, validation / test scoring) is inherited
unchanged from sed_trainer.SEDTask4, as in the reference where the two files differ only by the `embeddings` plumbing.
"""
from .sed_trainer import SEDTask4 as _SEDTask4


class SEDTask4(_SEDTask4):
    def __init__(self, hparams, encoder, sed_student, pretrained_model=None, opt=None, train_data=None, valid_data=None,
                 test_data=None, train_sampler=None, scheduler=None, fast_dev_run=False, evaluation=False, sed_teacher=None):
        if hparams.get("pretrained", {}).get("e2e", False):
            raise NotImplementedError("pretrained.e2e = True (embedding extractor inside the step) is not built: "
                                      "pre-compute the embeddings (SURVEY 8f rank 4)")
        super().__init__(hparams, encoder, sed_student, opt=opt, train_data=train_data, valid_data=valid_data,
                         test_data=test_data, train_sampler=train_sampler, scheduler=scheduler, fast_dev_run=fast_dev_run,
                         evaluation=evaluation, sed_teacher=sed_teacher)

    def _batch_embeddings(self, batch):
        if len(batch) < 4 or batch[3] is None:
            raise ValueError("the pretrained step expects batches (audio, labels, padded_indxs, embeddings)")
        return batch[3]

    # validation / test batches of the pretrained recipe are (audio, labels, padded_indxs, filenames, embeddings)
    def _eval_embeddings(self, batch):
        if len(batch) < 5 or batch[4] is None:
            raise ValueError("the pretrained validation / test step expects (audio, labels, padded_indxs, filenames, embeddings)")
        return batch[4]
