"""Bidirectional GRU container (state-dict layout of desed_task/nnet/RNN.py:7-30: `rnn.weight_ih_l{k}[_reverse]` ...).
nn.GRU is used as the parameter holder / initialiser only; the arithmetic is ops.BiGRULayerFn."""
import torch.nn as nn

from ..ops import BiGRULayerFn


class BidirectionalGRU(nn.Module):
    def __init__(self, n_in, n_hidden, dropout=0, num_layers=1):
        super().__init__()
        if n_hidden not in (128, 192):
            raise NotImplementedError("HIP GRU kernels are built for n_hidden = 128 and 192 (the 2023 / 2024 recipes' n_RNN_cell)")
        if dropout:
            raise NotImplementedError("inter-layer GRU dropout (dropout_recurrent) is not on the 2023 path")
        self.num_layers = num_layers
        self.rnn = nn.GRU(n_in, n_hidden, bidirectional=True, dropout=dropout, batch_first=True, num_layers=num_layers)

    def forward(self, input_feat, arena=None):
        x = input_feat
        cfg = dict(arena=arena)
        for k in range(self.num_layers):
            g = lambda n: getattr(self.rnn, "%s_l%d" % (n, k))          # noqa: E731
            r = lambda n: getattr(self.rnn, "%s_l%d_reverse" % (n, k))  # noqa: E731
            x = BiGRULayerFn.apply(x, g("weight_ih"), g("weight_hh"), g("bias_ih"), g("bias_hh"),
                                   r("weight_ih"), r("weight_hh"), r("bias_ih"), r("bias_hh"), cfg)
        return x
