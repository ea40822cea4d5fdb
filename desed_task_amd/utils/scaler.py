"""TorchScaler with the reference's interface (desed_task/utils/scaler.py:5-120).

The configuration on the hot path -- statistic="instance", normtype="minmax" -- runs on the HIP min/max kernels
(features.minmax_scale).  The other combinations are not on the 2023 recipe's path; they are kept functional
with plain device-side torch reductions so existing configs do not break."""
import torch

from .. import features


class TorchScaler(torch.nn.Module):
    def __init__(self, statistic="dataset", normtype="standard", dims=(1, 2), eps=1e-8):
        super().__init__()
        assert statistic in ["dataset", "instance", None]
        assert normtype in ["standard", "mean", "minmax", None]
        if statistic == "dataset" and normtype == "minmax":
            raise NotImplementedError("statistic==dataset and normtype==minmax is not currently implemented.")
        self.statistic, self.normtype, self.dims, self.eps = statistic, normtype, dims, eps

    def load_state_dict(self, state_dict, strict=True):
        if self.statistic == "dataset":
            super().load_state_dict(state_dict, strict)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self.statistic == "dataset":
            super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def fit(self, dataloader, transform_func=lambda x: x[0]):
        n = 0
        mean = mean_sq = None
        for batch in dataloader:
            feats = transform_func(batch)
            m = torch.mean(feats, self.dims, keepdim=True).mean(0).unsqueeze(0)
            m2 = torch.mean(feats ** 2, self.dims, keepdim=True).mean(0).unsqueeze(0)
            mean = m if mean is None else mean + m
            mean_sq = m2 if mean_sq is None else mean_sq + m2
            n += 1
        self.register_buffer("mean", mean / n)
        self.register_buffer("mean_squared", mean_sq / n)

    def forward(self, tensor):
        if self.statistic is None or self.normtype is None:
            return tensor
        if self.statistic == "instance" and self.normtype == "minmax" and tuple(self.dims) == tuple(range(1, tensor.dim())):
            return features.minmax_scale(tensor, eps=self.eps)
        if self.statistic == "dataset":
            assert hasattr(self, "mean") and hasattr(self, "mean_squared"), "TorchScaler should be fit before used if statistics=dataset"
            if self.normtype == "mean":
                return tensor - self.mean
            if self.normtype == "standard":
                return (tensor - self.mean) / (torch.sqrt(self.mean_squared - self.mean ** 2) + self.eps)
            raise NotImplementedError
        if self.normtype == "mean":
            return tensor - torch.mean(tensor, self.dims, keepdim=True)
        if self.normtype == "standard":
            return (tensor - torch.mean(tensor, self.dims, keepdim=True)) / (torch.std(tensor, self.dims, keepdim=True) + self.eps)
        if self.normtype == "minmax":
            mn = torch.amin(tensor, dim=self.dims, keepdim=True)
            mx = torch.amax(tensor, dim=self.dims, keepdim=True)
            return (tensor - mn) / (mx - mn + self.eps) * 2 - 1
        raise NotImplementedError
