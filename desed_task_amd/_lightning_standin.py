"""What SEDTask4 needs of `pytorch_lightning.LightningModule` when Lightning is not installed (this image has none): `hparams`,
`log`, and Lightning 1.9's DEFAULT implementations of the hooks its automatic-optimisation loop calls
(pytorch_lightning/core/module.py), so that a hand-written loop in that order (tests/lightning_order.py, bench.py --surface
lightning, desed_task_amd.launcher) drives the stand-in and the real class alike.  With Lightning installed this module is unused
except for `move_to_device`."""
import torch


class LightningModule(torch.nn.Module):
    current_epoch = 0       # LightningModule exposes the trainer's epoch here; a hand-written loop may set this attribute
    trainer = None

    def __init__(self):
        super().__init__()
        self.hparams = {}
        self.logged = {}

    def log(self, name, value, **kwargs):
        """Values stay where they are (device tensors are not synchronised)."""
        self.logged[name] = value

    def optimizer_step(self, epoch, batch_idx, optimizer, optimizer_idx=0, optimizer_closure=None, **kwargs):
        optimizer.step(closure=optimizer_closure)

    def optimizer_zero_grad(self, epoch, batch_idx, optimizer, optimizer_idx=0):
        optimizer.zero_grad()

    def backward(self, loss, optimizer=None, optimizer_idx=None, *args, **kwargs):
        loss.backward(*args, **kwargs)

    def transfer_batch_to_device(self, batch, device, dataloader_idx=0):
        return move_to_device(batch, device)


def move_to_device(batch, device):
    """lightning's move_data_to_device for the containers a DataLoader's default collate produces."""
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, (list, tuple)) and not hasattr(batch, "_fields"):
        return type(batch)(move_to_device(b, device) for b in batch)
    if isinstance(batch, dict):
        return {k: move_to_device(v, device) for k, v in batch.items()}
    return batch
