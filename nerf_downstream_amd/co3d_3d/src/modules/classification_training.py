"""Classification step logic (counterpart of the reference's
co3d_3d/src/modules/classification_training.py:19-97, without PyTorch-Lightning):
forward = model.process_input(batch) -> model(...), loss = F.cross_entropy(logits, labels),
top-1 / top-5 in percent, NaN guard on the logged loss."""
import numpy as np
import torch
import torch.nn.functional as F


def cross_entropy(logits, labels):
    """F.cross_entropy(logits, labels) (reference :33); device logits take the backend's one-launch kernels."""
    if logits.is_cuda:
        from nerf_downstream_amd.minkowski.functional import cross_entropy as hip_ce

        return hip_ce(logits, labels)
    return F.cross_entropy(logits, labels)


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """Percent of samples whose label is among the k highest logits (reference :83-97)."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    return [float(correct[: min(k, maxk)].any(0).float().sum() * (100.0 / target.numel())) for k in topk]


class ClassificationTraining:
    monitor = "val/acc1"

    def __init__(self, model):
        self.model = model

    def forward(self, batch_or_field):
        x = batch_or_field if hasattr(batch_or_field, "sparse") else self.model.process_input(batch_or_field)
        return self.model(x)

    def training_step(self, batch, field=None):
        out = self.forward(field if field is not None else batch)
        labels = batch["labels"].long()
        return cross_entropy(out, labels), out

    @staticmethod
    def check_finite(loss_float):
        if not np.isfinite(loss_float):
            raise ValueError(f"Invalid loss: {loss_float}")

    @torch.no_grad()
    def validation_step(self, batch):
        logits = self.forward(batch)
        labels = batch["labels"].long()
        loss = cross_entropy(logits, labels)
        top = logits.topk(min(5, logits.shape[1]), 1).indices
        c1 = (top[:, 0] == labels).sum()
        c5 = (top == labels[:, None]).any(1).sum()
        return loss.detach(), c1, c5, labels.numel()

    @torch.no_grad()
    def train_metrics(self, out, batch):
        acc1, acc5 = accuracy(out, batch["labels"].long(), topk=(1, 5))
        return {"train/acc1": acc1, "train/acc5": acc5}

    @torch.no_grad()
    def val_accumulate(self, batch):
        """-> float64 vector [loss * n, correct@1, correct@5, n] that validate() sums over batches and ranks."""
        loss, c1, c5, n = self.validation_step(batch)
        return torch.stack([loss.double() * n, c1.double(), c5.double(), torch.tensor(float(n), device=loss.device, dtype=torch.float64)])

    @staticmethod
    def val_metrics(tot):
        n = max(float(tot[3]), 1.0)
        return {"val/loss": float(tot[0]) / n, "val/acc1": 100.0 * float(tot[1]) / n, "val/acc5": 100.0 * float(tot[2]) / n}
