"""Segmentation step logic (counterpart of the reference's co3d_3d/src/modules/segmentation_training.py:27-238
without PyTorch-Lightning): per-point logits `model(field)` (Res16UNet: `out.slice(x).F`), cross entropy
with an ignore label and an optional weight on the last ("void") class (SegLoss :27-44), and the
confusion-matrix metrics of src/utils `fast_hist` / `per_class_iu` (:115-126) accumulated on the device."""
import numpy as np
import torch
import torch.nn.functional as F

from nerf_downstream_amd import gin_lite as gin

EPS = 1e-10


@torch.no_grad()
def confusion(pred, label, n):
    """hist[label, pred] over the points with 0 <= label < n (reference fast_hist)."""
    k = (label >= 0) & (label < n)
    return torch.bincount(n * label[k] + pred[k], minlength=n * n).reshape(n, n)


def iou_metrics(hist):
    """-> (mIoU, mAcc, OA) in percent; classes that never occur (no label, no prediction) do not count."""
    hist = hist.double()
    tp, rows, cols = hist.diag(), hist.sum(1), hist.sum(0)
    seen = (rows + cols) > 0
    iou = tp / (rows + cols - tp + EPS)
    acc = tp / (rows + EPS)
    has = rows > 0
    miou = float(iou[seen].mean()) if bool(seen.any()) else 0.0
    macc = float(acc[has].mean()) if bool(has.any()) else 0.0
    return 100.0 * miou, 100.0 * macc, 100.0 * float(tp.sum() / (hist.sum() + EPS))


@gin.configurable
class SegmentationTraining:
    monitor = "val/mIoU"

    def __init__(self, model, ignore_label=255, void_weight=None):
        self.model, self.ignore_label, self.void_weight = model, ignore_label, void_weight
        self._weight = None

    def forward(self, batch_or_field):
        x = batch_or_field if hasattr(batch_or_field, "sparse") else self.model.process_input(batch_or_field)
        return self.model(x)

    def loss(self, logits, labels):
        weight = None
        if self.void_weight is not None and self.void_weight > 0:  # reference SegLoss: weight[-1] = void_weight
            if self._weight is None or self._weight.device != logits.device or self._weight.numel() != logits.shape[1]:
                self._weight = torch.ones(logits.shape[1], device=logits.device)
                self._weight[-1] = self.void_weight
            weight = self._weight
        return F.cross_entropy(logits, labels, weight=weight, ignore_index=self.ignore_label)

    def training_step(self, batch, field=None):
        out = self.forward(field if field is not None else batch)
        return self.loss(out, batch["labels"].long()), out

    @staticmethod
    def check_finite(loss_float):
        if not np.isfinite(loss_float):
            raise ValueError(f"Invalid loss: {loss_float}")

    @torch.no_grad()
    def train_metrics(self, out, batch):
        labels = batch["labels"].long()
        miou, macc, oa = iou_metrics(confusion(out.argmax(1), labels, out.shape[1]))
        return {"train/mIoU": miou, "train/mAcc": macc, "train/OA": oa,
                "train/ignore_ratio": 100.0 * float((labels == self.ignore_label).float().mean())}

    @torch.no_grad()
    def val_accumulate(self, batch):
        """-> float64 vector [loss * points, points, confusion matrix...] that validate() sums over batches and ranks."""
        logits = self.forward(batch)
        labels = batch["labels"].long()
        n = int(((labels >= 0) & (labels < logits.shape[1])).sum())
        loss = self.loss(logits, labels) if n else logits.sum() * 0
        hist = confusion(logits.argmax(1), labels, logits.shape[1])
        return torch.cat([torch.stack([loss.double() * n, torch.tensor(float(n), device=logits.device, dtype=torch.float64)]),
                          hist.double().flatten()])

    def val_metrics(self, tot):
        n = max(float(tot[1]), 1.0)
        c = int(round((tot.numel() - 2) ** 0.5))
        miou, macc, oa = iou_metrics(tot[2:].reshape(c, c))
        return {"val/loss": float(tot[0]) / n, "val/mIoU": miou, "val/mAcc": macc, "val/OA": oa}
