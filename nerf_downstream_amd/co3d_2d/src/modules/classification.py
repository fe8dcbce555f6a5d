"""Training-step logic of the 2-D baseline (counterpart of the reference's LitModel,
co3d_2d/src/modules/classification.py:43-163, without PyTorch-Lightning): cross-entropy with label smoothing 0.005,
weight decay as an explicit loss term `wd * ||W||_2` summed over the conv / fc / downsample weights (:81-86), SGD
momentum 0.9 (:63), learning rate warmed up linearly over the first 10 % of the steps and then cos(pi/2 * progress)
(:129-151), top-1 accuracy in percent."""
import numpy as np
import torch
import torch.nn as nn

from nerf_downstream_amd import gin_lite as gin

from ..model.models import select_model


def lr_at(step, lr, num_training_steps):
    peak = int(num_training_steps * 0.1)
    if step <= peak:
        return lr * (step / max(peak, 1))
    return lr * float(np.cos((step - peak) / max(num_training_steps - peak, 1) * np.pi / 2))


@gin.configurable
class LitModel(nn.Module):
    def __init__(self, model_name=None, lr=0.1, weight_decay=1e-4):
        super().__init__()
        self.model_name, self.lr, self.weight_decay = model_name, lr, weight_decay
        self.model = select_model(model_name)
        self.loss = nn.CrossEntropyLoss(label_smoothing=0.005)

    def configure_optimizers(self):
        return torch.optim.SGD(self.parameters(), self.lr, momentum=0.9)

    def wd_loss(self):
        wd = 0
        for name, param in self.named_parameters():
            if ("conv" in name or "fc" in name or "downsample" in name) and "weight" in name:
                wd = wd + self.weight_decay * param.norm()
        return wd

    def training_step(self, batch):
        labels, imgs = batch["labels"], batch["images"]
        prediction = self.model(imgs)
        celoss = self.loss(prediction, labels)
        acc = (prediction.argmax(1) == labels).float().mean() * 100
        wdloss = self.wd_loss()
        return celoss + wdloss, {"train/celoss": celoss.detach(), "train/wdloss": wdloss.detach() if torch.is_tensor(wdloss) else wdloss,
                                  "train/acc": acc}

    @torch.no_grad()
    def evaluation(self, loader, device, prefix="val"):
        was = self.training
        self.eval()
        preds, labels = [], []
        for batch in loader:
            preds.append(self.model(batch["images"].to(device)).float())
            labels.append(batch["labels"].to(device))
        self.train(was)
        preds, labels = torch.cat(preds), torch.cat(labels)
        return {f"{prefix}/acc": float((preds.argmax(1) == labels).float().mean() * 100), f"{prefix}/loss": float(self.loss(preds, labels))}
