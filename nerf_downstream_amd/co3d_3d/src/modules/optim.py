"""gin-configurable optimizers / schedulers (counterpart of the reference's
co3d_3d/src/modules/optim.py:12-121: thin subclasses so `SGD.momentum = 0.9` style bindings
work, and CosineAnnealingLR taking T_max from `train.max_steps` / `train.max_epochs`)."""
import torch.optim as optim
import torch.optim.lr_scheduler as lr_scheduler

from nerf_downstream_amd import gin_lite as gin


@gin.configurable
class SGD(optim.SGD):
    pass


@gin.configurable
class Adam(optim.Adam):
    pass


@gin.configurable
class AdamW(optim.AdamW):
    pass


OPTIMIZERS = {c.__name__: c for c in (SGD, Adam, AdamW)}


def get_optimizer(optimizer_name, parameters, lr, weight_decay):
    if optimizer_name not in OPTIMIZERS:
        raise ValueError(f"optimizer {optimizer_name} not recognized in {sorted(OPTIMIZERS)}.")
    parameters = list(parameters)
    kw = {}
    if optimizer_name == "SGD" and parameters and all(p.is_cuda for p in parameters):
        kw["fused"] = True  # same update rule, one multi-tensor kernel instead of four foreach passes
    return OPTIMIZERS[optimizer_name](parameters, lr=lr, weight_decay=weight_decay, **kw)


@gin.configurable
class StepLR(lr_scheduler.StepLR):
    pass


@gin.configurable
class CosineAnnealingLR(lr_scheduler.CosineAnnealingLR):
    def __init__(self, optimizer, eta_min=0, last_epoch=-1):
        interval = gin.query_parameter("train.scheduler_interval")
        T_max = gin.query_parameter("train.max_steps" if interval == "step" else "train.max_epochs")
        super().__init__(optimizer, T_max, eta_min, last_epoch)


class _Poly:  # picklable (checkpoints store the scheduler state)
    def __init__(self, max_steps, poly_exp):
        self.max_steps, self.poly_exp = max_steps, poly_exp

    def __call__(self, step):
        return (1 - step / (self.max_steps + 1)) ** self.poly_exp


@gin.configurable
class PolyLR(lr_scheduler.LambdaLR):
    """lr * (1 - step / (max_steps + 1)) ** poly_exp -- the segmentation recipe's schedule (reference optim.py:180-205,
    configs/scannet_semseg.gin:55-61)."""

    def __init__(self, optimizer, poly_exp=0.9):
        self.max_steps, self.poly_exp = gin.query_parameter("train.max_steps"), poly_exp
        super().__init__(optimizer, _Poly(self.max_steps, poly_exp))


@gin.configurable
class MultiStepLR(lr_scheduler.MultiStepLR):
    def __init__(self, optimizer, milestones=(20000, 40000), gamma=0.1, last_epoch=-1):
        super().__init__(optimizer, list(milestones), gamma, last_epoch)


@gin.configurable
class ExponentialLR(lr_scheduler.ExponentialLR):
    def __init__(self, optimizer, gamma=0.99):
        super().__init__(optimizer, gamma)


SCHEDULERS = {c.__name__: c for c in (StepLR, MultiStepLR, ExponentialLR, CosineAnnealingLR, PolyLR)}


def get_scheduler(scheduler_name, optimizer, warmup_steps=-1):
    if scheduler_name.lower() == "none":
        return None
    if scheduler_name not in SCHEDULERS:
        raise ValueError(f"scheduler {scheduler_name} not recognized in {sorted(SCHEDULERS)}.")
    if warmup_steps and warmup_steps > 0:
        raise NotImplementedError("warm-up schedules are not used by the co3d classification configs")
    return SCHEDULERS[scheduler_name](optimizer)
