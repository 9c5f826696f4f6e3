import torch.nn as nn


class GeneratorLoss(nn.Module):
    def __init__(self, reduction="mean", override_train_ops=None):
        super().__init__()
        self.reduction = reduction
        self.override_train_ops = override_train_ops
        self.arg_map = {}


class DiscriminatorLoss(nn.Module):
    def __init__(self, reduction="mean", override_train_ops=None):
        super().__init__()
        self.reduction = reduction
        self.override_train_ops = override_train_ops
        self.arg_map = {}
