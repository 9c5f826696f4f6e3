import torch.nn as nn


class Generator(nn.Module):
    def __init__(self, encoding_dims, label_type="none"):
        super().__init__()
        self.encoding_dims = encoding_dims
        self.label_type = label_type

    def _weight_initializer(self):
        pass  # fixtures always load explicit seeded weights


class Discriminator(nn.Module):
    def __init__(self, input_dims, label_type="none"):
        super().__init__()
        self.input_dims = input_dims
        self.label_type = label_type

    def _weight_initializer(self):
        pass
