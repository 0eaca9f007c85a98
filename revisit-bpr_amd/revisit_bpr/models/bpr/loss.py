"""BPR pairwise loss, API of the reference's revisit_bpr/models/bpr/loss.py:5-21."""
import torch
from torch.nn.functional import softplus


class Loss(torch.nn.Module):
    """-log(sigmoid(x)) per element, or its mean when `size_average` (the default) is set."""

    def __init__(self, size_average: bool = True) -> None:
        super().__init__()
        self.size_average = size_average

    def forward(self, logits: torch.Tensor) -> torch.Tensor:
        per_pair = softplus(-logits)  # == -logsigmoid(logits), same stable formulation in ATen
        return per_pair.mean() if self.size_average else per_pair
