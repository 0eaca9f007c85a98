"""BPR pairwise loss with the interface of the reference's ``revisit_bpr/models/bpr/loss.py:5-21``
(``Loss(size_average)``; the model sums it over the batch, see ``Model.forward``).

Only the dense PyTorch restatement (``set_backend("torch")``) and the eval branch evaluate it as a
torch op; on the fused path the same quantity, softplus(-x), is accumulated inside the HIP kernels
(``neg_logsigmoid`` in ``csrc/bpr_device.h``)."""
import torch
from torch.nn.functional import softplus


class Loss(torch.nn.Module):
    def __init__(self, size_average: bool = True) -> None:
        super().__init__()
        # True: mean over all elements; False: the per-element values (the BPR model sums them)
        self.size_average = size_average

    def forward(self, logits: torch.Tensor) -> torch.Tensor:
        value = softplus(logits.neg())  # -log(sigmoid(x)) in its overflow-safe form
        if self.size_average:
            value = value.mean()
        return value
