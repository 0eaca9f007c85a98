"""Model zoo entry point.  Only the BPR family is implemented natively (the other
reference families — MultVAE/DAE, EASE, popularity — are paper baselines outside the hot path)."""
from revisit_bpr.models.bpr import Loss as BPRLoss
from revisit_bpr.models.bpr import Model as BPR

__all__ = ["BPR", "BPRLoss"]
