"""``experiments.bpr`` — the package the reference's BPR configs address with
``_target_: experiments.bpr.Experiment``; the experiment itself lives in ``exp.py``."""
from . import exp as _exp

Experiment = _exp.BPRExperiment
__all__ = ["Experiment"]
