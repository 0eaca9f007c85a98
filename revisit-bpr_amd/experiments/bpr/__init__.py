from experiments.bpr.exp import BPRExperiment as Experiment

__all__ = ["Experiment"]
