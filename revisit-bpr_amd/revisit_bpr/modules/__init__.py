from revisit_bpr.modules.neg_samplers import AdaptiveSampler, Sampler, UniformSampler

__all__ = ["Sampler", "UniformSampler", "AdaptiveSampler"]
