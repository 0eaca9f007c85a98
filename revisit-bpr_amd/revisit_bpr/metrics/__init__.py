from revisit_bpr.metrics.auc import RocAucMany, RocAucManySlow, RocAucOne
from revisit_bpr.metrics.metric import MaskedMetric, Metric, prepare_target, ranked_targets
from revisit_bpr.metrics.ranking import MAP, NDCG, FBeta, Precision, Recall

__all__ = ["Metric", "MaskedMetric", "NDCG", "Recall", "Precision", "MAP", "FBeta", "RocAucOne",
           "RocAucMany", "RocAucManySlow", "prepare_target", "ranked_targets"]
