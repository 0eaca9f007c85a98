from revisit_bpr.models.bpr.loss import Loss
from revisit_bpr.models.bpr.model import (MF, BaseLogitModel, FreeItemKNN, ItemKNN, Model, get_backend,
                                          set_backend)

__all__ = ["Model", "MF", "ItemKNN", "FreeItemKNN", "BaseLogitModel", "Loss", "set_backend", "get_backend"]
