"""Trainer-package compatibility: use pytorch_lightning when it is installed (the reference's
runtime, requirements.txt:172); otherwise a minimal stand-in so C_NETWORK is still an nn.Module
with the attributes the reference's step functions touch (self.hparams, self.config, log_dict)."""
import torch

try:                                                       # pragma: no cover - not in this image
    import pytorch_lightning as pl
    try:
        from pytorch_lightning.core.lightning import LightningModule
    except ImportError:
        from pytorch_lightning import LightningModule
    seed_everything = pl.seed_everything
    HAVE_LIGHTNING = True
except ImportError:
    HAVE_LIGHTNING = False

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._hparams = {}
            self.logged = {}

        @property
        def hparams(self):
            return self._hparams

        def save_hyperparameters(self, *args, **kwargs):
            pass

        def log_dict(self, metrics, **kwargs):
            self.logged.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in metrics.items()})

        @property
        def current_epoch(self):
            return 0

    def seed_everything(seed):
        import random
        import numpy as np
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        return seed
