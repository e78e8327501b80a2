"""Mirror of reference src/config.py:77-92 + src/networks/config.py:25-32 (model factory only)."""
from .networks.decoders import Decoders


def get_model(cfg):
    return Decoders(c_dim=cfg['model']['c_dim'], truncation=cfg['model']['truncation'],
                    learnable_beta=cfg['rendering']['learnable_beta'])
