"""lightningdot_amd — MI355X-native retrieval hot path behind LightningDOT's indexer / loss interfaces.

The compute lives in ``libldot.so`` (hand-written HIP for gfx950, C ABI in ``include/ldot.h``); this package is the
host-side mirror of the reference's Python surface for that path:

    indexer.DenseIndexer / DenseFlatIndexer      <- dvl/indexer/faiss_indexers.py:22-87
    loss.dot_product_scores / BiEncoderNllLoss / _calc_loss
                                                 <- dvl/models/bi_encoder.py:54-68,613-665 ; dvl/utils.py:114-169
    harness.eval_model_on_dataloader / get_indexer
                                                 <- dvl/trainer.py:93-190
    hn.sampled_hard_negatives                    <- dvl/hn.py:45-66
    sharded.ShardedFlatIndexer                   <- (new) row-sharded index over torch.distributed / RCCL

There is NO CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from ._lib import LdotError, lib_path, load_library  # noqa: F401

__all__ = ['LdotError', 'lib_path', 'load_library']
