"""Predictor data set on the device: encode -> BIC -> (mu, target) rows.

Mirror of the reference's ``prepare_predictor_data`` (experiments/01_bn_asia/main.py:268-303) and of
``generate_predictor_graphs_batch`` / ``create_predictor_dataset`` (src/predictors/utils.py:15-59).  The reference encodes
ONE graph per ``model.encode([g])`` call and starts one ``Rscript`` per graph for the target; here a batch of graphs stays
on the GPU as its row codec (``CompactBatch``), ``dvs_build_records`` + ``dvs_encode`` produce the posterior means and
``dvs_bic_parent_masks`` + ``dvs_bic_scores`` the BIC targets — four HIP launch sequences per batch, no host round trip
until the rows are written.  Output has the reference's schema: ``vector`` (float32[latent]) and ``target`` (float64).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from .records import CompactBatch, encode_graphs


def _as_compact(model, graphs) -> CompactBatch:
    if isinstance(graphs, CompactBatch):
        return graphs
    return encode_graphs(list(graphs), model.max_num_vertices - 3, model.graph_label_key)


def generate_predictor_graphs_batch(model, evaluator, graphs) -> Tuple[torch.Tensor, torch.Tensor]:
    """src/predictors/utils.py:15-34 for a whole batch: (vectors float32 [B, latent], targets float64 [B]), both on the
    model's device.  ``evaluator``: a ``bic.BNLearnWrapper`` or — the reference's call shape, experiments/01_bn_asia/
    main.py:296 — its bound ``.score`` method (both take the batched on-device path through ``score_compact``), or any other
    callable graph -> float (then the targets come from that callable, one graph at a time, on the host)."""
    dev = model.flat_params.device
    batch = _as_compact(model, graphs).to(dev)
    was_training = model.training
    model.eval()                                         # main.py:278: the reference encodes in eval mode
    try:
        mu, _ = model.encode_direct(batch)
    finally:
        model.train(was_training)
    owner = getattr(evaluator, "__self__", evaluator)      # `BNLearnWrapper(...).score` (a bound method) -> the wrapper
    if getattr(evaluator, "__name__", "score") == "score" and hasattr(owner, "score_compact"):
        y = owner.score_compact(batch)
    else:
        from .records import decode_graphs
        y = torch.tensor([float(evaluator(g)) for g in decode_graphs(batch)], dtype=torch.float64, device=dev)
    return mu.detach(), y


def create_predictor_dataset(model, graphs_dataloader: Iterable, output_dir: Optional[str], evaluator,
                             npartitions: int = 4):
    """src/predictors/utils.py:37-59: one ``part-{i}.parquet`` per batch of the loader under ``output_dir`` (columns
    ``vector``, ``target``).  Difference on purpose: the reference writes its parts into ``output_dir + '_tmp'``
    (utils.py:45) and leaves the re-partitioning into ``output_dir`` commented out; here the parts go to ``output_dir``
    itself, which is where its readers (main.py:315-330) look.  With ``output_dir=None`` nothing is written.  Returns (vectors [N, latent], targets [N]) on the
    device, rows in loader order.  ``npartitions`` is accepted for signature compatibility (the reference ignores it too)."""
    vecs: List[torch.Tensor] = []
    tgts: List[torch.Tensor] = []
    for batch_ind, batch in enumerate(graphs_dataloader):
        mu, y = generate_predictor_graphs_batch(model, evaluator, batch)
        vecs.append(mu)
        tgts.append(y)
        if output_dir is not None:
            import pandas as pd
            os.makedirs(output_dir, exist_ok=True)
            pd.DataFrame({"vector": list(mu.cpu().numpy()), "target": y.cpu().numpy()}).to_parquet(
                os.path.join(output_dir, f"part-{batch_ind}.parquet"), engine="pyarrow")
    return torch.cat(vecs), torch.cat(tgts)


def prepare_predictor_data(model, graphs: Sequence, evaluator, batch_size: int = 64, output_dir: Optional[str] = None,
                           checkpoint: Optional[str] = None):
    """experiments/01_bn_asia/main.py:268-303: (optionally) load a checkpoint, eval mode, batches of ``batch_size`` graphs
    (``drop_last=True`` like the reference's loader; no shuffle — pass shuffled graphs for the reference's order),
    encode + score, write the rows."""
    if checkpoint is not None:
        from .train import load_model_state
        load_model_state(model, checkpoint)
    model.eval()
    graphs = list(graphs)
    n_full = len(graphs) - len(graphs) % batch_size
    loader = (graphs[s:s + batch_size] for s in range(0, n_full, batch_size))
    return create_predictor_dataset(model, loader, output_dir, evaluator)
