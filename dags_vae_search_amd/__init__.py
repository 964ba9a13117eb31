"""dags_vae_search_amd — MI355X-native DAG-VAE (PACE) train-step hot path of rlog58/dags-vae-search.

Drop-in surface for that path: PaceVaeV3, train_batch, pace_collate_fn, collate_graph_batch, load_model_state,
LabeledDag (row codec).  All arithmetic runs in libdvs_hip.so (hand-written HIP for gfx950); there is no CPU path.
"""
from .features import LabeledDag, LabeledGraph, collate_graph_batch, pace_collate_fn, prepare_features  # noqa: F401
from .pace import PaceVaeV3  # noqa: F401
from .train import batch_test, load_model_state, model_test, train_batch, train_model  # noqa: F401
from . import optim  # noqa: F401
from .records import CompactBatch, CompactDagDataset, encode_graphs  # noqa: F401
from .bic import BNLearnWrapper  # noqa: F401
from .predictor_data import create_predictor_dataset, generate_predictor_graphs_batch, prepare_predictor_data  # noqa: F401
from .datasets import LabeledDagDatasetInMemory, LabeledDagDatasetInMemoryTest  # noqa: F401
